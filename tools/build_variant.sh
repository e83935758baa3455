#!/bin/bash
# tools/build_variant.sh <name> <flags...>: engine.hip compiled with extra flags, linked with the current objects of the
# other translation units into build/variants/lib<name>.so (run with RABE_HIP_LIB=build/variants/lib<name>.so)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/variants
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -c rabe_amd/csrc/engine.hip -o build/variants/$name.engine.o "$@"
hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib$name.so build/variants/$name.engine.o build/obj/engine_jobs.hip.o build/obj/schemes.cpp.o build/obj/host_abi.cpp.o build/obj/packed.cpp.o build/obj/pipeline.cpp.o
echo build/variants/lib$name.so
