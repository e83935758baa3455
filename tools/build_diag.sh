#!/bin/bash
# build/variants/libdiag.so: the engine with the region profile of k_miller_multi_rr compiled in (-DRB_MILLER_PROF: rhip_debug_miller_prof),
# linked with the current objects of the other translation units.  Used by tools/prof_miller.sh through RABE_HIP_LIB.
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -mllvm -misched-prera-direction=topdown -mllvm -greedy-reverse-local-assignment -DRB_MILLER_PROF -c rabe_amd/csrc/engine_rr.hip -o build/variants/diag.engine_rr.o
hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libdiag.so build/obj/engine.hip.o build/obj/engine_jobs.hip.o build/obj/engine_coop.hip.o build/obj/engine_coop_w1.hip.o \
  build/variants/diag.engine_rr.o build/obj/engine_sym.hip.o build/obj/schemes.cpp.o build/obj/host_abi.cpp.o build/obj/packed.cpp.o build/obj/pipeline.cpp.o build/obj/records.cpp.o
echo build/variants/libdiag.so
