"""Compiles the HIP engine for gfx950 in-tree: rabe_amd/csrc/*.hip + host/*.cpp -> rabe_amd/librabe_hip.so.

hipcc cross-compiles without a GPU; the resulting .so travels to the GPU box with the snapshot.  The translation
units are compiled in parallel (objects under build/obj, git-ignored) and linked into one shared library.
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librabe_hip.so")
OBJ = os.path.join(os.path.dirname(HERE), "build", "obj")
SOURCES = [os.path.join(CSRC, "engine.hip"), os.path.join(CSRC, "engine_jobs.hip"), os.path.join(CSRC, "engine_coop.hip"), os.path.join(CSRC, "engine_coop_w1.hip"), os.path.join(CSRC, "engine_rr.hip"), os.path.join(CSRC, "engine_rr2.hip"), os.path.join(CSRC, "engine_sym.hip"),
           os.path.join(CSRC, "host", "schemes.cpp"),
           os.path.join(CSRC, "host", "host_abi.cpp"), os.path.join(CSRC, "host", "packed.cpp"),
           os.path.join(CSRC, "host", "pipeline.cpp"), os.path.join(CSRC, "host", "records.cpp")]


def _headers(src=None):
    """headers a translation unit can see: the device units (csrc/*.hip) include bn254/, engine_internal.h and rabe_hip.h; the host
    units (csrc/host/*.cpp) their own directory and both public headers -- a host-header edit does not recompile the kernels"""
    inc = os.path.join(os.path.dirname(HERE), "include")
    deps = [os.path.join(inc, "rabe_hip.h")]
    host_dir = os.path.join(CSRC, "host")
    is_host = src is None or os.path.dirname(src) == host_dir
    is_dev = src is None or not is_host
    if is_host:
        deps.append(os.path.join(inc, "rabe_host.h"))
    if src is not None and os.path.basename(src) == "engine_coop_w1.hip":          # that unit IS engine_coop.hip, compiled for one wave per SIMD
        deps.append(os.path.join(CSRC, "engine_coop.hip"))
    for root, _dirs, files in os.walk(CSRC):
        if (root == host_dir and not is_host) or (root != host_dir and not is_dev):
            continue
        deps += [os.path.join(root, f) for f in files if f.endswith(".h")]
    return deps


LIB_SAFE = os.path.join(HERE, "librabe_hip_safe.so")          # the RB_SAFE_CARRY build of the same sources (fp.h): an A/B reference
DEVICE_SOURCES = SOURCES[:5]          # (the reduced-radix units have no carry chains: nothing for RB_SAFE_CARRY to change)


def _obj(src, safe=False):
    return os.path.join(OBJ, os.path.basename(src) + (".safe.o" if safe else ".o"))


def _flags():
    extra = os.environ.get("RABE_HIPCC_FLAGS")          # kernel-tuning experiments, e.g. -DRB_MIN_WAVES=3
    return extra.split() if extra else []


# Per-unit code-generation flags, MEASURED (tools/ab_variants.sh: fourteen scheduler / register-allocator settings of engine_rr.hip alternating
# with the default build on one MI355X).  The reduced-radix pairing kernels are one wave per SIMD of straight-line multiply-add code around
# ~13 k calls per lane, at the register limit: a top-down pre-RA schedule and the reversed order of local assignments give k_miller_multi_rr
# 17.3 -> 17.0 ms and k_final_exp_rr 5.95 -> 5.8 ms per 65 536 items (config 4's launch set 170 -> 166 ms); the other units do not react.
# They are also what lets the unit-y prepared lines (bn254/pairing29.h) keep their gain: docs/rr29.md.
UNIT_FLAGS = {"engine_rr.hip": ["-mllvm", "-misched-prera-direction=topdown", "-mllvm", "-greedy-reverse-local-assignment"]}


def _unit_flags(src):
    return [] if os.environ.get("RABE_NO_UNIT_FLAGS") else UNIT_FLAGS.get(os.path.basename(src), [])


def _flags_tag():
    return " ".join(_flags()) + " | " + repr(sorted(UNIT_FLAGS.items())) + (" | no-unit-flags" if os.environ.get("RABE_NO_UNIT_FLAGS") else "")


def stale():
    if not os.path.exists(LIB):
        return True
    tag_file = os.path.join(OBJ, ".flags")
    if not os.path.exists(tag_file) or open(tag_file).read() != _flags_tag():
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _headers() + [s for s in SOURCES if os.path.exists(s)])


def declared_symbols():
    """the C API: every function include/rabe_hip.h (rhip_*) and include/rabe_host.h (rabe_*) declare"""
    import re
    inc = os.path.join(os.path.dirname(HERE), "include")
    out = set()
    for header, prefix in (("rabe_hip.h", "rhip_"), ("rabe_host.h", "rabe_")):
        text = open(os.path.join(inc, header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"^\s*#.*$", "", text, flags=re.M)          # macros (rabe_host_open) are not symbols
        out |= set(re.findall(r"\b(%s[a-z0-9_]+)\s*\(" % prefix, text))
    return sorted(out)


def _version_script():
    """build/obj/exports.map: the dynamic symbol table of the library is EXACTLY the two headers' function lists -- no std:: / rabe:: C++
    symbols for a Rust or C++ host to interpose (the C convention of the reference's own FFI, src/ffi/bsw.rs:22-163)."""
    path = os.path.join(OBJ, "exports.map")
    # diagnostic builds (RABE_HIPCC_FLAGS=-DRB_MILLER_PROF / -DRB_UBENCH_CORES: tools/prof_miller.sh, tools/ubench_cores.py) also export their
    # rhip_debug_* entry points; the product build has none
    debug = "    rhip_debug_*;\n" if _flags() else ""
    text = "RABE_AMD {\n  global:\n" + "".join("    %s;\n" % s for s in declared_symbols()) + debug + "  local:\n    *;\n};\n"
    if not os.path.exists(path) or open(path).read() != text:
        open(path, "w").write(text)
    return path


LINK_FLAGS = ["-static-libstdc++", "-static-libgcc"] if os.environ.get("RABE_STATIC_CXX") else []


def _link(out, objs, verbose):
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + _version_script(), "-o", out] + LINK_FLAGS + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, timeout=600)


def _compile(src, verbose, safe=False):
    cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-c", src, "-o", _obj(src, safe)] + _unit_flags(src) + _flags() + (["-DRB_SAFE_CARRY"] if safe else [])
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, timeout=3000)


def build(force=False, verbose=False):
    if not (force or stale()):
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(s)]
    hdr_t = {s: max(os.path.getmtime(h) for h in _headers(s)) for s in srcs}
    flags_tag = os.path.join(OBJ, ".flags")
    tag = _flags_tag()
    same_flags = os.path.exists(flags_tag) and open(flags_tag).read() == tag
    todo = [s for s in srcs if force or not same_flags or not os.path.exists(_obj(s))
            or os.path.getmtime(_obj(s)) < max(hdr_t[s], os.path.getmtime(s))]
    if todo:
        with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, min(len(todo), os.cpu_count() or 2))) as ex:
            for f in [ex.submit(_compile, s, verbose) for s in todo]:
                f.result()
    open(flags_tag, "w").write(tag)
    _link(LIB, [_obj(s) for s in srcs], verbose)
    return LIB


def build_safe(force=False, verbose=False):
    """librabe_hip_safe.so: the two device translation units compiled with -DRB_SAFE_CARRY (every carry dependency padded with the
    wait states LLVM's gfx940+ hazard table asks for, compiler-scheduled additive chains), linked with the host objects of the
    normal build.  tests/test_gpu_carry_interlock.py runs the same vectors through both libraries and requires identical bytes."""
    build(force=force, verbose=verbose)
    todo = [s for s in DEVICE_SOURCES if force or not os.path.exists(_obj(s, True))
            or os.path.getmtime(_obj(s, True)) < max(max(os.path.getmtime(h) for h in _headers(s)), os.path.getmtime(s))]
    if not todo and os.path.exists(LIB_SAFE) and os.path.getmtime(LIB_SAFE) >= os.path.getmtime(LIB):
        return LIB_SAFE
    if todo:
        with concurrent.futures.ThreadPoolExecutor(max_workers=len(todo)) as ex:
            for f in [ex.submit(_compile, s, verbose, True) for s in todo]:
                f.result()
    srcs = [s for s in SOURCES if os.path.exists(s)]
    _link(LIB_SAFE, [_obj(s, s in DEVICE_SOURCES) for s in srcs], verbose)
    return LIB_SAFE


def build_tools(verbose=False):
    """build/ubench_addc: the carry-chain equality check tests/test_gpu_carry_interlock.py runs on the GPU box"""
    src = os.path.join(os.path.dirname(HERE), "tools", "ubench_addc.hip")
    exe = os.path.join(os.path.dirname(HERE), "build", "ubench_addc")
    if os.path.exists(exe) and os.path.getmtime(exe) >= os.path.getmtime(src):
        return exe
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    cmd = ["hipcc", "-O3", "--offload-arch=gfx950", src, "-o", exe]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, timeout=600)
    return exe


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--safe" in sys.argv:
        print(build_safe(force="--force" in sys.argv, verbose=True))
        print(build_tools(verbose=True))
