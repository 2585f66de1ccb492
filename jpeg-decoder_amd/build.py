"""In-tree build of the native parts (no JIT cache: the built .so files travel with the tree).

  libzj.so            HIP kernels + C ABI + host parser, hipcc --offload-arch=gfx950
  zj_decode           command-line driver (counterpart of examples/decode.rs)
  synth/libzjsynth.so synthetic JPEG generator (gcc, CPU only)

-ffp-contract=off is part of the arithmetic contract: the IDCT's fused multiply-adds are explicit
fmaf calls (src/dct.rs:95) and nothing else may be fused (src/decoder.rs:271-273).
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libzj.so")
CLI = os.path.join(HERE, "zj_decode")
SYNTH = os.path.join(HERE, "synth", "libzjsynth.so")

HIP_SOURCES = ["zj_kernels.hip", "zj_api.hip", "zj_host.cpp", "zj_decoder.cpp"]
HEADERS = ["zj_core.h", "zj_dev.h", "zj_hostutil.h", "zj_decoder.hpp", os.path.join("..", "..", "include", "zj.h")]
HIPCC_FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-Wno-unused-value"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def hipcc():
    for p in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.isabs(p) and not os.path.exists(p):
            continue
        return p
    return "hipcc"


def build_all(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS]
    if force or _stale(LIB, deps):
        cmd = [hipcc()] + HIPCC_FLAGS + ["-shared"] + srcs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    cli_src = os.path.join(CSRC, "zj_decode_cli.cpp")
    if force or _stale(CLI, [cli_src, LIB]):
        cmd = [hipcc(), "-O2", "-std=c++17", cli_src, "-o", CLI, "-L" + HERE, "-lzj", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    synth_src = os.path.join(HERE, "synth", "zj_synth.c")
    if force or _stale(SYNTH, [synth_src]):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", SYNTH, synth_src, "-lm"])
    return LIB


if __name__ == "__main__":
    build_all(force=True, verbose=True)
