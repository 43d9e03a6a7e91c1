"""Build the engine in-tree for gfx950:  python -m sextans_amd.build

  sextans_amd/lib/libsextans_amd.so   C-ABI shared library (include/sextans_amd.h)
  sextans_amd/bin/sextans             CLI with the reference's call surface

hipcc cross-compiles without a GPU.  Artefacts are git-ignored but travel to the GPU box.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sextans_amd", "csrc")
LIBDIR = os.path.join(ROOT, "sextans_amd", "lib")
BINDIR = os.path.join(ROOT, "sextans_amd", "bin")
LIB = os.path.join(LIBDIR, "libsextans_amd.so")
CLI = os.path.join(BINDIR, "sextans")

LIB_SOURCES = ["engine.hip", "synth.hip", "host_mtx.cpp", "panel_plan.cpp", "pack_api.cpp",
               "edge_stream.cpp"]
HEADERS = ["spmm_csr_kernels.h", "bell_kernels.h", "chan_kernels.h", "panel_plan.h", os.path.join("..", "..", "include", "sextans_amd.h")]

# -ffp-contract=off: the EXACT kernels and the CLI golden need "multiply, round, add" (the
# reference's arithmetic, sparse_helper.h:283); hipcc's default is to contract into FMA.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-pthread",
         "-Wall", "-Wno-unused-result", "-I", os.path.join(ROOT, "include"), "-I", CSRC]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the engine cannot be built (no CPU fallback exists)")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(BINDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in LIB_SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    if force or _stale(LIB, deps):
        cmd = [hipcc()] + FLAGS + ["-shared", "-o", LIB] + srcs
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    cli_src = os.path.join(CSRC, "cli_main.cpp")
    if force or _stale(CLI, [cli_src, LIB]):
        cmd = [hipcc()] + FLAGS + ["-o", CLI, cli_src, "-L", LIBDIR, "-lsextans_amd",
                                   "-Wl,-rpath,$ORIGIN/../lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB, CLI


if __name__ == "__main__":
    lib, cli = build(force="--force" in sys.argv, verbose=True)
    print("built", lib)
    print("built", cli)
