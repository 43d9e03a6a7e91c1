"""Build the engine in-tree for gfx950:  python -m sextans_amd.build

  sextans_amd/lib/libsextans_amd.so   C-ABI shared library (include/sextans_amd.h)
  sextans_amd/bin/sextans             CLI with the reference's call surface
  sextans_amd/bin/dist_spmm           examples/dist_spmm.cpp: multi-GPU SpMM through the C ABI alone (one thread per GPU)

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

LIB_SOURCES = ["engine.hip", "engine_plan.hip", "engine_bell.hip", "engine_dist.hip", "plan_device.hip", "row_cluster.hip", "graph_cluster.hip", "synth.hip", "host_mtx.cpp", "panel_plan.cpp", "window_plan.cpp", "pack_api.cpp",
               "edge_stream.cpp"]
# every header of csrc/ (a header that is not listed here would change without anything being recompiled -- round 6 measured two
# "new" kernels that were never built that way)
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "sextans_amd.h")]
OBJDIR = os.path.join(LIBDIR, "obj")

# -ffp-contract=off: the EXACT kernels and the CLI golden need "multiply, round, add" (the
# reference's arithmetic, sparse_helper.h:283); hipcc's default is to contract into FMA.
# -fgpu-default-stream=per-thread: everything the library does on "stream 0" -- the one-time plan builders (kernels, hipcub scans and
# sorts, synchronous copies) -- runs on the calling thread's own default stream instead of the process-wide LEGACY stream.  With the
# legacy stream, a plan build in one host thread made HIP fail a hipGraph capture that another thread's engine had open ("operation
# would make the legacy stream depend on a capturing blocking stream"): engines are re-entrant per handle only without it.
# Found by tests/test_concurrency_gpu.py in round 5 (tools/capture_race.py reproduces it on a build without the flag).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fgpu-default-stream=per-thread", "-fPIC", "-pthread",
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
    """One object per source (rebuilt only when it or a header changed, compiled in parallel), then link."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(BINDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]

    def compile_one(name):
        src = os.path.join(CSRC, name)
        obj = os.path.join(OBJDIR, name + ".o")
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc()] + FLAGS + ["-c", "-o", obj, src]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            return obj, True
        return obj, False

    with ThreadPoolExecutor(len(LIB_SOURCES)) as ex:
        res = list(ex.map(compile_one, LIB_SOURCES))
    objs = [o for o, _ in res]
    if force or any(ch for _, ch in res) or _stale(LIB, objs):
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    cli_src = os.path.join(CSRC, "cli_main.cpp")
    if force or _stale(CLI, [cli_src, LIB]):
        cmd = [hipcc()] + FLAGS + ["-o", CLI, cli_src, "-L", LIBDIR, "-lsextans_amd",
                                   "-Wl,-rpath,$ORIGIN/../lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    ex_src = os.path.join(ROOT, "examples", "dist_spmm.cpp")
    ex_bin = os.path.join(BINDIR, "dist_spmm")
    if os.path.exists(ex_src) and (force or _stale(ex_bin, [ex_src, LIB])):
        cmd = [hipcc()] + FLAGS + ["-o", ex_bin, ex_src, "-L", LIBDIR, "-lsextans_amd", "-Wl,-rpath,$ORIGIN/../lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB, CLI


if __name__ == "__main__":
    lib, cli = build(force="--force" in sys.argv, verbose=True)
    print("built", lib)
    print("built", cli)
