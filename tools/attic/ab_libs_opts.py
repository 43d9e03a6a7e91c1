"""tools/ab_libs_opts.py <spec> <N> <iters> <opts k=v,..> lib1.so lib2.so ...: tools/ab_opts.py (kernel / pre / post / wall us) once per library build, each in its own process, 2 rounds."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import sextans_amd.api as api
    api.LIB_PATH = os.path.join(ROOT, sys.argv[2])
    sys.argv = ["ab_opts"] + sys.argv[3:]
    exec(open(os.path.join(ROOT, "tools", "ab_opts.py")).read())
else:
    spec, N, iters, opts = sys.argv[1:5]
    for rnd in range(2):
        for lib in sys.argv[5:]:
            out = subprocess.run([sys.executable, __file__, "--child", lib, spec, N, iters, opts], capture_output=True, text=True).stdout
            print(lib.split("/")[-1], [l for l in out.splitlines() if "round 2" in l][-1:], flush=True)
