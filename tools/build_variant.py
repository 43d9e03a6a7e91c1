"""A variant build of the library for same-box A/B runs (tools/ab.py, tools/ab_libs_opts.py):
    python tools/build_variant.py <name> [-DMACRO=value ...]   ->  tools/bin/libsextans_<name>.so   (git-ignored, travels with gpurun)"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, ".")
from sextans_amd import build as b

name, defs = sys.argv[1], sys.argv[2:]
objdir = os.path.join(b.ROOT, "tools", "bin", "obj_" + name)
os.makedirs(objdir, exist_ok=True)
out = os.path.join(b.ROOT, "tools", "bin", f"libsextans_{name}.so")


def one(src):
    obj = os.path.join(objdir, src + ".o")
    subprocess.run([b.hipcc()] + b.FLAGS + defs + ["-c", "-o", obj, os.path.join(b.CSRC, src)], check=True)
    return obj


with ThreadPoolExecutor(len(b.LIB_SOURCES)) as ex:
    objs = list(ex.map(one, b.LIB_SOURCES))
subprocess.run([b.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", out] + objs, check=True)
print("built", out)
