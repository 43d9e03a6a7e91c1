"""Registers / scratch / occupancy of every kernel of one translation unit, from hipcc's own resource-usage remarks
(the same figures the code-object notes carry):

    python tools/kernel_resources.py sextans_amd/csrc/engine.hip [--spills] [--grep panel_v2] > profiles/r05_kernel_resources_engine.txt

Compiles for gfx950 with the flags of sextans_amd/build.py (no GPU needed)."""
import re
import subprocess
import sys

sys.path.insert(0, ".")
from sextans_amd import build as b

src = sys.argv[1]
only_spills = "--spills" in sys.argv
pat = sys.argv[sys.argv.index("--grep") + 1] if "--grep" in sys.argv else None
cmd = [b.hipcc()] + b.FLAGS + ["-c", "-o", "/dev/null", src, "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
names = re.findall(r"remark: Function Name: (\S+)", err)
if not names:   # the translation unit did not compile
    sys.exit("no kernels reported -- compiler output:\n" + err[-3000:])
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
blocks = re.split(r"remark: Function Name: \S+", err)[1:]
print(f"# {src}: {len(names)} kernels; columns: VGPRs AGPRs scratch[B/lane] spilledVGPRs waves/SIMD LDS[B]")
for name, blk in zip(dem, blocks):
    g = lambda k: int(re.search(k + r": (\d+)", blk).group(1))
    row = (g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g("VGPRs Spill"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]"))
    if only_spills and row[2] == 0 and row[3] == 0:
        continue
    name = re.sub(r"^void ", "", name).split("(")[0]
    if pat and pat not in name:
        continue
    print("%-120s %4d %3d %4d %3d %2d %6d" % ((name[:120],) + row))
