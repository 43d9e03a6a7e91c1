import sys
sys.path.insert(0, ".")
import numpy as np
from sextans_amd import api, meshgen
rp, ci, v = api.gen_fem3d_host(24, 22, 20, 3, 7)
M = 24 * 22 * 20 * 3
rp, ci, v = meshgen.permute_symmetric(rp, ci, v, M, meshgen.node_permutation(M // 3, 3, 3))
rows = np.repeat(np.arange(M), np.diff(rp))
with open("gpurun_out/fem_random.mtx", "w") as f:
    f.write("%%%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (M, M, len(ci)))
    np.savetxt(f, np.c_[rows + 1, ci + 1, v], fmt="%d %d %.9g")
print(M, len(ci))
