"""What would graph-compact row blocks buy?  The 4M-row FEM matrix with its ROWS permuted brick by brick (columns untouched: the
result is the same C with permuted rows), against the natural order.  tools/perm_exp.py [bx by bz]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from sextans_amd import api
nx = ny = nz = 110; dof = 3
bx, by, bz = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else (4, 4, 4)
M = K = nx * ny * nz * dof
p, i, v, nnz = api.gen_fem3d_device(0, nx, ny, nz, dof, 3)
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
def as_t(ptr, n, dt):
    import ctypes
    t = torch.empty(n, dtype=dt, device=dev)
    api.device_copy(0, t.data_ptr(), ptr, n * t.element_size()) if hasattr(api, "device_copy") else None
    return t
# wrap the generator's device arrays as tensors through the cuda array interface
class _W:
    def __init__(s, ptr, n, typestr): s.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}
rp = torch.as_tensor(_W(p, M + 1, "<i4"), device=dev).long()
ci = torch.as_tensor(_W(i, nnz, "<i4"), device=dev)
va = torch.as_tensor(_W(v, nnz, "<f4"), device=dev)
# brick order of the nodes, dofs of a node stay together
z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
key = ((z // bz) * ((ny + by - 1) // by) + (y // by)) * ((nx + bx - 1) // bx) + (x // bx)
inner = ((z % bz) * by + (y % by)) * bx + (x % bx)
order = np.lexsort((inner.ravel(), key.ravel()))           # node indices in brick order
perm = (order[:, None] * dof + np.arange(dof)[None, :]).ravel()
perm_t = torch.from_numpy(perm).to(dev)
lens = rp[1:] - rp[:-1]
nl = lens[perm_t]
nrp = torch.zeros(M + 1, dtype=torch.long, device=dev); nrp[1:] = torch.cumsum(nl, 0)
src0 = rp[:-1][perm_t]
idx = torch.arange(nnz, device=dev) - torch.repeat_interleave(nrp[:-1], nl) + torch.repeat_interleave(src0, nl)
nci = ci[idx].contiguous(); nva = va[idx].contiguous(); nrp32 = nrp.int().contiguous()
del idx
torch.cuda.synchronize()
def run(tag, prp, pci, pva):
    e = api.Engine(0); e.set_matrix_csr_device(M, K, nnz, prp, pci, pva)
    out = []
    for N in (16, 32, 128):
        B = torch.empty(K * N, device=dev); Cin = torch.empty(M * N, device=dev); Cout = torch.empty(M * N, device=dev)
        api.gen_uniform_device(0, B.data_ptr(), K * N, 41, st); api.gen_uniform_device(0, Cin.data_ptr(), M * N, 42, st)
        f = lambda: e.spmm_device(N, 0.85, B.data_ptr(), K, -2.06, Cin.data_ptr(), Cout.data_ptr(), M, st)
        for _ in range(3): f()
        e.set_option("profile", 1); e.profile_reset()
        for _ in range(20): f()
        torch.cuda.synchronize()
        k_ns, n, r_ns = e.profile_read(); e.set_option("profile", 0)
        out.append(f"N={N}: {k_ns / 1e3:.1f} us ({e.last_kernel()})")
        del B, Cin, Cout
    print(tag, " ".join(out), "panel blocks", int(e.get_stat("panel_blocks")), flush=True)
    e.close()
for rnd in range(2):
    run("natural order      ", p, i, v)
    run(f"bricks {bx}x{by}x{bz}      ", nrp32.data_ptr(), nci.data_ptr(), nva.data_ptr())
