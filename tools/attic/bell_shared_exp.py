"""Block-banded blocked-ELL, N = 256: shared-tile kernel vs per-wavefront kernel."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sextans_amd import api
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
ALPHA, BETA = 0.85, -2.06
def run(M, hw, opts, iters=5):
    K, N, W = M, 256, 2 * hw + 1
    dc, dv = api.gen_bell_banded_device(0, M, K, hw, 5)
    e = api.Engine(0)
    for k, v in opts.items():
        e.set_option(k, v)
    e.set_matrix_bell_device(M, K, W, dc, dv)
    api.device_free(0, dv)
    B = torch.empty(K * N, dtype=torch.int16, device=dev)
    Cin = torch.empty(M * N, dtype=torch.float32, device=dev); Cout = torch.empty(M * N, dtype=torch.float32, device=dev)
    api.gen_uniform_bf16_device(0, B.data_ptr(), K * N, 51, st)
    api.gen_uniform_device(0, Cin.data_ptr(), M * N, 52, st)
    f = lambda: e.spmm_bell_device(N, ALPHA, B.data_ptr(), K, BETA, Cin.data_ptr(), Cout.data_ptr(), M, st)
    f(); torch.cuda.synchronize()
    e.set_option("profile", 1); e.profile_reset()
    for _ in range(iters):
        f()
    torch.cuda.synchronize()
    k_ns, _, rp_ns = e.profile_read()
    k_ns /= 1.0
    nb = (M // 32) * W
    flops = 2.0 * N * (1024.0 * nb + M)
    by = 2048 * nb + 4 * nb + 2 * K * N + 8 * M * N
    print(f"M={M} half_width={hw} W={W} blocks={nb} opts={opts} kernel={e.last_kernel()} share={e.get_stat('bell_share'):.2f} "
          f"kernel_ms={k_ns/1e6:.3f} TFLOPs={flops/(k_ns*1e-9)/1e12:.1f} frac_of_2.5PF={flops/(k_ns*1e-9)/2.5e15:.4f} "
          f"alg_GBs={by/(k_ns*1e-9)/1e9:.0f} frac_hbm={by/(k_ns*1e-9)/8e12:.4f}", flush=True)
    e.close(); api.device_free(0, dc)
    del B, Cin, Cout; torch.cuda.empty_cache()
print("# block-banded blocked-ELL bf16 (32x32 blocks, band of 2*half_width+1 block columns), M=K=1048576, N=256: union-walk shared-tile kernel")
print("# (bell_shared=1) against the per-wavefront kernel (bell_shared=0); profile events, mean of 5 launches")
for hw in (7, 15, 31, 63, 127):
    for shared in (1, 0):
        run(1 << 20, hw, {"bell_shared": shared})
print("# ablation of the shared kernel at half_width=127 (option bell_debug, RESULTS ARE WRONG, timing only): 1 = no global-memory requests in")
print("# the steady state (LDS + MFMA only), 2 = no MFMAs, 4 = no per-step barrier, 8 = no step loop at all (setup + epilogue), 16 = no epilogue")
for dbg in (0, 1, 2, 4, 8, 16, 3):
    run(1 << 20, 127, {"bell_shared": 1, "bell_debug": dbg})
