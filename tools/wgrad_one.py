"""A few launches of one weight-gradient form for counter passes (rocprofv3 --pmc ... -- python tools/wgrad_one.py [tn4|forms]).
tn4 (default): gemm4t_kernel on the MoE experts' gate+up shape — 4 experts x [11008 x 2048], 9000 / 12500 / 20036 / 24000 live rows of a
24576-row capacity slab (1376 tiles: 5.4 rounds of the CUs).  forms: the round-2 comparison (NT on copies, X k-major, TN) at the qkv shape."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K
if (sys.argv[1] if len(sys.argv) > 1 else "tn4") == "tn4":
    E, C, M, N = 4, 24576, 11008, 2048
    rows = torch.tensor([9000, 12500, 20036, 24000], device="cuda", dtype=torch.int32)
    dy = torch.randn(E, C, M, device="cuda").to(torch.bfloat16); x = torch.randn(E, C, N, device="cuda").to(torch.bfloat16)
    g = torch.zeros(E, M, N, device="cuda", dtype=torch.float32)
    for _ in range(4):
        K.gemm_tn(dy, x, out=g, accumulate=True, k_valid=rows)
else:
    T, M, N = 32768, 6144, 2048
    dy = torch.randn(T, M, device="cuda").to(torch.bfloat16); x = torch.randn(T, N, device="cuda").to(torch.bfloat16)
    g = torch.zeros(M, N, device="cuda", dtype=torch.float32)
    dyt, xt = K.transpose(dy), K.transpose(x)
    for _ in range(3):
        K.gemm_wgrad(dyt, xt, g)
        K.gemm_wgrad(dyt, x, g, b_kmajor=True)
        K.gemm_tn(dy, x, out=g, accumulate=True)
torch.cuda.synchronize()
