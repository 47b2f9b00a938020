import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K
M, N, Kd = 32768, 12288, 4096   # teacher fused QKV projection at micro-batch 16
a = torch.randn(M, Kd, device="cuda").to(torch.bfloat16); b = torch.randn(N, Kd, device="cuda").to(torch.bfloat16)
o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(4):
    K.gemm_nt(a, b, out=o)
torch.cuda.synchronize()
