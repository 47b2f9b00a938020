"""Vendor F.linear on the fused [gate; up] weight vs our fused SwiGLU GEMM (same flops, same tiles) for rocprofv3 --pmc."""
import os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K
M, I, Kd = 32768, 11008, 4096
g = torch.Generator(device="cuda"); g.manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
x, w = rnd(M, Kd), rnd(2 * I, Kd)
with torch.no_grad():
    for _ in range(4): F.linear(x, w)
    torch.cuda.synchronize()
    for _ in range(4): K.gemm_swiglu(x, w)
    torch.cuda.synchronize()
