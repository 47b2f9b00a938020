"""One shape, vendor F.linear and ours, a few launches each: run under rocprofv3 --kernel-trace --pmc FETCH_SIZE (or TCC counters)."""
import os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K
M, N, Kd = [int(v) for v in os.environ.get("SHAPE", "32768,12288,4096").split(",")]
g = torch.Generator(device="cuda"); g.manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
x, w = rnd(M, Kd), rnd(N, Kd)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
with torch.no_grad():
    for _ in range(4):
        F.linear(x, w)
    torch.cuda.synchronize()
    for _ in range(4):
        K.gemm_nt(x, w, out=out)
    torch.cuda.synchronize()
