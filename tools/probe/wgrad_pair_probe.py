"""Would the student's qkv (192 tiles) and o (64 tiles) weight gradients fill the 256 CUs together?  Two streams, un-split TN launches."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K
T = 32768
dq = torch.randn(T, 6144, device="cuda").to(torch.bfloat16); x = torch.randn(T, 2048, device="cuda").to(torch.bfloat16)
do = torch.randn(T, 2048, device="cuda").to(torch.bfloat16); o = torch.randn(T, 2048, device="cuda").to(torch.bfloat16)
g1 = torch.zeros(6144, 2048, device="cuda"); g2 = torch.zeros(2048, 2048, device="cuda")
s2 = torch.cuda.Stream()
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def seq():
    K.gemm_tn(dq, x, out=g1, accumulate=True); K.gemm_wgrad(do, o, g2, a_kmajor=True)
def both():
    s2.wait_stream(torch.cuda.current_stream())
    K.gemm_tn(dq, x, out=g1, accumulate=True)
    with torch.cuda.stream(s2):
        K.gemm_tn(do, o, out=g2, accumulate=True)
    torch.cuda.current_stream().wait_stream(s2)
print({"qkv alone": t(lambda: K.gemm_tn(dq, x, out=g1, accumulate=True)), "o alone (split heuristic)": t(lambda: K.gemm_wgrad(do, o, g2, a_kmajor=True)),
       "o alone un-split": t(lambda: K.gemm_tn(do, o, out=g2, accumulate=True)), "sequential": t(seq), "two streams": t(both)})
