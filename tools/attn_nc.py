import math, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llava-mod_amd"))
from llavamod import kernels as K
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for B, S, nh in [(16, 2048, 16), (16, 2048, 32), (8, 4096, 32), (4, 8192, 32)]:
    hd = 128
    qkv = torch.randn(B * S, 3 * nh * hd, device="cuda").to(torch.bfloat16)
    q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:2 * nh * hd], qkv[:, 2 * nh * hd:]
    for causal in (False, True):
        fl = 4.0 * B * nh * S * S * hd * (0.5 if causal else 1.0)
        s = t(lambda: K.attn_fwd(q, k, v, B, S, nh, nh, hd, 1 / math.sqrt(hd), causal))
        print(f"B{B} S{S} nh{nh} causal={causal}: {s * 1e3:.3f} ms {fl / s / 1e12:.0f} TF", flush=True)
