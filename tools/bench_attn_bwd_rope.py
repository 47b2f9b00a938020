"""Attention backward + RoPE backward: two launches (lmod_attn_bwd, lmod_rope) against the fused epilogue (lmod_attn_bwd_rope)."""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K
BF = torch.bfloat16
B, S, nh, hd = 16, 2048, 16, 128
qkv = torch.randn(B * S, 3 * nh * hd, device="cuda").to(BF)
q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:2 * nh * hd], qkv[:, 2 * nh * hd:]
sc = 1 / math.sqrt(hd)
o, lse = K.attn_fwd(q, k, v, B, S, nh, nh, hd, sc, True)
do = torch.randn(B * S, nh * hd, device="cuda").to(BF)
d = torch.empty_like(qkv)
inv = 1.0 / (1e6 ** (torch.arange(0, hd, 2).float() / hd)); fr = torch.outer(torch.arange(4096).float(), inv)
emb = torch.cat((fr, fr), -1); cos, sin = emb.cos().to(BF).cuda(), emb.sin().to(BF).cuda()
pos = (torch.arange(B * S, device="cuda") % S).to(torch.int32)
def plain(): K.attn_bwd(q, k, v, o, do, lse, d[:, :nh * hd], d[:, nh * hd:2 * nh * hd], d[:, 2 * nh * hd:], B, S, nh, nh, hd, sc, True)
def two():
    plain(); K.rope_(d, cos, sin, pos, 2 * nh, hd, backward=True)
def fused(): K.attn_bwd(q, k, v, o, do, lse, d[:, :nh * hd], d[:, nh * hd:2 * nh * hd], d[:, 2 * nh * hd:], B, S, nh, nh, hd, sc, True, rope=(cos, sin, pos))
arms = [("bwd alone", plain), ("bwd + rope kernel", two)]
if "lmod_attn_bwd_rope" in __import__("llavamod._hip", fromlist=["x"]).SIGNATURES and "bwdhead" not in os.environ.get("LMOD_HIP_LIB", ""):
    arms.append(("bwd with fused rope", fused))       # (an older build loaded through LMOD_HIP_LIB has no such entry point)
for rnd in range(2):
    for name, f in arms:
        for _ in range(5): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f()
        e1.record(); torch.cuda.synchronize()
        print(f"{os.path.basename(os.environ.get('LMOD_HIP_LIB', 'current'))}: {name}: {e0.elapsed_time(e1) / 50:.4f} ms", flush=True)
