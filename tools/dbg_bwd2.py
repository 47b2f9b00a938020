"""Debug helper: hd-128 attention backward vs an fp32 torch reference, error maps per output tensor."""
import math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K
BF = torch.bfloat16

def run(B, S, nh, nkv, hd, causal):
    torch.manual_seed(S)
    ld = (nh + 2 * nkv) * hd
    qkv = torch.randn(B * S, ld, device="cuda").to(BF)
    q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
    sc = 1 / math.sqrt(hd)
    o, lse = K.attn_fwd(q, k, v, B, S, nh, nkv, hd, sc, causal, None)
    do = torch.randn(B * S, nh * hd, device="cuda").to(BF)
    dqkv = torch.zeros_like(qkv)
    K.attn_bwd(q, k, v, o, do, lse, dqkv[:, :nh * hd], dqkv[:, nh * hd:(nh + nkv) * hd], dqkv[:, (nh + nkv) * hd:], B, S, nh, nkv, hd, sc, causal, None)
    qf = q.float().reshape(B, S, nh, hd).requires_grad_(True)
    kf = k.float().reshape(B, S, nkv, hd).requires_grad_(True)
    vf = v.float().reshape(B, S, nkv, hd).requires_grad_(True)
    rep = nh // nkv
    s = torch.einsum("bqhd,bkhd->bhqk", qf, kf.repeat_interleave(rep, 2)) * sc
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool, device="cuda"), 1), float("-inf"))
    oo = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), vf.repeat_interleave(rep, 2))
    oo.backward(do.float().reshape(B, S, nh, hd))
    for name, got, ref, n in (("dQ", dqkv[:, :nh * hd], qf.grad, nh), ("dK", dqkv[:, nh * hd:(nh + nkv) * hd], kf.grad, nkv),
                              ("dV", dqkv[:, (nh + nkv) * hd:], vf.grad, nkv)):
        g = got.float().reshape(B, S, n, hd)
        err = (g - ref).abs()
        bad = err > 0.02 * ref.abs().max() + 0.05 * ref.abs()
        print(f"{name} B{B} S{S} causal={causal}: max err {err.max().item():.4g} ref max {ref.abs().max().item():.4g} bad {int(bad.sum())}/{bad.numel()}")
        if bad.any():
            rows = bad.any(-1).any(-1).any(0).nonzero().flatten().tolist()
            cols = bad.any(0).any(0).any(0).nonzero().flatten().tolist()
            print("   bad rows:", rows[:40], "..." if len(rows) > 40 else "", "n=", len(rows))
            print("   bad feats:", cols[:40], "n=", len(cols))
            i = bad.nonzero()[0].tolist()
            print("   first bad", i, g[tuple(i)].item(), ref[tuple(i)].item())
            r0 = rows[0]
            print("   row", r0, "got", g[0, r0, 0, :8].tolist(), "ref", ref[0, r0, 0, :8].tolist())

run(1, 32, 1, 1, 128, False)
run(1, 64, 1, 1, 128, False)
run(1, 256, 1, 1, 128, False)
run(1, 256, 1, 1, 128, True)
run(2, 512, 2, 2, 128, True)
