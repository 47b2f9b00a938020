"""Debug aid for the dS-spill attention backward: compares the spilled dS^T workspace (layout of attn_bwd2.hip) and the dQ GEMM's
result with fp32 torch, block by block.  usage: python tools/dbg_ds.py B S nh nkv causal"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K  # noqa: E402

B, S, nh, nkv, causal = (int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (2, 768, 2, 2, 0)))
hd = 128
torch.manual_seed(0)
ld = (nh + 2 * nkv) * hd
qkv = (torch.randn(B * S, ld, device="cuda")).to(torch.bfloat16)
q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
sc = 1 / math.sqrt(hd)
o, lse = K.attn_fwd(q, k, v, B, S, nh, nkv, hd, sc, bool(causal))
do = torch.randn(B * S, nh * hd, device="cuda").to(torch.bfloat16)
ws = K._ds_workspace(torch.device("cuda", 0), B * nh * S * S * 2)
ws.fill_(0xFF)
out = torch.zeros_like(qkv)
K.attn_bwd(q, k, v, o, do, lse, out[:, :nh * hd], out[:, nh * hd:(nh + nkv) * hd], out[:, (nh + nkv) * hd:], B, S, nh, nkv, hd, sc, bool(causal),
           split=False)
torch.cuda.synchronize()
w = ws[:B * nh * S * S * 2].view(torch.bfloat16).view(B * nh, S // 32, S // 32, 2, 32, 16)      # [bh][q tile][key strip][half][key][16 q]
dst = w.permute(0, 2, 4, 1, 3, 5).reshape(B * nh, S, S).float()                                 # [bh][key][query]
rep = nh // nkv
bad = 0
for b in range(B):
    for h in range(nh):
        qf = q.float().reshape(B, S, nh, hd)[b, :, h]
        kf = k.float().reshape(B, S, nkv, hd)[b, :, h // rep]
        vf = v.float().reshape(B, S, nkv, hd)[b, :, h // rep]
        dof = do.float().reshape(B, S, nh, hd)[b, :, h]
        of = o.float().reshape(B, S, nh, hd)[b, :, h]
        s = qf @ kf.T * sc
        if causal:
            s = s.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool, device="cuda"), 1), float("-inf"))
        p = torch.exp(s - lse[b, h][:, None])
        dp = dof @ vf.T
        delta = (dof * of).sum(-1)
        ds = p * (dp - delta[:, None])                       # [query][key]
        got = dst[b * nh + h].T                              # [query][key]
        fin = torch.isfinite(got)
        err = (torch.nan_to_num(got) - ds).abs()
        tol = 2 ** -6 * ds.abs().max()
        blk = err.reshape(S // 256, 256, S // 256, 256).amax(dim=(1, 3))
        nanb = (~fin).reshape(S // 256, 256, S // 256, 256).any(dim=3).any(dim=1)
        dq_ref = (ds.to(torch.bfloat16).float() @ kf) * sc
        dq_got = out[:, :nh * hd].float().reshape(B, S, nh, hd)[b, :, h]
        dqe = (dq_got - dq_ref).abs().reshape(S // 256, 256, hd).amax(dim=(1, 2))
        print(f"b{b} h{h}: dS err per (query block, key block) max {blk.max().item():.4f} (tol {tol.item():.4f}); unwritten blocks {nanb.nonzero().tolist()}; "
              f"dQ err per query block {[round(x, 4) for x in dqe.tolist()]} (|dQ| max {dq_ref.abs().max().item():.3f})")
        if blk.max() > tol:
            bad += 1
            if bad == 1:
                e32 = (err > tol).reshape(S // 32, 32, S // 32, 2, 16)          # [q tile][q in tile][key strip][half... no: keys
                em = (err > tol)                                                 # [query][key]
                print("   wrong elements:", int(em.sum()), "of", em.numel())
                # by (query tile, key strip, half = (query % 32) // 16)
                t = em.reshape(S // 32, 2, 16, S // 32, 32).permute(0, 3, 1, 2, 4).reshape(S // 32, S // 32, 2, 16 * 32).sum(-1)
                nz = t.nonzero()
                print("   (query tile, key strip, half) blocks with errors:", nz[:40].tolist(), "count", len(nz))
                if len(nz):
                    qt, ks, hf = nz[0].tolist()
                    sub = em[qt * 32 + hf * 16: qt * 32 + hf * 16 + 16, ks * 32: ks * 32 + 32]
                    print("   first bad block [16 queries x 32 keys]:\n", sub.int())
                    g = got[qt * 32 + hf * 16: qt * 32 + hf * 16 + 16, ks * 32: ks * 32 + 32]
                    r = ds[qt * 32 + hf * 16: qt * 32 + hf * 16 + 16, ks * 32: ks * 32 + 32]
                    print("   got[:, key0]", g[:, 0].tolist(), "\n   ref[:, key0]", [round(x, 4) for x in r[:, 0].tolist()])
                    print("   got[q0, :8]", g[0, :8].tolist(), "\n   ref[q0, :8]", [round(x, 4) for x in r[0, :8].tolist()])
print("BAD" if bad else "OK")
