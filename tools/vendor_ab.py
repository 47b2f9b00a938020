"""The reference's operator path on this GPU (PyTorch-ROCm: F.linear -> hipBLASLt / rocBLAS, F.scaled_dot_product_attention, eager
SwiGLU and RMSNorm as qwen2/modeling_qwen2.py writes them) against this repo's kernels on the step's shapes, in ONE process, arms
alternating.  A measurement tool: nothing in the product path calls a vendor library.

    python tools/vendor_ab.py [--rounds 3] [--iters 20]
"""
import argparse, json, math, os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
dev, bf = "cuda", torch.bfloat16
g = torch.Generator(device=dev); g.manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).to(bf)


def timed(fn):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(args.iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / args.iters


def ab(tag, flop, ours, ref, note=""):
    to, tr = [], []
    for _ in range(args.rounds):
        to.append(timed(ours)); tr.append(timed(ref))
    o, r = min(to), min(tr)
    rec = {"case": tag, "ms_ours": round(o, 4), "ms_pytorch_rocm": round(r, 4), "ours_over_ref": round(r / o, 3)}
    if flop:
        rec.update(tf_ours=round(flop / o / 1e9, 1), tf_pytorch_rocm=round(flop / r / 1e9, 1))
    if note:
        rec["note"] = note
    print(json.dumps(rec), flush=True)


with torch.no_grad():
    for tag, M, N, Kd in [("teacher QKV", 32768, 12288, 4096), ("teacher down", 32768, 4096, 11008), ("student QKV", 32768, 6144, 2048),
                          ("student down", 32768, 2048, 5504), ("student lm_head (loss rows)", 8208, 151936, 2048)]:
        x, w = rnd(M, Kd), rnd(N, Kd)
        out = torch.empty(M, N, device=dev, dtype=bf)
        ab(f"linear {tag} [{M} x {N} x {Kd}]", 2.0 * M * N * Kd, lambda: K.gemm_nt(x, w, out=out), lambda: F.linear(x, w))
        del x, w, out
    for tag, M, I, Kd in [("teacher", 32768, 11008, 4096), ("student", 32768, 5504, 2048)]:
        x, wgu = rnd(M, Kd), rnd(2 * I, Kd)
        wg, wu = wgu[:I], wgu[I:]
        ab(f"SwiGLU input half {tag}: silu(x Wg^T) * (x Wu^T) [{M} x 2*{I} x {Kd}]", 4.0 * M * I * Kd,
           lambda: K.gemm_swiglu(x, wgu), lambda: F.silu(F.linear(x, wg)) * F.linear(x, wu),
           "ours: one GEMM with the activation in its epilogue; reference: two GEMMs + two elementwise kernels (modeling_qwen2.py:186)")
        del x, wgu
    # attention: the reference calls SDPA on [B, nh, S, hd] tensors
    for B, S, nh in [(16, 2048, 16), (16, 2048, 32)]:
        hd = 128
        qkv = rnd(B * S, 3 * nh * hd)
        q2, k2, v2 = qkv[:, :nh * hd], qkv[:, nh * hd:2 * nh * hd], qkv[:, 2 * nh * hd:]
        q4 = qkv.view(B, S, 3, nh, hd).permute(2, 0, 3, 1, 4)
        qs, ks, vs = q4[0].contiguous(), q4[1].contiguous(), q4[2].contiguous()
        fl = 4.0 * B * nh * S * S * hd * 0.5
        sc = 1 / math.sqrt(hd)
        ab(f"attention forward causal B{B} S{S} nh{nh} hd{hd}", fl, lambda: K.attn_fwd(q2, k2, v2, B, S, nh, nh, hd, sc, True),
           lambda: F.scaled_dot_product_attention(qs, ks, vs, is_causal=True))
    del qkv, qs, ks, vs
B, S, nh, hd = 16, 2048, 16, 128
qkv = rnd(B * S, 3 * nh * hd)
q2, k2, v2 = qkv[:, :nh * hd], qkv[:, nh * hd:2 * nh * hd], qkv[:, 2 * nh * hd:]
sc = 1 / math.sqrt(hd)
o, lse = K.attn_fwd(q2, k2, v2, B, S, nh, nh, hd, sc, True)
do = rnd(B * S, nh * hd)
dqkv = torch.empty_like(qkv)
q4 = qkv.view(B, S, 3, nh, hd).permute(2, 0, 3, 1, 4)
qs, ks, vs = (q4[i].contiguous().requires_grad_(True) for i in range(3))
do4 = do.view(B, S, nh, hd).permute(0, 2, 1, 3).contiguous()


def ref_fb():
    out = F.scaled_dot_product_attention(qs, ks, vs, is_causal=True)
    out.backward(do4)
    qs.grad = ks.grad = vs.grad = None


def ours_fb():
    oo, ll = K.attn_fwd(q2, k2, v2, B, S, nh, nh, hd, sc, True)
    K.attn_bwd(q2, k2, v2, oo, do, ll, dqkv[:, :nh * hd], dqkv[:, nh * hd:2 * nh * hd], dqkv[:, 2 * nh * hd:], B, S, nh, nh, hd, sc, True)


ab(f"attention forward + backward causal B{B} S{S} nh{nh} hd{hd}", 3.5 * 4.0 * B * nh * S * S * hd * 0.5, ours_fb, ref_fb,
   "flops: forward + 2.5x forward")
with torch.no_grad():
    T, H = 32768, 4096
    x, w = rnd(T, H), rnd(H)

    def ref_norm():                                  # Qwen2RMSNorm.forward (modeling_qwen2.py:92-97), eager
        h = x.to(torch.float32)
        v = h.pow(2).mean(-1, keepdim=True)
        return w * (h * torch.rsqrt(v + 1e-6)).to(bf)
    ab(f"RMSNorm [{T} x {H}]", 0, lambda: K.rmsnorm_fwd(x, w, 1e-6), ref_norm)
