"""Functional (non-autograd) wrappers over the C-ABI kernels: allocate outputs, pass raw pointers.

Every function here launches hand-written HIP kernels through `_hip.call`; there is no torch
compute in this file (torch only allocates device memory).  Tensors are bf16 unless noted.
"""
import os

import torch

from . import _hip
from ._hip import call, ptr

BF16 = torch.bfloat16


def _chk(t, dtype=None):
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensor required"
    if dtype is not None:
        assert t.dtype == dtype, f"expected {dtype}, got {t.dtype}"
    return t


# ------------------------------------------------------------------------------------------ GEMM
def gemm_nt(a, b, bias=None, *, out=None, act=0, out_f32=False, accumulate=False, m_valid=None, k_valid=None,
            M=None, N=None, K=None, lda=None, ldb=None, ldc=None, batch=1, strides=(0, 0, 0)):
    """out[M,N] (+)= act(a[M,K] @ b[N,K]^T + bias).  2-D tensors by default; explicit dims/strides for
    sub-matrix and batched (grouped) use."""
    if M is None:
        M, K = a.shape[-2], a.shape[-1]
        N = b.shape[-2]
        lda, ldb = a.stride(-2), b.stride(-2)
        if a.dim() == 3:
            batch = a.shape[0]
            strides = (a.stride(0), b.stride(0) if b.dim() == 3 else 0, None)
    if out is None:
        shape = (batch, M, N) if (a.dim() == 3) else (M, N)
        out = torch.empty(shape, device=a.device, dtype=torch.float32 if out_f32 else BF16)
    if ldc is None:
        ldc = out.stride(-2)
    sA, sB, sC = strides
    if sC is None:
        sC = out.stride(0) if out.dim() == 3 else 0
    call("lmod_gemm_bf16_nt", ptr(a), ptr(b), ptr(out), ptr(bias), M, N, K, lda, ldb, ldc, batch, sA, sB, sC,
         ptr(m_valid), ptr(k_valid), act, int(out_f32), int(accumulate))
    return out


def gemm_res_fusable(M, w, res):
    """Can `gemm_nt_res` take the projection of a contiguous [M, K] operand through w [N, K] with residual res [M, N]?  Mirrors
    lmod_gemm_bf16_nt's choice of the 4-wave 256-tile kernel (gemm.hip) — the only kernel with the residual epilogue;
    LMOD_GEMM_RES=0 switches the fusion off (A/B runs)."""
    if os.environ.get("LMOD_GEMM_RES", "1") == "0" or os.environ.get("LMOD_GEMM_WAVES", "0") not in ("", "0"):
        return False
    if res is None or res.dim() != 2 or res.stride(1) != 1 or w.dim() != 2 or w.stride(1) != 1:
        return False
    N, Kd = w.shape
    if tuple(res.shape) != (M, N) or N % 8 or Kd % 8 or w.stride(0) % 8 or res.stride(0) % 8 or res.data_ptr() % 16:
        return False
    t256 = ((M + 255) // 256) * ((N + 255) // 256)
    big = M >= 512 and N >= 256 and (t256 >= 160 or (t256 >= 96 and Kd >= 8192)) and \
        ((M + 255) // 256 * 256) * ((N + 255) // 256 * 256) <= M * N * 115 // 100 + 65536
    return big and 255 * Kd * 2 + Kd * 2 < 0x7fffffff and 255 * w.stride(0) * 2 + Kd * 2 < 0x7fffffff


def gemm_nt_res(x, w, res, out=None):
    """out[M, N] = bf16(res + bf16(x @ w^T)): a bias-free projection with the residual add in the GEMM epilogue (bit-identical to
    gemm_nt followed by the residual path of rmsnorm_fwd)."""
    M, Kd = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=BF16)
    rc = call("lmod_gemm_bf16_nt_res", ptr(x), ptr(w), ptr(out), None, ptr(res), M, N, Kd, x.stride(0), w.stride(0), out.stride(0),
              res.stride(0), allow=(_hip.UNSUPPORTED,))
    if rc == _hip.UNSUPPORTED:
        # a shape (or a build: G4_ASM=0, LMOD_GEMM_WAVES) the residual-epilogue kernel does not take, although `gemm_res_fusable`
        # — a mirror of the library's test — said yes: the header's two-step form, same roundings (bf16(res + bf16(acc)))
        gemm_nt(x, w, out=out)
        add(out, res if res.is_contiguous() else res.contiguous(), out=out)
    return out


def qkv_rope_fusable(x, w, nrot_heads, hd, out=None):
    """Can `gemm_qkv_rope` take this projection?  (head dim 128; an even number of rotated heads so that the q|k block ends on
    a 256-column tile boundary; enough rows for the 256-tile kernel to be the right choice.)"""
    return hd == 128 and nrot_heads % 2 == 0 and w.shape[0] % 16 == 0 and x.shape[0] >= 512


def gemm_qkv_rope(x, w, bias, cos_t, sin_t, pos, nrot_heads, out=None):
    """out[T, N] = rope(x @ w^T + bias) on the first nrot_heads heads of 128 (q and k), v columns untouched: the fused
    QKV projection with the rotary embedding in the GEMM epilogue (bit-identical to gemm_nt + rope_)."""
    T, Kd = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((T, N), device=x.device, dtype=BF16)
    call("lmod_gemm_qkv_rope_bf16", ptr(x), ptr(w), ptr(out), ptr(bias), T, N, Kd, x.stride(0), w.stride(0), out.stride(0),
         ptr(cos_t), ptr(sin_t), ptr(pos), nrot_heads * 128)
    return out


def gemm_swiglu(x, w_gu, act=None, gu=None, want_gu=False, m_valid=None):
    """act[.., M, I] = silu(x @ Wg^T) * (x @ Wu^T) for the fused gate-over-up weight w_gu [.., 2I, K] in ONE GEMM launch
    (SwiGLU in the epilogue).  x: [M, K] or [E, M, K] (grouped, weights [E, 2I, K] or shared [2I, K]).
    gu (or want_gu): also store the [.., M, 2I] pre-activations for the backward.  Returns (act, gu-or-None)."""
    M, Kd = x.shape[-2], x.shape[-1]
    I = w_gu.shape[-2] // 2
    batch = x.shape[0] if x.dim() == 3 else 1
    lead = (batch,) if x.dim() == 3 else ()
    if act is None:
        act = torch.empty(lead + (M, I), device=x.device, dtype=BF16)
    if gu is None and want_gu:
        gu = torch.empty(lead + (M, 2 * I), device=x.device, dtype=BF16)
    call("lmod_gemm_swiglu_bf16", ptr(x), ptr(w_gu), ptr(act), ptr(gu), M, I, Kd, x.stride(-2), w_gu.stride(-2),
         act.stride(-2), gu.stride(-2) if gu is not None else 0, batch,
         x.stride(0) if x.dim() == 3 else 0, w_gu.stride(0) if w_gu.dim() == 3 else 0,
         act.stride(0) if act.dim() == 3 else 0, (gu.stride(0) if gu.dim() == 3 else 0) if gu is not None else 0,
         ptr(m_valid))
    return act, gu


def gemm_swiglu_bwd(dy, wt_down, gu, out=None, m_valid=None, K=None):
    """dgu[.., M, 2I] = SwiGLU backward of dact = dy[.., M, H] @ wt_down[.., I, Hpad]^T against the saved gu = [gate | up]:
    the down-projection dgrad GEMM with swiglu_bwd in its epilogue.  out may be gu itself (in place)."""
    M, Kd = dy.shape[-2], (K if K is not None else dy.shape[-1])
    I = wt_down.shape[-2]
    batch = dy.shape[0] if dy.dim() == 3 else 1
    if out is None:
        out = torch.empty_like(gu)
    call("lmod_gemm_swiglu_bwd_bf16", ptr(dy), ptr(wt_down), ptr(gu), ptr(out), M, I, Kd, dy.stride(-2), wt_down.stride(-2),
         gu.stride(-2), out.stride(-2), batch, dy.stride(0) if dy.dim() == 3 else 0,
         wt_down.stride(0) if wt_down.dim() == 3 else 0, gu.stride(0) if gu.dim() == 3 else 0,
         out.stride(0) if out.dim() == 3 else 0, ptr(m_valid))
    return out


_WGRAD_WS = {}
WGRAD_WS_BYTES = 512 << 20


def gemm_wgrad(at, b, out, b_kmajor=False, a_kmajor=False):
    """out (fp32 [M, N]) += at[M, K] @ X with X = b[N, K]^T (b_kmajor False) or b[K, N] as autograd holds the layer
    input (b_kmajor True: no transposed copy of it).  a_kmajor: `at` is dY [K, M] as autograd holds it too (dW = dY^T X with no
    transposed copy of either operand: the 4-wave TN kernel).  Deterministic split-K when the output has too few tiles to fill
    the GPU (K = tokens).  One zero-initialised workspace per device, reused by every call on the compute stream."""
    key = at.device.index
    ws = _WGRAD_WS.get(key)
    if ws is None:
        ws = _WGRAD_WS[key] = torch.zeros(WGRAD_WS_BYTES, device=at.device, dtype=torch.uint8)
    if a_kmajor:
        Kd, M = at.shape
        Kd = min(Kd, b.shape[0])
        N = b.shape[1]
        mode = 2
    else:
        M, Kd = at.shape
        if b_kmajor:
            Kd = min(Kd, b.shape[0])              # at is zero-padded to a multiple of 8 tokens by the transpose
        N = b.shape[1] if b_kmajor else b.shape[0]
        mode = int(b_kmajor)
    call("lmod_gemm_wgrad_bf16_nt", ptr(at), ptr(b), ptr(out), M, N, Kd, at.stride(0), b.stride(0),
         out.stride(0), mode, ptr(ws), ws.numel())
    return out


def gemm_tn(a, b, out=None, out_f32=True, accumulate=False, k_valid=None, K=None):
    """out[.., M, N] (+)= a^T @ b with a [.., Kd, M], b [.., Kd, N] (reduction-major: dW = dY^T X without transposes).
    K: reduce over the first K rows only."""
    Kd, M = a.shape[-2], a.shape[-1]
    N = b.shape[-1]
    if K is not None:
        Kd = K
    batch = a.shape[0] if a.dim() == 3 else 1
    if out is None:
        out = torch.empty(((batch,) if a.dim() == 3 else ()) + (M, N), device=a.device,
                          dtype=torch.float32 if out_f32 else BF16)
    call("lmod_gemm_bf16_tn", ptr(a), ptr(b), ptr(out), M, N, Kd, a.stride(-2), b.stride(-2), out.stride(-2), batch,
         a.stride(0) if a.dim() == 3 else 0, b.stride(0) if b.dim() == 3 else 0, out.stride(0) if out.dim() == 3 else 0,
         ptr(k_valid), int(out_f32), int(accumulate))
    return out


def transpose(x, ld_out=None, out=None, r_valid=None):
    """[.., R, C] -> [.., C, roundup(R,8)] (zero padded).  r_valid (int32 [batch]): live rows per batch entry; columns past
    roundup(r_valid, 8) of the result are left untouched (a k_valid GEMM never reads them)."""
    R, C = x.shape[-2], x.shape[-1]
    batch = x.shape[0] if x.dim() == 3 else 1
    if ld_out is None:
        ld_out = (R + 7) // 8 * 8
    if out is None:
        shape = (batch, C, ld_out) if x.dim() == 3 else (C, ld_out)
        out = torch.empty(shape, device=x.device, dtype=BF16)
    call("lmod_transpose_bf16", ptr(x), ptr(out), R, C, x.stride(-2), ld_out, batch,
         x.stride(0) if x.dim() == 3 else 0, out.stride(0) if out.dim() == 3 else 0, ptr(r_valid))
    return out


# ------------------------------------------------------------------------------------------ rows
def rmsnorm_fwd(x, w, eps, res=None, want_h=True):
    T, H = x.shape
    y = torch.empty_like(x)
    rstd = torch.empty(T, device=x.device, dtype=torch.float32)
    h = torch.empty_like(x) if (res is not None and want_h) else None
    call("lmod_rmsnorm_fwd", ptr(x), ptr(res), ptr(w), ptr(h), ptr(y), ptr(rstd), T, H, float(eps))
    return y, rstd, (h if res is not None else x)


def rmsnorm_bwd(dy, h, w, rstd, dres=None):
    T, H = h.shape
    dh = torch.empty_like(h)
    call("lmod_rmsnorm_bwd", ptr(dy), ptr(h), ptr(w), ptr(rstd), ptr(dres), ptr(dh), T, H)
    return dh


def rmsnorm_dw(dy, h, rstd, dw, accumulate=True):
    T, H = h.shape
    ws = torch.empty(64 * H, device=h.device, dtype=torch.float32)
    call("lmod_rmsnorm_dw", ptr(dy), ptr(h), ptr(rstd), ptr(dw), ptr(ws), T, H, int(accumulate))
    return dw


def embed_wgrad(d_embeds, idx, dW):
    call("lmod_embed_wgrad", ptr(d_embeds), ptr(idx), ptr(dW), d_embeds.shape[0], d_embeds.shape[1])
    return dW


def layernorm_fwd(x, w, b, eps):
    T, H = x.shape
    y = torch.empty_like(x)
    call("lmod_layernorm_fwd", ptr(x), ptr(w), ptr(b), ptr(y), T, H, float(eps))
    return y


def rope_(buf, cos_t, sin_t, pos, nheads, hd, backward=False):
    T, ld = buf.shape[0], buf.stride(0)
    call("lmod_rope", ptr(buf), ptr(cos_t), ptr(sin_t), ptr(pos), T, nheads, hd, ld, int(backward))
    return buf


def swiglu_fwd(gate, up, out=None, seg_rows=0, seg_valid=None):
    rows, I = gate.shape
    if out is None:
        out = torch.empty((rows, I), device=gate.device, dtype=BF16)
    call("lmod_swiglu_fwd", ptr(gate), ptr(up), ptr(out), rows, I, gate.stride(0), up.stride(0), out.stride(0),
         seg_rows, ptr(seg_valid))
    return out


def swiglu_bwd(dact, gate, up, dgate=None, dup=None, seg_rows=0, seg_valid=None):
    rows, I = gate.shape
    if dgate is None:
        dgate = torch.empty((rows, I), device=gate.device, dtype=BF16)
    if dup is None:
        dup = torch.empty((rows, I), device=gate.device, dtype=BF16)
    call("lmod_swiglu_bwd", ptr(dact), ptr(gate), ptr(up), ptr(dgate), ptr(dup), rows, I, dact.stride(0),
         gate.stride(0), up.stride(0), dgate.stride(0), dup.stride(0), seg_rows, ptr(seg_valid))
    return dgate, dup


def gelu_fwd(x):
    y = torch.empty_like(x)
    call("lmod_gelu_fwd", ptr(x), ptr(y), x.numel())
    return y


def gelu_bwd(dy, x):
    dx = torch.empty_like(x)
    call("lmod_gelu_bwd", ptr(dy), ptr(x), ptr(dx), x.numel())
    return dx


def add(a, b, out=None):
    if out is None:
        out = torch.empty_like(a)
    call("lmod_add_bf16", ptr(a), ptr(b), ptr(out), a.numel())
    return out


def gather_rows(src_a, src_b, idx, H):
    rows = idx.numel()
    out = torch.empty((rows, H), device=idx.device, dtype=BF16)
    call("lmod_gather_rows", ptr(src_a), ptr(src_b), ptr(idx), ptr(out), rows, H)
    return out


def im2col_patch(pixels, patch, kpad):
    B, _, S, _ = pixels.shape
    G = S // patch
    out = torch.empty((B * G * G, kpad), device=pixels.device, dtype=BF16)
    call("lmod_im2col_patch", ptr(pixels), ptr(out), B, S, patch, kpad)
    return out


def vit_embed(patch_emb, cls, pos, B, n_patches):
    D = patch_emb.shape[-1]
    out = torch.empty((B * (n_patches + 1), D), device=patch_emb.device, dtype=BF16)
    call("lmod_vit_embed", ptr(patch_emb), ptr(cls), ptr(pos), ptr(out), B, n_patches, D)
    return out


def adamw_step(master, param, grad, m, v, lr, beta1, beta2, eps, wd, step, grad_scale=1.0, zero_grad=False,
               dev_scale=None):
    call("lmod_adamw_step", ptr(master), ptr(param), ptr(grad), ptr(m), ptr(v), master.numel(), float(lr),
         float(beta1), float(beta2), float(eps), float(wd), int(step), float(grad_scale), int(zero_grad), ptr(dev_scale))


def sumsq(x, out, partials, accumulate=False):
    """out[0] (+)= sum(x^2) for a contiguous fp32 tensor (deterministic)."""
    call("lmod_sumsq_f32", ptr(x), x.numel(), ptr(partials), ptr(out), int(accumulate))
    return out


def clip_coef(sumsq_t, norm_scale, max_norm, coef, norm_out=None):
    call("lmod_clip_coef", ptr(sumsq_t), float(norm_scale), float(max_norm), ptr(coef), ptr(norm_out))
    return coef


def cast_f32_bf16(src, dst):
    """Elementwise fp32 -> bf16 or bf16 -> fp32 between two contiguous tensors of equal numel."""
    to_bf16 = src.dtype == torch.float32
    assert dst.dtype == (BF16 if to_bf16 else torch.float32) and src.numel() == dst.numel()
    call("lmod_cast_f32_bf16", ptr(src), ptr(dst), src.numel(), int(to_bf16))
    return dst


# ------------------------------------------------------------------------------------------ attention
def attn_fwd(q, k, v, B, S, nh, nkv, hd, scale, causal, seqlens=None, want_lse=True, cu=None):
    """q/k/v: 2-D views [T, ld] whose column 0 is head 0 (views into a fused QKV buffer are fine).  Padded layout: T = B*S,
    sample b at rows b*S.., keys >= seqlens[b] masked.  Packed (cu [B+1] int32, hd 128): T = cu[B], S = longest sample."""
    o = torch.empty((q.shape[0], nh * hd), device=q.device, dtype=BF16)
    lse = torch.empty((B, nh, S), device=q.device, dtype=torch.float32) if want_lse else None
    call("lmod_attn_fwd", ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), ptr(seqlens), ptr(cu), B, S, nh, nkv, hd, q.stride(0),
         k.stride(0), v.stride(0), o.stride(0), float(scale), int(causal))
    return o, lse


def attn_bwd_rope_fusable(hd):
    """lmod_attn_bwd_rope exists in the hd-128 kernels of attn_bwd2.hip only (not in the generic LMOD_ATTN_BWD=1 path)."""
    import os
    return hd == 128 and os.environ.get("LMOD_ATTN_BWD") != "1"


def attn_bwd(q, k, v, o, do, lse, dq, dk, dv, B, S, nh, nkv, hd, scale, causal, seqlens=None, cu=None, rope=None, split=True):
    """rope = (cos_t, sin_t, pos): also apply the rotary embedding's gradient map to dq / dk in the kernels' epilogues.
    split=False: never use the head-split form of the dK/dV launch (tests, A/B timing)."""
    delta = torch.empty((B, nh, S), device=q.device, dtype=torch.float32)
    args = (ptr(q), ptr(k), ptr(v), ptr(o), ptr(do), ptr(lse), ptr(delta), ptr(dq), ptr(dk), ptr(dv),
            ptr(seqlens), ptr(cu), B, S, nh, nkv, hd, q.stride(0), k.stride(0), v.stride(0), o.stride(0), do.stride(0),
            dq.stride(0), dk.stride(0), dv.stride(0), float(scale), int(causal))
    rp = (ptr(rope[0]), ptr(rope[1]), ptr(rope[2])) if rope is not None else (None, None, None)
    # few KV heads (grouped-query students): the dK/dV kernel's grid is cut along the query-head group, fp32 partial sums in `ws`
    ns = _hip.load().lmod_attn_bwd_nsplit(B, S, nh, nkv, hd, int(causal)) if (cu is None and split and rope is None) else 1
    if ns > 1:
        ws = torch.empty(ns * 2 * B * S * nkv * hd, device=q.device, dtype=torch.float32)
        call("lmod_attn_bwd_split", *args, None, None, None, ptr(ws), ws.numel() * 4)
    elif attn_bwd_ds_fusable(B, S, nh, hd, cu):
        # dS spill: the dK/dV kernel stores dS^T into the (cached, per device) workspace and dQ is one batched TN GEMM; the dQ kernel's
        # recomputation of S, dP and the exponentials is gone (include/lmod_hip.h, lmod_attn_bwd_split)
        ws = _ds_workspace(q.device, B * nh * S * S * 2)
        call("lmod_attn_bwd_split", *args, *rp, ptr(ws), ws.numel())
    elif rope is not None:
        call("lmod_attn_bwd_rope", *args, *rp)
    else:
        call("lmod_attn_bwd", *args)
    return dq, dk, dv


_DS_WS = {}
DS_WS_MAX_BYTES = 8 << 30


def attn_bwd_ds_fusable(B, S, nh, hd, cu=None):
    """The dS-spill form of the attention backward (5 matmuls instead of 7) applies to hd 128, dense layouts and whole 256-row blocks;
    its workspace is B * nh * S^2 bf16 (2.1 GB at B 16, 16 heads, S 2048), bounded here; LMOD_ATTN_DS=0 keeps the two-kernel form."""
    return (hd == 128 and cu is None and S % 256 == 0 and os.environ.get("LMOD_ATTN_DS", "1") != "0"
            and B * nh * S * S * 2 <= DS_WS_MAX_BYTES)


def _ds_workspace(device, nbytes):
    """One workspace per device, grown on demand, shared by every layer's backward (launches on one stream are ordered: the dQ GEMM of
    layer l has read it before layer l - 1's dK/dV kernel writes it)."""
    w = _DS_WS.get(device)
    if w is None or w.numel() < nbytes:
        _DS_WS[device] = w = torch.empty(nbytes, device=device, dtype=torch.uint8)
    return w


def row_argmax(logits):
    R, V = logits.shape
    out = torch.empty(R, device=logits.device, dtype=torch.int32)
    call("lmod_row_argmax_bf16", ptr(logits), logits.stride(0), V, ptr(out), R)
    return out


def attn_decode(q, kcache, vcache, lens, nh, nkv, hd, scale):
    """q [B, nh*hd] (views fine), caches [B, Smax, nkv*hd] with lens[b] valid keys -> out [B, nh*hd]."""
    B = q.shape[0]
    out = torch.empty((B, nh * hd), device=q.device, dtype=BF16)
    call("lmod_attn_decode", ptr(q), ptr(kcache), ptr(vcache), ptr(lens), ptr(out), B, nh, nkv, hd, kcache.shape[1],
         q.stride(0), kcache.stride(1), out.stride(0), float(scale))
    return out


# ------------------------------------------------------------------------------------------ MoE
def moe_router_fwd(x, wg):
    T, H = x.shape
    E = wg.shape[0]
    logits = torch.empty((T, E), device=x.device, dtype=torch.float32)
    call("lmod_moe_router_fwd", ptr(x), ptr(wg), ptr(logits), T, H, E)
    return logits


class GateState:
    """Index maps produced by lmod_moe_gate (all device tensors)."""
    __slots__ = ("T", "E", "k", "C", "gates", "idx1", "idx2", "slot1", "slot2", "w1", "w2", "slot_token", "slot_w",
                 "exp_counts", "gate_sum", "l_aux", "slots_used", "noise")


def moe_gate(logits, k, C, noise=None, seed=None, offset=0, want_noise=False):
    """noise: explicit [T,E] noise (k=2: added for the 2nd pick; k=1: random-token-selection priorities).  seed (with noise
    None): the kernel draws it (Gumbel for k=2, uniform for k=1) from Philox(seed, offset + token); want_noise returns it as
    st.noise."""
    T, E = logits.shape
    dev = logits.device
    st = GateState()
    st.T, st.E, st.k, st.C = T, E, k, C
    st.gates = torch.empty((T, E), device=dev, dtype=torch.float32)
    i32 = dict(device=dev, dtype=torch.int32)
    f32 = dict(device=dev, dtype=torch.float32)
    st.idx1 = torch.empty(T, **i32); st.slot1 = torch.empty(T, **i32); st.w1 = torch.empty(T, **f32)
    if k == 2:
        st.idx2 = torch.empty(T, **i32); st.slot2 = torch.empty(T, **i32); st.w2 = torch.empty(T, **f32)
    else:
        st.idx2 = st.slot2 = st.w2 = None
    st.slot_token = torch.empty(E * C, **i32); st.slot_w = torch.empty(E * C, **f32)
    st.exp_counts = torch.empty(E, **i32); st.gate_sum = torch.empty(E, **f32); st.l_aux = torch.empty(1, **f32)
    st.slots_used = torch.empty(E, **i32)
    me = 8 if E <= 8 else (16 if E <= 16 else 32)          # expert slots of the routing kernels' instantiation (csrc/moe.hip)
    scratch = torch.empty(4 * T + 3 * me * ((T + 511) // 512), **i32)
    mode = 0 if (noise is not None or seed is None) else (1 if k == 2 else 2)
    st.noise = torch.empty((T, E), **f32) if (mode and want_noise) else None
    call("lmod_moe_gate", ptr(logits), ptr(noise), T, E, k, C, ptr(st.gates), ptr(st.idx1), ptr(st.idx2),
         ptr(st.slot1), ptr(st.slot2), ptr(st.w1), ptr(st.w2), ptr(st.slot_token), ptr(st.slot_w),
         ptr(st.exp_counts), ptr(st.gate_sum), ptr(st.l_aux), ptr(st.slots_used), ptr(scratch), mode,
         int(seed or 0) & 0xFFFFFFFFFFFFFFFF, int(offset) & 0xFFFFFFFFFFFFFFFF, ptr(st.noise))
    return st


def moe_combine_fwd(y, st, H):
    out = torch.empty((st.T, H), device=y.device, dtype=BF16)
    call("lmod_moe_combine_fwd", ptr(y), ptr(st.slot1), ptr(st.slot2), ptr(st.w1), ptr(st.w2), ptr(out), st.T, H)
    return out


def moe_combine_bwd(dout, y, st, H):
    S = st.E * st.C
    dy = torch.empty((S, H), device=y.device, dtype=BF16)
    dw1 = torch.empty(st.T, device=y.device, dtype=torch.float32)
    dw2 = torch.empty(st.T, device=y.device, dtype=torch.float32) if st.k == 2 else None
    call("lmod_moe_combine_bwd", ptr(dout), ptr(y), ptr(st.slot1), ptr(st.slot2), ptr(st.slot_token), ptr(st.slot_w),
         ptr(dy), ptr(dw1), ptr(dw2), st.T, S, H)
    return dy, dw1, dw2


def moe_gate_bwd(st, dw1, dw2, d_laux):
    dlogits = torch.empty((st.T, st.E), device=dw1.device, dtype=torch.float32)
    call("lmod_moe_gate_bwd", ptr(st.gates), ptr(st.idx1), ptr(st.idx2), ptr(st.slot1), ptr(st.slot2), ptr(dw1),
         ptr(dw2), ptr(st.exp_counts), ptr(d_laux), ptr(dlogits), st.T, st.E, st.k)
    return dlogits


def moe_dispatch_bwd(d_in, st, dlogits, wg, H):
    dx = torch.empty((st.T, H), device=d_in.device, dtype=BF16)
    call("lmod_moe_dispatch_bwd", ptr(d_in), ptr(st.slot1), ptr(st.slot2), ptr(dlogits), ptr(wg), ptr(dx), st.T, H,
         st.E)
    return dx


def moe_router_wgrad(x, dlogits, dwg, accumulate):
    T, H = x.shape
    E = dlogits.shape[1]
    ws = torch.empty(((T + 63) // 64) * E * H, device=x.device, dtype=torch.float32)      # one partial per 64-token slab
    call("lmod_moe_router_wgrad", ptr(x), ptr(dlogits), ptr(dwg), ptr(ws), T, H, E, int(accumulate))
    return dwg


def residual_mix_fwd(moe_out, mlp_out, coef_logits, coef_bias):
    T, H = moe_out.shape
    out = torch.empty_like(moe_out)
    p = torch.empty((T, 2), device=moe_out.device, dtype=torch.float32)
    call("lmod_moe_residual_mix_fwd", ptr(moe_out), ptr(mlp_out), ptr(coef_logits), ptr(coef_bias), ptr(out), ptr(p), T, H)
    return out, p


def residual_mix_bwd(dout, moe_out, mlp_out, p):
    T, H = moe_out.shape
    d_moe, d_mlp = torch.empty_like(moe_out), torch.empty_like(moe_out)
    dc = torch.empty((T, 2), device=moe_out.device, dtype=torch.float32)
    call("lmod_moe_residual_mix_bwd", ptr(dout), ptr(moe_out), ptr(mlp_out), ptr(p), ptr(d_moe), ptr(d_mlp), ptr(dc), T, H)
    return d_moe, d_mlp, dc


def small_linear_dgrad(dlogits, w):
    T, E = dlogits.shape
    H = w.shape[1]
    dx = torch.empty((T, H), device=dlogits.device, dtype=BF16)
    call("lmod_small_linear_dgrad", ptr(dlogits), ptr(w), ptr(dx), T, H, E)
    return dx


# ------------------------------------------------------------------------------------------ losses
NSTAT = 8


def rowloss_fwd(s, Vs, t=None, Va=None, label=None):
    R = s.shape[0]
    stats = torch.empty((R, NSTAT), device=s.device, dtype=torch.float32)
    if Va is None:
        Va = Vs
    call("lmod_rowloss_fwd", ptr(s), s.stride(0), Vs, ptr(t), t.stride(0) if t is not None else 0, Va, ptr(label),
         ptr(stats), R)
    return stats


def rowloss_bwd(s, Vs, t, Va, label, stats, kd_w, ce_w, seg_id, kd_scale, ce_scale, ds=None):
    R = s.shape[0]
    if ds is None:
        ds = s                      # in place over the student logits
    call("lmod_rowloss_bwd", ptr(s), s.stride(0), Vs, ptr(t), t.stride(0) if t is not None else 0, Va, ptr(label),
         ptr(stats), ptr(kd_w), ptr(ce_w), ptr(seg_id), ptr(kd_scale), ptr(ce_scale), ptr(ds), ds.stride(0), R)
    return ds


def segment_wsum(stats, col, w, seg_off):
    nseg = seg_off.numel() - 1
    out = torch.empty(nseg, device=stats.device, dtype=torch.float32)
    outw = torch.empty(nseg, device=stats.device, dtype=torch.float32)
    call("lmod_segment_wsum", ptr(stats), stats.stride(0), col, ptr(w), ptr(seg_off), nseg, ptr(out), ptr(outw))
    return out, outw


LOSS_TYPES = {"sigmoid": 0, "hinge": 1, "ipo": 2, "kto_pair": 3}


def dpo_loss(pc, pr, rc, rr, beta, label_smoothing, loss_type):
    B = pc.numel()
    lt = LOSS_TYPES[loss_type]
    dev = pc.device
    losses = torch.empty(2 * B if lt == 3 else B, device=dev, dtype=torch.float32)
    cr = torch.empty(B, device=dev, dtype=torch.float32)
    rj = torch.empty(B, device=dev, dtype=torch.float32)
    dpc = torch.empty(B, device=dev, dtype=torch.float32)
    dpr = torch.empty(B, device=dev, dtype=torch.float32)
    call("lmod_dpo_loss", ptr(pc), ptr(pr), ptr(rc), ptr(rr), B, float(beta), float(label_smoothing), lt, ptr(losses),
         ptr(cr), ptr(rj), ptr(dpc), ptr(dpr))
    return losses, cr, rj, dpc, dpr


def row_softmax_f32(logits_f32, Va, log):
    """[R, V] fp32 -> fp32 softmax / log_softmax over the first Va columns (materialising API parity)."""
    R = logits_f32.shape[0]
    out = torch.empty((R, Va), device=logits_f32.device, dtype=torch.float32)
    call("lmod_row_softmax_f32", ptr(logits_f32), logits_f32.stride(0), Va, int(log), ptr(out), R)
    return out


def rowdot_masked(p, logp):
    R, V = p.shape
    x = torch.empty(R, device=p.device, dtype=torch.float32)
    call("lmod_rowdot_masked", ptr(p), ptr(logp), V, R, ptr(x))
    return x
