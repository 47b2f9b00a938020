"""Kernel-level numerics on a real MI355X: each hand-written HIP kernel (called through the C ABI)
against a plain PyTorch fp32 reference of the same op computed from the same bf16-rounded inputs.
Tolerances are stated per test: bf16 outputs are allowed 2 bf16 ulp (2^-7 relative) plus a small
absolute term; fp32 outputs 1e-4..1e-3 relative (summation order differs)."""
import math
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from llavamod import kernels as K  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import moe as omoe  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def rnd(*shape, scale=1.0, seed=None):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed if seed is not None else (hash(shape) & 0xFFFF))
    return (torch.randn(*shape, generator=g) * scale).to(BF).to(DEV)


def close(out, ref, name, rtol=2 ** -7, afrac=2 ** -8, scale=None):
    """|out - ref| <= rtol |ref| + afrac * S.  S is the magnitude the absolute term is taken from:
      * default: the maximum of |ref| over the tensor (attention gradients, router gradients, normalisations: quantities whose
        error is set by OTHER elements' magnitudes — a query row's dQ error by the keys it attends to, not by its own size);
      * scale="row" — the GEMM family: the maximum over the element's own ROW (last dimension), so that an error confined to a
        row of small-magnitude outputs (a wrong tail row of a grouped launch, say) cannot hide under 0.4 % of some other row's
        maximum;
      * scale=<tensor like ref>: the element's own natural scale, e.g. sum_k |a_mk| |b_nk| for a dot product — the bound of
        `test_gemm_scaled_rows_and_columns_elementwise_bound`."""
    out, ref = out.float(), ref.float()
    assert out.shape == ref.shape, f"{name}: shape {tuple(out.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(out).all(), f"{name}: non-finite output"
    err = (out - ref).abs()
    if torch.is_tensor(scale):
        S = scale.float()
    elif scale == "row" and ref.dim() >= 2:
        S = ref.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
    else:
        S = ref.abs().max().clamp_min(1e-30)
    bound = afrac * S + rtol * ref.abs()
    bad = err > bound
    if bad.any():
        i = torch.argmax(err - bound)
        idx = tuple(int(x) for x in torch.unravel_index(i, err.shape))
        raise AssertionError(f"{name}: {int(bad.sum())}/{err.numel()} outside tolerance; worst at {idx}: "
                             f"out={out[idx].item():.6g} ref={ref[idx].item():.6g} max_abs_err={err.max().item():.4g} "
                             f"ref_absmax={ref.abs().max().item():.4g}")


# ------------------------------------------------------------------------------------------ GEMM
def test_gemm_layout_identity_asymmetric():
    # A = I (so C[m, n] must equal B[n, m]); B asymmetric -> detects transposed / permuted C writes.
    M = N = K = 128
    a = torch.eye(M, K, device=DEV, dtype=BF)
    b = (torch.arange(N * K, device=DEV, dtype=torch.float32).reshape(N, K) % 251 - 125).to(BF)
    c = K_gemm(a, b)
    close(c, b.float().t(), "gemm identity", rtol=0, afrac=0, scale="row")


def K_gemm(a, b, **kw):
    return K.gemm_nt(a, b, **kw)


@pytest.mark.parametrize("M,N,K_", [(128, 128, 64), (256, 384, 512), (100, 200, 72), (1, 16, 8), (300, 136, 2048),
                                     (513, 1000, 1032)])
def test_gemm_random_shapes(M, N, K_):
    a, b = rnd(M, K_, seed=1), rnd(N, K_, seed=2)
    c = K_gemm(a, b)
    close(c, a.float() @ b.float().t(), f"gemm {M}x{N}x{K_}", scale="row")


@pytest.mark.parametrize("M,N,K_", [(4000, 2568, 200), (300, 136, 2048), (8192, 4096, 1024)])
def test_gemm_scaled_rows_and_columns_elementwise_bound(M, N, K_):
    """Rows of A and rows of B scaled by powers of two from 2^-8 to 2^8: the outputs span 2^-16 ... 2^16, so a tolerance tied to
    the tensor's (or even a row's) maximum would accept garbage in the small-magnitude rows and columns — ragged tails included.
    Each element is held to ITS OWN scale: 2^-7 |ref| (two bf16 roundings) + 2^-12 sum_k |a_mk| |b_nk| (accumulation order).
    Covers the 128-tile kernel, the 8-wave and the 4-wave asm kernels (persistent from 4 rounds of the CUs)."""
    a, b = rnd(M, K_, seed=90), rnd(N, K_, seed=91)
    sa = torch.pow(2.0, torch.arange(M, device=DEV) % 17 - 8.0).to(BF)[:, None]
    sb = torch.pow(2.0, (torch.arange(N, device=DEV) * 5) % 17 - 8.0).to(BF)[:, None]
    a, b = (a * sa).contiguous(), (b * sb).contiguous()        # exact: powers of two
    ref = a.float() @ b.float().t()
    mag = a.float().abs() @ b.float().abs().t()
    close(K_gemm(a, b), ref, f"scaled gemm {M}x{N}x{K_}", rtol=2 ** -7, afrac=2 ** -12, scale=mag)


def test_gemm_big_tile_path_all_epilogues():
    # large enough for the 256x256 four-wave kernel (>= 160 tiles), ragged edges in M, N and K
    M, N, K_ = 4000, 2568, 200
    a, b, bias = rnd(M, K_, seed=70), rnd(N, K_, seed=71), rnd(N, seed=72)
    ref = a.float() @ b.float().t()
    close(K_gemm(a, b), ref, "big gemm", scale="row")
    pre = (ref + bias.float()).to(BF).float()
    close(K_gemm(a, b, bias=bias, act=1), F.gelu(pre), "big gemm+bias+gelu", scale="row")
    close(K_gemm(a, b, out_f32=True), ref, "big gemm f32", rtol=1e-4, afrac=1e-5, scale="row")
    acc = torch.full((M, N), -1.5, device=DEV, dtype=torch.float32)
    K_gemm(a, b, out=acc, out_f32=True, accumulate=True)
    close(acc, ref - 1.5, "big gemm f32 accumulate", rtol=1e-4, afrac=1e-5, scale="row")
    eye = torch.eye(M, K_, device=DEV, dtype=BF)                      # layout check: C[m, n] = B[n, m] for m < K
    got = K_gemm(eye, b)
    assert torch.equal(got[:K_].float(), b.float().t()) and got[K_:].abs().max().item() == 0


@pytest.mark.parametrize("K_", [64, 128, 192, 448, 2048])
def test_gemm_big_tile_k64_hand_placed_loop(K_):
    # K % 64 == 0 on the four-wave kernel = the K loop as one asm statement (gemm4_loop_asm.h).  1, 2, 3, 7 and 32 K tiles: the
    # first cases end inside the prefetch distance (tiles past the end switch to a zero-record descriptor); ragged M / N edges;
    # plain, bias + activation, the fused QKV + RoPE epilogue's plain cousin (bias only) and the fused SwiGLU forward share the loop
    M, N = 4000, 2568
    a, b, bias = rnd(M, K_, seed=80 + K_), rnd(N, K_, seed=81 + K_), rnd(N, seed=82)
    ref = a.float() @ b.float().t()
    close(K_gemm(a, b), ref, f"k64 gemm K={K_}", scale="row")
    pre = (ref + bias.float()).to(BF).float()
    close(K_gemm(a, b, bias=bias, act=1), F.gelu(pre), f"k64 gemm+bias+gelu K={K_}", scale="row")
    assert torch.equal(K_gemm(a, b), K_gemm(a, b))
    eye = torch.eye(M, K_, device=DEV, dtype=BF)
    got = K_gemm(eye, b)
    assert torch.equal(got[:K_].float(), b.float().t()) and got[K_:].abs().max().item() == 0
    I = 1280
    wgu = rnd(2 * I, K_, seed=83 + K_)
    gu_ref = K.gemm_nt(a, wgu)
    act, gu = K.gemm_swiglu(a, wgu, want_gu=True)
    assert torch.equal(gu, gu_ref) and torch.equal(act, K.swiglu_fwd(gu_ref[:, :I], gu_ref[:, I:]))
    g, u = a.float() @ wgu[:I].float().t(), a.float() @ wgu[I:].float().t()
    close(act, F.silu(g.to(BF).float()).to(BF).float() * u.to(BF).float(), f"k64 fused swiglu K={K_}", scale="row")


@pytest.fixture
def persist_from_one_round(monkeypatch):
    """The launcher takes the persistent form from 10 rounds of the CUs up; tests lower that to "more tiles than CUs" (the switch is
    read per launch) so that small shapes walk several tiles per workgroup."""
    monkeypatch.setenv("LMOD_GEMM_PERSIST_ROUNDS", "1")
    monkeypatch.setenv("LMOD_GEMM_PERSIST", "1")


@pytest.mark.parametrize("K_", [256, 320, 1024])
def test_gemm_persistent_tile_walk(K_, persist_from_one_round):
    # more 256x256 tiles than CUs (18 x 17 = 306 > 256): the persistent four-wave kernel — one workgroup per CU walks several
    # tiles, its K loop's look-ahead runs on into the NEXT tile's operand windows.  Ragged M and N edges (rows / columns past the
    # edge are cut by the buffer descriptors' byte counts on this path), K of 4, 5 and 16 K tiles, second tiles of every parity;
    # plain, bias + activation, and the fused SwiGLU forward (128-column tiles: 18 x 18).
    M, N = 4500, 4300 - 4
    a, b, bias = rnd(M, K_, seed=90 + K_), rnd(N, K_, seed=91 + K_), rnd(N, seed=92)
    ref = a.float() @ b.float().t()
    got = K_gemm(a, b)
    close(got, ref, f"persistent gemm K={K_}", scale="row")
    assert torch.equal(got, K_gemm(a, b))
    pre = (ref + bias.float()).to(BF).float()
    close(K_gemm(a, b, bias=bias, act=1), F.gelu(pre), f"persistent gemm+bias+gelu K={K_}", scale="row")
    eye = torch.eye(M, K_, device=DEV, dtype=BF)
    g2 = K_gemm(eye, b)
    assert torch.equal(g2[:K_].float(), b.float().t()) and g2[K_:].abs().max().item() == 0
    I = 2296
    wgu = rnd(2 * I, K_, seed=93 + K_)
    gu_ref = K.gemm_nt(a, wgu)
    act, gu = K.gemm_swiglu(a, wgu, want_gu=True)
    assert torch.equal(gu, gu_ref) and torch.equal(act, K.swiglu_fwd(gu_ref[:, :I], gu_ref[:, I:]))
    # batched (3 x 9 x 10 = 270 tiles, a tile walk that crosses batch entries) with per-batch weights
    a3, b3 = rnd(3, 2300, K_, seed=94 + K_), rnd(3, 2560, K_, seed=95 + K_)
    got3 = K_gemm(a3, b3)
    close(got3, torch.einsum("bmk,bnk->bmn", a3.float(), b3.float()), f"persistent batched gemm K={K_}", scale="row")
    # the one-tile-per-workgroup form of the same launches: bit-identical
    os.environ["LMOD_GEMM_PERSIST"] = "0"
    assert torch.equal(got, K_gemm(a, b)) and torch.equal(act, K.gemm_swiglu(a, wgu)[0]) and torch.equal(got3, K_gemm(a3, b3))


@pytest.mark.parametrize("M,N,Kd", [(8192, 2048, 2048), (4096, 4096, 11008), (8232, 2176, 2048), (2048, 4096, 11008)])
def test_gemm_residual_epilogue_is_the_two_step_form(M, N, Kd, monkeypatch):
    """bf16(res + bf16(x W^T)) in the GEMM epilogue (gemm4_kernel<8>: the decoder layer's residual adds) = lmod_gemm_bf16_nt followed by
    the residual path of lmod_rmsnorm_fwd, bit for bit — one tile per workgroup and persistent, ragged M / N edges included."""
    x, w, res = rnd(M, Kd, seed=160), rnd(N, Kd, scale=0.03, seed=161), rnd(M, N, seed=162)
    assert K.gemm_res_fusable(M, w, res)
    delta = K_gemm(x, w)
    _, _, h = K.rmsnorm_fwd(delta, torch.ones(N, device=DEV, dtype=BF), 1e-6, res=res)
    assert torch.equal(h, (delta.float() + res.float()).to(BF))
    monkeypatch.setenv("LMOD_GEMM_PERSIST", "0")
    got = K.gemm_nt_res(x, w, res)
    assert torch.equal(got, h), f"one tile per workgroup: {(got.float() - h.float()).abs().max().item()}"
    monkeypatch.setenv("LMOD_GEMM_PERSIST", "1")
    monkeypatch.setenv("LMOD_GEMM_PERSIST_ROUNDS", "1")
    got = K.gemm_nt_res(x, w, res)
    assert torch.equal(got, h), f"persistent: {(got.float() - h.float()).abs().max().item()}"
    # a shape the 256-tile kernel does not take: the library answers LMOD_EUNSUPPORTED and the wrapper keeps the header's two-step
    # form (a drift between `gemm_res_fusable` and the library's own test must not raise inside a decoder layer)
    small, rsm = rnd(256, Kd, seed=163), rnd(256, N, seed=164)
    assert not K.gemm_res_fusable(256, w, rsm)
    from llavamod import _hip
    rc = _hip.call("lmod_gemm_bf16_nt_res", small.data_ptr(), w.data_ptr(), torch.empty(256, N, device=DEV, dtype=BF).data_ptr(), None,
                   rsm.data_ptr(), 256, N, Kd, Kd, Kd, N, N, allow=(_hip.UNSUPPORTED,))
    assert rc == _hip.UNSUPPORTED
    got = K.gemm_nt_res(small, w, rsm)
    assert torch.equal(got, (K.gemm_nt(small, w).float() + rsm.float()).to(BF))


def test_decoder_layer_with_fused_residual_adds_is_bit_identical(monkeypatch):
    """A full-width dense decoder layer (H 2048, I 5504, 16 heads, T = 8192) forward + backward with the residual adds in the o / down
    projections' epilogues (LMOD_GEMM_RES default) against the two-step form (LMOD_GEMM_RES=0): outputs and every gradient equal."""
    from types import SimpleNamespace
    from llavamod.engine import GradBuffer
    from llavamod.model.language_model import qwen2_hip as Q
    cfg = SimpleNamespace(hidden_size=2048, intermediate_size=5504, num_attention_heads=16, num_key_value_heads=16, head_dim=128,
                          rms_norm_eps=1e-6, rope_theta=1000000.0, max_position_embeddings=2048)
    torch.manual_seed(5)
    layer = Q.Qwen2DecoderLayer(cfg, DEV)
    for name, prm in layer.named_parameters():
        if "layernorm" in name:
            prm.data.fill_(1.0)
        else:
            prm.data.normal_(0, 0.02)
    B, S = 4, 2048
    cos, sin = Q.rope_tables(128, 2048, cfg.rope_theta, DEV)
    pos = torch.arange(S, device=DEV, dtype=torch.int32).repeat(B)
    rt = SimpleNamespace(B=B, S=S, cos=cos, sin=sin, pos=pos, seqlens=torch.full((B,), S, device=DEV, dtype=torch.int32))
    gb = GradBuffer(layer)
    delta0, res0 = rnd(B * S, 2048, seed=170), rnd(B * S, 2048, seed=171)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("LMOD_GEMM_RES", flag)
        gb.zero()
        delta, res = delta0.clone().requires_grad_(True), res0.clone().requires_grad_(True)
        m, h2, _ = layer(delta, res, rt)
        assert (h2 is None) == (flag == "1")
        total = m if h2 is None else (m.float() + h2.float()).to(BF)        # the next layer's add, done by hand for the two-step form
        total.backward(rnd(B * S, 2048, seed=172))
        torch.cuda.synchronize()
        outs.append((total.detach().clone(), delta.grad.clone(), res.grad.clone(), gb.flat.clone()))
    for a, b, name in zip(outs[0], outs[1], ("output", "d delta", "d res", "weight gradients")):
        assert torch.isfinite(a.float()).all() and a.float().abs().max() > 0, name
        assert torch.equal(a, b), f"{name}: fused != two-step ({(a.float() - b.float()).abs().max().item()})"


def test_fused_qkv_rope_persistent_is_bit_identical(persist_from_one_round):
    # fused q/k/v projection + bias + RoPE on the persistent kernel (16 x 24 = 384 tiles) against the one-tile-per-workgroup launch
    # and against GEMM + the separate RoPE kernel
    T, nh, nkv, Kd = 4000, 16, 16, 512
    hd, N = 128, (nh + 2 * nkv) * 128
    x, w, bias = rnd(T, Kd, seed=95), rnd(N, Kd, seed=96), rnd(N, seed=97)
    pos = torch.randint(0, 4096, (T,), device=DEV, dtype=torch.int32)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, device=DEV).float() / hd))
    fr = torch.arange(4096, device=DEV).float()[:, None] * inv[None]
    emb = torch.cat([fr, fr], -1)
    cos_t, sin_t = emb.cos().to(BF), emb.sin().to(BF)
    fused = K.gemm_qkv_rope(x, w, bias, cos_t, sin_t, pos, nh + nkv)
    os.environ["LMOD_GEMM_PERSIST"] = "0"
    plain = K.gemm_qkv_rope(x, w, bias, cos_t, sin_t, pos, nh + nkv)
    ref = K.gemm_nt(x, w, bias=bias)
    K.rope_(ref, cos_t, sin_t, pos, nh + nkv, hd)
    assert torch.equal(fused, plain) and torch.equal(fused, ref)


def test_gemm_bias_act_f32_accumulate():
    M, N, K_ = 200, 264, 320
    a, b, bias = rnd(M, K_, seed=3), rnd(N, K_, seed=4), rnd(N, seed=5)
    ref = a.float() @ b.float().t() + bias.float()
    close(K_gemm(a, b, bias=bias), ref, "gemm+bias", scale="row")
    pre = ref.to(BF).float()
    close(K_gemm(a, b, bias=bias, act=1), F.gelu(pre), "gemm+bias+gelu", scale="row")
    close(K_gemm(a, b, bias=bias, act=2), pre * torch.sigmoid(1.702 * pre), "gemm+bias+quickgelu", scale="row")
    c32 = K_gemm(a, b, out_f32=True)
    close(c32, a.float() @ b.float().t(), "gemm f32 out", rtol=1e-4, afrac=1e-5, scale="row")
    acc = torch.full((M, N), 2.0, device=DEV, dtype=torch.float32)
    K_gemm(a, b, out=acc, out_f32=True, accumulate=True)
    close(acc, a.float() @ b.float().t() + 2.0, "gemm f32 accumulate", rtol=1e-4, afrac=1e-5, scale="row")


@pytest.mark.parametrize("M,I,K_", [(200, 136, 320), (1024, 1152, 256), (300, 5504, 128)])
def test_gemm_fused_swiglu(M, I, K_):
    # ONE launch: silu(x Wg^T) * (x Wu^T) against the fused [2I, K] weight, bit-identical to GEMM + swiglu kernel
    x, w = rnd(M, K_, seed=30), rnd(2 * I, K_, seed=31)
    gu_ref = K.gemm_nt(x, w)
    ref = K.swiglu_fwd(gu_ref[:, :I], gu_ref[:, I:])
    act, gu = K.gemm_swiglu(x, w, want_gu=True)
    assert act.shape == (M, I) and gu.shape == (M, 2 * I)
    assert torch.equal(gu, gu_ref), "pre-activations differ from the plain GEMM"
    assert torch.equal(act, ref), float((act.float() - ref.float()).abs().max())
    act2, none = K.gemm_swiglu(x, w)
    assert none is None and torch.equal(act2, ref)
    gf, uf = x.float() @ w[:I].float().t(), x.float() @ w[I:].float().t()
    close(act, F.silu(gf.to(BF).float()).to(BF).float() * uf.to(BF).float(), "fused swiglu vs torch", scale="row")


def test_gemm_fused_swiglu_grouped():
    E, C, H, I = 4, 200, 128, 264
    x, w = rnd(E, C, H, seed=33), rnd(E, 2 * I, H, seed=34)
    mv = torch.tensor([200, 0, 77, 130], device=DEV, dtype=torch.int32)
    act = torch.full((E, C, I), 7.0, device=DEV, dtype=BF)
    K.gemm_swiglu(x, w, act=act, m_valid=mv)
    for e in range(E):
        n = int(mv[e]); n8 = min((n + 7) // 8 * 8, C)
        if n:
            gu = K.gemm_nt(x[e, :n].contiguous(), w[e])
            assert torch.equal(act[e, :n], K.swiglu_fwd(gu[:, :I], gu[:, I:])), f"expert {e}"
        assert act[e, n:n8].abs().max().item() == 0 if n8 > n else True, "rows up to the next multiple of 8 must be zeroed"
        assert n8 == C or (act[e, n8:] == 7.0).all(), "rows past the zeroed chunk must not be written"


@pytest.mark.parametrize("K_", [256, 1024])
@pytest.mark.parametrize("rows", [[2304, 0, 1100, 3000, 257, 2048], [3000] * 6, [0, 0, 5, 0, 0, 2999], [256] * 16])
def test_gemm_grouped_persistent_walk_is_bit_identical(K_, rows, monkeypatch):
    """Round 5: grouped launches (MoE capacity slabs, live rows per expert in `m_valid`) on the PERSISTENT four-wave kernel — one
    workgroup per CU walks the LIVE tiles only (live count computed in the kernel), operand stream continuous across tiles and
    across experts.  Plain grouped GEMM (the experts' down projection / dgrad) and the grouped fused SwiGLU forward, against the
    one-tile-per-workgroup form of the same launches: bit-identical outputs, dead rows untouched, rows up to the next multiple of 8
    zeroed by the SwiGLU form.  Cases: ragged counts with an empty expert and a 1-row tile tail; all slabs full; almost everything
    dead (fewer live tiles than CUs: most workgroups exit at once); 16 experts of exactly one row tile."""
    E, C = len(rows), 3000 if max(rows) > 256 else 256
    H, I = K_, 1280                                               # 12 x 5 = 60 column tiles... per row tile: N 1280 -> 5 tiles of 256
    x, w = rnd(E, C, H, seed=40 + K_), rnd(E, I, H, seed=41 + K_)
    wgu = rnd(E, 2 * I, H, seed=42 + K_)
    mv = torch.tensor(rows, device=DEV, dtype=torch.int32)
    monkeypatch.setenv("LMOD_GEMM_PERSIST", "1")
    monkeypatch.setenv("LMOD_GEMM_PERSIST_ROUNDS", "0")           # every grouped launch takes the persistent form

    def run():
        out = torch.full((E, C, I), 7.0, device=DEV, dtype=BF)
        K.gemm_nt(x, w, out=out, m_valid=mv)
        act = torch.full((E, C, I), 7.0, device=DEV, dtype=BF)
        gu = torch.full((E, C, 2 * I), 7.0, device=DEV, dtype=BF)
        K.gemm_swiglu(x, wgu, act=act, gu=gu, m_valid=mv)
        return out, act, gu
    monkeypatch.setenv("LMOD_GEMM_PERSIST_GROUPED", "2")           # 2: the plain grouped launches too (default 1: fused SwiGLU forward only)
    o1, a1, g1 = run()
    monkeypatch.setenv("LMOD_GEMM_PERSIST_GROUPED", "0")
    o0, a0, g0 = run()
    assert torch.equal(o1, o0) and torch.equal(a1, a0) and torch.equal(g1, g0)
    for e in range(E):
        n = rows[e]
        if n:
            close(o1[e, :n], x[e, :n].float() @ w[e].float().t(), f"grouped persistent expert {e}", scale="row")
        assert n == C or (o1[e, n:] == 7.0).all(), "dead rows must not be written"


def test_gemm_fused_swiglu_backward():
    # down-projection dgrad with swiglu_bwd in the epilogue == GEMM + swiglu_bwd kernel, bit for bit; grouped + dead rows
    M, H, I = 300, 136, 272
    dy, wd, gu = rnd(M, H, seed=60), rnd(H, I, seed=61), rnd(M, 2 * I, seed=62)
    wt = K.transpose(wd)                                  # [I, Hpad]
    dact = K.gemm_nt(dy, wt, M=M, N=I, K=H, lda=H, ldb=wt.stride(0))
    dg, du = K.swiglu_bwd(dact, gu[:, :I], gu[:, I:])
    dgu = K.gemm_swiglu_bwd(dy, wt, gu, K=H)
    assert torch.equal(dgu[:, :I], dg) and torch.equal(dgu[:, I:], du)
    inplace = gu.clone()
    K.gemm_swiglu_bwd(dy, wt, inplace, out=inplace, K=H)
    assert torch.equal(inplace, dgu)
    E, C = 3, 200
    dyg, wdg, gug = rnd(E, C, H, seed=63), rnd(E, H, I, seed=64), rnd(E, C, 2 * I, seed=65)
    wtg = K.transpose(wdg)
    mv = torch.tensor([200, 0, 77], device=DEV, dtype=torch.int32)
    out = torch.full((E, C, 2 * I), 7.0, device=DEV, dtype=BF)
    K.gemm_swiglu_bwd(dyg, wtg, gug, out=out, m_valid=mv, K=H)
    for e in range(E):
        n = int(mv[e]); n8 = min((n + 7) // 8 * 8, C)
        if n:
            ref = K.gemm_swiglu_bwd(dyg[e, :n].contiguous(), wtg[e], gug[e, :n].contiguous(), K=H)
            assert torch.equal(out[e, :n], ref), f"expert {e}"
        assert n8 == n or out[e, n:n8].abs().max().item() == 0
        assert n8 == C or (out[e, n8:] == 7.0).all()


def test_gemm_strided_output_and_subview():
    # write into a column slice of a wider buffer (QKV / gate-up fusion pattern)
    M, N, K_ = 130, 96, 128
    a, b = rnd(M, K_, seed=6), rnd(N, K_, seed=7)
    wide = torch.zeros((M, 3 * N), device=DEV, dtype=BF)
    K.gemm_nt(a, b, out=wide[:, N:2 * N], M=M, N=N, K=K_, lda=K_, ldb=K_, ldc=3 * N)
    close(wide[:, N:2 * N], a.float() @ b.float().t(), "gemm strided out", scale="row")
    assert wide[:, :N].abs().max() == 0 and wide[:, 2 * N:].abs().max() == 0


def test_gemm_grouped_valid_rows():
    E, C, H, I = 4, 192, 128, 256
    x = rnd(E, C, H, seed=8)
    w = rnd(E, I, H, seed=9)
    mv = torch.tensor([192, 0, 77, 130], device=DEV, dtype=torch.int32)
    out = torch.zeros((E, C, I), device=DEV, dtype=BF)
    K.gemm_nt(x, w, out=out, m_valid=mv)
    for e in range(E):
        n = int(mv[e])
        if n:
            close(out[e, :n], x[e, :n].float() @ w[e].float().t(), f"grouped e{e}", scale="row")
        assert n == C or out[e, n:].abs().max() == 0, "rows past m_valid must not be written"
    # k_valid: reduction extent per batch (wgrad over capacity slots)
    xt, dyt = rnd(E, H, C, seed=10), rnd(E, I, C, seed=11)
    kv = torch.tensor([192, 0, 77, 130], device=DEV, dtype=torch.int32)
    dw = K.gemm_nt(dyt, xt, k_valid=kv, out_f32=True)
    for e in range(E):
        n = (int(kv[e]) + 7) // 8 * 8     # k_valid is honoured at 8-element chunk granularity
        ref = dyt[e, :, :n].float() @ xt[e, :, :n].float().t()
        close(dw[e], ref, f"k_valid e{e}", rtol=1e-4, afrac=1e-5, scale="row")


@pytest.mark.parametrize("Kd,M,N", [(64, 8, 8), (300, 136, 264), (1000, 520, 256), (4096, 512, 1032)])
def test_gemm_tn_wgrad_form(Kd, M, N):
    # C = A^T B on reduction-major operands (dW = dY^T X), fp32 out / accumulate / bf16 out, strided views
    a_full, b_full = rnd(Kd, M + 16, seed=40), rnd(Kd, N + 8, seed=41)
    a, b = a_full[:, 8:8 + M], b_full[:, :N]                      # column sub-views: lda > M, 16-byte aligned offset
    ref = a.float().t() @ b.float()
    close(K.gemm_tn(a, b), ref, f"tn {Kd}x{M}x{N}", rtol=1e-4, afrac=1e-5, scale="row")
    acc = torch.full((M, N), 3.0, device=DEV, dtype=torch.float32)
    K.gemm_tn(a, b, out=acc, accumulate=True)
    close(acc, ref + 3.0, "tn accumulate", rtol=1e-4, afrac=1e-5, scale="row")
    close(K.gemm_tn(a, b, out_f32=False), ref, "tn bf16 out", scale="row")
    # asymmetric check against a transposed / permuted write: A = one-hot rows
    eye = torch.zeros(Kd, M, device=DEV, dtype=BF)
    idx = torch.arange(min(Kd, M), device=DEV)
    eye[idx, idx] = 1
    got = K.gemm_tn(eye, b)
    assert torch.equal(got[:min(Kd, M)], b[:min(Kd, M)].float()), "layout"


def test_gemm_tn_batched_kvalid_and_splitk():
    E, C, H, I = 4, 200, 136, 264
    dy, x = rnd(E, C, I, seed=42), rnd(E, C, H, seed=43)
    kv = torch.tensor([200, 0, 77, 130], device=DEV, dtype=torch.int32)
    dw = K.gemm_tn(dy, x, k_valid=kv)
    for e in range(E):
        n = int(kv[e])
        close(dw[e], dy[e, :n].float().t() @ x[e, :n].float(), f"tn k_valid e{e}", rtol=1e-4, afrac=1e-5, scale="row")
    # split-K through the batch dimension: 4 chunks of 256 rows into a [4, M, N] workspace
    a, b = rnd(1024, 264, seed=44), rnd(1024, 136, seed=45)
    ws = K.gemm_tn(a.view(4, 256, 264), b.view(4, 256, 136))
    close(ws.sum(0), a.float().t() @ b.float(), "tn split-K", rtol=1e-4, afrac=1e-5, scale="row")


@pytest.mark.parametrize("Kd,M,N", [(64, 8, 8), (37, 256, 256), (300, 136, 264), (1000, 520, 256), (4096 + 19, 512, 1032), (8192, 768, 512)])
def test_gemm_tn_accumulate_on_the_four_wave_asm_loop(Kd, M, N, monkeypatch):
    """fp32 C += A^T B on reduction-major operands takes gemm4t_kernel (4 waves, asm K loop, transposing LDS reads, descriptors cut
    the K edge at a ROW: any K, no padding).  Against the fp32 product of the same bf16 operands, against the 8-wave TN kernel
    (LMOD_GEMM_TN4=0), on column sub-views, and a one-hot operand pins the layout (which LDS row / fragment register / output column)."""
    a_full, b_full = rnd(Kd, M + 16, seed=140), rnd(Kd, N + 8, seed=141)
    a, b = a_full[:, 8:8 + M], b_full[:, :N]
    ref = a.float().t() @ b.float()
    acc = torch.full((M, N), 3.0, device=DEV, dtype=torch.float32)
    K.gemm_tn(a, b, out=acc, accumulate=True)
    close(acc, ref + 3.0, f"tn4 accumulate {Kd}x{M}x{N}", rtol=1e-4, afrac=1e-5, scale="row")
    monkeypatch.setenv("LMOD_GEMM_TN4", "0")
    old = torch.full((M, N), 3.0, device=DEV, dtype=torch.float32)
    K.gemm_tn(a, b, out=old, accumulate=True)
    monkeypatch.delenv("LMOD_GEMM_TN4")
    close(acc, old, "tn4 vs the 8-wave TN kernel", rtol=1e-4, afrac=1e-5, scale="row")
    # exact layout check: A = one-hot rows -> C[m] = B[k(m)]; a permutation so that a swapped k or m shows
    n1 = min(Kd, M)
    perm = torch.randperm(n1, device=DEV, generator=torch.Generator(device=DEV).manual_seed(7))
    eye = torch.zeros(Kd, M, device=DEV, dtype=BF)
    eye[perm, torch.arange(n1, device=DEV)] = 1
    got = torch.zeros(M, N, device=DEV, dtype=torch.float32)
    K.gemm_tn(eye, b, out=got, accumulate=True)
    assert torch.equal(got[:n1], b[perm].float()), "layout"
    assert got[n1:].abs().max().item() == 0 if n1 < M else True
    # rows past K (garbage, even NaN) are never read: the descriptors end at the last live row
    big_a, big_b = torch.full((Kd + 64, M), float("nan"), device=DEV, dtype=BF), torch.full((Kd + 64, N), float("nan"), device=DEV, dtype=BF)
    big_a[:Kd], big_b[:Kd] = a, b
    cut = torch.zeros(M, N, device=DEV, dtype=torch.float32)
    K.gemm_tn(big_a, big_b, out=cut, accumulate=True, K=Kd)
    close(cut, ref, "rows past K must not be read", rtol=1e-4, afrac=1e-5, scale="row")


def test_gemm_tn4_batched_kvalid_exact_rows():
    """MoE expert weight gradients without transposed copies: a live row count per expert (zero, ragged, full), garbage behind it."""
    for (E, C, H, I, kvl) in [(4, 200, 136, 264, [200, 0, 77, 130]), (4, 1024, 512, 256, [1024, 3, 640, 511]), (3, 512, 2048, 256, [512, 64, 200])]:
        dy, x = rnd(E, C, I, seed=142), rnd(E, C, H, seed=143)
        kv = torch.tensor(kvl, device=DEV, dtype=torch.int32)
        for e in range(E):
            dy[e, kvl[e]:] = float("nan")
            x[e, kvl[e]:] = float("inf")
        dw = torch.full((E, I, H), 1.0, device=DEV, dtype=torch.float32)
        K.gemm_tn(dy, x, out=dw, accumulate=True, k_valid=kv)
        for e in range(E):
            n = kvl[e]
            close(dw[e], dy[e, :n].float().t() @ x[e, :n].float() + 1.0, f"tn4 k_valid e{e} ({kvl})", rtol=1e-4, afrac=1e-5, scale="row")


@pytest.mark.parametrize("M,N,Kd", [(512, 512, 8192), (2048, 256, 4096 + 64), (264, 272, 5000), (128, 64, 512), (6144, 2048, 4096)])
def test_gemm_wgrad_tn_splitk_deterministic(M, N, Kd):
    """dW += dY^T X with both operands as autograd holds them (a_kmajor: lmod_gemm_wgrad_bf16_nt mode 2), deterministic split-K
    through the tile semaphores; the same numbers as the NT kernel on transposed copies (same products, same order inside a K tile)."""
    dy, x = rnd(Kd, M, seed=150), rnd(Kd, N, seed=151)
    ref = dy.float().t() @ x.float()
    outs = []
    for _ in range(3):
        g = torch.full((M, N), 0.5, device=DEV, dtype=torch.float32)
        K.gemm_wgrad(dy, x, g, a_kmajor=True)
        outs.append(g)
    close(outs[0], ref + 0.5, f"wgrad tn {M}x{N}x{Kd}", rtol=1e-4, afrac=1e-5, scale="row")
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "split-K must not depend on arrival order"
    K.gemm_wgrad(dy, x, outs[0], a_kmajor=True)
    close(outs[0], 2 * ref + 0.5, "wgrad tn accumulates", rtol=1e-4, afrac=1e-5, scale="row")
    nt = torch.full((M, N), 0.5, device=DEV, dtype=torch.float32)
    K.gemm_wgrad(K.transpose(dy), K.transpose(x), nt)
    close(outs[1], nt, "wgrad tn vs NT on transposed copies", rtol=1e-4, afrac=1e-5, scale="row")


def test_gemm_wgrad_tn_long_window_runs_in_k_chunks():
    """An operand window past 2 GiB (the lm_head's dY: [loss rows, vocab]) is walked in K chunks of whole launches."""
    Kd, M, N = 1536, 151936 * 5, 8            # row pitch 1.5 MB: 2 GiB / pitch = 1413 rows -> two launches
    dy = torch.zeros(Kd, M, device=DEV, dtype=BF)
    x = rnd(Kd, N, seed=152)
    cols = torch.tensor([0, 77, 4095, M - 1], device=DEV)
    dy[:, cols] = rnd(Kd, 4, seed=153)
    g = torch.zeros(M, N, device=DEV, dtype=torch.float32)
    K.gemm_wgrad(dy, x, g, a_kmajor=True)
    close(g[cols], dy[:, cols].float().t() @ x.float(), "k-chunked wgrad", rtol=1e-4, afrac=1e-5, scale="row")
    dead = torch.ones(M, device=DEV, dtype=torch.bool)
    dead[cols] = False
    assert g[dead].abs().max().item() == 0


@pytest.mark.parametrize("M,N,Kd", [(512, 512, 8192), (2048, 256, 4096 + 64), (264, 272, 5000 // 8 * 8), (128, 64, 512)])
def test_gemm_wgrad_splitk_deterministic(M, N, Kd):
    # fp32 accumulate with split-K through the tile semaphores: right answer, and bit-identical run to run
    at, bt = rnd(M, Kd, seed=50), rnd(N, Kd, seed=51)
    ref = at.float() @ bt.float().t()
    outs = []
    for _ in range(3):
        g = torch.full((M, N), 0.5, device=DEV, dtype=torch.float32)
        K.gemm_wgrad(at, bt, g)
        outs.append(g)
    close(outs[0], ref + 0.5, f"wgrad split-K {M}x{N}x{Kd}", rtol=1e-4, afrac=1e-5, scale="row")
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "split-K must not depend on arrival order"
    K.gemm_wgrad(at, bt, outs[0])                      # accumulates on top
    close(outs[0], 2 * ref + 0.5, "wgrad accumulate twice", rtol=1e-4, afrac=1e-5, scale="row")
    # X reduction-major as autograd holds it ([K, N], a column sub-view): same answer, same determinism
    x_full = torch.zeros(Kd, N + 8, device=DEV, dtype=BF)
    x_full[:, :N] = bt.t()
    x = x_full[:, :N]
    outs = []
    for _ in range(2):
        g = torch.full((M, N), 0.5, device=DEV, dtype=torch.float32)
        K.gemm_wgrad(at, x, g, b_kmajor=True)
        outs.append(g)
    close(outs[0], ref + 0.5, f"wgrad k-major X {M}x{N}x{Kd}", rtol=1e-4, afrac=1e-5, scale="row")
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("E,M,N,Kd,kv", [(3, 2048, 256, 512, [512, 64, 200]), (4, 1024, 512, 320, [8, 320, 0, 168]),
                                         (5, 768, 256, 256, [256, 256, 104, 32, 256])])
def test_gemm_batched_kvalid_balanced_xcd_mapping(E, M, N, Kd, kv):
    """Batched fp32-accumulate NT GEMM with a live reduction length per batch (MoE expert weight gradients).  With a tile
    count per batch divisible by 8 the kernel deals every XCD 1/8 of EVERY batch's tiles, longest reduction first (cases 1, 2);
    otherwise the generic mapping (case 3).  Every (batch, tile) must be computed exactly once, over its own k_valid (a multiple of 8 here: the NT form reads whole 16-byte groups, callers zero-pad)."""
    a, b = rnd(E, M, Kd, seed=31, scale=0.1), rnd(E, N, Kd, seed=32, scale=0.1)
    base = torch.randn(E, M, N, device=DEV)
    out = base.clone()
    rows = torch.tensor(kv, device=DEV, dtype=torch.int32)
    K.gemm_nt(a, b, out=out, out_f32=True, accumulate=True, k_valid=rows)
    for e in range(E):
        ref = base[e].double() + a[e, :, :kv[e]].double() @ b[e, :, :kv[e]].double().T
        close(out[e], ref.float(), f"batch {e}", rtol=1e-4, afrac=1e-5, scale="row")


def test_long_k_few_tile_dgrad_takes_split_k():
    """ops.linear_dgrad: 33 x 8 tiles (one full round of the chip + a nearly empty second one) over K >= 32768 — the lm_head
    dgrad shape class — goes through the deterministic split-K entry point + a cast; same result as the plain NT GEMM."""
    from llavamod import ops
    M, N, Kd = 8208, 2048, 32768
    dy, wt = rnd(M, Kd, seed=21, scale=0.05), rnd(N, Kd, seed=22, scale=0.05)
    fw = type("FW", (), {"transposed": lambda self: wt, "w": torch.empty(Kd, 1)})()
    got = ops.linear_dgrad(dy, fw)
    ref = K.gemm_nt(dy, wt)
    assert got.dtype == BF and got.shape == (M, N)
    close(got, ref.float(), "split-K dgrad vs NT", rtol=2 ** -7, afrac=2 ** -8, scale="row")
    again = ops.linear_dgrad(dy, fw)
    assert torch.equal(got, again)                           # deterministic
    r64 = dy[:64].double().cpu() @ wt.double().cpu().T
    close(got[:64], r64.float().to(DEV), "split-K dgrad vs fp64", rtol=2 ** -7, afrac=2 ** -8, scale="row")


def test_transpose():
    for (R, C) in [(64, 64), (100, 72), (7, 8), (513, 1032)]:
        x = rnd(R, C, seed=R)
        t = K.transpose(x)
        Rp = (R + 7) // 8 * 8
        assert t.shape == (C, Rp)
        close(t[:, :R], x.t(), f"transpose {R}x{C}", rtol=0, afrac=0, scale="row")
        assert t[:, R:].abs().max().item() == 0 if Rp > R else True
    xb = rnd(3, 50, 40, seed=77)
    tb = K.transpose(xb)
    close(tb[:, :, :50], xb.transpose(1, 2), "batched transpose", rtol=0, afrac=0, scale="row")


# ------------------------------------------------------------------------------------------ row kernels
@pytest.mark.parametrize("T,H", [(5, 64), (33, 2048), (16, 4096), (3, 1024)])
def test_rmsnorm_fwd_bwd(T, H):
    x, res, w = rnd(T, H, seed=1), rnd(T, H, seed=2), (1 + 0.1 * torch.randn(H)).to(BF).to(DEV)
    eps = 1e-6
    y, rstd, h = K.rmsnorm_fwd(x, w, eps, res=res)
    hr = (x.float() + res.float()).to(BF)
    close(h, hr, "add", rtol=0, afrac=0)
    hf = hr.float()
    var = hf.pow(2).mean(-1, keepdim=True)
    yr = w.float() * (hf * torch.rsqrt(var + eps)).to(BF).float()
    close(y, yr, "rmsnorm fwd")
    close(rstd, torch.rsqrt(var + eps).squeeze(-1), "rstd", rtol=1e-5, afrac=1e-6)
    y2, _, h2 = K.rmsnorm_fwd(x, w, eps)
    assert h2 is x
    # backward vs autograd of the fp32 op
    dy, dres = rnd(T, H, seed=3), rnd(T, H, seed=4)
    hh = hr.float().requires_grad_(True)
    out = w.float() * (hh * torch.rsqrt(hh.pow(2).mean(-1, keepdim=True) + eps))
    out.backward(dy.float())
    dh = K.rmsnorm_bwd(dy, hr, w, rstd, dres=dres)
    close(dh, hh.grad + dres.float(), "rmsnorm bwd", rtol=2 ** -6, afrac=2 ** -7)


@pytest.mark.parametrize("T,H", [(1, 64), (63, 136), (1000, 2048), (4097, 4096)])
def test_rmsnorm_scale_gradient(T, H):
    """dw[c] = sum_t dy[t,c] * bf16(h[t,c] * rstd[t]) (Qwen2RMSNorm, modeling_qwen2.py:92-97) vs the fp32 sum; accumulates into
    an fp32 buffer, deterministic (two launches give identical bits)."""
    h, dy = rnd(T, H, seed=11), rnd(T, H, seed=12)
    rstd = torch.rsqrt(h.float().pow(2).mean(-1) + 1e-6).contiguous()
    ref = (dy.double() * (h.float() * rstd[:, None]).to(BF).double()).sum(0)
    base = torch.randn(H, device=DEV)
    dw = base.clone()
    K.rmsnorm_dw(dy, h, rstd, dw, accumulate=True)
    err = (dw.double() - base.double() - ref).abs().max().item()
    assert err <= 1e-5 * ref.abs().max().item() + 1e-6 * (T ** 0.5), err
    dw2 = torch.full((H,), 7.0, device=DEV)
    K.rmsnorm_dw(dy, h, rstd, dw2, accumulate=False)
    dw3 = torch.empty(H, device=DEV)
    K.rmsnorm_dw(dy, h, rstd, dw3, accumulate=False)
    assert torch.equal(dw2, dw3)
    assert (dw2.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item() + 1e-6 * (T ** 0.5)


def test_embedding_gradient_scatter_add():
    """dW[idx[r]] += d_embeds[r] for text rows (idx >= 0); image rows (idx <= -2) and padding (-1) do not touch the table."""
    V, H, R = 300, 136, 5000
    g = torch.Generator().manual_seed(5)
    idx = torch.randint(0, V, (R,), generator=g).to(torch.int32)
    idx[torch.rand(R, generator=g) < 0.3] = -1
    idx[torch.rand(R, generator=g) < 0.2] = -7
    idx[:40] = 3                                               # heavy collisions on one row
    d = rnd(R, H, seed=6)
    dW = torch.zeros(V, H, device=DEV)
    K.embed_wgrad(d, idx.to(DEV), dW)
    ref = torch.zeros(V, H, dtype=torch.float64)
    keep = idx >= 0
    ref.index_add_(0, idx[keep].long(), d.cpu().double()[keep])
    assert (dW.cpu().double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    K.embed_wgrad(d, idx.to(DEV), dW)                          # accumulates
    assert (dW.cpu().double() - 2 * ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_layernorm_fwd():
    T, H = 37, 1024
    x, w, b = rnd(T, H, seed=1), rnd(H, seed=2), rnd(H, seed=3)
    y = K.layernorm_fwd(x, w, b, 1e-5)
    close(y, F.layer_norm(x.float(), (H,), w.float(), b.float(), 1e-5), "layernorm")


def _rope_tables(maxpos, hd, theta=10000.0):
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2).float() / hd))
    fr = torch.outer(torch.arange(maxpos).float(), inv)
    emb = torch.cat((fr, fr), dim=-1)
    return emb.cos().to(BF).to(DEV), emb.sin().to(BF).to(DEV)


def _rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def test_rope_fwd_bwd():
    T, nh, nkv, hd = 50, 4, 2, 128
    ld = (nh + 2 * nkv) * hd
    qkv = rnd(T, ld, seed=1)
    cos, sin = _rope_tables(64, hd)
    pos = (torch.arange(T) % 37).to(torch.int32).to(DEV)
    ref = qkv.clone()
    x = qkv[:, :(nh + nkv) * hd].reshape(T, nh + nkv, hd)
    c, s = cos[pos.long()][:, None, :], sin[pos.long()][:, None, :]
    emb = (x * c) + (_rot_half(x) * s)            # bf16 arithmetic, like the reference
    ref[:, :(nh + nkv) * hd] = emb.reshape(T, -1)
    out = qkv.clone()
    K.rope_(out, cos, sin, pos, nh + nkv, hd)
    close(out, ref, "rope fwd", rtol=2 ** -8, afrac=2 ** -9)
    # backward = transpose map: <rope(x), g> == <x, rope_bwd(g)>
    g = rnd(T, ld, seed=2)
    gb = g.clone()
    K.rope_(gb, cos, sin, pos, nh + nkv, hd, backward=True)
    xf = x.float().requires_grad_(True)
    yy = xf * c.float() + _rot_half(xf) * s.float()
    yy.backward(g[:, :(nh + nkv) * hd].reshape(T, nh + nkv, hd).float())
    close(gb[:, :(nh + nkv) * hd], xf.grad.reshape(T, -1), "rope bwd", rtol=2 ** -6, afrac=2 ** -7)
    close(gb[:, (nh + nkv) * hd:], g[:, (nh + nkv) * hd:], "rope bwd leaves V", rtol=0, afrac=0)


@pytest.mark.parametrize("B,S,nh,nkv,causal,ragged", [(2, 300, 4, 2, True, False), (1, 1100, 4, 1, True, False),
                                                       (2, 520, 2, 2, False, True), (3, 257, 3, 3, True, True)])
def test_attn_bwd_with_fused_rope_is_bit_identical_to_bwd_plus_rope(B, S, nh, nkv, causal, ragged):
    """lmod_attn_bwd_rope (the rotary embedding's gradient map applied to dQ / dK in the hd-128 backward kernels' epilogues, where a
    lane holds both halves of every rotate_half pair) == lmod_attn_bwd followed by lmod_rope(backward) on the d(QKV) buffer, bit for
    bit: causal / non-causal, GQA, ragged key masks, odd lengths, per-sample positions; dV untouched."""
    hd = 128
    ld = (nh + 2 * nkv) * hd
    qkv = rnd(B * S, ld, seed=S + 1, scale=1.0)
    q2, k2, v2 = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
    seqlens = torch.tensor([S - 17, max(1, S // 3), S][:B], dtype=torch.int32, device=DEV) if ragged else None
    scale = 1.0 / math.sqrt(hd)
    o, lse = K.attn_fwd(q2, k2, v2, B, S, nh, nkv, hd, scale, causal, seqlens)
    do = rnd(B * S, nh * hd, seed=S + 2)
    cos, sin = _rope_tables(4096, hd, theta=1e6)
    pos = ((torch.arange(B * S) * 3) % 2500).to(torch.int32).to(DEV)
    ref = torch.zeros(B * S, ld, device=DEV, dtype=BF)
    K.attn_bwd(q2, k2, v2, o, do, lse, ref[:, :nh * hd], ref[:, nh * hd:(nh + nkv) * hd], ref[:, (nh + nkv) * hd:], B, S, nh, nkv, hd,
               scale, causal, seqlens, split=False)      # the fused form is an unsplit launch; a head-split one re-associates fp32 sums
    K.rope_(ref, cos, sin, pos, nh + nkv, hd, backward=True)
    out = torch.zeros(B * S, ld, device=DEV, dtype=BF)
    assert K.attn_bwd_rope_fusable(hd)
    K.attn_bwd(q2, k2, v2, o, do, lse, out[:, :nh * hd], out[:, nh * hd:(nh + nkv) * hd], out[:, (nh + nkv) * hd:], B, S, nh, nkv, hd,
               scale, causal, seqlens, rope=(cos, sin, pos))
    assert torch.equal(out, ref), (out.float() - ref.float()).abs().max().item()
    assert float(out.float().abs().max()) > 0


@pytest.mark.parametrize("B,S,nh,nkv,causal,lens,rope", [
    (2, 512, 4, 4, True, None, False),
    (2, 512, 4, 4, True, None, True),              # + the rotary embedding's gradient map in the dQ GEMM's epilogue
    (1, 2048, 2, 2, True, None, True),             # 8 query blocks: reductions of 256 .. 2048 keys
    (2, 768, 2, 2, False, None, False),            # non-causal: every block reduces over all keys
    (3, 1024, 4, 2, True, [1007, 300, 1024], True),   # GQA group of 2, ragged: a length inside a block, one that leaves whole key blocks unwritten
    (2, 768, 2, 2, False, [500, 768], False),      # non-causal ragged (grouped-query launches WITHOUT fused RoPE at test sizes take the
                                                   # head-split form: the library prefers filling the CUs; GQA + spill is the case above)
    (1, 256, 8, 8, True, None, False),             # one block per head, 8 heads (the XCD-aligned id order)
])
def test_attn_bwd_ds_spill_form_matches_the_two_kernel_form(monkeypatch, B, S, nh, nkv, causal, lens, rope):
    """Round 6: with the workspace the dK/dV kernel spills dS^T and dQ = scale * dS K is one batched TN GEMM (5 matmuls instead of 7).
    Against the two-kernel form on the same inputs: dK and dV BIT-IDENTICAL (the same kernel body; the spill only stores), dQ equal
    to fp32 summation order (the same bf16 dS values, reduced in a different order) and within the usual tolerance of the fp32 torch
    reference.  The workspace is filled with NaN patterns first: every element the GEMM reads must have been written by this launch."""
    hd = 128
    ld = (nh + 2 * nkv) * hd
    qkv = rnd(B * S, ld, seed=S + nh, scale=1.0)
    q2, k2, v2 = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
    seqlens = torch.tensor(lens, dtype=torch.int32, device=DEV) if lens else None
    scale = 1.0 / math.sqrt(hd)
    o, lse = K.attn_fwd(q2, k2, v2, B, S, nh, nkv, hd, scale, causal, seqlens)
    do = rnd(B * S, nh * hd, seed=S + 2)
    if lens:                                         # padded query rows carry no gradient in the step (their loss weight is zero)
        rowmask = (torch.arange(S, device=DEV)[None, :] < seqlens[:, None]).reshape(B * S, 1)
        do = do * rowmask.to(BF)
    rp = None
    if rope:
        cos, sin = _rope_tables(4096, hd, theta=1e6)
        rp = (cos, sin, ((torch.arange(B * S) * 3) % 2500).to(torch.int32).to(DEV))

    def run(ds):
        monkeypatch.setenv("LMOD_ATTN_DS", "1" if ds else "0")
        assert K.attn_bwd_ds_fusable(B, S, nh, hd) == ds
        if ds:
            K._ds_workspace(torch.device(DEV, torch.cuda.current_device()), B * nh * S * S * 2).fill_(0xFF)       # bf16 NaN everywhere
        out = torch.zeros(B * S, ld, device=DEV, dtype=BF)
        K.attn_bwd(q2, k2, v2, o, do, lse, out[:, :nh * hd], out[:, nh * hd:(nh + nkv) * hd], out[:, (nh + nkv) * hd:], B, S, nh, nkv, hd,
                   scale, causal, seqlens, rope=rp, split=False)
        return out

    two, one = run(False), run(True)
    assert torch.isfinite(one.float()).all()
    assert torch.equal(one[:, nh * hd:], two[:, nh * hd:]), "dK / dV must not change"
    dq1, dq2 = one[:, :nh * hd].float(), two[:, :nh * hd].float()
    tol = 2 ** -7 * dq2.abs().max().item()           # one bf16 ulp of the largest element: a different fp32 summation order
    assert (dq1 - dq2).abs().max().item() <= tol, ((dq1 - dq2).abs().max().item(), tol)
    assert dq2.abs().max().item() > 0
    if not rope:
        qf = q2.float().reshape(B, S, nh, hd).requires_grad_(True)
        kf = k2.float().reshape(B, S, nkv, hd).requires_grad_(True)
        vf = v2.float().reshape(B, S, nkv, hd).requires_grad_(True)
        oref, _ = _attn_ref(qf, kf, vf, scale, causal, seqlens)
        oref = torch.nan_to_num(oref)
        oref.backward(do.float().reshape(B, S, nh, hd))
        close(one[:, :nh * hd].reshape(B, S, nh, hd), torch.nan_to_num(qf.grad), "attn dQ (dS spill)", rtol=2 ** -5, afrac=2 ** -6)


@pytest.mark.parametrize("rows", [[3000, 0, 1793, 257], [3000] * 4, [5, 0, 0, 300], [256] * 16])
def test_grouped_swiglu_backward_on_the_persistent_walk_is_bit_identical(rows, monkeypatch):
    """Round 6: the GROUPED fused SwiGLU backward (MoE experts' down-projection dgrad with dgate / dup in the epilogue) on the persistent
    grouped walk of the four-wave kernel (LMOD_GEMM_SB4G, default on) against the 8-wave one-tile-per-workgroup instantiation: bit-identical
    [dgate | dup], rows up to the next multiple of 8 zeroed, dead rows untouched; ragged counts with an empty expert and a 1-row tile tail,
    full slabs, almost everything dead, 16 experts of one row tile."""
    E, C = len(rows), 3000 if max(rows) > 256 else 256
    H, I = 512, 1280
    dy, wt, gu = rnd(E, C, H, seed=70), rnd(E, I, H, seed=71), rnd(E, C, 2 * I, seed=72)
    mv = torch.tensor(rows, device=DEV, dtype=torch.int32)
    monkeypatch.setenv("LMOD_GEMM_PERSIST", "1")
    monkeypatch.setenv("LMOD_GEMM_PERSIST_ROUNDS", "0")

    def run():
        out = torch.full((E, C, 2 * I), 7.0, device=DEV, dtype=BF)
        K.gemm_swiglu_bwd(dy, wt, gu, out=out, m_valid=mv, K=H)
        return out
    monkeypatch.setenv("LMOD_GEMM_SB4G", "1")
    a = run()
    monkeypatch.setenv("LMOD_GEMM_SB4G", "0")
    b = run()
    assert torch.equal(a, b)
    for e in range(E):
        n = rows[e]; n8 = min((n + 7) // 8 * 8, C)
        assert n8 == n or a[e, n:n8].abs().max().item() == 0
        assert n8 == C or (a[e, n8:] == 7.0).all(), "dead rows must not be written"
        if n:
            ref = K.gemm_swiglu_bwd(dy[e, :n].contiguous(), wt[e], gu[e, :n].contiguous(), K=H)
            assert torch.equal(a[e, :n], ref), f"expert {e}"


@pytest.mark.parametrize("T,nh,nkv,Kd", [(1000, 4, 4, 512), (777, 6, 2, 256), (2048, 16, 16, 2048)])
def test_fused_qkv_rope_gemm_is_bit_identical_to_gemm_plus_rope(T, nh, nkv, Kd):
    """lmod_gemm_qkv_rope_bf16 (rotary embedding in the QKV GEMM's epilogue: rounded acc + bias swapped between neighbouring
    wave columns through LDS, rope_kernel's roundings) == lmod_gemm_bf16_nt + lmod_rope, bit for bit: rows not a multiple of the
    tile, GQA widths, the V columns untouched, and against the bf16 reference arithmetic of apply_rotary_pos_emb."""
    hd = 128
    N = (nh + 2 * nkv) * hd
    x = rnd(T, Kd, seed=3) * 0.5
    w = rnd(N, Kd, seed=4) * 0.05
    b = rnd(N, seed=5)
    cos, sin = _rope_tables(4096, hd)
    pos = ((torch.arange(T) * 7) % 3000).to(torch.int32).to(DEV)
    assert K.qkv_rope_fusable(x, w, nh + nkv, hd)
    ref = K.gemm_nt(x, w, bias=b)
    plain = ref.clone()
    K.rope_(ref, cos, sin, pos, nh + nkv, hd)
    out = K.gemm_qkv_rope(x, w, b, cos, sin, pos, nh + nkv)
    assert torch.equal(out, ref), (out.float() - ref.float()).abs().max().item()
    assert torch.equal(out[:, (nh + nkv) * hd:], plain[:, (nh + nkv) * hd:])
    xq = plain[:, :(nh + nkv) * hd].reshape(T, nh + nkv, hd)
    c, s_ = cos[pos.long()][:, None, :], sin[pos.long()][:, None, :]
    close(out[:, :(nh + nkv) * hd], ((xq * c) + (_rot_half(xq) * s_)).reshape(T, -1), "fused qkv rope vs bf16 reference",
          rtol=2 ** -8, afrac=2 ** -9)
    # without a bias, into a preallocated strided buffer
    big = torch.zeros(T, N + 64, device=DEV, dtype=BF)
    K.gemm_qkv_rope(x, w, None, cos, sin, pos, nh + nkv, out=big[:, :N])
    ref2 = K.gemm_nt(x, w)
    K.rope_(ref2, cos, sin, pos, nh + nkv, hd)
    assert torch.equal(big[:, :N], ref2) and float(big[:, N:].abs().max()) == 0.0


def test_swiglu_gelu_add():
    T, I = 77, 5504
    gu = rnd(T, 2 * I, seed=1)
    gate, up = gu[:, :I], gu[:, I:]
    out = K.swiglu_fwd(gate, up)
    ref = (F.silu(gate.float()).to(BF).float() * up.float())
    close(out, ref, "swiglu fwd")
    d = rnd(T, I, seed=2)
    gf, uf = gate.float().requires_grad_(True), up.float().requires_grad_(True)
    (F.silu(gf) * uf).backward(d.float())
    dg, du = K.swiglu_bwd(d, gate, up)
    close(dg, gf.grad, "swiglu dgate", rtol=2 ** -6, afrac=2 ** -8)
    close(du, uf.grad, "swiglu dup", rtol=2 ** -6, afrac=2 ** -8)
    x = rnd(64, 2048, seed=3)
    close(K.gelu_fwd(x), F.gelu(x.float()), "gelu fwd")
    xf = x.float().requires_grad_(True)
    dy = rnd(64, 2048, seed=4)
    F.gelu(xf).backward(dy.float())
    close(K.gelu_bwd(dy, x), xf.grad, "gelu bwd", rtol=2 ** -6, afrac=2 ** -8)
    a, b = rnd(9, 2048, seed=5), rnd(9, 2048, seed=6)
    close(K.add(a, b), (a.float() + b.float()).to(BF), "add", rtol=0, afrac=0)


def test_gather_im2col_vitembed_adamw():
    H = 256
    ta, tb = rnd(50, H, seed=1), rnd(20, H, seed=2)
    idx = torch.tensor([3, -1, -2, 49, -21, 0, -5], dtype=torch.int32, device=DEV)
    out = K.gather_rows(ta, tb, idx, H)
    ref = torch.stack([ta[3], torch.zeros(H, device=DEV, dtype=BF), tb[0], ta[49], tb[19], ta[0], tb[3]])
    close(out, ref, "gather", rtol=0, afrac=0)
    B, S, P = 2, 28, 14
    pix = rnd(B, 3, S, S, seed=3)
    kp = 592
    cols = K.im2col_patch(pix, P, kp)
    ref = F.unfold(pix.float(), kernel_size=P, stride=P).transpose(1, 2).reshape(B * 4, 3 * P * P)
    close(cols[:, :588], ref, "im2col", rtol=0, afrac=0)
    assert cols[:, 588:].abs().max() == 0
    D, NP = 64, 4
    pe, cls, pos = rnd(B * NP, D, seed=4), rnd(D, seed=5), rnd(NP + 1, D, seed=6)
    tok = K.vit_embed(pe, cls, pos, B, NP).reshape(B, NP + 1, D)
    ref = torch.cat([cls.float().expand(B, 1, D), pe.float().reshape(B, NP, D)], 1) + pos.float()
    close(tok, ref.to(BF), "vit_embed", rtol=0, afrac=0)
    n = 1000
    p0 = torch.randn(n, device=DEV)
    g = torch.randn(n, device=DEV)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref_p], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1)
    master, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    pb = p0.to(BF)
    for step in (1, 2, 3):
        ref_p.grad = g.clone()
        opt.step()
        K.adamw_step(master, pb, g, m, v, 1e-2, 0.9, 0.999, 1e-8, 0.1, step)
    close(master, ref_p.detach(), "adamw", rtol=1e-5, afrac=1e-6)
    close(pb, master.to(BF), "adamw bf16 copy", rtol=0, afrac=0)
    # zero_grad: same update, gradient cleared in the same pass
    m2, v2, ms2, pb2, g2 = m.clone(), v.clone(), master.clone(), pb.clone(), g.clone()
    K.adamw_step(master, pb, g, m, v, 1e-2, 0.9, 0.999, 1e-8, 0.1, 4)
    K.adamw_step(ms2, pb2, g2, m2, v2, 1e-2, 0.9, 0.999, 1e-8, 0.1, 4, zero_grad=True)
    assert torch.equal(master, ms2) and torch.equal(pb, pb2) and torch.equal(m, m2) and torch.equal(v, v2)
    assert g2.abs().max().item() == 0 and torch.equal(g, g2 + g)


# ------------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v, scale, causal, seqlens):
    # q [B,S,nh,hd], k/v [B,S,nkv,hd] fp32
    B, S, nh, hd = q.shape
    nkv = k.shape[2]
    rep = nh // nkv
    kk = k.repeat_interleave(rep, dim=2)
    vv = v.repeat_interleave(rep, dim=2)
    s = torch.einsum("bqhd,bkhd->bhqk", q, kk) * scale
    mask = torch.zeros(B, 1, S, S, dtype=torch.bool, device=q.device)
    if causal:
        mask |= torch.triu(torch.ones(S, S, dtype=torch.bool, device=q.device), 1)[None, None]
    if seqlens is not None:
        mask |= (torch.arange(S, device=q.device)[None, :] >= seqlens[:, None].long())[:, None, None, :]
    s = s.masked_fill(mask, float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = torch.einsum("bhqk,bkhd->bqhd", p, vv)
    return o, torch.logsumexp(s, dim=-1)


@pytest.mark.parametrize("B,S,nh,nkv,hd,causal,ragged", [
    (2, 256, 4, 4, 128, True, False),
    (2, 200, 4, 2, 128, True, True),
    (1, 577, 4, 4, 64, False, False),
    (2, 96, 2, 2, 64, True, True),
    (1, 2048, 2, 2, 128, True, False),
    (2, 300, 2, 1, 128, False, True),          # hd-128 forward kernel: non-causal, ragged keys, S % 32 != 0, GQA
    (1, 1000, 3, 3, 128, True, False),         # S % 64 != 0, several query blocks, odd head count
    (2, 77, 2, 2, 128, False, False),          # shorter than one query block
    (1, 1100, 4, 1, 128, True, False),         # 5 owner blocks (odd: the middle one is a single pass), GQA group of 4
    (2, 520, 4, 1, 128, False, True),          # non-causal GQA group of 4 with a ragged key mask, 3 owner blocks
    (1, 1100, 4, 1, 64, True, False),          # head dim 64 on the one-wave-per-SIMD backward: odd block count, GQA group of 4
    (2, 520, 4, 2, 64, False, True),           # hd 64, non-causal, ragged key mask, GQA
    (1, 2048, 2, 2, 64, True, False),          # hd 64, 8 owner blocks (4 causal pairs)
    (2, 333, 14, 2, 64, True, True),           # the Qwen2-0.5B student's head geometry (14 heads, 2 KV heads), ragged
])
def test_attn_fwd_bwd(B, S, nh, nkv, hd, causal, ragged):
    ld = (nh + 2 * nkv) * hd
    qkv = rnd(B * S, ld, seed=S, scale=1.0)
    q2, k2, v2 = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
    seqlens = None
    if ragged:
        seqlens = torch.tensor([S - 17, max(1, S // 3)][:B], dtype=torch.int32, device=DEV)
    scale = 1.0 / math.sqrt(hd)
    o, lse = K.attn_fwd(q2, k2, v2, B, S, nh, nkv, hd, scale, causal, seqlens)
    qf = q2.float().reshape(B, S, nh, hd).requires_grad_(True)
    kf = k2.float().reshape(B, S, nkv, hd).requires_grad_(True)
    vf = v2.float().reshape(B, S, nkv, hd).requires_grad_(True)
    oref, lref = _attn_ref(qf, kf, vf, scale, causal, seqlens)
    close(o.reshape(B, S, nh, hd), oref, "attn fwd", rtol=2 ** -6, afrac=2 ** -7)
    close(lse, lref, "attn lse", rtol=1e-3, afrac=1e-3)
    do = rnd(B * S, nh * hd, seed=S + 1)
    oref.backward(do.float().reshape(B, S, nh, hd))
    dqkv = torch.zeros_like(qkv)
    K.attn_bwd(q2, k2, v2, o, do, lse, dqkv[:, :nh * hd], dqkv[:, nh * hd:(nh + nkv) * hd],
               dqkv[:, (nh + nkv) * hd:], B, S, nh, nkv, hd, scale, causal, seqlens)
    # the kernel recomputes P from bf16 O / lse and rounds dS to bf16: allow 2^-5 of the tensor's scale
    close(dqkv[:, :nh * hd].reshape(B, S, nh, hd), qf.grad, "attn dQ", rtol=2 ** -5, afrac=2 ** -6)
    close(dqkv[:, nh * hd:(nh + nkv) * hd].reshape(B, S, nkv, hd), kf.grad, "attn dK", rtol=2 ** -5, afrac=2 ** -6)
    close(dqkv[:, (nh + nkv) * hd:].reshape(B, S, nkv, hd), vf.grad, "attn dV", rtol=2 ** -5, afrac=2 ** -6)
    if nh > nkv:
        # grouped-query shapes: the launch above cut every group of query heads into parts (fp32 partial sums + a reduction
        # kernel, lmod_attn_bwd_split); the unsplit launch passes the same test, leaves dQ bit-identical and dK / dV within fp32
        # re-association (a few bf16 ulps of the largest element)
        from llavamod import _hip
        ns = _hip.load().lmod_attn_bwd_nsplit(B, S, nh, nkv, hd, int(causal))
        assert ns > 1 or os.environ.get("LMOD_ATTN_BWD") == "1" or (hd == 64 and os.environ.get("LMOD_ATTN_BWD64") == "1"), ns
        one = torch.zeros_like(qkv)
        K.attn_bwd(q2, k2, v2, o, do, lse, one[:, :nh * hd], one[:, nh * hd:(nh + nkv) * hd],
                   one[:, (nh + nkv) * hd:], B, S, nh, nkv, hd, scale, causal, seqlens, split=False)
        close(one[:, nh * hd:(nh + nkv) * hd].reshape(B, S, nkv, hd), kf.grad, "attn dK (unsplit)", rtol=2 ** -5, afrac=2 ** -6)
        close(one[:, (nh + nkv) * hd:].reshape(B, S, nkv, hd), vf.grad, "attn dV (unsplit)", rtol=2 ** -5, afrac=2 ** -6)
        assert torch.equal(one[:, :nh * hd], dqkv[:, :nh * hd])
        d = (one[:, nh * hd:].float() - dqkv[:, nh * hd:].float()).abs().max().item()
        assert d <= 2 ** -7 * dqkv[:, nh * hd:].float().abs().max().item(), d


def test_attn_forward_one_wave_per_simd_kernel_passes_the_same_tests():
    """attn_fwd3.hip (round 5: 4 waves x 64 queries, one wave per SIMD, O^T and Q in asm-owned accumulators, K / V by LDS-DMA) is selected
    by LMOD_ATTN_FWD=3, read once per process: every attention test of this file — forward values and lse against fp32 torch, and the
    backward fed by ITS outputs — is re-run in a child process under the switch."""
    import subprocess
    from llavamod import _hip
    if os.environ.get("LMOD_ATTN_FWD"):
        pytest.skip("already inside a run that selects a forward kernel")
    # a LAB arm since round 6 (1 - 6 % behind the shipped forward): it lives in liblmod_hip_lab.so (`make LAB=1`), not in the product library
    lab = os.path.join(os.path.dirname(_hip.LIB_PATH), "liblmod_hip_lab.so")
    if not os.path.exists(lab):
        pytest.skip("lab library not built (make -C llava-mod_amd/csrc LAB=1)")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k", "attn and not one_wave and not lab_arm"],
                       env=dict(os.environ, LMOD_ATTN_FWD="3", LMOD_HIP_LIB=lab), capture_output=True, text=True, timeout=900)
    tail = r.stdout[-600:]
    assert r.returncode == 0 and " passed" in tail and "failed" not in tail, tail


def test_product_library_refuses_lab_arm_requests():
    """The product library does not contain the A/B arms (generic attention kernels, attn_fwd3): a process that asks for one through the
    environment gets LMOD_EUNSUPPORTED from the entry point — loud, never a silent run of the default kernel under an A/B label."""
    import subprocess
    code = ("import sys, torch; sys.path.insert(0, %r); from llavamod import kernels as K\n"
            "q = torch.zeros(256, 128, device='cuda', dtype=torch.bfloat16)\n"
            "try:\n    K.attn_fwd(q, q, q, 1, 256, 1, 1, 128, 0.1, True)\n    print('RAN')\n"
            "except RuntimeError as e:\n    print('REFUSED' if 'UNSUPPORTED' in str(e) else repr(e))\n") % os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llava-mod_amd")
    env = {k: v for k, v in os.environ.items() if k != "LMOD_HIP_LIB"}
    for arm in ("1", "3"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(env, LMOD_ATTN_FWD=arm), capture_output=True, text=True, timeout=300)
        assert "REFUSED" in r.stdout, (arm, r.stdout[-300:], r.stderr[-600:])


def test_attn_online_softmax_rescale_branch():
    # force a late, large max: one key far down the sequence dominates one query row (guide rule 26)
    B, S, nh, hd = 1, 512, 1, 128
    q = rnd(B * S, hd, seed=1, scale=0.3)
    k = rnd(B * S, hd, seed=2, scale=0.3)
    v = rnd(B * S, hd, seed=3)
    k[400] = (q[450].float() * 40).to(BF)
    o, lse = K.attn_fwd(q, k, v, B, S, nh, nh, hd, 1 / math.sqrt(hd), True)
    oref, lref = _attn_ref(q.float().reshape(B, S, nh, hd), k.float().reshape(B, S, nh, hd),
                           v.float().reshape(B, S, nh, hd), 1 / math.sqrt(hd), True, None)
    close(o.reshape(B, S, nh, hd), oref, "attn spike", rtol=2 ** -6, afrac=2 ** -7)
    close(lse, lref, "attn spike lse", rtol=1e-3, afrac=1e-3)


def test_attn_fwd_hd128_edge_cases():
    """The 32x32x16 forward kernel (attn_fwd2.hip): a sample with NO visible key (seqlens 0) must give zeros / lse = -inf,
    a spike well above the deferred-rescale threshold late in the sequence must rescale correctly for every row of the wave,
    and a max that creeps up by less than the threshold per tile (deferred all the way) must still be exact."""
    B, S, nh, hd = 2, 320, 2, 128
    qkv = rnd(B * S, 3 * nh * hd, seed=11)
    q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:2 * nh * hd], qkv[:, 2 * nh * hd:]
    sl = torch.tensor([0, 200], dtype=torch.int32, device=DEV)
    o, lse = K.attn_fwd(q, k, v, B, S, nh, nh, hd, 1 / math.sqrt(hd), True, sl)
    assert o.reshape(B, S, nh * hd)[0].abs().max().item() == 0 and torch.isinf(lse[0]).all() and (lse[0] < 0).all()
    oref, lref = _attn_ref(q.float().reshape(B, S, nh, hd)[1:], k.float().reshape(B, S, nh, hd)[1:],
                           v.float().reshape(B, S, nh, hd)[1:], 1 / math.sqrt(hd), True, sl[1:])
    close(o.reshape(B, S, nh, hd)[1:], oref, "attn len0/len200", rtol=2 ** -6, afrac=2 ** -7)
    close(lse[1:], lref, "attn len0/len200 lse", rtol=1e-3, afrac=1e-3)
    # creeping max: key t's score for every query grows by ~1.5 (log2 units ~2.2 < threshold 6) per 64-key tile
    B, S, nh = 1, 1024, 1
    g = torch.Generator().manual_seed(5)
    qv = torch.randn(hd, generator=g)
    qv = qv / qv.norm()
    q = (qv[None, :] * math.sqrt(hd) + 0.05 * torch.randn(S, hd, generator=g)).to(BF).to(DEV)
    kk = (qv[None, :] * (torch.arange(S)[:, None] // 64).float() * 1.5 + 0.05 * torch.randn(S, hd, generator=g)).to(BF).to(DEV)
    vv = rnd(S, hd, seed=6)
    for causal in (True, False):
        o, lse = K.attn_fwd(q, kk, vv, B, S, nh, nh, hd, 1 / math.sqrt(hd), causal)
        oref, lref = _attn_ref(q.float().reshape(B, S, nh, hd), kk.float().reshape(B, S, nh, hd),
                               vv.float().reshape(B, S, nh, hd), 1 / math.sqrt(hd), causal, None)
        close(o.reshape(B, S, nh, hd), oref, f"attn creeping max causal={causal}", rtol=2 ** -6, afrac=2 ** -7)
        close(lse, lref, "attn creeping max lse", rtol=1e-3, afrac=1e-3)


# ------------------------------------------------------------------------------------------ MoE
@pytest.mark.parametrize("T,E,cf,noise", [(64, 4, 1.5, False), (1000, 4, 1.5, True), (777, 8, 1.0, True),
                                          (2048, 4, 0.5, True)])
def test_moe_gate_top2_matches_deepspeed_restatement(T, E, cf, noise):
    H = 128
    x = rnd(T, H, seed=T)
    wg = (torch.randn(E, H) * 0.5).to(DEV)
    logits = K.moe_router_fwd(x, wg)
    close(logits, x.float() @ wg.t(), "router logits", rtol=1e-4, afrac=1e-5)
    nz = omoe.gumbel_noise((T, E), torch.Generator().manual_seed(2)).to(DEV) if noise else None
    C = omoe.capacity(T, E, cf * 2, 0)
    st = K.moe_gate(logits, 2, C, nz)
    l_aux, combine, dispatch, cnt = omoe.top2gating(logits, cf, 0, nz)
    assert combine.shape[2] == C
    close(st.l_aux[0], l_aux, "l_aux", rtol=1e-4, afrac=1e-6)
    assert torch.equal(st.exp_counts.long(), cnt.long())
    # rebuild the dense [S,E,C] combine tensor from the index maps
    dense = torch.zeros(T, E * C, device=DEV)
    for slot, w in ((st.slot1, st.w1), (st.slot2, st.w2)):
        keep = slot >= 0
        dense[torch.nonzero(keep).squeeze(1), slot[keep].long()] = w[keep]
    close(dense.reshape(T, E, C), combine, "combine weights", rtol=1e-5, afrac=1e-6)
    # slot_token is the inverse map
    live = st.slot_token >= 0
    assert int(live.sum()) == int(dispatch.sum())
    tok = st.slot_token[live].long()
    sl = torch.nonzero(live).squeeze(1)
    assert torch.all((st.slot1[tok].long() == sl) | (st.slot2[tok].long() == sl))


def test_moe_combine_dispatch_and_backward():
    T, E, H, cf = 300, 4, 256, 1.0
    x = rnd(T, H, seed=1)
    wg = (torch.randn(E, H) * 0.5).to(DEV)
    logits = K.moe_router_fwd(x, wg)
    C = omoe.capacity(T, E, cf * 2, 0)
    st = K.moe_gate(logits, 2, C, None)
    disp = K.gather_rows(x, None, st.slot_token, H)                 # [E*C, H]
    _, combine, dispatch, _ = omoe.top2gating(logits, cf, 0, None)
    ref_disp = torch.einsum("sec,sm->ecm", dispatch.float(), x.float()).reshape(E * C, H)
    close(disp, ref_disp, "dispatch", rtol=0, afrac=0)
    y = rnd(E * C, H, seed=2)
    out = K.moe_combine_fwd(y, st, H)
    ref = torch.einsum("sec,ecm->sm", combine.to(BF).float(), y.float().reshape(E, C, H))
    close(out, ref, "combine fwd")
    # backward pieces vs autograd of the dense formulation
    dout = rnd(T, H, seed=3)
    lg = logits.clone().requires_grad_(True)
    yf = y.float().requires_grad_(True)
    l_aux, comb, _, _ = omoe.top2gating(lg, cf, 0, None)
    o = torch.einsum("sec,ecm->sm", comb, yf.reshape(E, C, H))
    (o * dout.float()).sum().backward(retain_graph=True)
    g_main = lg.grad.clone()
    lg.grad = None
    (l_aux * 3.0).backward()
    g_aux = lg.grad.clone()
    dy, dw1, dw2 = K.moe_combine_bwd(dout, y, st, H)
    close(dy, yf.grad, "combine dy", rtol=2 ** -6, afrac=2 ** -7)
    dla = torch.tensor([3.0], device=DEV)
    dlog = K.moe_gate_bwd(st, dw1, dw2, dla)
    close(dlog, g_main + g_aux, "gate dlogits", rtol=2e-3, afrac=2e-3)
    # router wgrad / dispatch bwd
    dwg = torch.zeros(E, H, device=DEV)
    K.moe_router_wgrad(x, dlog, dwg, False)
    close(dwg, dlog.t() @ x.float(), "router wgrad", rtol=1e-4, afrac=1e-5)
    d_in = rnd(E * C, H, seed=4)
    dx = K.moe_dispatch_bwd(d_in, st, dlog, wg, H)
    ref_dx = torch.einsum("sec,ecm->sm", dispatch.float(), d_in.float().reshape(E, C, H)) + (dlog @ wg).to(BF).float()
    close(dx, ref_dx, "dispatch bwd")


# ------------------------------------------------------------------------------------------ losses
def test_rowloss_fwd_bwd():
    R, Vs, Vt, Va = 9, 1024, 1152, 1000 // 8 * 8
    s = rnd(R, Vs, seed=1, scale=2.0)
    t = rnd(R, Vt, seed=2, scale=2.0)
    s[0, 5] = float("-inf")                        # exercises the isinf(logp) mask
    label = torch.tensor([3, -100, 1023, 7, -100, 0, 999, 1000, 12], dtype=torch.int32, device=DEV)
    st = K.rowloss_fwd(s, Vs, t, Va, label)
    sf, tf = s.float(), t.float()
    logp = F.log_softmax(sf[:, :Va], -1)
    p = F.softmax(tf[:, :Va], -1)
    x = torch.where(torch.isinf(logp), torch.zeros_like(logp), p * logp).sum(-1)
    close(st[:, 3], x, "x_kd", rtol=2e-4, afrac=1e-5)
    lse_full = torch.logsumexp(sf, -1)
    close(st[:, 0], lse_full, "lse full", rtol=1e-5, afrac=1e-6)
    close(st[:, 1], torch.logsumexp(sf[:, :Va], -1), "lse align", rtol=1e-5, afrac=1e-6)
    close(st[:, 2], torch.logsumexp(tf[:, :Va], -1), "lse teacher", rtol=1e-5, afrac=1e-6)
    valid = label >= 0
    ce = torch.zeros(R, device=DEV)
    ce[valid] = lse_full[valid] - sf[valid, label[valid].long()]
    close(st[:, 4], ce, "ce", rtol=1e-4, afrac=1e-5)
    # backward on finite logits
    s2 = rnd(R, Vs, seed=3, scale=2.0)
    st2 = K.rowloss_fwd(s2, Vs, t, Va, label)
    kd_w = torch.tensor([1, 1, 0, 1, 0, 1, 1, 0, 1], dtype=torch.float32, device=DEV)
    ce_w = valid.float()
    kd_scale = torch.tensor([0.37], device=DEV)
    ce_scale = torch.tensor([-0.21], device=DEV)
    ds = torch.empty_like(s2)
    K.rowloss_bwd(s2, Vs, t, Va, label, st2, kd_w, ce_w, None, kd_scale, ce_scale, ds)
    sg = s2.float().requires_grad_(True)
    lp = F.log_softmax(sg[:, :Va], -1)
    xk = (p * lp).sum(-1)
    lsf = torch.logsumexp(sg, -1)
    cer = torch.zeros(R, device=DEV)
    cer = torch.where(valid, lsf - sg.gather(1, label.clamp_min(0).long()[:, None]).squeeze(1), cer)
    # d/ds of  sum_r [-kd_scale*kd_w*x  (sign: ds = ckd*(q - p) = -ckd * dx/ds)] + ce_scale*ce_w*ce
    obj = (-(kd_scale * kd_w) * xk).sum() + ((ce_scale * ce_w) * cer).sum()
    obj.backward()
    close(ds, sg.grad, "rowloss bwd", rtol=2 ** -6, afrac=2 ** -8)


def test_segment_wsum_and_dpo_loss():
    R = 1000
    vals = torch.randn(R, 8, device=DEV)
    w = (torch.rand(R, device=DEV) > 0.3).float()
    off = torch.tensor([0, 10, 10, 400, 1000], dtype=torch.int32, device=DEV)
    s, ws = K.segment_wsum(vals, 4, w, off)
    for b in range(4):
        lo, hi = int(off[b]), int(off[b + 1])
        assert abs(s[b].item() - (vals[lo:hi, 4] * w[lo:hi]).sum().item()) < 1e-3
        assert abs(ws[b].item() - w[lo:hi].sum().item()) < 1e-4
    # known answers captured from the reference's DPOTrainer.dpo_loss
    pc = torch.tensor([-10.0, -12.0], device=DEV); pr = torch.tensor([-11.0, -11.0], device=DEV)
    rc = torch.tensor([-10.5, -12.5], device=DEV); rr = torch.tensor([-10.0, -12.0], device=DEV)
    known = {"sigmoid": [0.6210, 0.7185], "hinge": [0.85, 1.05], "ipo": [12.25, 30.25],
             "kto_pair": [0.4875, 0.4875, 0.4626, 0.5125]}
    for lt, exp in known.items():
        losses, cr, rj, dpc, dpr = K.dpo_loss(pc, pr, rc, rr, 0.1, 0.0, lt)
        assert torch.allclose(losses.cpu(), torch.tensor(exp), atol=2e-4), (lt, losses)
        a, b = pc.clone().requires_grad_(True), pr.clone().requires_grad_(True)
        z = (a - b) - (rc - rr)
        if lt == "sigmoid":
            L = -F.logsigmoid(0.1 * z)
        elif lt == "hinge":
            L = torch.relu(1 - 0.1 * z)
        elif lt == "ipo":
            L = (z - 1 / (2 * 0.1)) ** 2
        else:
            ckl = (a - rc).mean().clamp(min=0); rkl = (b - rr).mean().clamp(min=0)
            L = torch.cat((1 - torch.sigmoid(0.1 * ((a - rc) - rkl)), 1 - torch.sigmoid(0.1 * (ckl - (b - rr)))), 0)
        L.mean().backward()
        assert torch.allclose(dpc, a.grad, atol=1e-5), (lt, dpc, a.grad)
        assert torch.allclose(dpr, b.grad, atol=1e-5), (lt, dpr, b.grad)
        assert torch.allclose(cr, 0.1 * (pc - rc)) and torch.allclose(rj, 0.1 * (pr - rr))


def test_abi_error_codes_and_empty_inputs():
    """C-ABI behaviour at the edges: empty problems are a no-op (0), bad shapes / alignment / pointers are LMOD_EINVAL (-1),
    shapes outside the compiled envelope LMOD_EUNSUPPORTED (-3); the Python binding turns them into exceptions."""
    from llavamod import _hip
    lib = _hip.load()
    s = torch.cuda.current_stream().cuda_stream
    a, b = rnd(64, 64, seed=80), rnd(64, 64, seed=81)
    c = torch.zeros(64, 64, device=DEV, dtype=BF)
    g = lambda M, N, Kd, lda=64, ldb=64, ldc=64, A=a, B=b, C=c: lib.lmod_gemm_bf16_nt(
        A.data_ptr() if A is not None else None, B.data_ptr(), C.data_ptr(), None, M, N, Kd, lda, ldb, ldc, 1, 0, 0, 0, None, None, 0, 0, 0, s)
    assert g(0, 64, 64) == 0 and g(64, 0, 64) == 0                      # empty: nothing launched, C untouched
    assert c.abs().max().item() == 0
    assert g(64, 64, 60) == -1                                          # K % 8
    assert g(64, 64, 64, lda=60) == -1 and g(64, 64, 64, ldc=32) == -1  # leading dims
    assert g(64, 64, 64, A=None) == -1                                  # NULL pointer
    assert lib.lmod_gemm_bf16_nt(a.data_ptr() + 2, b.data_ptr(), c.data_ptr(), None, 64, 64, 64, 64, 64, 64, 1, 0, 0, 0,
                                 None, None, 0, 0, 0, s) == -1          # misaligned A
    assert g(64, 64, 64) == 0
    q = rnd(128, 3 * 96, seed=82)
    o = torch.empty(128, 96, device=DEV, dtype=BF)
    assert lib.lmod_attn_fwd(q.data_ptr(), q.data_ptr(), q.data_ptr(), o.data_ptr(), None, None, None, 1, 128, 1, 1, 96, 288, 288, 288,
                             96, 0.1, 1, s) == -3                       # head dim 96 is not compiled
    with pytest.raises(RuntimeError, match="LMOD_EINVAL"):
        K.gemm_nt(rnd(8, 12, seed=83), rnd(8, 12, seed=84))             # K = 12 through the binding
    assert K.gemm_nt(a[:0], b).shape == (0, 64)                         # zero rows: allocates, launches nothing
    # fused QKV + RoPE: the rotated block must end on a 256-column tile, N % 16, tables and positions present
    x, w = rnd(64, 64, seed=85), rnd(512, 64, seed=86)
    out = torch.zeros(64, 512, device=DEV, dtype=BF)
    cos, sin = _rope_tables(16, 128)
    pos = torch.zeros(64, device=DEV, dtype=torch.int32)
    r = lambda rope_cols, N=512, cos_p=cos.data_ptr(), M=64: lib.lmod_gemm_qkv_rope_bf16(
        x.data_ptr(), w.data_ptr(), out.data_ptr(), None, M, N, 64, 64, 64, 512, cos_p, sin.data_ptr(), pos.data_ptr(), rope_cols, s)
    assert r(256) == 0 and r(0) == 0 and r(512) == 0
    assert r(128) == -1 and r(768) == -1                               # not a tile multiple / past N
    assert r(256, N=504) == -1                                          # N % 16
    assert r(256, cos_p=None) == -1                                     # no table
    assert r(256, M=0) == 0                                             # empty
    # collectives: bad handles / dtypes are refused before RCCL is touched
    assert lib.lmod_allreduce_grads(None, c.data_ptr(), 16, 0, s) == -1
    assert lib.lmod_comm_init(None, None, 0, 1) == -1
    torch.cuda.synchronize()


def test_sumsq_clip_cast_and_adamw_device_scale():
    """Gradient-clipping pieces: deterministic sum of squares (accumulating over spans), clip coefficient with
    torch.nn.utils.clip_grad_norm_ arithmetic, AdamW reading that coefficient from the device; fp32 <-> bf16 casts."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1_000_003, generator=g).to(DEV)
    y = torch.randn(4099, generator=g).to(DEV)
    part = torch.empty(1024, device=DEV)
    out = torch.zeros(1, device=DEV)
    K.sumsq(x, out, part)
    K.sumsq(y[3:4099 - 0].contiguous(), out, part, accumulate=True)
    ref = (x.double() ** 2).sum() + (y[3:].double() ** 2).sum()
    assert abs(out.item() - ref.item()) <= 1e-5 * ref.item()
    out2 = torch.zeros(1, device=DEV)
    K.sumsq(x, out2, part)
    K.sumsq(y[3:].contiguous(), out2, part, accumulate=True)
    assert torch.equal(out, out2)                               # fixed reduction order: bit-identical run to run
    coef, nrm = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    K.clip_coef(out, 0.5, 1.0, coef, nrm)
    n_ref = math.sqrt(ref.item()) * 0.5
    assert abs(nrm.item() - n_ref) <= 1e-5 * n_ref and abs(coef.item() - min(1.0, 1.0 / (n_ref + 1e-6))) <= 1e-6
    K.clip_coef(out, 1e-6, 1.0, coef)
    assert coef.item() == 1.0                                   # small norm: no clipping
    # AdamW with the coefficient read on the device == AdamW with the product folded into grad_scale on the host
    n = 10007
    master = torch.randn(n, generator=g).to(DEV)
    grad = torch.randn(n, generator=g).to(DEV)
    K.clip_coef(out, 0.5, 1.0, coef)
    runs = []
    for dev_scale, gs in ((coef, 0.5), (None, 0.5 * coef.item())):
        ms, gr = master.clone(), grad.clone()
        m, v, pb = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), torch.empty(n, device=DEV, dtype=BF)
        K.adamw_step(ms, pb, gr, m, v, 1e-2, 0.9, 0.999, 1e-8, 0.1, 1, gs, dev_scale=dev_scale)
        runs.append((ms, m, v))
    for a, b in zip(*runs):
        close(a, b, "adamw dev scale", rtol=1e-5, afrac=1e-7)
    # casts
    xb = torch.empty(x.numel(), device=DEV, dtype=BF)
    K.cast_f32_bf16(x, xb)
    assert torch.equal(xb, x.to(BF))
    xf = torch.empty_like(x)
    K.cast_f32_bf16(xb, xf)
    assert torch.equal(xf, xb.float())


def _dense_from_state(st, T, E, C):
    dense = torch.zeros(T, E * C, device=DEV)
    for slot, w in ((st.slot1, st.w1),) + (((st.slot2, st.w2),) if st.k == 2 else ()):
        keep = slot >= 0
        dense[torch.nonzero(keep).squeeze(1), slot[keep].long()] = w[keep]
    return dense.reshape(T, E, C)


def test_moe_gate_noise_drawn_in_kernel():
    """Philox noise drawn inside the gating kernel: Gumbel(0,1) statistics (mean = Euler's gamma, var = pi^2/6),
    reproducible from (seed, offset), different per offset, and the picks are exactly those of a second call that is FED the
    drawn noise (so the oracle, given the same tensor, reproduces the routing)."""
    T, E = 8192, 4
    logits = torch.randn(T, E, generator=torch.Generator().manual_seed(1)).to(DEV)
    C = omoe.capacity(T, E, 1.5 * 2, 0)
    a = K.moe_gate(logits, 2, C, None, seed=1234, offset=77, want_noise=True)
    b = K.moe_gate(logits, 2, C, None, seed=1234, offset=77, want_noise=True)
    c = K.moe_gate(logits, 2, C, None, seed=1234, offset=77 + (1 << 24), want_noise=True)
    assert torch.equal(a.noise, b.noise) and torch.equal(a.idx2, b.idx2) and not torch.equal(a.noise, c.noise)
    n = a.noise.double()
    assert abs(n.mean().item() - 0.5772) < 0.03 and abs(n.var().item() - math.pi ** 2 / 6) < 0.08
    assert torch.isfinite(a.noise).all()
    fed = K.moe_gate(logits, 2, C, a.noise)
    assert torch.equal(fed.idx1, a.idx1) and torch.equal(fed.idx2, a.idx2) and torch.equal(fed.slot2, a.slot2)
    _, combine, _, _ = omoe.top2gating(logits, 1.5, 0, a.noise)
    close(_dense_from_state(a, T, E, C), combine, "combine with in-kernel noise", rtol=1e-5, afrac=1e-6)
    u = K.moe_gate(logits, 1, omoe.capacity(T, E, 1.0, 0), None, seed=5, offset=0, want_noise=True).noise
    assert 0.0 < u.min().item() and u.max().item() < 1.0 and abs(u.mean().item() - 0.5) < 0.01


@pytest.mark.parametrize("T,E,cf", [(1000, 4, 1.0), (4096, 8, 0.5), (300, 4, 4.0)])
def test_moe_gate_top1_random_token_selection(T, E, cf):
    """top1gating with use_rts (DeepSpeed's default): per expert the C tokens with the largest uniform priority keep their
    slot, survivors are numbered in token order; l_aux / exp_counts see every token.  Against the oracle's restatement,
    with explicit priorities, incl. an exact tie at the capacity boundary."""
    g = torch.Generator().manual_seed(T)
    logits = torch.randn(T, E, generator=g).to(DEV)
    prio = torch.rand(T, E, generator=g).to(DEV)
    C = omoe.capacity(T, E, cf, 0)
    st = K.moe_gate(logits, 1, C, prio)
    l_aux, combine, dispatch, counts = omoe.top1gating(logits, cf, 0, prio)
    close(_dense_from_state(st, T, E, C), combine, "rts combine", rtol=1e-5, afrac=1e-6)
    assert torch.equal(st.exp_counts.long(), counts.long())
    close(st.l_aux, l_aux.reshape(1), "rts l_aux", rtol=1e-5, afrac=1e-7)
    assert int((st.slot_token >= 0).sum()) == int(dispatch.sum())
    # token order (use_rts=False) through the same entry point
    st0 = K.moe_gate(logits, 1, C, None)
    _, combine0, _, _ = omoe.top1gating(logits, cf, 0, None)
    close(_dense_from_state(st0, T, E, C), combine0, "token-order combine", rtol=1e-5, afrac=1e-6)
    # ties exactly at the boundary: every token of expert 0 has the same priority -> the first C in token order survive
    prio2 = prio.clone()
    prio2[:, 0] = 0.25
    st2 = K.moe_gate(logits, 1, C, prio2)
    kept = torch.nonzero((st2.idx1 == 0) & (st2.slot1 >= 0)).squeeze(1)
    allt = torch.nonzero(st2.idx1 == 0).squeeze(1)
    assert torch.equal(kept, allt[:min(C, allt.numel())])


def test_residual_moe_layer_vs_oracle():
    """deepspeed.moe.layer.MoE(use_residual=True): out = moe(x) * c0 + mlp(x) * c1, (c0, c1) = softmax(coefficient(x)).
    The HIP layer against the oracle's layer on the same weights and routing noise: output, l_aux and every gradient."""
    import copy
    from oracle.decoder import MLP as OMLP, DecoderConfig
    from llavamod.model.language_model.qwen2_hip import Qwen2Config, Qwen2MLP, init_normal_
    from llavamod.model.moe_layer import MoE
    H, I, E, T = 256, 512, 4, 384
    cfg = Qwen2Config(hidden_size=H, intermediate_size=I, num_hidden_layers=1, num_attention_heads=2, vocab_size=64)
    expert = init_normal_(Qwen2MLP(cfg, DEV), 0.05, 3)
    layer = MoE(H, expert, num_experts=E, k=2, capacity_factor=1.5, min_capacity=0, use_residual=True)
    init_normal_(layer, 0.05, 4)                                   # experts / residual mlp / coefficient all different
    ocfg = DecoderConfig(vocab_size=64, hidden_size=H, intermediate_size=I, num_hidden_layers=1, num_attention_heads=2,
                         num_key_value_heads=2)
    olayer = omoe.OracleMoE(H, OMLP(ocfg), num_experts=E, k=2, capacity_factor=1.5, min_capacity=0, use_residual=True)
    sd = {k: v.detach().float().cpu() for k, v in layer.state_dict().items()}
    assert set(sd) == set(olayer.state_dict()), sorted(set(sd) ^ set(olayer.state_dict()))
    olayer.load_state_dict(sd)
    x = rnd(T, H, seed=9, scale=1.0)
    noise = omoe.gumbel_noise((T, E), torch.Generator().manual_seed(2))
    layer.train(); olayer.train()
    layer.gate_noise = noise
    olayer.noise = noise
    for p in layer.parameters():
        p.requires_grad_(True)
    from llavamod.engine import GradBuffer
    GradBuffer(layer)
    xh = x.clone().requires_grad_(True)
    out, l_aux, counts = layer(xh)
    dout = rnd(T, H, seed=10)
    ((out.float() * dout.float()).sum() + 3.0 * l_aux).backward()
    xo = x.float().cpu().requires_grad_(True)
    oo, ol, oc = olayer(xo)
    ((oo * dout.float().cpu()).sum() + 3.0 * ol).backward()
    close(out, oo.to(DEV), "residual moe out", rtol=2 ** -6, afrac=2 ** -7)
    close(l_aux.reshape(1), ol.reshape(1).to(DEV), "residual moe l_aux", rtol=1e-4, afrac=1e-6)
    close(xh.grad, xo.grad.to(DEV), "residual moe dx", rtol=2 ** -5, afrac=2 ** -6)
    for (n, p), (_, po) in zip(layer.named_parameters(), olayer.named_parameters()):
        close(p.main_grad, po.grad.to(DEV), f"residual moe grad {n}", rtol=2 ** -5, afrac=2 ** -5)


def test_attn_packed_varlen_equals_padded():
    """cu_seqlens (unpadded) layout of the hd-128 attention kernels: forward and backward on packed ragged samples equal the
    padded + key-masked launch on the same samples, row for row."""
    B, S, nh, nkv, hd = 3, 700, 4, 2, 128
    lens = [700, 333, 64]
    ld = (nh + 2 * nkv) * hd
    qkv = rnd(B * S, ld, seed=5)
    do = rnd(B * S, nh * hd, seed=6)
    sl = torch.tensor(lens, dtype=torch.int32, device=DEV)
    sc = 1 / math.sqrt(hd)
    q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
    o, lse = K.attn_fwd(q, k, v, B, S, nh, nkv, hd, sc, True, sl)
    dqkv = torch.zeros_like(qkv)
    K.attn_bwd(q, k, v, o, do, lse, dqkv[:, :nh * hd], dqkv[:, nh * hd:(nh + nkv) * hd], dqkv[:, (nh + nkv) * hd:],
               B, S, nh, nkv, hd, sc, True, sl)
    keep = torch.cat([torch.arange(L, device=DEV) + b * S for b, L in enumerate(lens)])
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=DEV)
    pq = qkv[keep].contiguous()
    pdo = do[keep].contiguous()
    q2, k2, v2 = pq[:, :nh * hd], pq[:, nh * hd:(nh + nkv) * hd], pq[:, (nh + nkv) * hd:]
    o2, lse2 = K.attn_fwd(q2, k2, v2, B, S, nh, nkv, hd, sc, True, None, cu=cu)
    assert torch.equal(o2, o[keep])
    for b, L in enumerate(lens):
        assert torch.equal(lse2[b, :, :L], lse[b, :, :L])
    d2 = torch.zeros_like(pq)
    K.attn_bwd(q2, k2, v2, o2, pdo, lse2, d2[:, :nh * hd], d2[:, nh * hd:(nh + nkv) * hd], d2[:, (nh + nkv) * hd:],
               B, S, nh, nkv, hd, sc, True, None, cu=cu)
    # padded backward: gradients of padded query rows are garbage-free zeros only for K/V; compare the kept rows
    assert torch.equal(d2[:, :nh * hd], dqkv[keep][:, :nh * hd])
    # dK/dV of kept keys receive contributions from padded QUERY rows in the padded launch (q >= len attends keys < len):
    # zero those contributions by re-running the padded launch with dO = 0 on padded rows
    do_m = do.clone()
    padrows = torch.ones(B * S, dtype=torch.bool, device=DEV); padrows[keep] = False
    do_m[padrows] = 0
    dq3 = torch.zeros_like(qkv)
    K.attn_bwd(q, k, v, o, do_m, lse, dq3[:, :nh * hd], dq3[:, nh * hd:(nh + nkv) * hd], dq3[:, (nh + nkv) * hd:],
               B, S, nh, nkv, hd, sc, True, sl)
    close(d2[:, nh * hd:], dq3[keep][:, nh * hd:], "packed dK/dV", rtol=2 ** -7, afrac=2 ** -9)
