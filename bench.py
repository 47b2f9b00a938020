#!/usr/bin/env python
"""bench.py — distillation training-step throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1: either launched one process per GPU by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or started plainly — `python bench.py --gpus N` then
re-executes itself under torch.distributed.run on 127.0.0.1 (the reference's one-command launch,
shells/train/qwen/dense2sparse_distillation.sh:48 `deepspeed llavamod/train/align_train.py ...`).
`--launch-check`: every rank rendezvouses over the selected backend (RCCL; `LMOD_DIST_BACKEND=gloo` on a box without GPUs),
builds the REAL config's span / shard / optimizer-state plan on `meta` tensors, proves the world with one collective and
prints the `exchange` object — the N>1 plumbing without touching a GPU.

One "step" = one full mimic-distillation OPTIMIZER step (config 2 of BASELINE.json: CLIP-ViT-L/14-336 +
Qwen-1.8B-MoE top-2/4-expert student, Qwen-7B dense teacher, bf16): `--grad-accum` (default 2) micro-batches
of `--micro-batch` (default 16) samples per GPU — config 3's 8 x 32 = 256 global batch at N=8 — each a frozen
teacher forward + student forward/backward + KD/LM/aux losses (the d2s recipe `kd_lm` + moe_loss), then ONE
gradient exchange over the DP group (RCCL reduce-scatter, ZeRO-2 style sharded optimizer state; all-reduce with
`--no-zero2`), global-norm clipping at 1.0 and the fused AdamW (+ in-place all-gather of the bf16 weights).
Synthetic data of the real shape (336x336 image -> 576 patches + 1472 text tokens = 2048 context, 512 response
tokens; `--ragged`: text lengths U[600,1473]), random init of the real architecture.  value = image-text
samples / s over ALL ranks (weak scaling: fixed work per GPU).  Prints ONE JSON line on rank 0 with `roofline`
(dominant kernel, timed live with HIP events) and `cpu_baseline` (the fp32 oracle's full-depth step on the host cores,
rank 0 at N=1, with the GPU-vs-oracle loss difference).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
sys.path.insert(0, ROOT)

TFLOP_PER_SAMPLE_LEDGER = 52.98          # BASELINE.md §2 / SURVEY.md §8(d): algorithmic TFLOP per mimic sample
PEAK_BF16_TFLOPS = 2500.0                # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)


def student_cfg(experts):
    from llavamod.model import CLIPVisionConfig, LLaVAMoDQwen2Config
    return LLaVAMoDQwen2Config(vocab_size=151936, hidden_size=2048, intermediate_size=5504, num_hidden_layers=24,
                               num_attention_heads=16, num_key_value_heads=16, rms_norm_eps=1e-6, rope_theta=1000000.0,
                               max_position_embeddings=4096, mm_image_tower=CLIPVisionConfig(),
                               image_projector_type="mlp2x_gelu", mm_hidden_size=1024, mm_vision_select_layer=-2,
                               mm_vision_select_feature="patch", init_seed=1)


def teacher_cfg():
    from llavamod.model import CLIPVisionConfig, LlavaQwen2Config
    return LlavaQwen2Config(vocab_size=151936, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                            num_attention_heads=32, num_key_value_heads=32, rms_norm_eps=1e-6, rope_theta=1000000.0,
                            max_position_embeddings=4096, mm_image_tower=CLIPVisionConfig(),
                            image_projector_type="mlp2x_gelu", mm_hidden_size=1024, mm_vision_select_layer=-2,
                            mm_vision_select_feature="patch", init_seed=2)


def moe_model_args(experts, k=2, ep_size=1):
    from types import SimpleNamespace
    # shells/train/qwen/dense2sparse_distillation.sh:27-42
    return SimpleNamespace(moe_enable=True, train_modules=["mlp.gate_proj", "mlp.up_proj", "mlp.down_proj", "wg"],
                           moe_mode="sparse", moe_layers_idx=None, ep_size=ep_size, top_k_experts=k, capacity_factor=1.5,
                           eval_capacity_factor=2.0, min_capacity=0, use_residual=False, router_aux_loss_coef=0.01,
                           num_experts=[experts])


def synthetic_batch(B, seed, text_len=1473, response=512, vocab=151643, ragged=False):
    """ids U[0,151643), one -200 at index 14, labels on the last `response` tokens.
    ragged: the §8(d) variant — text lengths U[600, 1473], right-padded, labels on each sample's own last `response` tokens."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, vocab, (B, text_len), generator=g)
    ids[:, 14] = -200
    labels = torch.full((B, text_len), -100, dtype=torch.long)
    mask = torch.ones(B, text_len, dtype=torch.bool)
    if ragged:
        lens = torch.randint(600, text_len + 1, (B,), generator=g)
        lens[0] = text_len                                    # the batch maximum stays 2048 after the splice
        for b in range(B):
            L = int(lens[b])
            labels[b, L - response:L] = ids[b, L - response:L]
            mask[b, L:] = False
            ids[b, L:] = vocab                                # pad id (unused: stripped by the mask)
    else:
        labels[:, -response:] = ids[:, -response:]
    images = torch.randn(B, 3, 336, 336, generator=g).to(torch.bfloat16)
    return dict(input_ids=ids, attention_mask=mask, labels=labels, images=images)


def mimic_tflop(t_layers=32, s_dense=12, s_moe=12, vit_layers=23, S=2048, head_rows=None, V=151936, E=4, k=2):
    """Algorithmic TFLOP of one mimic-distillation sample, BASELINE.md §2 / SURVEY.md §A.2 conventions
    (2mnk per GEMM, causal attention at 1/2, backward attention 2x forward, wgrad only for FFN/router).
    Full depth, head_rows=None reproduces the 52.98 TFLOP ledger figure."""
    rows = S if head_rows is None else head_rows

    def layer(H, I, moe):
        ffn = (k * 6 * H * I + 2 * H * E) if moe else 6 * H * I
        return 8 * H * H + 2 * S * H, ffn                     # (attention block incl. QKVO, ffn) per token

    ta, tf = layer(4096, 11008, False)
    teacher = S * t_layers * (ta + tf) + rows * 2 * 4096 * V
    sa, sf_d = layer(2048, 5504, False)
    _, sf_m = layer(2048, 5504, True)
    s_layers = s_dense + s_moe
    ffn = S * (s_dense * sf_d + s_moe * sf_m)
    proj = 2 * 576 * (1024 * 2048 + 2048 * 2048)
    s_fwd = S * s_layers * sa + ffn + rows * 2 * 2048 * V + proj
    s_dgrad = s_fwd + S * s_layers * 2 * S * 2048              # + one extra attention (backward = 2x forward)
    s_wgrad = ffn + proj
    vit = vit_layers * (8 * 577 * 1024 ** 2 + 4 * 577 ** 2 * 1024 + 4 * 577 * 1024 * 4096) + 2 * 576 * 588 * 1024
    return (teacher + s_fwd + s_dgrad + s_wgrad + 2 * vit) / 1e12


def executed_tflop_per_sample(S=2048, rows=513, E=4):
    """What this implementation issues: lm_head GEMMs (teacher fwd, student fwd + dgrad) and the o_proj + MLP of each
    model's LAST (dense) decoder layer on the 513 loss rows only."""
    skipped = S - rows
    t_last = skipped * (2 * 4096 * 4096 + 6 * 4096 * 11008)                       # teacher: forward only
    s_last = skipped * (2 * (2 * 2048 * 2048 + 6 * 2048 * 5504) + 6 * 2048 * 5504)  # student: fwd + dgrad (+ FFN wgrad)
    return mimic_tflop(head_rows=rows, E=E) - (t_last + s_last) / 1e12


def gemm_algorithmic_bytes(M, N, K, elem=2):
    """Algorithmic HBM bytes of one C[M,N] = A[M,K] . B[N,K]^T launch: every operand and the result once."""
    return (M * K + N * K + M * N) * elem


def workload_name(stage, experts, world, ep):
    """Exactly ONE BASELINE.json config per bench line."""
    tail = f"CLIP-ViT-L/14-336 + Qwen1.5-1.8B-MoE ({experts} experts, top-2, cf 1.5, 12 MoE layers, ep_size {ep}) student, Qwen1.5-7B teacher"
    if stage == "dpo":
        return "config 4: preference distillation (kto_pair), value = chosen/rejected PAIRS per second, " + tail
    if experts == 8:
        return "config 5: mimic distillation (kd_lm + moe aux), 8-expert top-2 student" + (" with expert-parallel all-to-all" if ep > 1 else " (ep_size 1: no exchange)") + ", " + tail
    if world > 1:
        return f"config 3: mimic distillation (kd_lm + moe aux), data parallel over {world} GPUs, " + tail
    return "config 2: mimic distillation (kd_lm + moe aux), " + tail


def whole_step_object(units_per_s_per_gpu, stage, E=4):
    """Whole-step MFMA utilisation.  PRIMARY: on the flops this implementation actually issues (lm_head and each model's last
    dense layer only on the loss rows); beside it the figure on BASELINE.md's algorithmic ledger."""
    mult = 2.0 if stage == "dpo" else 1.0
    ex, led = executed_tflop_per_sample(E=E) * mult, TFLOP_PER_SAMPLE_LEDGER * mult
    return {"frac_executed": round(ex * units_per_s_per_gpu / PEAK_BF16_TFLOPS, 4), "achieved_executed": round(ex * units_per_s_per_gpu, 1),
            "executed_tflop_per_unit": round(ex, 2),
            "frac_ledger": round(led * units_per_s_per_gpu / PEAK_BF16_TFLOPS, 4), "achieved_ledger": round(led * units_per_s_per_gpu, 1),
            "ledger_tflop_per_unit": round(led, 2),
            "basis": "TFLOP per unit x units/s per GPU / 2500 TFLOP/s; executed = what the kernels run, ledger = BASELINE.md's algorithmic count"}


def dominant_kernel_from_profile(root=ROOT):
    """The `roofline` object describes the DOMINANT kernel, and which kernel that is comes from the committed rocprofv3 summary of
    this very workload, not from a constant here: the top row (by total duration) of the newest
    `profiles/rNN_final_bench_kernel_stats.csv`.  Returns {"name", "pct", "file", "family_pct"} (family = every `gemm4_kernel` /
    `gemm4t_kernel` instantiation: one K loop, different epilogues) or None when no summary is committed."""
    import csv
    for r in range(20, 0, -1):
        f = os.path.join(root, "profiles", f"r{r:02d}_final_bench_kernel_stats.csv")
        if not os.path.exists(f):
            continue
        rows = [x for x in csv.DictReader(open(f)) if x.get("Name")]
        if not rows:
            continue
        top = max(rows, key=lambda x: float(x["TotalDurationNs"]))
        fam = sum(float(x["Percentage"]) for x in rows if "gemm4_kernel" in x["Name"] or "gemm4t_kernel" in x["Name"])
        return {"name": top["Name"], "pct": float(top["Percentage"]), "file": os.path.basename(f), "family_pct": round(fam, 1)}
    return None


# rocprofv3 kernel name (prefix) -> how bench.py times that kernel live: (label, which launch).  The shapes are the ones each
# instantiation spends most of its time on in config 2 (teacher layers: hidden 4096, intermediate 11008, 32 heads).
ROOFLINE_RECIPES = {
    "void gemm4_kernel<1, true, false>": ("swiglu", "gemm4_kernel<1> (fused SwiGLU forward: act = silu(A Wg^T) * (A Wu^T) on the stored [2I x K] "
                                          "weight, 256x256x64 tile = 128 gate + 128 up columns, 4 waves of 128x128, K loop as one hand-placed "
                                          "asm statement, persistent workgroups) @ teacher MLP gate+up"),
    "void gemm4_kernel<7, true, false>": ("nt", "gemm4_kernel<7> (bf16 NT GEMM, 256x256x64 tile, 4 waves of 128x128, K loop as one hand-placed asm "
                                          "statement, persistent workgroups from 4 rounds of the CUs up) @ teacher QKV"),
    "void gemm4_kernel<7, false, false>": ("nt", "gemm4_kernel<7> (bf16 NT GEMM, 256x256x64 tile, 4 waves of 128x128) @ teacher QKV"),
}


def roofline_recipe(root=ROOT):
    """(dominant-kernel record, recipe key, label) — falls back to the plain NT GEMM when the top row has no live recipe."""
    dom = dominant_kernel_from_profile(root)
    if dom:
        for prefix, (key, label) in ROOFLINE_RECIPES.items():
            if dom["name"].startswith(prefix):
                return dom, key, label
    return dom, "nt", ROOFLINE_RECIPES["void gemm4_kernel<7, true, false>"][1]


def time_dominant_kernel(key, B, dev, reps=10):
    """Live micro-launch of the dominant kernel with HIP events on the launch stream (torch's current stream): returns
    (ms per launch, flops per launch, algorithmic bytes per launch, shape)."""
    from llavamod import kernels as K
    gm, gk = B * 2048, 4096
    a = torch.randn(gm, gk, device=dev).to(torch.bfloat16)
    if key == "swiglu":
        I = 11008
        w = torch.randn(2 * I, gk, device=dev).to(torch.bfloat16)
        o = torch.empty(gm, I, device=dev, dtype=torch.bfloat16)
        run = lambda: K.gemm_swiglu(a, w, act=o)               # the frozen teacher keeps no pre-activations
        flops, algo, shape = 2.0 * gm * 2 * I * gk, (gm * gk + 2 * I * gk + gm * I) * 2, [gm, 2 * I, gk]
    else:
        gn = 12288
        w = torch.randn(gn, gk, device=dev).to(torch.bfloat16)
        o = torch.empty(gm, gn, device=dev, dtype=torch.bfloat16)
        run = lambda: K.gemm_nt(a, w, out=o)
        flops, algo, shape = 2.0 * gm * gn * gk, gemm_algorithmic_bytes(gm, gn, gk), [gm, gn, gk]
    for _ in range(2):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, flops, algo, shape


# entry point -> (family label, argument positions) for the in-step aggregate.  Positions index the integer-argument tuple that
# `_hip.call` records (pointers appear as 0 / 1 = NULL / set).
_IN_STEP = {
    "lmod_gemm_bf16_nt": "gemm4_kernel<7> plain bf16 store",
    "lmod_gemm_bf16_nt_res": "gemm4_kernel<8> + residual add",
    "lmod_gemm_qkv_rope_bf16": "gemm4_kernel<5> fused QKV + RoPE",
    "lmod_gemm_swiglu_bf16": "gemm4_kernel<1> fused SwiGLU forward",
    "lmod_gemm_swiglu_bwd_bf16": "gemm4_kernel<4> / gemm_256_kernel<4> fused SwiGLU backward",
    "lmod_gemm_bf16_tn": "gemm4t_kernel (TN weight gradient)",
    "lmod_gemm_wgrad_bf16_nt": "split-K weight gradient",
    "lmod_attn_fwd": "attn_fwd2_kernel",
    "lmod_attn_bwd": "attn_bwd2_kernel (dQ + dK/dV + delta)",
    "lmod_attn_bwd_rope": "attn_bwd2_kernel (dQ + dK/dV + delta)",
    "lmod_attn_bwd_split": "attn_bwd2_kernel (dQ + dK/dV + delta)",
}


def _in_step_row(name, a):
    """(family, shape key, flops or None) of one traced launch; None = not counted.  Grouped launches (live rows only known on the
    device) are timed but carry no flop count."""
    if name == "lmod_gemm_bf16_nt_res":       # (A, W, C, bias, res, M, N, K, ...)
        M, N, Kd, batch, grouped = a[5], a[6], a[7], 1, False
    elif name == "lmod_gemm_bf16_nt":
        M, N, Kd, batch, mv, kv, act, f32, accu = a[4], a[5], a[6], a[10], a[14], a[15], a[16], a[17], a[18]
        if f32 or accu or act == 3:
            return None
        grouped = bool(mv or kv)
    elif name == "lmod_gemm_qkv_rope_bf16":
        M, N, Kd, batch, grouped = a[4], a[5], a[6], 1, False
    elif name == "lmod_gemm_swiglu_bf16":     # N = output columns, the GEMM is 2N wide
        M, N, Kd, batch, grouped = a[4], 2 * a[5], a[6], a[11], bool(a[16])
    elif name == "lmod_gemm_swiglu_bwd_bf16":
        M, N, Kd, batch, grouped = a[4], a[5], a[6], a[11], bool(a[16])
    elif name == "lmod_gemm_bf16_tn":
        M, N, Kd, batch, grouped = a[3], a[4], a[5], a[9], bool(a[13])
    elif name == "lmod_gemm_wgrad_bf16_nt":
        M, N, Kd, batch, grouped = a[3], a[4], a[5], 1, False
    elif name == "lmod_attn_fwd":
        B, S, nh, hd, causal = a[7], a[8], a[9], a[11], a[17]
        return _IN_STEP[name], f"B{B} S{S} nh{nh} hd{hd}" + (" causal" if causal else ""), 4.0 * B * nh * S * S * hd * (0.5 if causal else 1.0)
    elif name in ("lmod_attn_bwd", "lmod_attn_bwd_rope", "lmod_attn_bwd_split"):
        B, S, nh, hd, causal = a[12], a[13], a[14], a[16], a[26]
        return _IN_STEP[name], f"B{B} S{S} nh{nh} hd{hd}" + (" causal" if causal else ""), 10.0 * B * nh * S * S * hd * (0.5 if causal else 1.0)
    else:
        return None
    if name in ("lmod_gemm_bf16_nt", "lmod_gemm_bf16_nt_res"):
        if M < 512 or N < 256 or ((M + 255) // 256) * ((N + 255) // 256) * batch < 160:
            return None                           # routed to the small-tile kernels, not this family
    key = f"{M}x{N}x{Kd}" + (f" x{batch}" if batch > 1 else "") + (" grouped" if grouped else "")
    return _IN_STEP[name], key, (None if grouped else 2.0 * M * N * Kd * batch)


def in_step_gemm_aggregate(step_fn, step_index):
    """The MFMA kernels INSIDE the step: one extra, untimed optimizer step with every GEMM-family and attention launch bracketed
    by events on its stream (`_hip.TRACE`).  Per family (one entry point = one epilogue of the shared K loop): launches, ms,
    sum of flops / sum of durations over the launches whose flop count the host knows (grouped MoE launches are timed, their
    live rows are device-side), and the top shapes.  The top-level `achieved` / `by_shape` stay those of the plain NT family
    (`gemm4_kernel<7>` / `<8>`) so that the figure is comparable with earlier rounds' lines.  The rocprofv3 figure of the same
    quantities is profiles/*_kernel_stats.md."""
    from llavamod import _hip
    _hip.TRACE = {"names": set(_IN_STEP), "rows": []}
    try:
        torch.cuda.synchronize()
        step_fn(step_index, pipelined=False)       # the frozen model's pass inline: no second stream sharing the chip with the timed launches
        torch.cuda.synchronize()
        rows = _hip.TRACE["rows"]
    finally:
        _hip.TRACE = None
    fams = {}
    for name, a, e0, e1 in rows:
        r = _in_step_row(name, a)
        if r is None:
            continue
        fam, key, fl = r
        t = e0.elapsed_time(e1)
        f = fams.setdefault(fam, {"launches": 0, "ms": 0.0, "fl": 0.0, "fl_ms": 0.0, "shapes": {}})
        f["launches"] += 1; f["ms"] += t
        if fl is not None:
            f["fl"] += fl; f["fl_ms"] += t
        sh = f["shapes"].setdefault(key, [0, 0.0, 0.0])
        sh[0] += 1; sh[1] += t; sh[2] += fl or 0.0
    if not fams:
        return None

    def shapes_of(f, n=6):
        top = sorted(f["shapes"].items(), key=lambda kv: -kv[1][1])[:n]
        return {k: {"launches": v[0], "ms": round(v[1], 2), "tflops": (round(v[2] / v[1] / 1e9, 1) if v[2] else None)} for k, v in top}

    fl = ms = 0.0
    n = 0
    plain = {}
    for fam in ("gemm4_kernel<7> plain bf16 store", "gemm4_kernel<8> + residual add"):
        if fam in fams:
            fl += fams[fam]["fl"]; ms += fams[fam]["fl_ms"]; n += fams[fam]["launches"]
            for k, v in fams[fam]["shapes"].items():
                s_ = plain.setdefault(k, [0, 0.0, 0.0]); s_[0] += v[0]; s_[1] += v[1]; s_[2] += v[2]
    out = {"kernel": "gemm4_kernel<7> / <8> (plain bf16 store / + residual add in the epilogue) launches of ONE optimizer step (events around every launch; teacher pass inline for this step, so no second stream shares the chip)",
           "launches": n, "ms": round(ms, 2),
           "achieved": round(fl / ms / 1e9, 1) if ms else None, "frac": round(fl / ms / 1e9 / PEAK_BF16_TFLOPS, 4) if ms else None,
           "by_shape": shapes_of({"shapes": plain})}
    out["families"] = {fam: {"launches": f["launches"], "ms": round(f["ms"], 2),
                             "achieved": round(f["fl"] / f["fl_ms"] / 1e9, 1) if f["fl_ms"] else None,
                             "frac": round(f["fl"] / f["fl_ms"] / 1e9 / PEAK_BF16_TFLOPS, 4) if f["fl_ms"] else None,
                             "by_shape": shapes_of(f, 4)}
                       for fam, f in sorted(fams.items(), key=lambda kv: -kv[1]["ms"])}
    return out


def time_optimizer(gb, opt, grad_div, reps=3):
    """the optimizer step "reported separately" — global-norm clipping (sum of squares, coefficient) + the fused
    AdamW over every trainable span of this rank, HIP events, after the timed region (zero gradients: same bytes, same kernels)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gb.zero()
    opt.step(grad_scale=1.0 / grad_div, lr=0.0, clear_grads=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        opt.step(grad_scale=1.0 / grad_div, lr=0.0, clear_grads=True)
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 3)


def cpu_baseline_config1():
    """the oracle's config-1 step (tiny ViT + 2-layer / 4-expert MoE student, 2-layer dense teacher, B = 2, only_kd,
    fp32) on the host cores: 1 warm-up + 3 timed steps."""
    from oracle.decoder import DecoderConfig
    from oracle.llava import LlavaOracle, freeze_like_d2s, mimic_step
    from oracle.vision import IGNORE_INDEX, IMAGE_TOKEN_INDEX, VisionConfig
    vc = VisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=3, num_attention_heads=4, image_size=28, patch_size=14,
                      select_layer=-2)
    sc = DecoderConfig(vocab_size=512, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                       num_key_value_heads=2, moe_layers_idx=[0], num_experts=4, top_k_experts=2, capacity_factor=1.5,
                       eval_capacity_factor=2.0, min_capacity=0, router_aux_loss_coef=0.01)
    tc = DecoderConfig(vocab_size=512, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                       num_key_value_heads=2)
    torch.manual_seed(0)
    st, te = LlavaOracle(sc, vc, moe=True), LlavaOracle(tc, vc, moe=False)
    freeze_like_d2s(st)
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, 500, (2, 8), generator=g)
    ids[:, 2] = IMAGE_TOKEN_INDEX
    labels = ids.clone()
    labels[:, :3] = IGNORE_INDEX
    b = dict(input_ids=ids, attention_mask=torch.ones(2, 8, dtype=torch.bool), labels=labels, images=torch.randn(2, 3, 28, 28, generator=g))
    st.train(); te.eval()
    st.set_gate_noise(None)
    ts = []
    for i in range(4):
        st.zero_grad()
        t0 = time.perf_counter()
        mimic_step(st, te, b, loss_type="only_kd", align_vocab=512)
        ts.append(time.perf_counter() - t0)
    ms = sum(ts[1:]) / 3 * 1e3
    return {"ms_per_step": round(ms, 2), "samples_per_s": round(2 / (ms * 1e-3), 1), "steps": "1 warm-up + 3 timed, B = 2, only_kd, fp32"}


def _mem_total_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemTotal:"):
                return int(line.split()[1]) / 2 ** 20
    except OSError:
        pass
    return 0.0


def cpu_baseline(student=None, teacher=None, trainer=None, mode="auto", gb=None, stage="mimic", twin_device="cuda", dpo_pairs=1,
                 forced_arm=False):
    """The oracle (fp32 PyTorch restatement of the reference, oracle/) timed on this box's host cores on ONE sample of the
    same workload (config 2, B=1, S=2048): teacher forward + student forward/backward + losses (no optimizer step).

    mode "full" (auto: host RAM >= 96 GB): FULL depth — 32-layer 7B teacher, 24-layer MoE student, 23 ViT
    layers — carrying the very weights of the GPU models; the same batch is then run through the GPU path (routing noise off
    on both sides) and the four logged loss scalars are compared: `loss_delta`.
    mode "sample": depth-reduced (2 + 2 decoder layers, 2 ViT layers, full-vocabulary heads), scaled to the full-depth
    sample by algorithmic FLOPs; used when the host is too small, and said so.  A reported baseline, not a target."""
    from oracle.decoder import DecoderConfig
    from oracle.llava import LlavaOracle, freeze_like_d2s, load_from_product_state, mimic_step
    from oracle.vision import VisionConfig
    cores = min(os.cpu_count() or 1, 128)
    torch.set_num_threads(cores)
    V = 151936
    mem = _mem_total_gb()
    full = (mode == "full") or (mode == "auto" and mem >= 96 and student is not None)
    vit_l, t_l, s_l = (23, 32, 24) if full else (2, 2, 2)
    vc = VisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=vit_l + 1, num_attention_heads=16,
                      image_size=336, patch_size=14, select_layer=-2)
    sc = DecoderConfig(vocab_size=V, hidden_size=2048, intermediate_size=5504, num_hidden_layers=s_l,
                       num_attention_heads=16, num_key_value_heads=16, moe_layers_idx=list(range(0, s_l, 2)), num_experts=4,
                       top_k_experts=2, capacity_factor=1.5, min_capacity=0, max_position_embeddings=4096,
                       rope_theta=1000000.0)
    tc = DecoderConfig(vocab_size=V, hidden_size=4096, intermediate_size=11008, num_hidden_layers=t_l,
                       num_attention_heads=32, num_key_value_heads=32, max_position_embeddings=4096, rope_theta=1000000.0)
    b = synthetic_batch(1, 3)
    loss_delta = None
    if full:
        with torch.device("meta"):                             # no host-side random init of 10.7 B parameters
            o_student, o_teacher = LlavaOracle(sc, vc, moe=True), LlavaOracle(tc, vc, moe=False)
        o_student.to_empty(device="cpu"); o_teacher.to_empty(device="cpu")
        load_from_product_state(o_student, student.state_dict())
        load_from_product_state(o_teacher, teacher.state_dict())
    else:
        o_student, o_teacher = LlavaOracle(sc, vc, moe=True), LlavaOracle(tc, vc, moe=False)
    freeze_like_d2s(o_student)
    cb = dict(b, images=b["images"].float())
    o_student.train(); o_teacher.eval()
    o_student.set_gate_noise(None)
    if stage == "dpo":
        # config 4 (VERDICT r03 next #6b): ONE chosen / rejected pair through the oracle's preference step at full depth (two
        # student forward/backward passes + two frozen-model forwards, kto_pair, beta 0.1) and the same pair through the GPU
        # DPOTrainer — loss, reward term, balance term, rewards and sequence log-probabilities side by side
        from oracle.llava import dpo_step
        rj = synthetic_batch(1, 5003)
        pair = dict(chosen_input_ids=b["input_ids"], chosen_labels=b["labels"], chosen_attention_mask=b["attention_mask"],
                    rejected_input_ids=rj["input_ids"], rejected_labels=rj["labels"], rejected_attention_mask=rj["attention_mask"],
                    images=b["images"])
        hist, handles = _record_picks(o_student)
        from oracle import losses as olosses
        fpair = dict(pair, images=pair["images"].float())
        och = dict(input_ids=fpair["chosen_input_ids"], labels=fpair["chosen_labels"], attention_mask=fpair["chosen_attention_mask"], images=fpair["images"])
        orj = dict(input_ids=fpair["rejected_input_ids"], labels=fpair["rejected_labels"], attention_mask=fpair["rejected_attention_mask"], images=fpair["images"])
        t0 = time.time()                 # oracle.llava.dpo_step, spelled out so that the four outputs stay in hand (per-token log-probs below)
        o_sch, o_srj = o_student(**och), o_student(**orj)
        with torch.no_grad():
            o_tch, o_trj = o_teacher(**och), o_teacher(**orj)
        loss_o, logs = olosses.preference_loss(o_sch, o_srj, o_tch, o_trj, 0.1, 0.0, "kto_pair", True)
        loss_o.backward()
        t_cpu = time.time() - t0
        for h in handles:
            h.remove()
        tok_ref = {"policy/chosen": _token_logps(o_sch.logits, o_sch.labels), "policy/rejected": _token_logps(o_srj.logits, o_srj.labels),
                   "reference/chosen": _token_logps(o_tch.logits, o_tch.labels), "reference/rejected": _token_logps(o_trj.logits, o_trj.labels)}
        del o_sch, o_srj, o_tch, o_trj, loss_o
        tf = 2.0 * mimic_tflop(t_layers=t_l, s_dense=s_l // 2, s_moe=s_l // 2, vit_layers=vit_l)
        out = {"value": round(1.0 / t_cpu if full else (tf / t_cpu) / (2 * TFLOP_PER_SAMPLE_LEDGER), 5), "unit": "pairs/s", "cores": cores, "kind": "port",
               "sample": (f"oracle fp32 torch-CPU preference step (kto_pair) on ONE chosen/rejected pair, S=2048 each, "
                          f"{'FULL depth, the GPU models weights' if full else f'DEPTH-REDUCED to teacher {t_l}/32, student {s_l}/24, ViT {vit_l}/23 layers, scaled by algorithmic FLOPs'}: "
                          f"{tf:.2f} algorithmic TFLOP in {t_cpu:.1f} s = {tf / t_cpu:.2f} TFLOP/s")}
        if full:
            moes = student.moe_layers()
            old = [(m.deterministic, m.gate_noise) for m in moes]
            for m in moes:
                m.deterministic, m.gate_noise = True, None
            with torch.no_grad():
                _, outs = trainer.compute_loss(student, pair, return_outputs=True)
            for m, (d, n) in zip(moes, old):
                m.deterministic, m.gate_noise = d, n
            # FULL-DEPTH bf16 noise floor (VERDICT r04 next #1a): the oracle's bf16 twin on the same pair, once with the fp32
            # run's expert picks forced (rounding only) and once routing on its own (rounding + its own near-tie flips)
            tw_logs, twin_note, tok_tw = {}, None, {}
            if twin_device != "off":
                try:
                    o_student.zero_grad(set_to_none=True)
                    tdev = torch.device(twin_device if twin_device == "cpu" else next(student.parameters()).device)
                    t0 = time.time()
                    tw_s, tw_t = cpu_baseline_twin(sc, vc, True, student, tdev), cpu_baseline_twin(tc, vc, False, teacher, tdev)
                    tw_s.train(); tw_t.eval(); tw_s.set_gate_noise(None)
                    tp = _to_dev(pair, tdev)
                    ch = dict(input_ids=tp["chosen_input_ids"], labels=tp["chosen_labels"], attention_mask=tp["chosen_attention_mask"], images=tp["images"])
                    rj_ = dict(input_ids=tp["rejected_input_ids"], labels=tp["rejected_labels"], attention_mask=tp["rejected_attention_mask"], images=tp["images"])
                    with torch.device(tdev), torch.no_grad():
                        t_ch, t_rj = tw_t(**ch), tw_t(**rj_)
                        tok_tw["reference/chosen"] = {"twin": _token_logps(t_ch.logits, t_ch.labels)}
                        tok_tw["reference/rejected"] = {"twin": _token_logps(t_rj.logits, t_rj.labels)}
                        for tag, forced in (("forced", True), ("free", False)):
                            for m, h in zip(_oracle_moes(tw_s), hist):
                                m.forced = [(a.to(tdev), b.to(tdev)) for a, b in h] if forced else None
                            s_ch, s_rj = tw_s(**ch), tw_s(**rj_)
                            _, tw_logs[tag] = olosses.preference_loss(s_ch, s_rj, t_ch, t_rj, 0.1, 0.0, "kto_pair", True)
                            tok_tw.setdefault("policy/chosen", {})["twin_" + tag] = _token_logps(s_ch.logits, s_ch.labels)
                            tok_tw.setdefault("policy/rejected", {})["twin_" + tag] = _token_logps(s_rj.logits, s_rj.labels)
                            del s_ch, s_rj
                    twin_note = (f"oracle bf16 twin (bf16 weights + activations, fp32 router) of both models at FULL depth on the same pair, "
                                 f"eager torch on {tdev.type}, {time.time() - t0:.1f} s: floor_forced = |twin - fp32| / |fp32| with the fp32 run's expert picks "
                                 "forced (rounding only), floor = the twin routing on its own (rounding + its own near-tie flips); the GPU product path routes on its own")
                    del tw_s, tw_t, t_ch, t_rj
                    if tdev.type == "cuda":
                        torch.cuda.empty_cache()
                except Exception as e:
                    twin_note = "twin failed: " + repr(e)[:300]
            ld = {}
            for k in ("loss", "loss/reward", "loss/moe_balance", "rewards/chosen", "rewards/rejected", "rewards/margins",
                      "logps/chosen", "logps/rejected"):
                g, c = float(outs[k].detach()), float(logs[k].detach())
                ld[k] = _floor_entry(g, c, float(tw_logs["forced"][k]) if "forced" in tw_logs else None,
                                     float(tw_logs["free"][k]) if "free" in tw_logs else None)
            out["loss_delta"] = ld
            if twin_note:
                out["floor_note"] = twin_note
            # The statistic that CAN be tested: per-token log p(label) of the four sequences (512 labelled tokens each) against the
            # fp32 oracle — root-mean-square and mean of the differences for the GPU product path and for the twin.  A logged scalar
            # above is ONE draw of a sum of 512 such differences (and the loss a sigmoid of differences of four such sums), so the
            # ratio of two single draws says little; the per-token RMS is an average over 512 draws.
            try:
                dev = next(student.parameters()).device
                dp_ = _to_dev(pair, dev)
                seqs = {"chosen": dict(input_ids=dp_["chosen_input_ids"], labels=dp_["chosen_labels"], attention_mask=dp_["chosen_attention_mask"], images=dp_["images"]),
                        "rejected": dict(input_ids=dp_["rejected_input_ids"], labels=dp_["rejected_labels"], attention_mask=dp_["rejected_attention_mask"], images=dp_["images"])}
                tok, raw_err = {}, {}
                for m in moes:
                    m.deterministic, m.gate_noise = True, None
                with torch.no_grad():
                    for who, mdl in (("policy", student), ("reference", teacher)):
                        for side, b_ in seqs.items():
                            o_ = mdl(**b_)
                            r_ = tok_ref[f"{who}/{side}"]
                            g_ = _token_logps(o_.logits, o_.labels)
                            e = {"gpu_bf16": _token_stats(g_, r_)}
                            raw_err[f"{who}/{side}"] = {"gpu_bf16": [(g_ - r_).double()]}
                            for kname, tv in tok_tw.get(f"{who}/{side}", {}).items():
                                e[kname] = _token_stats(tv, r_)
                                raw_err[f"{who}/{side}"].setdefault(kname, []).append((tv - r_).double())
                            tw_rms = max([v["rms"] for kk, v in e.items() if kk != "gpu_bf16"] or [0.0])
                            e["gpu_rms_over_twin_rms"] = round(e["gpu_bf16"]["rms"] / tw_rms, 3) if tw_rms > 0 else None
                            tok[f"{who}/{side}"] = e
                            del o_
                for m, (d, n) in zip(moes, old):
                    m.deterministic, m.gate_noise = d, n
                out["token_logp_vs_fp32"] = tok
                out["token_logp_note"] = ("per labelled token: log p(label) of the GPU product path (full-logits forward of the same models) and of the oracle's "
                                          "bf16 twin minus the fp32 oracle's; rms / bias over the sequence's 512 tokens, sum_dev = the deviation of the "
                                          "sequence log-probability, rms_x_sqrt_n = what a sum of independent errors of that rms would deviate by")
                if dpo_pairs > 1:
                    out["token_logp_pooled"] = dpo_token_logp_more_pairs(student, teacher, o_student, o_teacher, sc, tc, vc, dpo_pairs, twin_device, raw_err)
            except Exception as e:
                out["token_logp_vs_fp32"] = {"error": repr(e)[:300]}
        return out
    hist, handles = _record_picks(o_student) if full else ([], [])
    t0 = time.time()
    _, logs, _, _ = mimic_step(o_student, o_teacher, cb, loss_type="kd_lm")
    t_cpu = time.time() - t0
    for h in handles:
        h.remove()
    tf = mimic_tflop(t_layers=t_l, s_dense=s_l // 2, s_moe=s_l // 2, vit_layers=vit_l)
    rate = tf / t_cpu
    if full:
        moes = student.moe_layers()
        old = [(m.deterministic, m.gate_noise) for m in moes]
        for m in moes:
            m.deterministic, m.gate_noise = True, None
        grad_cmp, gpu_picks, hg_keep = None, [], {}
        if gb is not None:
            # FREE-RUNNING full-depth gradients too: the GPU step's backward (both sides route on their own, no forced picks)
            # against the oracle's, relative Frobenius error of a sample of trainable tensors
            from oracle.llava import hip_to_oracle_key
            gb.flat.zero_()
            gpu_picks, pick_hooks = [], []
            for m in moes:                                    # the product path's own expert picks, layer by layer (one forward)
                pick_hooks.append(m.register_forward_hook(
                    lambda mod, a, o, dst=gpu_picks: dst.append((mod.last_state.idx1.detach().cpu(), mod.last_state.idx2.detach().cpu()))))
            loss_g, outs = trainer.compute_loss(student, b, return_outputs=True)
            loss_g.backward()
            torch.cuda.synchronize()
            for h_ in pick_hooks:
                h_.remove()
            ograd = {n: p.grad for n, p in o_student.named_parameters() if p.grad is not None}
            want = ("model.mm_projector.image_spatial_proj.0.weight", "model.mm_projector.image_spatial_proj.2.weight",
                    "model.layers.23.mlp.down_proj.weight", "model.layers.1.mlp.gate_proj.weight",
                    "model.layers.0.mlp.deepspeed_moe.gate.wg.weight", "model.layers.22.mlp.deepspeed_moe.gate.wg.weight",
                    "model.layers.0.mlp.deepspeed_moe.experts.deepspeed_experts.0.up_proj.weight",
                    "model.layers.12.mlp.deepspeed_moe.experts.deepspeed_experts.3.down_proj.weight")
            named = dict(student.named_parameters())
            grad_cmp, hg_keep = {}, {}
            for n in want:
                og = ograd.get(hip_to_oracle_key(n))
                hg = getattr(named.get(n), "main_grad", None)
                if og is None or hg is None:
                    continue
                hg = hg.detach().float().cpu()
                hg_keep[n] = hg                                # (the GPU buffer is zeroed below; the forced-picks arm compares against these)
                grad_cmp[n] = round(float((hg - og).norm() / og.norm().clamp_min(1e-30)), 5)
            gb.flat.zero_()
        else:
            with torch.no_grad():
                _, outs = trainer.compute_loss(student, b, return_outputs=True)
        for m, (d, n) in zip(moes, old):
            m.deterministic, m.gate_noise = d, n
        # full-depth bf16 noise floor of the same quantities: the oracle's bf16 twin, forced picks and free-running
        tw_logs, tw_grads, twin_note = {}, {}, None
        if twin_device != "off":
            try:
                tdev = torch.device(twin_device if twin_device == "cpu" else next(student.parameters()).device)
                t0 = time.time()
                tw_s, tw_t = cpu_baseline_twin(sc, vc, True, student, tdev), cpu_baseline_twin(tc, vc, False, teacher, tdev)
                freeze_like_d2s(tw_s)
                tw_s.train(); tw_t.eval(); tw_s.set_gate_noise(None)
                tb = _to_dev(b, tdev)
                for tag, forced in (("forced", True), ("free", False)):
                    for m, h in zip(_oracle_moes(tw_s), hist):
                        m.forced = (h[0][0].to(tdev), h[0][1].to(tdev)) if forced else None
                    tw_s.zero_grad(set_to_none=True)
                    with torch.device(tdev):
                        _, tw_logs[tag], _, _ = mimic_step(tw_s, tw_t, tb, loss_type="kd_lm")
                    if grad_cmp:
                        tg = {n: p.grad for n, p in tw_s.named_parameters() if p.grad is not None}
                        tw_grads[tag] = {}
                        for n in grad_cmp:
                            og, g_ = ograd.get(hip_to_oracle_key(n)), tg.get(hip_to_oracle_key(n))
                            if og is not None and g_ is not None:
                                tw_grads[tag][n] = round(float((g_.detach().float().cpu() - og).norm() / og.norm().clamp_min(1e-30)), 5)
                twin_note = (f"oracle bf16 twin (bf16 weights + activations, fp32 router) at FULL depth on the same sample, eager torch on {tdev.type}, "
                             f"{time.time() - t0:.1f} s: floor_forced = |twin - fp32| / |fp32| with the fp32 run's expert picks forced, floor = the twin routing on its own")
                del tw_s, tw_t
                if tdev.type == "cuda":
                    torch.cuda.empty_cache()
            except Exception as e:
                twin_note = "twin failed: " + repr(e)[:300]
        loss_delta = {}
        for k in ("loss", "loss/align", "loss/lm", "loss/moe_balance"):
            g, c = float(outs[k].detach()), float(logs[k].detach())
            loss_delta[k] = _floor_entry(g, c, float(tw_logs["forced"][k]) if "forced" in tw_logs else None,
                                         float(tw_logs["free"][k]) if "free" in tw_logs else None)
        if grad_cmp and forced_arm and len(gpu_picks) == len(_oracle_moes(o_student)):
            # the arm that isolates ARITHMETIC from ROUTING at full depth — the fp32 oracle run again with the GPU
            # product path's expert picks forced into every MoE layer (a second host step), its gradients against the GPU's: what is
            # left is rounding alone, to be held against the twin's forced-picks floor
            try:
                t0 = time.time()
                for m, pk_ in zip(_oracle_moes(o_student), gpu_picks):
                    m.forced = pk_
                o_student.zero_grad(set_to_none=True)
                _, logs_f, _, _ = mimic_step(o_student, o_teacher, cb, loss_type="kd_lm")
                fgrad = {n: p.grad for n, p in o_student.named_parameters() if p.grad is not None}
                forced_cmp = {}
                for n in grad_cmp:
                    og = fgrad.get(hip_to_oracle_key(n))
                    if og is not None and hg_keep.get(n) is not None:
                        forced_cmp[n] = round(float((hg_keep[n] - og).norm() / og.norm().clamp_min(1e-30)), 5)
                loss_delta["grad_rel_frobenius_gpu_picks_forced_into_fp32_oracle"] = forced_cmp
                fl = tw_grads.get("forced") or {}
                loss_delta["grad_gpu_picks_forced_within_2x_twin_forced_floor"] = {n: bool(v <= 2.0 * fl[n]) for n, v in forced_cmp.items() if n in fl}
                loss_delta["loss_gpu_picks_forced"] = {k: {"gpu_bf16": round(float(outs[k].detach()), 6), "cpu_fp32_gpu_picks": round(float(logs_f[k].detach()), 6),
                                                           "rel": round(abs(float(outs[k].detach()) - float(logs_f[k].detach())) / max(abs(float(logs_f[k].detach())), 1e-30), 6)}
                                                       for k in ("loss", "loss/align", "loss/lm", "loss/moe_balance")}
                loss_delta["forced_arm_note"] = f"second fp32 oracle step on the host with the GPU's picks forced: {time.time() - t0:.1f} s"
            except Exception as e:
                loss_delta["forced_arm_note"] = "forced arm failed: " + repr(e)[:300]
            finally:
                for m in _oracle_moes(o_student):
                    m.forced = None
        if grad_cmp:
            loss_delta["grad_rel_frobenius_free_running"] = grad_cmp
            if tw_grads:
                loss_delta["grad_rel_frobenius_twin_forced"] = tw_grads.get("forced")
                loss_delta["grad_rel_frobenius_twin_free_running"] = tw_grads.get("free")
        if twin_note:
            loss_delta["floor_note"] = twin_note
        sample = (f"oracle fp32 torch-CPU mimic step at FULL depth (32-layer teacher fwd + 24-layer MoE student fwd/bwd + "
                  f"23-layer ViT x2 + losses; no optimizer), B=1 S=2048, same weights and batch as the GPU models "
                  f"(host RAM {mem:.0f} GB): {tf:.2f} algorithmic TFLOP in {t_cpu:.1f} s = {rate:.2f} TFLOP/s")
        value = 1.0 / t_cpu
    else:
        sample = (f"oracle fp32 torch-CPU mimic step, B=1 S=2048, config-2 widths, DEPTH-REDUCED ("
                  f"{'--cpu-baseline sample' if mode == 'sample' else f'host RAM {mem:.0f} GB < 96 GB'}) to teacher {t_l}/32, student {s_l}/24, ViT {vit_l}/23 layers, full-vocab heads: "
                  f"{tf:.2f} algorithmic TFLOP in {t_cpu:.1f} s = {rate:.2f} TFLOP/s; scaled to the 52.98 TFLOP full-depth sample")
        value = rate / TFLOP_PER_SAMPLE_LEDGER
    out = {"value": round(value, 5), "unit": "samples/s", "cores": cores, "kind": "port", "sample": sample}
    if loss_delta is not None:
        out["loss_delta"] = loss_delta
    try:
        out["config1"] = cpu_baseline_config1()
    except Exception as e:
        out["config1"] = {"error": repr(e)[:200]}
    return out


def dpo_token_logp_more_pairs(student, teacher, o_student, o_teacher, sc, tc, vc, k_pairs, twin_device, raw_err):
    """the per-token log-probability statistic over k >= 4 INDEPENDENT chosen / rejected pairs at full depth.  The
    first pair's errors come from the caller; every further pair costs four forward passes of the fp32 oracle on the host (no backward),
    the same four through the GPU product path and through the oracle's bf16 twin (free-running, and with the fp32 run's expert picks
    forced).  Pooled per sequence kind (policy / reference x chosen / rejected): n tokens, mean error +- sigma / sqrt(n), the mean in
    units of that standard error, and whether |mean| <= 3 sigma / sqrt(n) — the quantity a sequence log-probability (a sum over tokens)
    and with it every reward inherits — for the product and for the twin."""
    dev = next(student.parameters()).device
    moes = student.moe_layers()
    old = [(m.deterministic, m.gate_noise) for m in moes]
    tw_s = tw_t = None
    note = []
    t0 = time.time()
    try:
        if twin_device != "off":
            tdev = torch.device(twin_device if twin_device == "cpu" else dev)
            tw_s, tw_t = cpu_baseline_twin(sc, vc, True, student, tdev), cpu_baseline_twin(tc, vc, False, teacher, tdev)
            tw_s.train(); tw_t.eval(); tw_s.set_gate_noise(None)
        o_student.train(); o_teacher.eval(); o_student.set_gate_noise(None)
        for m in moes:
            m.deterministic, m.gate_noise = True, None
        for k in range(1, k_pairs):
            ch, rj = synthetic_batch(1, 3 + 101 * k), synthetic_batch(1, 5003 + 101 * k)
            seqs = {"chosen": dict(input_ids=ch["input_ids"], labels=ch["labels"], attention_mask=ch["attention_mask"], images=ch["images"]),
                    "rejected": dict(input_ids=rj["input_ids"], labels=rj["labels"], attention_mask=rj["attention_mask"], images=ch["images"])}
            for m in _oracle_moes(o_student):
                m.forced = None
            hist, handles = _record_picks(o_student)
            ref = {}
            with torch.no_grad():
                for side, b_ in seqs.items():
                    fb = dict(b_, images=b_["images"].float())
                    o_ = o_student(**fb); ref[f"policy/{side}"] = _token_logps(o_.logits, o_.labels); del o_
                    o_ = o_teacher(**fb); ref[f"reference/{side}"] = _token_logps(o_.logits, o_.labels); del o_
            for h in handles:
                h.remove()
            with torch.no_grad():
                for who, mdl in (("policy", student), ("reference", teacher)):
                    for side, b_ in seqs.items():
                        o_ = mdl(**_to_dev(b_, dev))
                        raw_err[f"{who}/{side}"]["gpu_bf16"].append((_token_logps(o_.logits, o_.labels) - ref[f"{who}/{side}"]).double())
                        del o_
                if tw_s is not None:
                    for side, b_ in seqs.items():
                        tb = _to_dev(b_, tdev)
                        with torch.device(tdev):
                            o_ = tw_t(**tb)
                            raw_err[f"reference/{side}"].setdefault("twin", []).append((_token_logps(o_.logits, o_.labels) - ref[f"reference/{side}"]).double())
                            del o_
                    for tag, forced in (("twin_forced", True), ("twin_free", False)):
                        # the student ran chosen then rejected: call c of every layer's history belongs to sequence c
                        for ci, (side, b_) in enumerate(seqs.items()):
                            for m, h in zip(_oracle_moes(tw_s), hist):
                                m.forced = [(h[ci][0].to(tdev), h[ci][1].to(tdev))] if forced else None
                            with torch.device(tdev):
                                o_ = tw_s(**_to_dev(b_, tdev))
                            raw_err[f"policy/{side}"].setdefault(tag, []).append((_token_logps(o_.logits, o_.labels) - ref[f"policy/{side}"]).double())
                            del o_
            note.append(round(time.time() - t0, 1))
    finally:
        for m, (d, n) in zip(moes, old):
            m.deterministic, m.gate_noise = d, n
        for m in _oracle_moes(o_student):
            m.forced = None
        del tw_s, tw_t
        if dev.type == "cuda":
            torch.cuda.empty_cache()
    pooled = {}
    for name, arms in raw_err.items():
        pooled[name] = {}
        for arm, parts in arms.items():
            e = torch.cat(parts)
            n, mean, sd = e.numel(), float(e.mean()), float(e.std())
            se = sd / n ** 0.5
            pooled[name][arm] = {"pairs": len(parts), "n": n, "mean": round(mean, 6), "sigma": round(sd, 6), "stderr": round(se, 6),
                                 "mean_in_stderr": round(abs(mean) / se, 2) if se > 0 else None, "within_3_stderr": bool(abs(mean) <= 3.0 * se),
                                 "rms": round(float(e.pow(2).mean().sqrt()), 6),
                                 "sequence_sum_devs": [round(float(p_.sum()), 4) for p_ in parts]}
    pooled["note"] = (f"{k_pairs} independent chosen / rejected pairs, 512 labelled tokens per sequence; pairs 2.. are forward-only on the fp32 oracle "
                      f"(cumulative seconds after each: {note}); error = log p(label) minus the fp32 oracle's; a reward is 0.1 x a difference of two "
                      "sequence sums of such errors, so a mean inside 3 standard errors on every sequence kind is the statement that no arm carries a bias")
    return pooled


def cpu_baseline_twin(cfg, vcfg, moe, product_model, device):
    """The oracle's bf16 twin of a product model (tests/test_step_parity_gpu.py::_bf16_twin at full depth): the SAME oracle
    module with bf16 weights and activations (router `wg` kept fp32, as DeepSpeed keeps it), i.e. what the reference's own bf16
    run computes through eager torch ops.  |twin - fp32 oracle| is the bf16 NOISE FLOOR of a logged quantity.  The twin is
    test infrastructure like the oracle; by default it executes on the GPU (eager torch = the vendor BLAS the reference itself
    would call there, seconds instead of ~10 minutes of host bf16 GEMMs); `--twin-device cpu` keeps it on the host cores."""
    from oracle.llava import LlavaOracle, hip_to_oracle_key
    with torch.device("meta"):
        tw = LlavaOracle(cfg, vcfg, moe=moe)
    sd = {}
    for k, v in product_model.state_dict().items():
        ok = hip_to_oracle_key(k)
        sd[ok] = v.detach().to(device=device, dtype=torch.float32 if "gate.wg" in ok else torch.bfloat16, copy=True)
    tw.load_state_dict(sd, strict=True, assign=True)
    return tw


def _oracle_moes(o):
    return [l.mlp for l in o.lm.model.layers if hasattr(l.mlp, "deepspeed_moe")]


def _record_picks(o_student):
    """Forward hooks that keep every call's (idx1, idx2) of every MoE layer of an oracle student (the preference step calls
    the student twice; `last_picks` alone would only hold the second call)."""
    hist, handles = [], []
    for m in _oracle_moes(o_student):
        h = []
        hist.append(h)
        handles.append(m.register_forward_hook(lambda mod, inp, out, h=h: h.append((mod.last_picks[0].clone(), mod.last_picks[1].clone()))))
    return hist, handles


def _to_dev(batch, device):
    return {k: (v.to(device=device, dtype=torch.bfloat16) if v.is_floating_point() else v.to(device)) if torch.is_tensor(v) else v
            for k, v in batch.items()}


def _token_logps(logits, labels):
    """Per-token log p(label) on the shifted, labelled positions of ONE sequence (dpo_trainer.py:483-495 before the sum): fp32 [n]."""
    lb = labels[0, 1:]
    keep = lb != -100
    lg = logits[0, :-1][keep].float()
    return torch.gather(lg.log_softmax(-1), 1, lb[keep].unsqueeze(1).to(lg.device)).squeeze(1).detach().cpu()


def _token_stats(x, ref):
    d = (x - ref).double()
    return {"n": int(d.numel()), "rms": round(float(d.pow(2).mean().sqrt()), 6), "bias": round(float(d.mean()), 6),
            "sum_dev": round(float(d.sum()), 4), "rms_x_sqrt_n": round(float(d.pow(2).mean().sqrt()) * d.numel() ** 0.5, 4)}


def _floor_entry(g, c, tw_forced, tw_free):
    """One logged scalar: GPU product path vs fp32 oracle, beside the twin's own distance from the fp32 oracle with the fp32
    run's routing forced (pure rounding) and free-running (rounding + the twin's own near-tie flips)."""
    den = max(abs(c), 1e-30)
    e = {"gpu_bf16": round(g, 6), "cpu_fp32": round(c, 6), "abs": round(abs(g - c), 6), "rel": round(abs(g - c) / den, 6)}
    if tw_forced is not None:
        e["floor_forced"] = round(abs(tw_forced - c) / den, 6)
    if tw_free is not None:
        e["floor"] = round(abs(tw_free - c) / den, 6)
        fl = max(e.get("floor_forced", 0.0), e["floor"])
        e["within_1e-3"] = bool(e["rel"] <= 1e-3)
        e["within_2x_floor"] = bool(e["rel"] <= max(2.0 * fl, 1e-3))
    return e


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def self_launch(n):
    """`python bench.py --gpus N` with no launcher environment: become `torch.distributed.run` with one process per GPU
    on this node (rendezvous on 127.0.0.1 — the container hostname may not resolve)."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL needs dmabuf IPC on this driver
    os.environ.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def exchange_object(dp, opt, steps, comm0=None, comm1=None):
    """What the data-parallel step exchanges: backend and world as torch.distributed reports them (so the line shows RCCL
    had N ranks), the static per-step plan of this rank, and — after a timed run — the collectives actually issued per step."""
    from llavamod import engine
    on = dist.is_available() and dist.is_initialized()
    ex = {"backend": ("rccl (torch 'nccl')" if dist.get_backend() == "nccl" else dist.get_backend()) if on else None,
          "world_seen_by_backend": dist.get_world_size() if on else 1,
          "zero2": bool(dp.zero2), "grad_dtype": "bf16" if dp.grad_dtype == torch.bfloat16 else "fp32",
          "ep_size": dp.ep_size, "overlap": "spans exchanged from wgrad-ready hooks on RCCL's own high-priority stream",
          "collectives": ("C-ABI (csrc/comm.hip: lmod_reduce_scatter_grads / lmod_allreduce_grads / lmod_allgather_params / lmod_moe_all_to_all, own RCCL "
                          "communicator, side stream)") if getattr(dp, "native", False) else "torch.distributed"}
    if dp.enabled:
        ex["plan"] = dp.exchange_plan()
    if steps and dp.enabled:
        base = comm0 or {}
        ex["issued_per_step"] = {k: {"calls": round((v[0] - base.get(k, [0, 0])[0]) / steps, 2),
                                     "bytes": int((v[1] - base.get(k, [0, 0])[1]) / steps)}
                                 for k, v in sorted((comm1 if comm1 is not None else engine.COMM).items())}
    return ex


def launch_check(args):
    """N>1 plumbing without hardware: rendezvous, world proof, span plan of the real architecture on `meta`."""
    from llavamod.engine import DataParallel, GradBuffer, HipAdamW, init_distributed
    from llavamod.model import LLaVAMoDQwen2ForCausalLM
    rank, local, world = init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert world == 1 or (dist.is_initialized() and dist.get_world_size() == world)
    cdev = torch.device("cuda", local) if (world > 1 and dist.get_backend() == "nccl") else torch.device("cpu")
    if world > 1:                                             # one real collective over the chosen backend: every rank is there
        seen = torch.zeros(world, device=cdev)
        seen[rank] = 1.0
        dist.all_reduce(seen)
        assert bool((seen == 1).all()), seen
    student = LLaVAMoDQwen2ForCausalLM(student_cfg(args.experts), device="meta")
    student.initialize_moe_modules(moe_model_args(args.experts, ep_size=args.ep))
    for p in student.get_model().mm_projector.parameters():
        p.requires_grad = True
    gb = GradBuffer(student)
    dp = DataParallel(zero2=not args.no_zero2, grad_dtype=torch.bfloat16 if args.grad_dtype == "bf16" else torch.float32
                      ).attach(gb, args.ep)
    opt = HipAdamW(gb, lr=2e-5, weight_decay=0.0, dp=dp, max_grad_norm=args.max_grad_norm or None)
    if world > 1:
        # every rank must have laid out the same spans, and the owned chunks must tile each span exactly once per group
        sizes = torch.tensor([n for _, _, n in gb.spans] + [gb.numel], dtype=torch.int64, device=cdev)
        ref = sizes.clone()
        dist.broadcast(ref, 0)
        assert torch.equal(ref, sizes), "ranks disagree on the gradient-buffer layout"
        owned = torch.tensor([float(i["hi"] - i["lo"]) if i["sharded"] else 0.0 for i in dp.plan.values()],
                             dtype=torch.float64, device=cdev)
        dist.all_reduce(owned)
        full = torch.tensor([float(i["n"]) * (world // i["gsize"]) if i["sharded"] else 0.0 for i in dp.plan.values()],
                            dtype=torch.float64, device=cdev)
        assert torch.equal(owned, full), "sharded chunks do not tile their spans"
    n_train = sum(p.numel() for p in student.parameters() if p.requires_grad)
    if rank == 0:
        print(json.dumps({"launch_check": "ok", "n_gpus": world, "stage": args.stage, "trainable_params": n_train,
                          "grad_buffer_elems": gb.numel, "optimizer_state_elems_rank0": opt.n_state,
                          "exchange": exchange_object(dp, opt, 0)}), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--micro-batch", type=int, default=16)
    ap.add_argument("--grad-accum", type=int, default=2,
                    help="micro-batches per optimizer step (reference regime: --gradient_accumulation_steps, "
                         "dense2sparse_distillation.sh:70-72); 16 x 2 per GPU = global batch 256 on 8 GPUs (config 3)")
    ap.add_argument("--no-zero2", action="store_true",
                    help="N>1: all-reduce + full optimizer on every rank instead of reduce-scatter / sharded AdamW / all-gather")
    ap.add_argument("--grad-dtype", default="bf16", choices=["fp32", "bf16"],
                    help="dtype of the gradient exchange at N > 1.  bf16 (default) is what the reference's bf16 DeepSpeed engine moves (SURVEY "
                         "config 3: 4.07 GB per optimizer step); gradients ACCUMULATE in fp32 here either way (wgrad epilogues, master weights), "
                         "only the exchanged copy is bf16.  fp32 doubles the xGMI bytes for an exact sum")
    ap.add_argument("--max-grad-norm", type=float, default=1.0, help="global-norm clipping (HF Trainer default 1.0); 0 = off")
    ap.add_argument("--experts", type=int, default=4)
    ap.add_argument("--ragged", action="store_true", help="SURVEY §8(d) ragged variant: text lengths U[600,1473], right-padded")
    ap.add_argument("--unpad", action="store_true",
                    help="with --ragged: unpadded (cu_seqlens) execution — no padding rows in any kernel (the MoE gate then does "
                         "not see padding rows either, unlike the reference)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the short config-4 (DPO) and config-5 (8-expert student) legs appended under `extra` at N=1")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "full", "sample"],
                    help="auto: full-depth oracle step (same weights, loss compared with the GPU) when host RAM >= 96 GB")
    ap.add_argument("--twin-device", default="cuda", choices=["cuda", "cpu", "off"],
                    help="where the oracle's full-depth bf16 twin (the noise floor printed beside every loss_delta) executes: eager "
                         "torch on the GPU (seconds), the host cores (minutes), or not at all")
    ap.add_argument("--cpu-forced-arm", action="store_true",
                    help="full-depth cpu_baseline (mimic): a second fp32 oracle step on the host with the GPU's expert picks forced, so that the "
                         "gradient comparison isolates arithmetic from routing (+ ~3 min of host time; off in the default run)")
    ap.add_argument("--dpo-pairs", type=int, default=1,
                    help="--stage dpo, full-depth cpu_baseline: pool the per-token log-prob statistic over this many independent pairs "
                         "(pairs beyond the first are forward-only on the fp32 oracle: ~2 min of host time each)")
    ap.add_argument("--optimizer-overlap", action="store_true",
                    help="just-in-time AdamW on a second stream (bit-identical; measured 0.3 %% slower than serial, off by default)")
    ap.add_argument("--separate-towers", action="store_true", help="different random CLIP towers: both are run (no feature sharing)")
    ap.add_argument("--no-teacher-prefetch", action="store_true", help="run the teacher forward inline (A/B of the pipelining)")
    ap.add_argument("--ep", type=int, default=1, help="expert-parallel group size (config 5: --experts 8 --ep 8)")
    ap.add_argument("--stage", default="mimic", choices=["mimic", "dpo"],
                    help="mimic = configs 2/3/5 (headline metric); dpo = config 4 (preference distillation, pairs/s)")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous + world proof + the real config's exchange plan on meta tensors, then exit (no GPU work)")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args.gpus)                                # does not return
    if args.launch_check:
        return launch_check(args)

    from llavamod import engine
    from llavamod.engine import DataParallel, GradBuffer, HipAdamW, init_distributed, warmup_cosine
    from llavamod.model import LLaVAMoDQwen2ForCausalLM, LlavaQwen2ForCausalLM
    from llavamod.train.align_trainer import AlignTrainer
    from llavamod import kernels as K

    rank, local, world = init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local)
    B = args.micro_batch

    def build_student(experts):
        # the routers are created by `MoE(...)` like DeepSpeed's TopKGate: nn.Linear's default init from torch's GLOBAL generator.
        # Seed it — the same on every rank (data-parallel replicas must start equal; a checkpoint load / DeepSpeed's rank-0
        # broadcast does that for the reference) and the same in every process (the synthetic benchmark's routing, and with it
        # the live rows of every grouped launch, must not depend on what the process drew before)
        torch.manual_seed(20240 + experts)
        st = LLaVAMoDQwen2ForCausalLM(student_cfg(experts), device=dev)
        st.initialize_moe_modules(moe_model_args(experts, ep_size=args.ep))
        for p in st.get_model().mm_projector.parameters():
            p.requires_grad = True                            # initialize_vision_modules (llava_arch.py:115-120)
        st.unpad = bool(args.unpad)
        st.train()
        gb_ = GradBuffer(st)
        dp_ = DataParallel(zero2=not args.no_zero2, grad_dtype=torch.bfloat16 if args.grad_dtype == "bf16" else torch.float32
                           ).attach(gb_, args.ep)
        opt_ = HipAdamW(gb_, lr=2e-5, weight_decay=0.0, dp=dp_, max_grad_norm=args.max_grad_norm or None)
        return st, gb_, dp_, opt_

    student, gb, dp, opt = build_student(args.experts)
    teacher = LlavaQwen2ForCausalLM(teacher_cfg(), device=dev)
    if not args.separate_towers:
        # both models load the same frozen CLIP checkpoint in the reference's recipe (one --image_tower flag): give the
        # random-init teacher tower the student's weights, which lets the trainer compute the image features once per batch
        teacher.get_image_tower().load_state_dict(student.get_image_tower().state_dict())
    teacher.unpad = bool(args.unpad)
    teacher.eval()
    n_train = sum(p.numel() for p in student.parameters() if p.requires_grad)
    A = args.grad_accum
    total = args.steps + args.warmup
    pipelined = not args.no_teacher_prefetch

    def make_workload(stage, st, gb_, dp_, opt_, mb, seed0=0):
        """(trainer, step function) of one workload: stage "mimic" (configs 2 / 3 / 5) or "dpo" (config 4) on student `st`.

        Teacher pipelining: the frozen model's forward for batch i+1 is issued on a side stream before the student's step i, so the
        two streams' MFMA-bound and HBM-bound kernels overlap.  Every step still runs exactly one frozen-model pass per micro-batch
        and one student forward/backward/optimizer update; the device-wide synchronize() that closes a timed region also waits for
        the last frozen-model pass."""
        if stage == "mimic":
            tr = AlignTrainer(st, teacher, args=type("A", (), dict(moe_enable=True, distill_all_tokens=False,
                                                                   loss_type="kd_lm", moe_loss_enable=True))())
            bs = [synthetic_batch(mb, seed0 + 1000 * rank + i, ragged=args.ragged) for i in range(max(2, A))]
            pf, key = tr.prefetch_teacher, "teacher"
        else:   # config 4: chosen / rejected pairs sharing the image; kto_pair is the shell default (preference_distillation.sh:29)
            from llavamod.train.dpo_trainer import DPOTrainer
            tr = DPOTrainer(st, teacher, beta=0.1, loss_type="kto_pair")
            bs = []
            for i in range(max(2, A)):
                ch, rj = synthetic_batch(mb, seed0 + 1000 * rank + i), synthetic_batch(mb, seed0 + 5000 + 1000 * rank + i)
                bs.append(dict(chosen_input_ids=ch["input_ids"], chosen_labels=ch["labels"],
                               chosen_attention_mask=ch["attention_mask"], rejected_input_ids=rj["input_ids"],
                               rejected_labels=rj["labels"], rejected_attention_mask=rj["attention_mask"],
                               images=ch["images"]))
            pf, key = tr.prefetch_reference, "reference"
        state = {"h": pf(bs[0]) if pipelined else None}
        nb_ = len(bs)

        def step_fn(i, pipelined=pipelined):
            """One optimizer step = A micro-batches (gradients accumulate in the fp32 buffer; the exchange is armed on the
            last one only) + gradient exchange + clipping + AdamW."""
            gb_.zero()
            for a in range(A):
                k = i * A + a
                dp_.armed = (a == A - 1)
                if pipelined:
                    nxt = pf(bs[(k + 1) % nb_])
                    loss = tr.training_step(st, bs[k % nb_], **{key: state["h"]})
                    state["h"] = nxt
                else:
                    loss = tr.training_step(st, bs[k % nb_])
            dp_.finish()                     # spans were exchanged asynchronously as the last backward produced them
            # the reference averages the loss over the accumulation window and the ranks: fold both means into the scale
            opt_.step(grad_scale=1.0 / (world * A), lr=warmup_cosine(i, max(total, 100), 2e-5), overlap=args.optimizer_overlap,
                      clear_grads=True)      # gradients are cleared inside the AdamW pass: no separate 8 GB memset
            return loss
        return tr, step_fn

    def timed(step_fn, warm, steps, first=0):
        """`warm` untimed + `steps` timed optimizer steps, barrier + synchronize on both sides, MAX over ranks: seconds."""
        for i in range(warm):
            step_fn(first + i)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = None
        for i in range(steps):
            last = step_fn(first + warm + i)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, last

    trainer, step = make_workload(args.stage, student, gb, dp, opt, B)

    if pipelined:
        # the student step is the critical path: it runs on a HIGH-priority stream, so the dispatcher serves its kernels
        # first and the prefetched teacher pass (default priority, side stream) fills what is left (+1.0 % measured)
        hp = torch.cuda.Stream(device=dev, priority=-1)
        hp.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(hp)
    for i in range(args.warmup):
        step(i)
    comm0 = {k: list(v) for k, v in engine.COMM.items()}
    dt, last = timed(step, 0, args.steps, first=args.warmup)
    comm1 = {k: list(v) for k, v in engine.COMM.items()}      # the collectives of the TIMED steps only (later legs issue more)
    loss_val = float(last)

    # Everything below that LAUNCHES work runs on EVERY rank: the in-step aggregate is one more optimizer step and the optimizer timing
    # runs the sharded AdamW with its all-gathers — at N > 1 both contain collectives, and a rank that ran them alone would wait for its
    # peers forever (rounds 4's bench did exactly that on rank 0; the N > 1 lines of round 3 predate it).  Only rank 0 reports.
    # dominant kernel = the top row of the committed rocprofv3 summary of this workload (`roofline_recipe`), at the shape it spends
    # most of its time on, measured live with HIP events on the launch stream (torch's current stream)
    dom, dom_key, dom_label = roofline_recipe()
    gemm_ms, dom_flops, algo_bytes, dom_shape = time_dominant_kernel(dom_key, B, dev)
    gemm_tf = dom_flops / (gemm_ms * 1e-3) / 1e12
    in_step = in_step_gemm_aggregate(step, args.warmup + args.steps)
    optimizer_ms = time_optimizer(gb, opt, world * A)
    if world > 1:
        dist.barrier()

    if rank == 0:
        sps = args.steps * B * A * world / dt
        ledger = TFLOP_PER_SAMPLE_LEDGER * (2.0 if args.stage == "dpo" else 1.0)      # DPO: 105.96 TFLOP per pair
        achieved = ledger * sps / world
        # HBM-side bytes per launch of that kernel come from the committed PMC passes (they cannot be collected live); the
        # ALGORITHMIC bytes are those of the launch timed above (operands and result once, bf16)
        traffic, traffic_note = None, f"no committed PMC pass for this kernel and shape; algorithmic {algo_bytes / 1e9:.2f} GB"
        for r_ in range(20, 0, -1):
            tp = os.path.join(ROOT, "profiles", f"r{r_:02d}_final_gemm_traffic.json")
            if not os.path.exists(tp):
                continue
            tj = json.load(open(tp))
            if tj["shape"] == dom_shape and tj["kernel"].split("<")[1][:1] == dom_label.split("<")[1][:1]:
                traffic = round((tj["fetch_bytes_corrected"] + tj["write_bytes"]) / 1e9, 2)
                traffic_note = (f"GB per launch at the L2/fabric boundary (Infinity-Cache hits included), {tj['source']}; "
                                f"algorithmic (operands + result once, bf16) {algo_bytes / 1e9:.2f} GB => {traffic * 1e9 / algo_bytes:.2f}x — see "
                                f"profiles/{os.path.basename(tp).replace('gemm_traffic.json', 'pmc.md')}")
                break
        out = {
            "metric": "distillation samples/sec (336px img + 2k ctx), 2B-MoE student / 7B teacher",
            "value": round(sps, 4), "unit": "samples/s" if args.stage == "mimic" else "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic (random-init weights, seeded image/token batches)",
            # dtype of the gradient EXCHANGE at N > 1 (accumulation is fp32 either way).  The default moved from fp32 to bf16 in round 5 (the
            # reference's bf16 DeepSpeed engine): N > 1 lines before that moved twice the bytes and are not like-for-like
            "grad_dtype": args.grad_dtype,
            "config": {"workload": workload_name(args.stage, args.experts, world, args.ep),
                       "batch_shape": ("dense: every sample 2048 tokens" if not args.ragged else
                                       "ragged: text lengths U[600,1473] right-padded (SURVEY 8d variant), " +
                                       ("unpadded cu_seqlens execution" if args.unpad else "padded execution with key masks")),
                       "micro_batch_per_gpu": B, "grad_accum": A, "global_batch": B * A * world, "seq_len": 2048,
                       "response_tokens": 512, "parallelism": f"dp{world}",
                       "optimizer": ("fused AdamW once per step (fp32 master), global-norm clipping "
                                     f"{args.max_grad_norm if args.max_grad_norm else 'off'}" +
                                     (f", ZeRO-2 style: {args.grad_dtype} reduce-scatter -> sharded AdamW -> bf16 all-gather"
                                      if (world > 1 and not args.no_zero2) else (", fp32 all-reduce" if world > 1 else "")) +
                                     ("" if not args.optimizer_overlap else ", just-in-time on a second stream")),
                       "image_tower": ("different weights per model, run twice" if args.separate_towers or args.stage != "mimic" else
                                       "student and teacher towers bit-identical (same checkpoint): features computed once per batch, shared"),
                       "teacher_pipelining": ("teacher fwd of batch i+1 on a side stream under the student's step i (student on a "
                                              "high-priority stream)") if pipelined else "off",
                       "trainable_params": n_train, "lm_head_rows": "loss rows only (513 of 2048 per sample)",
                       "final_loss": round(loss_val, 4),
                       "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)},
            "roofline": {"bound": "mfma", "kernel": f"{dom_label} [{dom_shape[0]}x{dom_shape[1]}x{dom_shape[2]}]",
                         "dominant_by": (f"top row of profiles/{dom['file']}: `{dom['name']}` {dom['pct']} % of GPU time; "
                                         f"gemm4 / gemm4t family (one K loop, different epilogues) {dom['family_pct']} %") if dom else "no committed kernel summary",
                         "achieved": round(gemm_tf, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(gemm_tf / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_note": traffic_note,
                         "launch_ms": round(gemm_ms, 4),
                         "achieved_note": "live micro-launch: 10 launches at this shape, HIP events on the launch stream",
                         "in_step": in_step,
                         "whole_step": whole_step_object(sps / world, args.stage, args.experts)},
            "optimizer_ms": optimizer_ms,
            "exchange": exchange_object(dp, opt, args.steps, comm0, comm1),
        }
        want_extras = (world == 1 and not args.no_extras and args.stage == "mimic" and args.experts == 4 and not args.ragged
                       and B == 16 and args.ep == 1)
        if want_extras:
            # config 4 on the SAME models (VERDICT r04 next #1b): the preference stage's own step — 8 pairs x accum (two student
            # forward/backward passes + two frozen-model passes per pair), kto_pair, optimizer inside — 1 warm-up + 3 timed steps
            try:
                _, step4 = make_workload("dpo", student, gb, dp, opt, 8, seed0=70000)
                dt4, l4 = timed(step4, 1, 3, first=total + 2)
                pps = 3 * 8 * A / dt4
                out["extra"] = {"config4_pairs_per_s": {
                    "value": round(pps, 4), "unit": "pairs/s", "ms_per_step": round(dt4 / 3 * 1e3, 2), "steps": 3, "warmup": 1,
                    "workload": workload_name("dpo", 4, 1, 1), "micro_batch_pairs": 8, "grad_accum": A,
                    "frac_executed": whole_step_object(pps, "dpo")["frac_executed"], "frac_ledger": whole_step_object(pps, "dpo")["frac_ledger"],
                    "final_loss": round(float(l4), 4)}}
                del step4
            except Exception as e:
                out["extra"] = {"config4_pairs_per_s": {"error": repr(e)[:300]}}
        if world == 1 and not args.no_cpu_baseline and not args.ragged:
            try:
                out["cpu_baseline"] = cpu_baseline(student, teacher, trainer, args.cpu_baseline, gb=gb, stage=args.stage, twin_device=args.twin_device,
                                                   dpo_pairs=args.dpo_pairs, forced_arm=args.cpu_forced_arm)
            except Exception as e:                              # never lose the GPU number to a host-side problem
                out["cpu_baseline"] = {"value": None, "error": repr(e)[:200]}
        if want_extras:
            # config 5's student (8 experts, top-2; ep_size 1 on one GPU) through the headline's own step: the 4-expert student
            # and its optimizer state are released first, 1 warm-up + 3 timed steps
            try:
                import gc
                trainer = step = student = gb = dp = opt = None
                gc.collect(); torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
                st8, gb8, dp8, opt8 = build_student(8)
                if not args.separate_towers:
                    st8.get_image_tower().load_state_dict(teacher.get_image_tower().state_dict())
                _, step8 = make_workload("mimic", st8, gb8, dp8, opt8, B, seed0=90000)
                dt8, l8 = timed(step8, 1, 3, first=total + 8)
                sps8 = 3 * B * A / dt8
                out.setdefault("extra", {})["config5_samples_per_s"] = {
                    "value": round(sps8, 4), "unit": "samples/s", "ms_per_step": round(dt8 / 3 * 1e3, 2), "steps": 3, "warmup": 1,
                    "workload": workload_name("mimic", 8, 1, 1), "micro_batch_per_gpu": B, "grad_accum": A,
                    "frac_executed": whole_step_object(sps8, "mimic", 8)["frac_executed"], "frac_ledger": whole_step_object(sps8, "mimic", 8)["frac_ledger"],
                    "trainable_params": sum(p.numel() for p in st8.parameters() if p.requires_grad), "final_loss": round(float(l8), 4),
                    "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
            except Exception as e:
                out.setdefault("extra", {})["config5_samples_per_s"] = {"error": repr(e)[:300]}
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
