"""End-to-end parity of the HIP distillation step against (a) the committed golden vectors and (b) the
oracle run on the same seeded inputs, on a real MI355X, through the reference's model / trainer API.

Tolerance (north_star: "within 1e-3 bf16 tolerance"): the GPU path keeps activations in bf16 exactly
where the reference's bf16 run does, the oracle computes in fp32 on the same bf16-rounded weights, so
the residual is bf16 activation rounding.  Loss scalars and sequence log-probabilities must agree to
1e-3 relative.  Quantities whose natural scale is not their own magnitude — gradients, DPO loss /
reward (differences of ~-150 sums), materialised log-probabilities, logits — are held to 2x their measured
bf16 NOISE FLOOR: the oracle's own bf16 twin (`_bf16_twin`) against the fp32 oracle, same inputs, same
routing (`_twin_run`, `_check_grads_floor`).  No flat gradient / logit bound is left in this file.
"""
import os
import sys
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import _util as U  # noqa: E402
from oracle import losses as olosses  # noqa: E402
from oracle import moe as omoe  # noqa: E402
from oracle.decoder import DecoderConfig  # noqa: E402
from oracle.llava import LlavaOracle, dpo_step, freeze_like_d2s, init_weights, mimic_step, sync_experts_from_dense  # noqa: E402
from oracle.vision import IGNORE_INDEX, IMAGE_TOKEN_INDEX, VisionConfig  # noqa: E402

DEV = "cuda"


def small_cfgs():
    vc = VisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=1,
                      image_size=28, patch_size=14, select_layer=-2)
    sc = DecoderConfig(vocab_size=512, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                       num_attention_heads=2, num_key_value_heads=1, moe_layers_idx=[0], num_experts=4,
                       top_k_experts=2, capacity_factor=1.5, eval_capacity_factor=2.0, min_capacity=0,
                       router_aux_loss_coef=0.01)
    tc = DecoderConfig(vocab_size=512, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                       num_attention_heads=2, num_key_value_heads=2)
    return vc, sc, tc


def _batch_from(g, tag):
    b = {k.split(".")[-1]: v for k, v in g.items() if k.startswith(f"{tag}.batch.")}
    return dict(input_ids=b["input_ids"], attention_mask=b["attention_mask"].bool(), labels=b["labels"],
                images=b["images"].to(DEV).to(torch.bfloat16))


def _froerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _grads_of(student):
    return {n: p.main_grad for n, p in student.named_parameters() if p.requires_grad and getattr(p, "main_grad", None) is not None}


@pytest.mark.parametrize("tag", ["plain", "ragged_kdlm"])
def test_golden_small_mimic_step(tag):
    from llavamod.engine import GradBuffer
    from llavamod.train.align_trainer import AlignTrainer
    vc, sc, tc = small_cfgs()
    ssd, tsd = U.load_golden("gpusmall_student.safetensors"), U.load_golden("gpusmall_teacher.safetensors")
    g, meta = U.load_golden("gpusmall_mimic.safetensors"), U.load_json("gpusmall_mimic.json")[tag]
    student, teacher = U.build_hip_pair(ssd, tsd, sc, tc, vc, DEV)
    for m in student.moe_layers():
        m.deterministic = True
    gb = GradBuffer(student)
    batch = _batch_from(g, tag)
    # (1) model API: logits / labels
    student.train(); teacher.eval()
    with torch.no_grad():
        so, to = student(**batch), teacher(**batch)
    assert torch.equal(so.labels.cpu(), g[f"{tag}.labels"])
    live = (batch["attention_mask"].sum(1) - 1 + 4)
    # bf16 noise floors of this very case: the oracle (which must reproduce the committed golden values) and its bf16 twin
    o_student, o_teacher = _oracle_pair_from(ssd, tsd, sc, tc, vc)
    ob = {k: g[f"{tag}.batch." + k] for k in ("input_ids", "attention_mask", "labels", "images")}
    ob["attention_mask"] = ob["attention_mask"].bool()
    o_student.train(); o_teacher.eval(); o_student.set_gate_noise([None])
    _, logs_o, s_o, t_o = mimic_step(o_student, o_teacher, ob, loss_type=meta["loss_type"], align_vocab=512)
    assert torch.allclose(s_o.logits, g[f"{tag}.student_logits"], atol=1e-5) and torch.allclose(t_o.logits, g[f"{tag}.teacher_logits"], atol=1e-5)
    fgrads, _, s_f, t_f = _twin_run(o_student, o_teacher, ob, meta["loss_type"], 512, [None])
    for name, out, twin in (("student", so, s_f), ("teacher", to, t_f)):
        ref = g[f"{tag}.{name}_logits"]
        for b in range(ref.shape[0]):
            n = int(live[b])
            d = (out.logits[b, :n].float().cpu() - ref[b, :n]).abs().max().item()
            floor = (twin.logits[b, :n].float() - ref[b, :n]).abs().max().item()
            assert d <= 2.0 * floor, (name, b, d, floor)
    # (2) trainer API: fused loss step
    tr = AlignTrainer(student, teacher, args=type("A", (), dict(moe_enable=True, distill_all_tokens=False,
                                                               loss_type=meta["loss_type"], moe_loss_enable=True))(),
                      align_vocab=512)
    gb.zero()
    loss, outs = tr.compute_loss(student, batch, return_outputs=True)
    loss.backward()
    for k in ("loss", "loss/align", "loss/moe_balance", "loss/lm"):
        got, exp = float(outs[k]), meta[k]
        assert abs(got - exp) <= 1e-3 * abs(exp), (k, got, exp)
    # (3) gradients of every trainable tensor
    grads = _grads_of(student)
    golden = {U.oracle_to_hip_key(k[len(f"{tag}.grad."):]): ref for k, ref in g.items() if k.startswith(f"{tag}.grad.")}
    assert set(golden) <= set(grads) and len(golden) > 10
    worst = _check_grads_floor(grads, golden, fgrads, f"golden small ({tag})")
    assert worst[0] > 0


def _bf16_twin(model):
    """The same oracle module with bf16 weights and activations (router kept fp32, as DeepSpeed keeps it): what the
    reference's own bf16 training run computes.  |twin - fp32 oracle| is the bf16 NOISE FLOOR of a quantity; the GPU path,
    which keeps bf16 exactly where the reference does, is held to 2x that floor wherever north_star's flat 1e-3 is not the
    natural scale of the quantity (differences of large numbers, gradients of discontinuously routed experts)."""
    import copy
    m = copy.deepcopy(model)
    for n, p in m.named_parameters():
        if "gate.wg" not in n:
            p.data = p.data.to(torch.bfloat16)
        p.grad = None
    return m


def _bf16_batch(batch):
    return {k: (v.to(torch.bfloat16) if (torch.is_tensor(v) and v.is_floating_point()) else v) for k, v in batch.items()}


def _twin_run(o_student, o_teacher, ob, loss_type, align_vocab, noises):
    """The bf16 twin of an oracle pair stepped on the same batch with the fp32 run's routing picks forced (the floor is a
    rounding question; routing is gated separately).  Call AFTER the fp32 `mimic_step` (its `last_picks` are read).
    Returns (twin gradients by product name, twin logs, twin student output, twin teacher output)."""
    tw_s, tw_t = _bf16_twin(o_student), _bf16_twin(o_teacher)
    tw_s.train(); tw_t.eval(); tw_s.set_gate_noise(noises)
    for om, ot in zip(_oracle_moes(o_student), _oracle_moes(tw_s)):
        if om.last_picks is not None:
            ot.forced = (om.last_picks[0], om.last_picks[1])
    _, logs_f, s_out, t_out = mimic_step(tw_s, tw_t, _bf16_batch(ob), loss_type=loss_type, align_vocab=align_vocab)
    fgrads = {U.oracle_to_hip_key(n): p.grad for n, p in tw_s.named_parameters() if p.grad is not None}
    return fgrads, logs_f, s_out, t_out


def _check_grads_floor(hgrads, ograds, fgrads, what=""):
    """Every gradient within 2x its own bf16 noise floor (5e-3 of the tensor's max where the floor is below that: the
    GPU path's MFMA accumulation order differs from the twin's)."""
    worst = (0.0, None, 0.0)
    for n, ref in ograds.items():
        if ref.abs().max() == 0:
            continue
        e, floor = U.relerr(hgrads[n], ref), U.relerr(fgrads[n].float(), ref)
        assert e <= max(2.0 * floor, 5e-3), (what, n, e, floor)
        worst = max(worst, (e, n, floor))
    print(f"{what}: worst gradient error {worst[0]:.4f} ({worst[1]}), its bf16 floor {worst[2]:.4f}")
    return worst


def _oracle_pair_from(ssd, tsd, sc, tc, vc):
    o_student, o_teacher = LlavaOracle(sc, vc, moe=True), LlavaOracle(tc, vc, moe=False)
    o_student.load_state_dict(ssd); o_teacher.load_state_dict(tsd)
    freeze_like_d2s(o_student)
    return o_student, o_teacher


def _seeded_pair(seed, sc, tc, vc):
    teacher = init_weights(LlavaOracle(tc, vc, moe=False), seed=seed + 100)
    student = sync_experts_from_dense(init_weights(LlavaOracle(sc, vc, moe=True), seed=seed))
    for m in (student, teacher):
        with torch.no_grad():
            for n, p in m.named_parameters():
                if "gate.wg" not in n:
                    p.copy_(p.to(torch.bfloat16).float())
    freeze_like_d2s(student)
    return student, teacher


def _mid_cfgs():
    """Real head geometry (hd 128 decoder, hd 64 ViT), few layers: the oracle finishes in seconds on the host."""
    vc = VisionConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=3, num_attention_heads=4,
                      image_size=56, patch_size=14, select_layer=-2)          # 16 patches
    sc = DecoderConfig(vocab_size=2048, hidden_size=512, intermediate_size=1024, num_hidden_layers=4,
                       num_attention_heads=4, num_key_value_heads=4, moe_layers_idx=[0, 2], num_experts=4,
                       top_k_experts=2, capacity_factor=1.5, min_capacity=0)
    tc = DecoderConfig(vocab_size=2048, hidden_size=768, intermediate_size=1536, num_hidden_layers=3,
                       num_attention_heads=6, num_key_value_heads=6)
    return vc, sc, tc


def _mid_batch(seed, B, T, vocab, img, ragged):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, vocab - 8, (B, T), generator=g)
    ids[:, 5] = IMAGE_TOKEN_INDEX
    labels = ids.clone()
    labels[:, : T // 2] = IGNORE_INDEX
    labels[:, T // 2 + 3: T // 2 + 6] = IGNORE_INDEX              # multi-span label mask
    mask = torch.ones(B, T, dtype=torch.bool)
    if ragged:
        for b in range(1, B):
            cut = T - 5 * b
            ids[b, cut:] = vocab - 1; labels[b, cut:] = IGNORE_INDEX; mask[b, cut:] = False
    images = torch.randn(B, 3, img, img, generator=g).to(torch.bfloat16).float()
    return dict(input_ids=ids, attention_mask=mask, labels=labels, images=images)


def _oracle_moes(o_student):
    return [l.mlp for l in o_student.lm.model.layers if hasattr(l.mlp, "deepspeed_moe")]


def _routing_report(student, o_student):
    """Fraction of tokens whose (1st, 2nd) expert picks agree between the bf16 GPU path and the fp32 oracle,
    and the largest decision margin (in the oracle's own scores) among the disagreements."""
    agree, total, worst_margin = 0, 0, 0.0
    for hm, om in zip(student.moe_layers(), _oracle_moes(o_student)):
        i1, i2, g, lw = om.last_picks
        h1, h2 = hm.last_state.idx1.cpu().long(), hm.last_state.idx2.cpu().long()
        same = (h1 == i1) & (h2 == i2)
        agree += int(same.sum()); total += same.numel()
        for t in torch.nonzero(~same).flatten().tolist():
            m1 = (g[t, i1[t]] - g[t, h1[t]]).abs().item()
            m2 = (lw[t, i2[t]] - lw[t, h2[t]]).abs().item() if h1[t] == i1[t] else 0.0
            worst_margin = max(worst_margin, m1, m2 / max(1.0, lw[t].abs().max().item()))
    return agree / max(1, total), worst_margin


def _mimic_parity_case(vc, sc, tc, seed, batch, noises, tag, min_agree=0.97, distill_all=False, margs=None):
    """One mimic step vs the oracle on the same seeded inputs.  The router argmax is discontinuous: bf16 activations can
    break a near-tie differently from the fp32 oracle for a few tokens.  So the case (1) requires >= `min_agree` of the routing
    decisions to agree and every disagreement to be a near-tie, then (2) re-runs the oracle with the GPU path's expert picks
    forced and requires tight agreement of the losses (1e-3) and of every trainable tensor's gradient (2x its bf16 floor)."""
    from llavamod.engine import GradBuffer
    from llavamod.train.align_trainer import AlignTrainer
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    o_student, o_teacher = _seeded_pair(seed, sc, tc, vc)
    student, teacher = U.build_hip_pair(o_student.state_dict(), o_teacher.state_dict(), sc, tc, vc, DEV, margs=margs)
    for m, nz in zip(student.moe_layers(), noises):
        m.deterministic = nz is None
        m.gate_noise = nz
    GradBuffer(student)
    hb = dict(batch, images=batch["images"].to(DEV).to(torch.bfloat16))
    Va = min(olosses.ALIGN_VOCAB, sc.vocab_size, tc.vocab_size)
    tr = AlignTrainer(student, teacher, args=type("A", (), dict(moe_enable=True, distill_all_tokens=distill_all,
                                                               loss_type="kd_lm", moe_loss_enable=True))(), align_vocab=Va)
    student.train()
    loss, outs = tr.compute_loss(student, hb, return_outputs=True)
    loss.backward()
    # (1) free-running oracle: routing agreement
    o_student.train(); o_teacher.eval(); o_student.set_gate_noise(noises)
    mimic_step(o_student, o_teacher, batch, loss_type="kd_lm", align_vocab=Va, distill_all_tokens=distill_all)
    frac, margin = _routing_report(student, o_student)
    assert frac >= min_agree, frac
    assert margin <= 2e-2, margin
    # (2) oracle with the GPU picks forced: tight parity
    o_student.zero_grad()
    for hm, om in zip(student.moe_layers(), _oracle_moes(o_student)):
        om.forced = (hm.last_state.idx1.cpu(), hm.last_state.idx2.cpu())
    loss_o, logs_o, _, _ = mimic_step(o_student, o_teacher, batch, loss_type="kd_lm", align_vocab=Va, distill_all_tokens=distill_all)
    for k in ("loss", "loss/align", "loss/moe_balance", "loss/lm"):
        got, exp = float(outs[k]), float(logs_o[k])
        assert abs(got - exp) <= 1e-3 * abs(exp), (k, got, exp)
    ograds = {U.oracle_to_hip_key(n): p.grad for n, p in o_student.named_parameters() if p.grad is not None}
    hgrads = _grads_of(student)
    assert set(ograds) == set(hgrads), sorted(set(ograds) ^ set(hgrads))[:8]
    # bf16 noise floor of every gradient: the oracle's bf16 twin with the same forced picks
    tw_s, tw_t = _bf16_twin(o_student), _bf16_twin(o_teacher)
    tw_s.train(); tw_t.eval(); tw_s.set_gate_noise(noises)
    for hm, om in zip(student.moe_layers(), _oracle_moes(tw_s)):
        om.forced = (hm.last_state.idx1.cpu(), hm.last_state.idx2.cpu())
    _, logs_f, _, _ = mimic_step(tw_s, tw_t, _bf16_batch(batch), loss_type="kd_lm", align_vocab=Va, distill_all_tokens=distill_all)
    fgrads = {U.oracle_to_hip_key(n): p.grad for n, p in tw_s.named_parameters() if p.grad is not None}
    worst = _check_grads_floor(hgrads, ograds, fgrads, tag)
    print(f"{tag}: routing agreement {frac * 100:.1f} %; loss floor "
          f"{abs(float(logs_f['loss']) - float(logs_o['loss'])) / abs(float(logs_o['loss'])):.2e}")
    return SimpleNamespace(worst=worst, student=student, teacher=teacher, logs_o=logs_o)


@pytest.mark.parametrize("ragged,noise", [(False, True), (True, False)])
def test_seeded_mid_mimic_step_vs_oracle(ragged, noise):
    """Mid-size mimic step (hd 128 decoder, hd 64 ViT, multi-span labels; ragged / Gumbel-noise variants) vs the oracle."""
    vc, sc, tc = _mid_cfgs()
    batch = _mid_batch(11, 3, 48, sc.vocab_size, vc.image_size, ragged)
    Sp = batch["input_ids"].shape[1] - 1 + vc.num_patches
    noises = [omoe.gumbel_noise((3 * Sp, sc.num_experts), torch.Generator().manual_seed(20 + i)) if noise else None
              for i in range(len(sc.moe_layers_idx))]
    _mimic_parity_case(vc, sc, tc, 3, batch, noises, f"mid mimic (ragged={ragged}, noise={noise})")


def test_qwen2_geometry_mimic_step_vs_oracle():
    """The Qwen2 shells' geometry (VERDICT r02 missing #4 / next #6b; dense2sparse_distillation.sh:20 trains
    `llavaqwen-2-0.5b`): student = Qwen2-0.5B attention shape — H 896, 14 heads of 64, 2 KV heads (GQA group 7), rope theta
    1e6, QKV bias — with V 151936; teacher = Qwen2-7B attention shape — H 3584, 28 heads of 128, 4 KV heads (group 7) — with
    V 152064 > 151936, so `get_p` / `get_logp`'s `[:, :, :151936]` slice (align_trainer.py:473,497) is live.  Two layers each
    (MoE on the student's layer 0), FFN widths cut 4x (1216 / 4736) to keep the CPU oracle in seconds; ragged batch."""
    vc = VisionConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=3, num_attention_heads=4,
                      image_size=56, patch_size=14, select_layer=-2)
    sc = DecoderConfig(vocab_size=151936, hidden_size=896, intermediate_size=1216, num_hidden_layers=2,
                       num_attention_heads=14, num_key_value_heads=2, moe_layers_idx=[0], num_experts=4,
                       top_k_experts=2, capacity_factor=1.5, min_capacity=0, rope_theta=1000000.0)
    tc = DecoderConfig(vocab_size=152064, hidden_size=3584, intermediate_size=4736, num_hidden_layers=2,
                       num_attention_heads=28, num_key_value_heads=4, rope_theta=1000000.0)
    assert sc.hidden_size // sc.num_attention_heads == 64 and tc.hidden_size // tc.num_attention_heads == 128
    batch = _mid_batch(31, 2, 44, 151000, vc.image_size, True)
    _mimic_parity_case(vc, sc, tc, 9, batch, [None], "Qwen2 geometry (hd 64 GQA-7 student, V 152064 teacher)")


def test_mid_mimic_step_free_running_routing():
    """No routing hook at all: a seed (found offline with the CPU oracle, 600 candidates) for which EVERY routing decision of
    both MoE layers — first and second pick — has a margin > 4e-2 of the token's largest |router logit| in the oracle.  bf16
    hidden states carry ~1 % relative noise after a few layers (tests/test_full_width_gpu.py measures it), so the GPU path
    must then make exactly the same picks on its own, and losses / gradients must agree without forcing anything."""
    from llavamod.engine import GradBuffer
    from llavamod.train.align_trainer import AlignTrainer
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    vc, sc, tc = _mid_cfgs()
    o_student, o_teacher = _seeded_pair(613, sc, tc, vc)
    with torch.no_grad():
        for m in _oracle_moes(o_student):
            m.deepspeed_moe.gate.wg.weight.mul_(32.0)
    batch = _mid_batch(620, 1, 12, sc.vocab_size, vc.image_size, False)
    o_student.train(); o_teacher.eval(); o_student.set_gate_noise([None, None])
    loss_o, logs_o, _, _ = mimic_step(o_student, o_teacher, batch, loss_type="kd_lm", align_vocab=sc.vocab_size)
    for m in _oracle_moes(o_student):                          # the premise: no near-tie anywhere
        i1, i2, g, lw = m.last_picks
        ls = lw.sort(dim=1, descending=True).values
        scale = lw.abs().max(dim=1).values.clamp_min(1.0)
        assert ((ls[:, 0] - ls[:, 1]) / scale).min().item() > 4e-2 and ((ls[:, 1] - ls[:, 2]) / scale).min().item() > 4e-2
    student, teacher = U.build_hip_pair(o_student.state_dict(), o_teacher.state_dict(), sc, tc, vc, DEV)
    for m in student.moe_layers():
        m.deterministic = True
    GradBuffer(student)
    hb = dict(batch, images=batch["images"].to(DEV).to(torch.bfloat16))
    tr = AlignTrainer(student, teacher, args=type("A", (), dict(moe_enable=True, distill_all_tokens=False,
                                                               loss_type="kd_lm", moe_loss_enable=True))(),
                      align_vocab=sc.vocab_size)
    student.train()
    loss, outs = tr.compute_loss(student, hb, return_outputs=True)
    loss.backward()
    for hm, om in zip(student.moe_layers(), _oracle_moes(o_student)):
        i1, i2, _, _ = om.last_picks
        assert torch.equal(hm.last_state.idx1.cpu().long(), i1) and torch.equal(hm.last_state.idx2.cpu().long(), i2)
    for k in ("loss", "loss/align", "loss/moe_balance", "loss/lm"):
        got, exp = float(outs[k].detach()), float(logs_o[k].detach())
        assert abs(got - exp) <= 1e-3 * abs(exp), (k, got, exp)
    tw_s, tw_t = _bf16_twin(o_student), _bf16_twin(o_teacher)
    tw_s.train(); tw_t.eval(); tw_s.set_gate_noise([None, None])
    mimic_step(tw_s, tw_t, _bf16_batch(batch), loss_type="kd_lm", align_vocab=sc.vocab_size)
    ograds = {U.oracle_to_hip_key(n): p.grad for n, p in o_student.named_parameters() if p.grad is not None}
    fgrads = {U.oracle_to_hip_key(n): p.grad for n, p in tw_s.named_parameters() if p.grad is not None}
    hgrads = _grads_of(student)
    assert set(ograds) == set(hgrads)
    for n, ref in ograds.items():
        e, floor = U.relerr(hgrads[n], ref), U.relerr(fgrads[n].float(), ref)
        assert e <= max(2.0 * floor, 5e-3), (n, e, floor)


def test_mid_mimic_step_free_running_270_tokens():
    """Free-running at 270 tokens with the PLAIN router init (no scaling of `wg`, no seed search, no forced picks anywhere;
    VERDICT r02 next #6d).  With N(0, 0.02) routers the median decision margin is ~0.1 and the bf16 noise of a router logit is
    ~0.003-0.01 (the oracle's own bf16 twin flips 2-4 of the 270 x 2 decisions per layer), so identical picks cannot be demanded
    at this size — a seed for which all ~1080 decisions clear the noise does not exist in practice (P ~ 0.9^1080).  What is
    demanded instead, of two runs that each route on their own:
      * >= 98 % of the decisions agree and every disagreement is a near-tie (margin <= 2e-2 of the token's largest logit);
      * the four logged loss scalars agree to north_star's 1e-3 relative — flips move single tokens between near-equal experts,
        the losses are averages over hundreds of tokens;
      * gradients of everything that is not an expert / router tensor (projector, dense FFNs) are within 2x their bf16 floor
        (the twin's floor is taken free-running as well, so it contains the twin's own flips);
      * expert and router gradients: relative Frobenius error within 2x the twin's own free-running Frobenius error + the mass
        the flipped tokens can move, 2 * sqrt(flips / routed tokens)."""
    from llavamod.engine import GradBuffer
    from llavamod.train.align_trainer import AlignTrainer
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    vc, sc, tc = _mid_cfgs()
    o_student, o_teacher = _seeded_pair(1001, sc, tc, vc)
    batch = _mid_batch(1008, 2, 120, sc.vocab_size, vc.image_size, False)
    T = 2 * (120 - 1 + vc.num_patches)
    assert T >= 256
    o_student.train(); o_teacher.eval(); o_student.set_gate_noise([None, None])
    loss_o, logs_o, _, _ = mimic_step(o_student, o_teacher, batch, loss_type="kd_lm", align_vocab=sc.vocab_size)
    student, teacher = U.build_hip_pair(o_student.state_dict(), o_teacher.state_dict(), sc, tc, vc, DEV)
    for m in student.moe_layers():
        m.deterministic = True
    GradBuffer(student)
    hb = dict(batch, images=batch["images"].to(DEV).to(torch.bfloat16))
    tr = AlignTrainer(student, teacher, args=type("A", (), dict(moe_enable=True, distill_all_tokens=False,
                                                               loss_type="kd_lm", moe_loss_enable=True))(),
                      align_vocab=sc.vocab_size)
    student.train()
    loss, outs = tr.compute_loss(student, hb, return_outputs=True)
    loss.backward()
    frac, margin = _routing_report(student, o_student)
    flips = []
    for hm, om in zip(student.moe_layers(), _oracle_moes(o_student)):
        i1, i2, _, _ = om.last_picks
        flips.append(int(((hm.last_state.idx1.cpu().long() != i1) | (hm.last_state.idx2.cpu().long() != i2)).sum()))
    assert frac >= 0.98 and margin <= 2e-2, (frac, margin, flips)
    for k in ("loss", "loss/align", "loss/moe_balance", "loss/lm"):
        got, exp = float(outs[k].detach()), float(logs_o[k].detach())
        assert abs(got - exp) <= 1e-3 * abs(exp), (k, got, exp, flips)
    # the bf16 twin, free-running too
    tw_s, tw_t = _bf16_twin(o_student), _bf16_twin(o_teacher)
    tw_s.train(); tw_t.eval(); tw_s.set_gate_noise([None, None])
    mimic_step(tw_s, tw_t, _bf16_batch(batch), loss_type="kd_lm", align_vocab=sc.vocab_size)
    ograds = {U.oracle_to_hip_key(n): p.grad for n, p in o_student.named_parameters() if p.grad is not None}
    fgrads = {U.oracle_to_hip_key(n): p.grad for n, p in tw_s.named_parameters() if p.grad is not None}
    hgrads = _grads_of(student)
    assert set(ograds) == set(hgrads)
    routed = 2 * T
    allow = 2.0 * (max(flips) / routed) ** 0.5
    worst_dense, worst_moe = (0.0, None, 0.0), (0.0, None, 0.0)
    for n, ref in ograds.items():
        if ref.abs().max() == 0:
            continue
        if "deepspeed_moe" in n:
            e, floor = _froerr(hgrads[n], ref), _froerr(fgrads[n].float(), ref)
            assert e <= 2.0 * floor + allow + 5e-3, (n, e, floor, allow)
            worst_moe = max(worst_moe, (e, n, floor))
        else:
            e, floor = U.relerr(hgrads[n], ref), U.relerr(fgrads[n].float(), ref)
            assert e <= max(2.0 * floor, 5e-3), (n, e, floor)
            worst_dense = max(worst_dense, (e, n, floor))
    print(f"free-running 270 tokens: {frac * 100:.2f} % of picks agree (flips per layer {flips}, worst margin {margin:.2e}); "
          f"loss {float(loss):.5f} vs {float(loss_o):.5f}; worst dense grad err {worst_dense[0]:.4f} (floor {worst_dense[2]:.4f}), "
          f"worst expert/router grad Frobenius err {worst_moe[0]:.4f} (twin {worst_moe[2]:.4f}, flip allowance {allow:.4f})")


def test_finetune_student_and_top1_step():
    """`LLaVAMoDQwen2ForCausalLMFineTune` (the class the preference stage constructs, llava_qwen2_moe.py:564-626): built
    from a saved config.moe it has the up-cycled student's exact state-dict layout, loads its weights, reproduces its
    outputs bit for bit and sets trainability by substring.  Then a top-1 student (k = 1, token-order selection) steps
    against the oracle's top1gating."""
    import copy
    from llavamod.engine import GradBuffer
    from llavamod.model.language_model.llava_qwen2_moe import LLaVAMoDQwen2ForCausalLMFineTune
    from llavamod.train.align_trainer import AlignTrainer
    vc, sc, tc = small_cfgs()
    ssd, tsd = U.load_golden("gpusmall_student.safetensors"), U.load_golden("gpusmall_teacher.safetensors")
    g = U.load_golden("gpusmall_mimic.safetensors")
    student, teacher = U.build_hip_pair(ssd, tsd, sc, tc, vc, DEV)
    cfg = copy.deepcopy(student.config)
    ft = LLaVAMoDQwen2ForCausalLMFineTune(cfg, device=DEV)
    assert list(ft.state_dict().keys()) == list(student.state_dict().keys())
    ft.load_state_dict(student.state_dict())
    ft.initialize_moe_modules(type("A", (), dict(train_modules=["mlp.gate_proj", "mlp.up_proj", "mlp.down_proj", "wg"]))())
    names = {n for n, p in ft.named_parameters() if p.requires_grad}
    assert names and all(any(t in n for t in ("mlp.gate_proj", "mlp.up_proj", "mlp.down_proj", "wg")) for n in names)
    batch = _batch_from(g, "plain")
    for m in list(student.moe_layers()) + list(ft.moe_layers()):
        m.deterministic = True
    student.train(); ft.train()
    with torch.no_grad():
        a, b = student(**batch), ft(**batch)
    assert torch.equal(a.logits, b.logits) and float(a.loss) == float(b.loss) and float(a.moe_loss) == float(b.moe_loss)
    # top-1 gating, k = 1 (token-order capacity selection; DeepSpeed's random token selection needs explicit noise)
    sc1 = copy.deepcopy(sc)
    sc1.top_k_experts = 1
    o_student = LlavaOracle(sc1, vc, moe=True)
    o_student.load_state_dict(ssd)
    o_teacher = LlavaOracle(tc, vc, moe=False)
    o_teacher.load_state_dict(tsd)
    freeze_like_d2s(o_student)
    ob = {k: g["plain.batch." + k] for k in ("input_ids", "attention_mask", "labels", "images")}
    ob["attention_mask"] = ob["attention_mask"].bool()
    o_student.train(); o_teacher.eval()
    for m in _oracle_moes(o_student):
        m.rts_noise = None                                      # token order
    loss_o, logs_o, _, _ = mimic_step(o_student, o_teacher, ob, loss_type="kd_lm", align_vocab=512)
    s1, t1 = U.build_hip_pair(ssd, tsd, sc1, tc, vc, DEV)
    for m in s1.moe_layers():
        m.deterministic = True                                  # token order (random token selection: kernel tests)
    GradBuffer(s1)
    tr = AlignTrainer(s1, t1, args=type("A", (), dict(moe_enable=True, distill_all_tokens=False, loss_type="kd_lm",
                                                      moe_loss_enable=True))(), align_vocab=512)
    s1.train()
    loss, outs = tr.compute_loss(s1, batch, return_outputs=True)
    loss.backward()
    for k in ("loss", "loss/align", "loss/moe_balance", "loss/lm"):
        got, exp = float(outs[k].detach()), float(logs_o[k].detach())
        assert abs(got - exp) <= 1e-3 * abs(exp), (k, got, exp)
    ograds = {U.oracle_to_hip_key(n): p.grad for n, p in o_student.named_parameters() if p.grad is not None}
    fgrads, _, _, _ = _twin_run(o_student, o_teacher, ob, "kd_lm", 512, [None])
    _check_grads_floor(_grads_of(s1), ograds, fgrads, "top-1 student")


@pytest.mark.parametrize("everything", [False, True])
def test_trainable_norms_and_embedding_step(everything):
    """everything=True: empty `train_modules` (the reference's default: every decoder / projector parameter trainable —
    attention q/k/v/o with biases, lm_head, norms, embeddings; the CLIP tower stays frozen as in clip_encoder.py:31).
    A run that unfreezes the decoder's RMSNorm scales and embed_tokens (not the distillation shells' train_modules, but
    reachable through --train_modules / FineTune): their gradients come from lmod_rmsnorm_dw / lmod_embed_wgrad and match the
    oracle's, alongside the usual set; a parameter with no gradient kernel (CLIP tower) raises at GradBuffer construction."""
    from llavamod.engine import GradBuffer
    from llavamod.train.align_trainer import AlignTrainer
    vc, sc, tc = small_cfgs()
    ssd, tsd = U.load_golden("gpusmall_student.safetensors"), U.load_golden("gpusmall_teacher.safetensors")
    g = U.load_golden("gpusmall_mimic.safetensors")
    student, teacher = U.build_hip_pair(ssd, tsd, sc, tc, vc, DEV)
    extra = ("layernorm.weight", "model.norm.weight", "embed_tokens.weight")
    for n, p in student.named_parameters():
        if (everything or any(n.endswith(e) for e in extra)) and "image_tower" not in n:
            p.requires_grad = True
    for m in student.moe_layers():
        m.deterministic = True
    gb = GradBuffer(student)
    o_student, o_teacher = LlavaOracle(sc, vc, moe=True), LlavaOracle(tc, vc, moe=False)
    o_student.load_state_dict(ssd); o_teacher.load_state_dict(tsd)
    freeze_like_d2s(o_student)
    for n, p in o_student.named_parameters():
        if (everything or any(n.endswith(e) for e in extra)) and "image_tower" not in n:
            p.requires_grad_(True)
    ob = {k: g["ragged_kdlm.batch." + k] for k in ("input_ids", "attention_mask", "labels", "images")}
    ob["attention_mask"] = ob["attention_mask"].bool()
    o_student.train(); o_teacher.eval(); o_student.set_gate_noise([None])
    loss_o, logs_o, _, _ = mimic_step(o_student, o_teacher, ob, loss_type="kd_lm", align_vocab=512)
    tr = AlignTrainer(student, teacher, args=type("A", (), dict(moe_enable=True, distill_all_tokens=False, loss_type="kd_lm",
                                                               moe_loss_enable=True))(), align_vocab=512)
    student.train(); gb.zero()
    loss, outs = tr.compute_loss(student, _batch_from(g, "ragged_kdlm"), return_outputs=True)
    loss.backward()
    assert abs(float(loss) - float(loss_o)) <= 1e-3 * abs(float(loss_o))
    ograds = {U.oracle_to_hip_key(n): p.grad for n, p in o_student.named_parameters() if p.grad is not None}
    hgrads = _grads_of(student)
    assert set(ograds) == set(hgrads), sorted(set(ograds) ^ set(hgrads))[:8]
    new = [n for n in ograds if any(n.endswith(e) for e in extra)]
    assert len(new) == 2 * sc.num_hidden_layers + 2
    fgrads, _, _, _ = _twin_run(o_student, o_teacher, ob, "kd_lm", 512, [None])
    _check_grads_floor(hgrads, ograds, fgrads, f"trainable norms/embedding (everything={everything})")
    # second backward accumulates (gradient accumulation contract of main_grad)
    before = {n: hgrads[n].clone() for n in new}
    loss2 = tr.compute_loss(student, _batch_from(g, "ragged_kdlm"))
    loss2.backward()
    for n in new:
        assert U.relerr(hgrads[n], 2 * before[n]) <= 1e-3, n
    # no silent zero-gradient parameters: a trainable CLIP tower is refused
    s2, _ = U.build_hip_pair(ssd, tsd, sc, tc, vc, DEV)
    next(s2.get_image_tower().parameters()).requires_grad = True
    with pytest.raises(NotImplementedError):
        GradBuffer(s2)


@pytest.mark.parametrize("loss_type", ["sigmoid", "kto_pair"])
def test_seeded_mid_dpo_step_vs_oracle(loss_type):
    from llavamod.engine import GradBuffer
    from llavamod.train.dpo_trainer import DPOTrainer
    vc, sc, tc = _mid_cfgs()
    o_student, o_teacher = _seeded_pair(5, sc, tc, vc)
    ch = _mid_batch(21, 2, 40, sc.vocab_size, vc.image_size, False)
    rj = _mid_batch(22, 2, 40, sc.vocab_size, vc.image_size, True)
    batch = dict(chosen_input_ids=ch["input_ids"], chosen_labels=ch["labels"], chosen_attention_mask=ch["attention_mask"],
                 rejected_input_ids=rj["input_ids"], rejected_labels=rj["labels"],
                 rejected_attention_mask=rj["attention_mask"], images=ch["images"])
    student, teacher = U.build_hip_pair(o_student.state_dict(), o_teacher.state_dict(), sc, tc, vc, DEV)
    picks = {}
    for li, m in enumerate(student.moe_layers()):            # record the routing picks of every call (chosen, rejected)
        m.deterministic = True
        picks[li] = []
        m.register_forward_hook(lambda mod, a, o, li=li: picks[li].append((mod.last_state.idx1.cpu(), mod.last_state.idx2.cpu())))
    GradBuffer(student)
    hb = dict(batch, images=batch["images"].to(DEV).to(torch.bfloat16))
    tr = DPOTrainer(student, teacher, beta=0.1, loss_type=loss_type)
    student.train()
    loss, outs = tr.compute_loss(student, hb, return_outputs=True)
    loss.backward()
    # oracle with the GPU path's picks forced, call by call (a discontinuous argmax is not a rounding question; the mimic
    # tests gate the agreement rate, this one checks the arithmetic around it)
    o_student.train(); o_teacher.eval(); o_student.set_gate_noise([None, None])
    for li, om in enumerate(_oracle_moes(o_student)):
        om.forced = list(picks[li])
    loss_o, logs_o = dpo_step(o_student, o_teacher, batch, beta=0.1, loss_type=loss_type)
    # bf16 noise floor: the oracle's bf16 twin on the same batch with the same picks
    tw_s, tw_t = _bf16_twin(o_student), _bf16_twin(o_teacher)
    tw_s.train(); tw_t.eval(); tw_s.set_gate_noise([None, None])
    for li, om in enumerate(_oracle_moes(tw_s)):
        om.forced = list(picks[li])
    bb = {k: (v.to(torch.bfloat16) if (torch.is_tensor(v) and v.is_floating_point()) else v) for k, v in batch.items()}
    _, logs_f = dpo_step(tw_s, tw_t, bb, beta=0.1, loss_type=loss_type)
    # sequence log-probs are sums of ~20 token logps (~ -150): 1e-3 relative; loss / reward are differences of those sums:
    # their natural scale is the noise floor of the difference
    for k in ("logps/chosen", "logps/rejected"):
        assert abs(float(outs[k]) - float(logs_o[k])) <= 1e-3 * abs(float(logs_o[k])), (k, float(outs[k]), float(logs_o[k]))
    for k in ("loss", "loss/reward", "loss/moe_balance"):
        floor = abs(float(logs_f[k]) - float(logs_o[k]))
        assert abs(float(outs[k]) - float(logs_o[k])) <= max(2.0 * floor, 1e-3 * abs(float(logs_o[k]))), \
            (k, float(outs[k]), float(logs_o[k]), floor)
    ograds = {U.oracle_to_hip_key(n): p.grad for n, p in o_student.named_parameters() if p.grad is not None}
    fgrads = {U.oracle_to_hip_key(n): p.grad for n, p in tw_s.named_parameters() if p.grad is not None}
    hgrads = _grads_of(student)
    assert set(ograds) == set(hgrads)
    for n, ref in ograds.items():
        e, floor = _froerr(hgrads[n], ref), _froerr(fgrads[n].float(), ref)
        assert e <= max(2.0 * floor, 1e-2), (n, e, floor)


def _token_logps(logits, labels):
    """log p(label) at every labelled, shifted position of every sample (dpo_trainer.py:483-495 before the sum): fp32 [n]."""
    lb = labels[:, 1:]
    keep = lb != IGNORE_INDEX
    lg = logits[:, :-1].float().log_softmax(-1)
    return torch.gather(lg, 2, lb.clamp_min(0).unsqueeze(2).to(lg.device)).squeeze(2)[keep.to(lg.device)].detach().cpu()


def test_per_token_logp_error_is_at_the_bf16_floor():
    """The preference stage's scalars are sigmoids of differences of SUMS of per-token log-probabilities, so one sample of them says
    little about a kernel's accuracy (bench.py --stage dpo prints why).  The testable statistic is the per-token error itself, over
    k = 4 INDEPENDENT 2 x 160-token batches (n > 1): for the dense teacher (no routing) and for the MoE student
    (oracle routed with the product's own picks), against the fp32 oracle —
      * root mean square of log p(label) errors, pooled over all batches, held to 1.5x the same statistic of the oracle's bf16 twin
        (the reference's own bf16 arithmetic);
      * NO SYSTEMATIC OFFSET: |mean error| <= 3 sigma / sqrt(n) with sigma the product's own per-token spread and n every labelled
        token of the four batches — the mean is the quantity a sequence log-probability (a sum over tokens) inherits;
      * every sequence's summed error (what a DPO reward sees) within 4 sigma sqrt(T) of zero."""
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    vc, sc, tc = _mid_cfgs()
    o_student, o_teacher = _seeded_pair(17, sc, tc, vc)
    student, teacher = U.build_hip_pair(o_student.state_dict(), o_teacher.state_dict(), sc, tc, vc, DEV)
    picks = {}
    for li, m in enumerate(student.moe_layers()):
        m.deterministic = True
        picks[li] = []
        m.register_forward_hook(lambda mod, a, o, li=li: picks[li].append((mod.last_state.idx1.cpu(), mod.last_state.idx2.cpu())))
    student.train(); teacher.eval()
    o_student.train(); o_teacher.eval(); o_student.set_gate_noise([None, None])
    tw_s, tw_t = _bf16_twin(o_student), _bf16_twin(o_teacher)
    tw_s.train(); tw_t.eval(); tw_s.set_gate_noise([None, None])
    pooled = {"teacher": ([], []), "student": ([], [])}
    seq_sums = {"teacher": [], "student": []}
    for seed in (33, 34, 35, 36):
        batch = _mid_batch(seed, 2, 160, sc.vocab_size, vc.image_size, True)
        hb = dict(batch, images=batch["images"].to(DEV).to(torch.bfloat16))
        bb = _bf16_batch(batch)
        for li in picks:
            picks[li].clear()
        with torch.no_grad():
            hs, ht = student(**hb), teacher(**hb)
        for li, (om, ot) in enumerate(zip(_oracle_moes(o_student), _oracle_moes(tw_s))):
            om.forced = list(picks[li]); ot.forced = list(picks[li])
        with torch.no_grad():
            os_, ot_ = o_student(**batch), o_teacher(**batch)
            ts_, tt_ = tw_s(**bb), tw_t(**bb)
        for name, prod, ref, twin in (("teacher", ht, ot_, tt_), ("student", hs, os_, ts_)):
            assert torch.equal(prod.labels.cpu(), ref.labels)
            r = _token_logps(ref.logits, ref.labels)
            e_prod = (_token_logps(prod.logits, ref.labels) - r).double()
            pooled[name][0].append(e_prod)
            pooled[name][1].append((_token_logps(twin.logits, ref.labels) - r).double())
            keep = (ref.labels[:, 1:] != IGNORE_INDEX)
            off = 0
            for row in keep:                                     # per sequence: the summed error a reward would inherit
                n_row = int(row.sum())
                if n_row:
                    seq_sums[name].append((float(e_prod[off:off + n_row].sum()), n_row))
                off += n_row
    for name, (ep, et) in pooled.items():
        e_prod, e_twin = torch.cat(ep), torch.cat(et)
        n = e_prod.numel()
        rms_p, rms_t = float(e_prod.pow(2).mean().sqrt()), float(e_twin.pow(2).mean().sqrt())
        mean_p, sd_p = float(e_prod.mean()), float(e_prod.std())
        mean_t, sd_t = float(e_twin.mean()), float(e_twin.std())
        print(f"{name}: {n} labelled tokens over 4 batches; per-token log-prob error vs fp32: product rms {rms_p:.5f} mean {mean_p:+.6f} "
              f"+- {sd_p / n ** 0.5:.6f} ({abs(mean_p) / (sd_p / n ** 0.5):.2f} standard errors); bf16 twin rms {rms_t:.5f} mean {mean_t:+.6f} "
              f"+- {sd_t / n ** 0.5:.6f} ({abs(mean_t) / (sd_t / n ** 0.5):.2f} standard errors)")
        assert n >= 480 and rms_p <= 1.5 * rms_t + 1e-4, (name, rms_p, rms_t)
        assert abs(mean_p) <= 3.0 * sd_p / n ** 0.5 + 1e-5, (name, mean_p, sd_p, n)
        for dev, t in seq_sums[name]:
            assert abs(dev) <= 4.0 * max(rms_p, rms_t) * t ** 0.5 + 1e-4, (name, dev, t, rms_p)


def test_materialising_api_matches_oracle():
    """get_p / get_logp / compute_align_loss on materialised tensors (slow-path API parity)."""
    from llavamod.train.align_trainer import AlignTrainer
    vc, sc, tc = small_cfgs()
    ssd, tsd = U.load_golden("gpusmall_student.safetensors"), U.load_golden("gpusmall_teacher.safetensors")
    g = U.load_golden("gpusmall_mimic.safetensors")
    student, teacher = U.build_hip_pair(ssd, tsd, sc, tc, vc, DEV)
    for m in student.moe_layers():
        m.deterministic = True
    batch = _batch_from(g, "plain")
    tr = AlignTrainer(student, teacher, align_vocab=512)
    with torch.no_grad():
        p, _, _ = tr.get_p(teacher, batch)
        lp, sft, moe, labels = tr.get_logp(student, batch)
        al = tr.compute_align_loss(lp, p, labels)
    ref_p = olosses.get_p(g["plain.teacher_logits"], 512)
    ref_lp = olosses.get_logp(g["plain.student_logits"], 512)
    assert (p.cpu() - ref_p).abs().max() < 1e-4
    # log-probabilities ~ -6: the student's bf16 noise floor is measured on the oracle's bf16 twin of the same model
    o_student = LlavaOracle(sc, vc, moe=True)
    o_student.load_state_dict(ssd)
    tw = _bf16_twin(o_student)
    tw.train(); tw.set_gate_noise([None])
    ob = {k: g["plain.batch." + k] for k in ("input_ids", "attention_mask", "labels", "images")}
    ob["attention_mask"] = ob["attention_mask"].bool()
    with torch.no_grad():
        floor = (olosses.get_logp(tw(**_bf16_batch(ob)).logits.float(), 512) - ref_lp).abs().max().item()
    assert (lp.cpu() - ref_lp).abs().max().item() <= max(2.0 * floor, 1e-3), ((lp.cpu() - ref_lp).abs().max().item(), floor)
    ref_al = olosses.compute_align_loss(ref_lp, ref_p, g["plain.labels"])
    assert abs(float(al) - float(ref_al)) <= 1e-3 * abs(float(ref_al))


def test_teacher_row_trimmed_forward_is_exact():
    """no-grad forward with a loss plan returns only the plan's rows, bit-identical to gathering the full forward."""
    from llavamod.model.language_model.llava_qwen2 import build_loss_plan
    vc, sc, tc = small_cfgs()
    ssd, tsd = U.load_golden("gpusmall_student.safetensors"), U.load_golden("gpusmall_teacher.safetensors")
    g = U.load_golden("gpusmall_mimic.safetensors")
    _, teacher = U.build_hip_pair(ssd, tsd, sc, tc, vc, DEV)
    batch = _batch_from(g, "ragged_kdlm")
    mk = lambda info: build_loss_plan(info.labels_np, info.lens_np, device=DEV)
    with torch.no_grad():
        full, _, info = teacher.forward_hidden(**batch)
        plan = mk(info)
        rows, _, info2 = teacher.forward_hidden(**batch, plan_fn=mk)
    assert info2.plan.pregathered and rows.shape == (plan.R, full.shape[1]) and plan.R < full.shape[0]
    assert torch.equal(rows, full[plan.row_idx.long()])
    # with autograd on the same rows come back, and the gradient w.r.t. the inputs equals the untrimmed one
    emb = torch.randn(full.shape[0], teacher.config.hidden_size, device=DEV).to(torch.bfloat16)
    B, S = info.B, info.S
    grads = []
    for trimmed in (False, True):
        e = emb.clone().requires_grad_(True)
        if trimmed:
            y, _ = teacher.model(e, B, S, None, out_rows=plan.row_idx, inv_rows=plan.inv_row_idx)
        else:
            y, _ = teacher.model(e, B, S, None)
            y = y[plan.row_idx.long()]
        w = torch.linspace(-1, 1, y.numel(), device=DEV).view_as(y)
        (y.float() * w).sum().backward()
        grads.append(e.grad.float())
    assert U.relerr(grads[1], grads[0]) < 2e-2, U.relerr(grads[1], grads[0])


def test_prefetched_teacher_equals_inline():
    """The side-stream teacher pass (engine pipelining) feeds compute_loss the same plan and logits as the inline one."""
    from llavamod.train.align_trainer import AlignTrainer
    vc, sc, tc = small_cfgs()
    ssd, tsd = U.load_golden("gpusmall_student.safetensors"), U.load_golden("gpusmall_teacher.safetensors")
    g = U.load_golden("gpusmall_mimic.safetensors")
    student, teacher = U.build_hip_pair(ssd, tsd, sc, tc, vc, DEV)
    for m in student.moe_layers():
        m.deterministic = True
    tr = AlignTrainer(student, teacher, args=type("A", (), dict(moe_enable=True, distill_all_tokens=False,
                                                               loss_type="kd_lm", moe_loss_enable=True))(), align_vocab=512)
    b0, b1 = _batch_from(g, "plain"), _batch_from(g, "ragged_kdlm")
    h0, h1 = tr.prefetch_teacher(b0), tr.prefetch_teacher(b1)          # two passes in flight on the side stream
    for batch, h in ((b0, h0), (b1, h1)):
        grads = []
        for teacher_arg in (h, None):
            for p in student.parameters():
                if getattr(p, "main_grad", None) is not None:
                    p.main_grad.zero_()
            loss = tr.compute_loss(student, batch, teacher=teacher_arg)
            loss.backward()
            grads.append((float(loss), {n: p.main_grad.clone() for n, p in student.named_parameters()
                                        if getattr(p, "main_grad", None) is not None}))
        assert grads[0][0] == grads[1][0]
        assert all(torch.equal(grads[0][1][n], grads[1][1][n]) for n in grads[0][1])


def test_shared_image_tower_features_are_exact():
    """With bit-identical frozen towers the trainer runs CLIP once per batch; loss and gradients do not change."""
    from llavamod.train.align_trainer import AlignTrainer
    vc, sc, tc = small_cfgs()
    ssd, tsd = U.load_golden("gpusmall_student.safetensors"), U.load_golden("gpusmall_teacher.safetensors")
    g = U.load_golden("gpusmall_mimic.safetensors")
    student, teacher = U.build_hip_pair(ssd, tsd, sc, tc, vc, DEV)
    for m in student.moe_layers():
        m.deterministic = True
    for p in student.get_image_tower().parameters():
        p.requires_grad = False
    args = type("A", (), dict(moe_enable=True, distill_all_tokens=False, loss_type="kd_lm", moe_loss_enable=True))()
    batch = _batch_from(g, "plain")
    tr = AlignTrainer(student, teacher, args=args, align_vocab=512)
    assert not tr._towers_identical()                         # the golden pair has different towers: nothing is shared
    teacher.get_image_tower().load_state_dict(student.get_image_tower().state_dict())
    res = []
    for share in (True, False):
        tr = AlignTrainer(student, teacher, args=args, align_vocab=512)
        tr.share_image_tower = share
        assert tr._towers_identical() == share
        for p in student.parameters():
            if getattr(p, "main_grad", None) is not None:
                p.main_grad.zero_()
        calls = {"n": 0}
        tower = student.get_image_tower()
        orig = tower.forward
        tower.forward = lambda *a, **k: (calls.__setitem__("n", calls["n"] + 1), orig(*a, **k))[1]
        loss = tr.compute_loss(student, batch)
        loss.backward()
        tower.forward = orig
        assert calls["n"] == (0 if share else 1)              # the student's tower only runs when nothing is shared
        res.append((float(loss), {n: p.main_grad.clone() for n, p in student.named_parameters()
                                  if getattr(p, "main_grad", None) is not None}))
    assert res[0][0] == res[1][0]
    assert all(torch.equal(res[0][1][n], res[1][1][n]) for n in res[0][1])


def test_prefetched_reference_equals_inline_dpo():
    from llavamod.train.dpo_trainer import DPOTrainer
    vc, sc, tc = small_cfgs()
    ssd, tsd = U.load_golden("gpusmall_student.safetensors"), U.load_golden("gpusmall_teacher.safetensors")
    g = U.load_golden("gpusmall_mimic.safetensors")
    student, teacher = U.build_hip_pair(ssd, tsd, sc, tc, vc, DEV)
    for m in student.moe_layers():
        m.deterministic = True
    a, b = _batch_from(g, "plain"), _batch_from(g, "ragged_kdlm")
    pair = dict(chosen_input_ids=a["input_ids"], chosen_labels=a["labels"], chosen_attention_mask=a["attention_mask"],
                rejected_input_ids=b["input_ids"], rejected_labels=b["labels"], rejected_attention_mask=b["attention_mask"],
                images=a["images"])
    tr = DPOTrainer(student, teacher, beta=0.1, loss_type="sigmoid")
    h = tr.prefetch_reference(pair)
    l1 = float(tr.compute_loss(student, pair, reference=h))
    l2 = float(tr.compute_loss(student, pair))
    assert l1 == l2


def test_overlapped_optimizer_is_exact_over_steps():
    """Just-in-time AdamW (second stream, forward order, gradients cleared in the same pass) == serial AdamW + memset,
    bit for bit, over several optimizer steps (weights, moments, losses)."""
    from llavamod.engine import GradBuffer, HipAdamW
    from llavamod.train.align_trainer import AlignTrainer
    vc, sc, tc = small_cfgs()
    ssd, tsd = U.load_golden("gpusmall_student.safetensors"), U.load_golden("gpusmall_teacher.safetensors")
    g = U.load_golden("gpusmall_mimic.safetensors")
    batches = [_batch_from(g, "plain"), _batch_from(g, "ragged_kdlm")]
    runs = []
    for overlap in (False, True, "serial+clear"):
        student, teacher = U.build_hip_pair(ssd, tsd, sc, tc, vc, DEV)
        for m in student.moe_layers():
            m.deterministic = True
        tr = AlignTrainer(student, teacher, args=type("A", (), dict(moe_enable=True, distill_all_tokens=False,
                                                                   loss_type="kd_lm", moe_loss_enable=True))(), align_vocab=512)
        gb = GradBuffer(student)
        opt = HipAdamW(gb, lr=1e-3, weight_decay=0.01)
        losses = []
        for i in range(4):
            gb.zero()
            losses.append(float(tr.training_step(student, batches[i % 2])))
            if overlap == "serial+clear":
                opt.step(grad_scale=1.0, clear_grads=True)
                assert gb.clean and gb.flat.abs().max().item() == 0
                continue
            opt.step(grad_scale=1.0, overlap=overlap)
            if overlap and i >= 1:
                assert gb.clean and any(fw._ready is not None for _, fw, _ in gb.spans if hasattr(fw, "_ready"))
        opt.sync()
        torch.cuda.synchronize()
        assert gb.flat.abs().max().item() == 0 if overlap is True else True    # gradients were cleared by the optimizer pass
        runs.append((losses, opt.master.clone(), opt.m.clone(), opt.v.clone(),
                     {k: v.clone() for k, v in student.state_dict().items()}))
    assert runs[0][0][0] != runs[0][0][2]                                  # the weights really moved between steps
    for other in runs[1:]:
        assert runs[0][0] == other[0], (runs[0][0], other[0])
        for a, b in zip(runs[0][1:4], other[1:4]):
            assert torch.equal(a, b)
        assert all(torch.equal(runs[0][4][k], other[4][k]) for k in runs[0][4])


def test_clipped_accumulated_step_matches_torch_reference():
    """One optimizer step over a 2-micro-batch accumulation window with global-norm clipping: the gradient buffer holds
    the SUM of the micro-batch gradients, the norm/coefficient are those of torch.nn.utils.clip_grad_norm_ on the mean
    gradient, and the update equals torch.optim.AdamW on the clipped mean gradient."""
    from llavamod.engine import GradBuffer, HipAdamW
    from llavamod.train.align_trainer import AlignTrainer
    vc, sc, tc = small_cfgs()
    ssd, tsd = U.load_golden("gpusmall_student.safetensors"), U.load_golden("gpusmall_teacher.safetensors")
    g = U.load_golden("gpusmall_mimic.safetensors")
    batches = [_batch_from(g, "plain"), _batch_from(g, "ragged_kdlm")]
    student, teacher = U.build_hip_pair(ssd, tsd, sc, tc, vc, DEV)
    for m in student.moe_layers():
        m.deterministic = True
    tr = AlignTrainer(student, teacher, args=type("A", (), dict(moe_enable=True, distill_all_tokens=False,
                                                               loss_type="kd_lm", moe_loss_enable=True))(), align_vocab=512)
    gb = GradBuffer(student)
    singles = []
    for b in batches:
        gb.zero()
        tr.training_step(student, b)
        singles.append(gb.flat.clone())
    gb.zero()
    for b in batches:                                             # accumulation window: no zero in between
        tr.training_step(student, b)
    total = singles[0] + singles[1]
    assert (gb.flat - total).abs().max().item() <= 1e-5 * total.abs().max().item()
    max_norm = 0.25 * float((gb.flat * 0.5).norm())               # force clipping
    opt = HipAdamW(gb, lr=1e-3, weight_decay=0.01, max_grad_norm=max_norm)
    p0 = opt.master.clone()
    mean_grad = gb.flat.clone() * 0.5
    opt.step(grad_scale=0.5, clear_grads=True)
    torch.cuda.synchronize()
    assert abs(float(opt.grad_norm) - float(mean_grad.norm())) <= 1e-4 * float(mean_grad.norm())
    ref_p = torch.nn.Parameter(p0.clone())
    ref_p.grad = mean_grad.clone()
    torch.nn.utils.clip_grad_norm_([ref_p], max_norm)
    ropt = torch.optim.AdamW([ref_p], lr=1e-3, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8)
    ropt.step()
    err = (opt.master - ref_p.data).abs().max().item()
    assert err <= 2e-6, err
    assert gb.flat.abs().max().item() == 0


def _nccl_zero2_worker(rank, world, port, q):
    import os
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from llavamod.engine import DataParallel, GradBuffer, HipAdamW, init_distributed
    from llavamod.train.align_trainer import AlignTrainer
    init_distributed()
    dev = f"cuda:{rank}"
    vc, sc, tc = small_cfgs()
    ssd, tsd = U.load_golden("gpusmall_student.safetensors"), U.load_golden("gpusmall_teacher.safetensors")
    g = U.load_golden("gpusmall_mimic.safetensors")
    batches = [_batch_from(g, "plain"), _batch_from(g, "ragged_kdlm")]
    res, engine_comm_calls = {}, {}
    for mode in ("allreduce", "zero2"):
        student, teacher = U.build_hip_pair(ssd, tsd, sc, tc, vc, dev)
        for m in student.moe_layers():
            m.deterministic = True
        tr = AlignTrainer(student, teacher, args=type("A", (), dict(moe_enable=True, distill_all_tokens=False,
                                                                   loss_type="kd_lm", moe_loss_enable=True))(), align_vocab=512)
        gb = GradBuffer(student)
        dp = DataParallel(zero2=(mode == "zero2"), min_shard_numel=1).attach(gb)
        opt = HipAdamW(gb, lr=1e-3, weight_decay=0.01, dp=dp, max_grad_norm=1.0)
        for i in range(3):
            gb.zero()
            tr.training_step(student, batches[(i + rank) % 2])   # the two ranks see different data
            dp.finish()
            opt.step(grad_scale=1.0 / world, clear_grads=True)
        torch.cuda.synchronize()
        if mode.endswith("native"):
            engine_comm_calls[mode] = dp._ncomm is not None      # the communicator of csrc/comm.hip was created and used
        res[mode] = {k: v.detach().float().cpu() for k, v in student.state_dict().items()}
    bad = [k for k in res["allreduce"] if not torch.allclose(res["allreduce"][k], res["zero2"][k], rtol=0, atol=1e-6)]
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bad))


def _rccl_world1_worker(port, q):
    import os
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0", LMOD_FORCE_DIST="1", LMOD_DP_FORCE="1")
    import torch.distributed as dist
    from llavamod.engine import DataParallel, GradBuffer, HipAdamW, init_distributed
    from llavamod.train.align_trainer import AlignTrainer
    rank, local, world = init_distributed()
    assert dist.get_backend() == "nccl" and world == 1
    vc, sc, tc = small_cfgs()
    ssd, tsd = U.load_golden("gpusmall_student.safetensors"), U.load_golden("gpusmall_teacher.safetensors")
    g = U.load_golden("gpusmall_mimic.safetensors")
    batches = [_batch_from(g, "plain"), _batch_from(g, "ragged_kdlm")]
    res, engine_comm_calls = {}, {}
    for mode in ("plain", "allreduce", "zero2", "zero2_bf16", "allreduce_native", "zero2_native"):
        student, teacher = U.build_hip_pair(ssd, tsd, sc, tc, vc, "cuda:0")
        for m in student.moe_layers():
            m.deterministic = True
        tr = AlignTrainer(student, teacher, args=type("A", (), dict(moe_enable=True, distill_all_tokens=False,
                                                                   loss_type="kd_lm", moe_loss_enable=True))(), align_vocab=512)
        gb = GradBuffer(student)
        if mode == "plain":
            dp = None
        else:
            # *_native (VERDICT r04 next #9a): the same engine with its world exchanges on the C-ABI collectives of the kernel library
            # (lmod_allreduce_grads / lmod_reduce_scatter_grads / lmod_allgather_params: their own RCCL communicator, a side stream)
            dp = DataParallel(zero2=mode.startswith("zero2"), min_shard_numel=1, native=mode.endswith("native"),
                              grad_dtype=torch.bfloat16 if mode.endswith("bf16") else torch.float32).attach(gb)
            assert dp.enabled and dp.zero2 == mode.startswith("zero2")
        opt = HipAdamW(gb, lr=1e-3, weight_decay=0.01, dp=dp, max_grad_norm=1.0)
        hp = torch.cuda.Stream(priority=-1)                      # the step runs on a high-priority stream, as in bench.py
        hp.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(hp):
            for i in range(3):
                gb.zero()
                for a in range(2):                               # accumulation window of 2
                    if dp is not None:
                        dp.armed = (a == 1)
                    tr.training_step(student, batches[(i + a) % 2])
                if dp is not None:
                    dp.finish()
                opt.step(grad_scale=0.5, clear_grads=True)
        torch.cuda.synchronize()
        if mode.endswith("native"):
            engine_comm_calls[mode] = dp._ncomm is not None      # the communicator of csrc/comm.hip was created and used
        res[mode] = {k: v.detach().float().cpu() for k, v in student.state_dict().items()}
    bad = {}
    for mode in ("allreduce", "zero2", "allreduce_native", "zero2_native"):   # one rank: the exchange is the identity -> identical weights
        bad[mode] = [k for k in res["plain"] if not torch.equal(res["plain"][k], res[mode][k])]
    bad["native_used"] = [m_ for m_ in ("allreduce_native", "zero2_native") if not engine_comm_calls.get(m_)]
    far = [k for k in res["plain"] if (res["plain"][k] - res["zero2_bf16"][k]).abs().max() > 2.5e-3]   # lr-sized Adam steps
    bad["zero2_bf16"] = far
    # the expert-parallel exchange through RCCL (VERDICT r03 next #6c): with LMOD_FORCE_DIST the decomposed MoE path sends its
    # counts and rows through dist.all_to_all_single on the world-of-one group — packed live rows (unequal splits) and whole
    # slabs — and must reproduce the fused single-GPU block bit for bit (output, aux loss, counts, input gradient)
    import copy
    from llavamod import engine
    from llavamod.model.language_model.qwen2_hip import Qwen2Config, Qwen2MLP, init_normal_
    from llavamod.model.moe_layer import MoE
    torch.manual_seed(0)
    mlp = init_normal_(Qwen2MLP(Qwen2Config(hidden_size=256, intermediate_size=512), "cuda"), std=0.05, seed=1)
    fused = MoE(256, mlp, num_experts=8, k=2, capacity_factor=1.0, min_capacity=0)
    with torch.no_grad():
        for i, e in enumerate(fused.deepspeed_moe.experts.deepspeed_experts):
            for p_ in e.parameters():
                p_.mul_(1.0 + 0.1 * i)
        fused.deepspeed_moe.gate.wg.weight.normal_(0, 0.5)
    x = (torch.randn(1000, 256, device="cuda") * 0.5).to(torch.bfloat16)
    dout = torch.randn(1000, 256, device="cuda").to(torch.bfloat16)
    outs = []
    before = {k: list(v) for k, v in engine.COMM.items()}
    for live in (None, True, False):
        m = copy.deepcopy(fused)
        if live is not None:
            m.force_decomposed, m.ep_live_rows = True, live
        m.train(); m.deterministic = True
        xi = x.clone().requires_grad_(True)
        o, l_aux, counts = m(xi)
        (o.float() * dout.float()).sum().backward()
        outs.append((o.detach(), l_aux.detach(), counts, xi.grad))
    sent = engine.COMM.get("all_to_all", [0, 0])[0] - before.get("all_to_all", [0, 0])[0]
    bad["ep_rccl"] = [] if sent >= 8 else [f"only {sent} all_to_all calls reached the backend"]
    for tag, r in (("live", outs[1]), ("slabs", outs[2])):
        if not (torch.equal(outs[0][0], r[0]) and torch.equal(outs[0][1], r[1]) and torch.equal(outs[0][2], r[2]) and torch.equal(outs[0][3], r[3])):
            bad["ep_rccl"].append(tag)
    # and the live-row exchange through the C-ABI `lmod_moe_all_to_all` (LMOD_DP_NATIVE=1: the expert-parallel group is the world)
    os.environ["LMOD_DP_NATIVE"] = "1"
    m = copy.deepcopy(fused)
    m.force_decomposed, m.ep_live_rows = True, True
    m.train(); m.deterministic = True
    xi = x.clone().requires_grad_(True)
    o, l_aux, counts = m(xi)
    (o.float() * dout.float()).sum().backward()
    from llavamod import comm as _comm_mod
    bad["ep_native"] = [] if (_comm_mod._SHARED is not None and torch.equal(outs[0][0], o.detach()) and torch.equal(outs[0][3], xi.grad)) else ["native live-row exchange"]
    os.environ.pop("LMOD_DP_NATIVE")
    dist.barrier()
    dist.destroy_process_group()
    q.put(bad)


def test_single_rank_rccl_exchange_paths_are_exact():
    """The N>1 code on ONE GPU (the GPU box has one): an RCCL process group of world size 1 with the exchange forced on
    (LMOD_DP_FORCE) runs the bucketed all-reduce, the in-place reduce-scatter / sharded AdamW / in-place all-gather and the
    bf16-staged exchange through real RCCL calls on the real streams (high-priority compute stream, RCCL's own stream,
    readiness hooks, 2-micro-batch accumulation window).  With one rank every collective is the identity, so the weights
    after 3 optimizer steps must equal the no-DP run bit for bit (bf16 exchange: up to bf16 rounding of the gradients)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_world1_worker, args=(port, q))
    p.start()
    p.join(600)
    assert p.exitcode == 0, p.exitcode
    bad = q.get(timeout=5)
    assert not any(bad.values()), {k: v[:4] for k, v in bad.items()}


def test_two_gpu_zero2_equals_allreduce_over_rccl():
    """N = 2 over RCCL on real devices: ZeRO-2 style sharded step == all-reduce step (self-skips on a 1-GPU box; the
    CPU twin is tests/test_dp_gloo.py)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_nccl_zero2_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    for _ in range(2):
        rank, bad = q.get(timeout=5)
        assert not bad, (rank, bad[:5])
