"""CPU suite: the oracle against the committed golden vectors (written by oracle/validate_vs_reference.py
after it pinned the oracle to the imported reference), property tests of the DeepSpeed-MoE restatement
(parity unpinned there), and the host-side index logic of the product path.  No GPU."""
import math
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import _util as U  # noqa: E402
from oracle import losses as olosses  # noqa: E402
from oracle import moe as omoe  # noqa: E402
from oracle.decoder import DecoderConfig  # noqa: E402
from oracle.llava import LlavaOracle, freeze_like_d2s, mimic_step  # noqa: E402
from oracle.vision import IGNORE_INDEX, IMAGE_TOKEN_INDEX, VisionConfig, splice  # noqa: E402


def _cfg1():
    vc = VisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=3, num_attention_heads=4,
                      image_size=28, patch_size=14, select_layer=-2)
    sc = DecoderConfig(vocab_size=512, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                       num_attention_heads=4, num_key_value_heads=2, moe_layers_idx=[0], num_experts=4,
                       top_k_experts=2, capacity_factor=1.5, eval_capacity_factor=2.0, min_capacity=0,
                       router_aux_loss_coef=0.01)
    tc = DecoderConfig(vocab_size=512, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                       num_attention_heads=4, num_key_value_heads=2)
    return vc, sc, tc


@pytest.mark.parametrize("tag", ["plain", "ragged_noise_kdlm"])
def test_oracle_reproduces_config1_golden(tag):
    """BASELINE.json configs[0]: tiny ViT + 2-layer/4-expert MoE student, 2-layer dense teacher, B=2, CPU."""
    vc, sc, tc = _cfg1()
    student, teacher = LlavaOracle(sc, vc, moe=True), LlavaOracle(tc, vc, moe=False)
    student.load_state_dict(U.load_golden("config1_student.safetensors"))
    teacher.load_state_dict(U.load_golden("config1_teacher.safetensors"))
    freeze_like_d2s(student)
    g, meta = U.load_golden("config1_mimic.safetensors"), U.load_json("config1_mimic.json")[tag]
    b = {k.split(".")[-1]: v for k, v in g.items() if k.startswith(f"{tag}.batch.")}
    batch = dict(input_ids=b["input_ids"], attention_mask=b["attention_mask"].bool(), labels=b["labels"], images=b["images"])
    student.train(); teacher.eval()
    student.set_gate_noise([g.get(f"{tag}.gate_noise")])
    loss, logs, s_out, t_out = mimic_step(student, teacher, batch, loss_type=meta["loss_type"], align_vocab=512)
    assert torch.allclose(s_out.logits, g[f"{tag}.student_logits"], atol=1e-5)
    assert torch.allclose(t_out.logits, g[f"{tag}.teacher_logits"], atol=1e-5)
    assert torch.equal(s_out.labels, g[f"{tag}.labels"])
    for k in ("loss", "loss/align", "loss/moe_balance", "loss/lm"):
        assert abs(float(logs[k]) - meta[k]) < 1e-5 * max(1.0, abs(meta[k])), k
    params = dict(student.named_parameters())
    for k, ref in g.items():
        if k.startswith(f"{tag}.grad."):
            assert torch.allclose(params[k[len(f"{tag}.grad."):]].grad, ref, atol=1e-6), k


def test_dpo_known_answers_from_reference():
    ka = U.load_json("dpo_known_answers.json")
    pc, pr, rc, rr = (torch.tensor(ka[k]) for k in ("pc", "pr", "rc", "rr"))
    for lt, exp in ka["losses"].items():
        got, cr, rj = olosses.dpo_loss(pc, pr, rc, rr, ka["beta"], 0.0, lt)
        assert torch.allclose(got, torch.tensor(exp), atol=1e-5), lt
    # the values quoted in SURVEY.md §8c
    got, _, _ = olosses.dpo_loss(pc, pr, rc, rr, 0.1, 0.0, "sigmoid")
    assert torch.allclose(got, torch.tensor([0.6210, 0.7185]), atol=1e-4)


def test_moe_gating_golden_and_properties():
    g = U.load_golden("moe_gating.safetensors")
    l_aux, comb, disp, cnt = omoe.top2gating(g["logits"], 1.0, 0, g["noise"])
    assert torch.allclose(l_aux.reshape(1), g["top2.l_aux"]) and torch.equal(comb, g["top2.combine"])
    l1, c1, _, n1 = omoe.top1gating(g["logits"], 1.0, 0, None)
    assert torch.allclose(l1.reshape(1), g["top1.l_aux"]) and torch.equal(c1, g["top1.combine"])
    # properties (the pins we can give an unpinned third-party restatement)
    torch.manual_seed(0)
    for T, E, cf in [(64, 4, 1.5), (200, 8, 1.0), (50, 4, 0.25)]:
        logits = torch.randn(T, E)
        l_aux, comb, disp, cnt = omoe.top2gating(logits, cf, 0, omoe.gumbel_noise((T, E)))
        C = omoe.capacity(T, E, cf * 2, 0)
        assert comb.shape == (T, E, C)
        assert disp.sum(dim=(0, 2)).max() <= C                     # capacity never exceeded
        assert disp.sum(dim=0).max() <= 1                          # a slot holds at most one token
        w = comb.sum(dim=(1, 2))
        kept = disp.sum(dim=(1, 2))
        assert torch.all((kept == 0) | ((w - 1).abs() < 1e-5))     # surviving picks renormalise to 1, dropped -> 0
        gates = torch.softmax(logits, 1)
        me, ce = gates.mean(0), torch.nn.functional.one_hot(gates.argmax(1), E).float().mean(0)
        assert torch.allclose(l_aux, (me * ce).mean() * E * E)
        assert int(cnt.sum()) == T
    # no drops when capacity >= T: equals the dense top-2 mixture
    T, E, H = 30, 4, 16
    x = torch.randn(T, H)
    ffn = torch.nn.Linear(H, H, bias=False)
    m = omoe.OracleMoE(H, ffn, num_experts=E, k=2, capacity_factor=float(E), min_capacity=0)
    m.train()
    out, _, _ = m(x)
    logits = x @ m.deepspeed_moe.gate.wg.weight.t()
    gates = torch.softmax(logits, 1)
    top2 = gates.topk(2, dim=1).values
    ref = ffn(x) * 1.0                                             # identical experts: mixture weights sum to 1
    assert torch.allclose(out, ref, atol=1e-5) and torch.all(top2.sum(1) <= 1 + 1e-6)


def test_align_loss_edge_cases():
    V = 16
    lp = torch.log_softmax(torch.randn(1, 4, V), -1)
    p = torch.softmax(torch.randn(1, 4, V), -1)
    lp[0, 1, 3] = float("-inf")                                    # isinf mask: that term contributes 0
    labels = torch.tensor([[-100, 5, 7, -100]])
    got = olosses.compute_align_loss(lp, p, labels)
    x = (p * lp).masked_fill(torch.isinf(lp), 0).sum(-1)[0]
    assert torch.allclose(got, -(x[1] + x[2]) / 2)
    allm = olosses.compute_align_loss(lp, p, labels, distill_all_tokens=True)
    assert torch.allclose(allm, -x.sum() / 4)
    nan = olosses.compute_align_loss(lp, p, torch.full((1, 4), -100))   # reference divides by zero -> NaN
    assert torch.isnan(nan)


# ------------------------------------------------------------------------------------------ host logic of the product
def test_splice_plan_matches_oracle_splice():
    from llavamod.model.llava_arch import build_splice_plan
    torch.manual_seed(0)
    P, H = 4, 8
    emb = torch.nn.Embedding(100, H)
    ids = torch.randint(0, 90, (4, 11))
    ids[0, 2] = IMAGE_TOKEN_INDEX; ids[1, 0] = IMAGE_TOKEN_INDEX; ids[2, 10] = IMAGE_TOKEN_INDEX   # sample 3: no image
    labels = ids.clone(); labels[:, :3] = IGNORE_INDEX
    am = torch.ones(4, 11, dtype=torch.bool); am[1, 8:] = False; am[3, 5:] = False
    feats = torch.randn(4, P, H)                                    # the no-image sample still consumes a slot
    o_emb, o_am, o_lab = splice(emb, feats, ids, am, labels)
    plan = build_splice_plan(ids, am, labels, P, device="cpu")
    assert plan.S == o_emb.shape[1] and plan.n_images == 4
    assert torch.equal(plan.labels, o_lab) and torch.equal(plan.attention_mask, o_am)
    table, img = emb.weight.detach(), feats.reshape(-1, H)
    idx = plan.idx.long()
    rows = torch.where((idx >= 0)[:, None], table[idx.clamp_min(0)], torch.zeros(1, H))
    rows = torch.where((idx <= -2)[:, None], img[(-(idx + 2)).clamp_min(0)], rows)
    assert torch.allclose(rows.view(4, plan.S, H), o_emb)
    inv = plan.inv_idx.long()                                      # inverse map: image row -> output row
    for k in range(4 * P):
        if inv[k] >= 0:
            assert int(idx[inv[k]]) == -(k + 2)
    assert (inv[3 * P:] == -1).all()                               # the unused slot gets no gradient
    assert plan.seqlens is not None and plan.seqlens.tolist() == o_am.sum(1).tolist()


def test_loss_plan_rows():
    from llavamod.model.language_model.llava_qwen2 import build_loss_plan
    lab = np.array([[-100, -100, 5, 6, -100, 7], [-100, 3, -100, -100, -100, -100]])
    plan = build_loss_plan(lab, None, device="cpu", align_vocab=16)
    S = 6
    rows = plan.row_idx.tolist()
    assert rows == [1, 2, 3, 4, 5, S + 0, S + 1]                    # union of KD rows (t) and CE rows (t-1)
    assert plan.kd_w.tolist() == [0, 1, 1, 0, 1, 0, 1]
    assert plan.ce_w.tolist() == [1, 1, 0, 1, 0, 1, 0]
    assert plan.ce_label.tolist() == [5, 6, -1, 7, -1, 3, -1]
    assert plan.seg_off.tolist() == [0, 5, 7] and plan.seg_id.tolist() == [0] * 5 + [1] * 2
    inv = plan.inv_row_idx.tolist()
    assert all(inv[r] == i for i, r in enumerate(rows)) and sum(v >= 0 for v in inv) == len(rows)
    allp = build_loss_plan(lab, None, distill_all_tokens=True, device="cpu")
    assert allp.kd_w.sum() == 12


def test_moe_capacity_formula_and_api_errors():
    from llavamod.model.moe_layer import MoE
    from llavamod.model.language_model.qwen2_hip import Qwen2Config, Qwen2MLP
    cfg = Qwen2Config(hidden_size=64, intermediate_size=128)
    mlp = Qwen2MLP(cfg, "cpu")
    m = MoE(64, mlp, num_experts=4, k=2, capacity_factor=1.5, eval_capacity_factor=2.0, min_capacity=0)
    m.train()
    assert m.capacity(2048) == math.ceil(2048 / 4 * 1.5 * 2) == 1536
    m.eval()
    assert m.capacity(2048) == 2048
    assert len(m.deepspeed_moe.experts.deepspeed_experts) == 4
    assert any("wg" in n for n, _ in m.named_parameters())
    with pytest.raises(ValueError):
        MoE(64, mlp, num_experts=4, k=3)
    with pytest.raises(ValueError):
        MoE(64, mlp, num_experts=4, ep_size=3)


def test_warmup_cosine_and_flops_ledger():
    from llavamod.engine import warmup_cosine
    assert warmup_cosine(0, 1000, 2e-5) == pytest.approx(2e-5 / 30)
    assert warmup_cosine(29, 1000, 2e-5) == pytest.approx(2e-5)
    assert warmup_cosine(999, 1000, 2e-5) < 1e-9 + 2e-5 * 1e-4
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert abs(bench.mimic_tflop() - 52.98) < 0.02                   # BASELINE.md §2 ledger


# ---------------------------------------------------------------------------------------------- X1: two restatements
def _adversarial_logits(case, S, E, g):
    """Routings chosen to separate readings of DeepSpeed's gating that agree on friendly inputs."""
    if case == "random":
        return torch.randn(S, E, generator=g)
    if case == "ties":                       # exact ties between experts on most tokens: first maximum must win everywhere
        base = torch.randint(0, 2, (S, E), generator=g).float()
        return base
    if case == "all_to_one":                 # every token's first pick is expert 1: its queue overflows, second picks spread
        l = torch.randn(S, E, generator=g) * 0.1
        l[:, 1] += 5.0
        return l
    if case == "two_hot":                    # first picks all on 0, second picks all on 2: second queue starts BEHIND nothing,
        l = torch.full((S, E), -3.0)         # while expert 0's second-pick offset (sum of its first picks) is never used
        l[:, 0] = 2.0
        l[:, min(2, E - 1)] = 1.0
        return l + torch.randn(S, E, generator=g) * 1e-3
    if case == "second_behind_first":        # half the tokens pick (0 then 1), half (1 then 0): second picks queue behind ALL firsts
        l = torch.full((S, E), -4.0)
        l[0::2, 0], l[0::2, 1] = 2.0, 1.0
        l[1::2, 1], l[1::2, 0] = 2.0, 1.0
        return l
    raise KeyError(case)


@pytest.mark.parametrize("case", ["random", "ties", "all_to_one", "two_hot", "second_behind_first"])
@pytest.mark.parametrize("S,E,cf,min_cap", [(24, 4, 1.5, 0), (16, 4, 1.0, 0), (9, 4, 1.5, 0), (3, 8, 1.5, 0), (12, 4, 0.5, 7),
                                            (32, 8, 1.25, 4)])
@pytest.mark.parametrize("k,noisy", [(2, False), (2, True), (1, False), (1, True)])
def test_two_independent_restatements_of_deepspeed_moe_agree(case, S, E, cf, min_cap, k, noisy):
    """VERDICT r04 missing #1: X1 (`deepspeed.moe.layer.MoE`) has no reference-held vector, so the tensor-program restatement
    (`oracle/moe.py`: one-hot masks, cumsums, [S,E,C] einsums) is cross-checked against a per-token scalar restatement written
    separately (`oracle/moe_scalar.py`: python loops, per-expert queues, no [S,E,C] tensor): same capacity, same surviving picks
    in the same slots with the same weights, same l_aux / exp_counts, same layer output — on ties, exactly-full and overflowing
    queues, `min_capacity` above the computed capacity (12 tokens, factor 0.5), all tokens on one expert and S < E.
    (16 tokens x factor 1.0 x top-2 = capacity 8 = exactly the two-hot queues' length.)"""
    from oracle import moe_scalar
    g = torch.Generator().manual_seed(S * 131 + E * 17 + k)
    H = 8
    logits_target = _adversarial_logits(case, S, E, g)
    # a router and inputs that PRODUCE those logits exactly: x = [one-hot token code | 0], wg rows read the code
    x = torch.zeros(S, max(H, S))
    x[torch.arange(S), torch.arange(S)] = 1.0
    wg = torch.zeros(E, x.shape[1])
    wg[:, :S] = logits_target.t()
    H = x.shape[1]
    expert = torch.nn.Linear(H, H, bias=False)
    layer = omoe.OracleMoE(H, expert, num_experts=E, k=k, capacity_factor=cf, eval_capacity_factor=cf, min_capacity=min_cap)
    with torch.no_grad():
        layer.deepspeed_moe.gate.wg.weight.copy_(wg)
        for i, e in enumerate(layer.deepspeed_moe.experts.deepspeed_experts):
            e.weight.copy_(torch.randn(H, H, generator=g) * 0.3 + 0.1 * i)
    noise = rts = None
    if noisy and k == 2:
        noise = omoe.gumbel_noise((S, E), g)
    if noisy and k == 1:
        rts = torch.rand(S, E, generator=g)
    layer.train()
    layer.noise, layer.rts_noise = noise, rts
    with torch.no_grad():
        out_t, laux_t, counts_t = layer(x)
        experts = [lambda r, e=e: e(r) for e in layer.deepspeed_moe.experts.deepspeed_experts]
        out_s, laux_s, counts_s, picks, C = moe_scalar.forward(x, wg, experts, k, cf, min_cap, noise, rts)
        # the tensor program's decisions, read back from its combine tensor
        logits = x @ wg.t()
        if k == 2:
            _, combine, dispatch, _ = omoe.top2gating(logits, cf, min_cap, noise)
        else:
            _, combine, dispatch, _ = omoe.top1gating(logits, cf, min_cap, rts)
    assert combine.shape[2] == C == omoe.capacity(S, E, cf * (2 if k == 2 else 1), min_cap)
    for t in range(S):
        got = {(int(e), int(c)): float(combine[t, e, c]) for e, c in torch.nonzero(combine[t]).tolist()}
        want = {(e, c): w for e, c, w in picks[t] if w != 0.0}
        assert set(got) == set(want), (case, t, got, want)
        for key in got:
            assert abs(got[key] - want[key]) <= 1e-6, (case, t, key, got[key], want[key])
    assert counts_t.tolist() == counts_s
    assert abs(float(laux_t) - laux_s) <= 1e-6 * max(1.0, abs(laux_s))
    assert torch.allclose(out_t, out_s, atol=1e-5, rtol=1e-5)
