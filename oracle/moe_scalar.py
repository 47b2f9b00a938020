"""Second, independently written restatement of DeepSpeed-0.9.5 MoE routing — TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED like `oracle/moe.py`: `deepspeed==0.9.5` (reference requirements.txt:9) is neither vendored under
/root/reference nor installable here, so no reference-held vector exists for `deepspeed.moe.layer.MoE` (constructed at
llava_qwen2_moe.py:536-546, result consumed at :161-167).  What this file adds is insurance against a shared MISREADING
(VERDICT r04 missing #1): `oracle/moe.py` follows the published tensor program (one-hot masks, cumsums, `[S, E, C]`
dispatch / combine einsums); this file states the same published semantics as what happens to ONE TOKEN AT A TIME — plain
Python loops over tokens and per-expert queues, no one-hot tensor, no `[S, E, C]` tensor, no einsum — and
tests/test_oracle_cpu.py holds the two against each other on adversarial routings (ties, a capacity that is exactly full,
`min_capacity` above the computed capacity, every token on one expert, fewer tokens than experts, random token selection).

Semantics restated (sharded_moe.top1gating / top2gating / MOELayer.forward, layer.MoE.forward):
  * every token takes its arg-max expert (first maximum on ties); top-2 takes a second, different expert: the arg-max of
    logits + noise over the remaining experts;
  * an expert's queue is filled in TOKEN ORDER: all first picks, then (top-2) all second picks behind them; a pick whose
    position in the queue is >= capacity is dropped;  capacity = ceil(tokens / experts * factor) (factor doubled for
    top-2), raised to min_capacity;
  * top-1 with random token selection: of the tokens that picked an expert, the `capacity` with the largest uniform
    priority survive and are then numbered in token order;
  * combine weight: top-1 — the softmax gate of the pick, NOT renormalised; top-2 — the gates of the SURVIVING picks
    divided by their sum (clamped at fp32 eps);
  * l_aux = E * sum_e mean_t(gate[t, e]) * mean_t(first pick of t is e)   (top-2's mean(me*ce)*E*E is the same number);
  * the layer output of a token is sum over its surviving picks of weight * expert(token row); dropped tokens get zeros.
"""
import math

import torch


def _capacity(tokens, experts, factor, min_capacity):
    return max(int(math.ceil(tokens / experts * factor)), int(min_capacity))


def _argmax_first(values, skip=None):
    best, arg = None, -1
    for e, v in enumerate(values):
        if e == skip:
            continue
        if best is None or v > best:
            best, arg = v, e
    return arg


def route(logits, k, capacity_factor, min_capacity, noise=None, rts_noise=None):
    """logits [S, E] fp32.  Returns (picks, l_aux, exp_counts, capacity) with picks[t] = list of (expert, slot, weight) for the
    surviving picks of token t (weight a python float computed in fp32 steps)."""
    S, E = logits.shape
    gates = torch.softmax(logits.float(), dim=1)
    C = _capacity(S, E, capacity_factor * (2 if k == 2 else 1), min_capacity)
    first = [_argmax_first(gates[t].tolist()) for t in range(S)]
    second = None
    if k == 2:
        lw = logits.float() if noise is None else logits.float() + noise.float()
        second = [_argmax_first(lw[t].tolist(), skip=first[t]) for t in range(S)]
    exp_counts = [sum(1 for t in range(S) if first[t] == e) for e in range(E)]
    me = [float(gates[:, e].mean()) for e in range(E)]
    l_aux = E * sum(me[e] * (exp_counts[e] / S) for e in range(E))
    # queues in token order
    kept_first = [True] * S
    if k == 1 and rts_noise is not None:
        for e in range(E):
            mine = [t for t in range(S) if first[t] == e]
            mine.sort(key=lambda t: -float(rts_noise[t, e]))
            for t in mine[C:]:
                kept_first[t] = False
    fill = [0] * E
    slot1 = [-1] * S
    for t in range(S):
        if not kept_first[t]:
            continue
        e = first[t]
        if fill[e] < C:
            slot1[t] = fill[e]
        fill[e] += 1                              # a dropped first pick still takes its place in the count (cumsum semantics)
    slot2 = [-1] * S
    if k == 2:
        fill2 = list(exp_counts)                  # second picks queue behind ALL first picks of the expert, dropped ones included
        for t in range(S):
            e = second[t]
            if fill2[e] < C:
                slot2[t] = fill2[e]
            fill2[e] += 1
    picks = []
    eps = torch.finfo(torch.float32).eps
    for t in range(S):
        row = []
        if k == 1:
            if slot1[t] >= 0:
                row.append((first[t], slot1[t], float(gates[t, first[t]])))
        else:
            g1 = gates[t, first[t]] if slot1[t] >= 0 else torch.zeros(())
            g2 = gates[t, second[t]] if slot2[t] >= 0 else torch.zeros(())
            den = torch.clamp(g1 + g2, min=eps)
            if slot1[t] >= 0:
                row.append((first[t], slot1[t], float(g1 / den)))
            if slot2[t] >= 0:
                row.append((second[t], slot2[t], float(g2 / den)))
        picks.append(row)
    return picks, l_aux, exp_counts, C


def forward(x, wg_weight, experts, k, capacity_factor, min_capacity, noise=None, rts_noise=None):
    """x [S, H]; experts: list of callables on [n, H] rows.  Returns (out [S, H], l_aux, exp_counts, picks, capacity)."""
    logits = x.float() @ wg_weight.float().t()
    picks, l_aux, exp_counts, C = route(logits, k, capacity_factor, min_capacity, noise, rts_noise)
    out = torch.zeros_like(x)
    for t, row in enumerate(picks):
        for e, _slot, w in row:
            y = experts[e](x[t:t + 1])[0]
            out[t] = out[t] + torch.tensor(w, dtype=x.dtype) * y
    return out, l_aux, exp_counts, picks, C
