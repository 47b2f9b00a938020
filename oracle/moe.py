"""Oracle restatement of DeepSpeed-0.9.5 MoE (deepspeed.moe.layer.MoE, sharded_moe.{TopKGate,
top1gating, top2gating, MOELayer}, experts.Experts).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: DeepSpeed is a third-party dependency pinned at `deepspeed==0.9.5`
(reference requirements.txt:9), is not vendored under /root/reference and is not installed in
the authoring container.  The reference's only contact points are the constructor call
`MoE(hidden, expert=<dense Qwen2MLP>, num_experts, ep_size, k, capacity_factor,
eval_capacity_factor, min_capacity, use_residual)` (llava_qwen2_moe.py:536-546) and the 3-tuple
result `(out, l_aux, exp_counts)` (llava_qwen2_moe.py:161-167).  The algorithm below is the
published one (dense one-hot dispatch/combine einsums); randomness (Gumbel noise of the 2nd
choice, uniform noise of top-1 random-token-selection) is an explicit input so results are
reproducible.
"""
import copy
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def capacity(num_tokens, num_experts, capacity_factor, min_capacity):
    """sharded_moe._capacity: ceil(tokens / experts * factor), floored at min_capacity."""
    cap = int(math.ceil((num_tokens / num_experts) * capacity_factor))
    return max(cap, int(min_capacity))


def _one_hot_float(idx, n):
    return F.one_hot(idx, num_classes=n).float()


def top2gating(logits, capacity_factor, min_capacity, noise=None, forced=None):
    """sharded_moe.top2gating.  logits [S,E] fp32; noise [S,E] added to the logits for the 2nd pick.
    forced=(idx1, idx2): use these expert picks instead of the argmaxes (test hook: lets a parity test
    separate argmax tie-breaks under bf16 noise from everything else)."""
    S, E = logits.shape
    gates = F.softmax(logits, dim=1)
    C = capacity(S, E, capacity_factor * 2, min_capacity)
    idx1 = torch.argmax(gates, dim=1) if forced is None else forced[0].long()
    mask1 = F.one_hot(idx1, num_classes=E)
    lw = logits if noise is None else logits + noise
    idx2 = torch.argmax(lw.masked_fill(mask1.bool(), float("-inf")), dim=1) if forced is None else forced[1].long()
    mask2 = F.one_hot(idx2, num_classes=E)
    loc1 = torch.cumsum(mask1, dim=0) - 1
    loc2 = torch.cumsum(mask2, dim=0) - 1
    loc2 = loc2 + torch.sum(mask1, dim=0, keepdim=True)
    exp_counts = torch.sum(mask1, dim=0).detach()
    me = torch.mean(gates, dim=0)
    ce = torch.mean(mask1.float(), dim=0)
    l_aux = torch.mean(me * ce) * E * E
    mask1 = mask1 * torch.lt(loc1, C)
    mask2 = mask2 * torch.lt(loc2, C)
    loc1_s = torch.sum(loc1 * mask1, dim=1)
    loc2_s = torch.sum(loc2 * mask2, dim=1)
    m1f, m2f = mask1.float(), mask2.float()
    g1 = torch.einsum("se,se->s", gates, m1f)
    g2 = torch.einsum("se,se->s", gates, m2f)
    den = torch.clamp(g1 + g2, min=torch.finfo(gates.dtype).eps)
    g1, g2 = g1 / den, g2 / den
    gates1 = torch.einsum("s,se->se", g1, m1f)
    gates2 = torch.einsum("s,se->se", g2, m2f)
    combine = torch.einsum("se,sc->sec", gates1, _one_hot_float(loc1_s, C)) + \
        torch.einsum("se,sc->sec", gates2, _one_hot_float(loc2_s, C))
    dispatch = combine.bool()
    return l_aux, combine, dispatch, exp_counts


def top1gating(logits, capacity_factor, min_capacity, rts_noise=None, forced=None):
    """sharded_moe.top1gating with noisy_gate_policy=None, drop_tokens=True.  rts_noise [S,E] uniform
    noise for random token selection (use_rts=True); None -> token order (use_rts=False).
    forced: idx1 test hook (bf16-twin noise floors reuse the fp32 run's picks), not part of DeepSpeed."""
    S, E = logits.shape
    gates = F.softmax(logits, dim=1)
    C = capacity(S, E, capacity_factor, min_capacity)
    idx1 = torch.argmax(gates, dim=1) if forced is None else forced
    mask1 = F.one_hot(idx1, num_classes=E)
    exp_counts = torch.sum(mask1, dim=0).detach()
    me = torch.mean(gates, dim=0)
    ce = torch.mean(mask1.float(), dim=0)
    l_aux = torch.sum(me * ce) * E
    if rts_noise is not None:
        mask1_rand = mask1 * rts_noise
        top_idx = torch.topk(mask1_rand, k=min(C, S), dim=0)[1]
        mask1 = mask1 * torch.zeros_like(mask1).scatter_(0, top_idx, 1)
        loc1 = torch.cumsum(mask1, dim=0) - 1
    else:
        loc1 = torch.cumsum(mask1, dim=0) - 1
        mask1 = mask1 * torch.lt(loc1, C)
    loc1_s = torch.sum(loc1 * mask1, dim=1)
    gates = gates * mask1.float()
    combine = torch.einsum("se,sc->sec", gates, _one_hot_float(loc1_s, C))
    dispatch = combine.bool()
    return l_aux, combine, dispatch, exp_counts


class OracleMoE(nn.Module):
    """Same constructor and return signature as deepspeed.moe.layer.MoE (ep_size=1 semantics; expert
    parallelism only changes where experts live, not the arithmetic).  `deepspeed_moe.gate.wg` and
    `deepspeed_moe.experts.deepspeed_experts[i]` exist because the reference reads them
    (llava_qwen2_moe.py:547; utils.py:41)."""

    def __init__(self, hidden_size, expert, num_experts=1, ep_size=1, k=1, capacity_factor=1.0,
                 eval_capacity_factor=1.0, min_capacity=4, use_residual=False):
        super().__init__()
        assert k in (1, 2), "DeepSpeed 0.9.5 supports top-1 and top-2 gating only"
        self.num_experts, self.k = num_experts, k
        self.capacity_factor, self.eval_capacity_factor, self.min_capacity = capacity_factor, eval_capacity_factor, min_capacity
        self.use_residual = use_residual
        self.deepspeed_moe = nn.Module()
        self.deepspeed_moe.gate = nn.Module()
        self.deepspeed_moe.gate.wg = nn.Linear(hidden_size, num_experts, bias=False).float()
        self.deepspeed_moe.experts = nn.Module()
        self.deepspeed_moe.experts.deepspeed_experts = nn.ModuleList([copy.deepcopy(expert) for _ in range(num_experts)])
        if use_residual:
            self.mlp = copy.deepcopy(expert)
            self.coefficient = nn.Linear(hidden_size, 2)
        self.noise = None        # [S,E] gumbel noise for the next top-2 forward (explicit input)
        self.forced = None       # (idx1, idx2) test hook, see top2gating
        self.last_picks = None
        self.rts_noise = None    # [S,E] uniform noise for the next top-1 forward

    def forward(self, hidden_states, used_token=None):
        d = hidden_states.shape[-1]
        x = hidden_states.reshape(-1, d)
        wg = self.deepspeed_moe.gate.wg
        logits = F.linear(x.float(), wg.weight.float())
        cf = self.capacity_factor if self.training else self.eval_capacity_factor
        if self.k == 2:
            forced = self.forced.pop(0) if isinstance(self.forced, list) else self.forced     # list: one entry per call
            l_aux, combine, dispatch, exp_counts = top2gating(logits, cf, self.min_capacity, self.noise, forced)
            g = F.softmax(logits.detach(), dim=1)
            i1 = torch.argmax(g, dim=1)
            lw = logits.detach() if self.noise is None else logits.detach() + self.noise
            i2 = torch.argmax(lw.masked_fill(F.one_hot(i1, g.shape[1]).bool(), float("-inf")), dim=1)
            self.last_picks = (i1, i2, g, lw)
        else:
            forced = self.forced.pop(0) if isinstance(self.forced, list) else self.forced
            l_aux, combine, dispatch, exp_counts = top1gating(logits, cf, self.min_capacity, self.rts_noise,
                                                              None if forced is None else forced[0])
            self.last_picks = (torch.argmax(logits.detach(), dim=1), None, F.softmax(logits.detach(), dim=1), logits.detach())
        dispatched = torch.einsum("sec,sm->ecm", dispatch.type_as(x), x)
        outs = [e(dispatched[i]) for i, e in enumerate(self.deepspeed_moe.experts.deepspeed_experts)]
        expert_out = torch.stack(outs, dim=0)
        out = torch.einsum("sec,ecm->sm", combine.type_as(x), expert_out).reshape(hidden_states.shape)
        if self.use_residual:
            res = self.mlp(hidden_states)
            coef = F.softmax(self.coefficient(hidden_states), dim=-1)
            out = out * coef[..., 0:1] + res * coef[..., 1:]
        return out, l_aux, exp_counts


def gumbel_noise(shape, generator=None):
    """Gumbel(0,1) sample, what sharded_moe.gumbel_rsample draws."""
    u = torch.rand(shape, generator=generator).clamp_(1e-20, 1.0)
    return -torch.log(-torch.log(u))
