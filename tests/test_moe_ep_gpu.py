"""The expert-parallel (decomposed) MoE path — route / all-to-all / local experts / all-to-all / combine —
must reproduce the fused single-GPU MoE block exactly when the exchange is the identity (ep_size == 1),
for forward output, aux loss, input gradient and every weight gradient.  (The exchange itself and the
expert-aware gradient all-reduce are covered on CPU by tests/test_dp_gloo.py.)"""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("live", [True, False])
@pytest.mark.parametrize("E,k,T", [(4, 2, 300), (8, 2, 1000), (4, 1, 257)])
def test_decomposed_equals_fused(E, k, T, live):
    from llavamod.model.language_model.qwen2_hip import Qwen2Config, Qwen2MLP, init_normal_
    from llavamod.model.moe_layer import MoE
    torch.manual_seed(0)
    cfg = Qwen2Config(hidden_size=256, intermediate_size=512)
    mlp = init_normal_(Qwen2MLP(cfg, "cuda"), std=0.05, seed=1)
    a = MoE(256, mlp, num_experts=E, k=k, capacity_factor=1.0, min_capacity=0)
    with torch.no_grad():
        for i, e in enumerate(a.deepspeed_moe.experts.deepspeed_experts):     # make experts differ
            for p in e.parameters():
                p.mul_(1.0 + 0.1 * i)
        a.deepspeed_moe.gate.wg.weight.normal_(0, 0.5)
    b = copy.deepcopy(a)
    b.force_decomposed = True
    b.ep_live_rows = live              # packed live rows through the unequal-split exchange vs whole [E_local*C, H] slabs
    a.train(); b.train()
    a.deterministic = b.deterministic = True
    x = (torch.randn(T, 256, device="cuda") * 0.5).to(torch.bfloat16)
    dout = torch.randn(T, 256, device="cuda").to(torch.bfloat16)
    res = []
    for m in (a, b):
        xi = x.clone().requires_grad_(True)
        out, l_aux, counts = m(xi)
        (out.float() * dout.float()).sum().backward(retain_graph=True)
        (l_aux * 2.0).backward()
        grads = {n: p.main_grad.clone() for n, p in m.named_parameters() if getattr(p, "main_grad", None) is not None}
        res.append((out.detach(), l_aux.detach(), counts, xi.grad, grads))
    (o1, l1, c1, g1, w1), (o2, l2, c2, g2, w2) = res
    if live:
        pl = b.last_ep_plan
        assert sum(pl.in_splits) == int(b.last_state.slots_used.sum()) <= E * b.last_state.C
    assert torch.equal(o1, o2) and torch.equal(l1, l2) and torch.equal(c1, c2)
    assert torch.equal(g1, g2)
    assert set(w1) == set(w2) and len(w1) == 3 * E + 1
    for n in w1:
        d = (w1[n] - w2[n]).abs().max().item()
        assert d <= 1e-5 * max(1.0, w1[n].abs().max().item()), (n, d)


def test_full_slab_exchange_runs_without_a_host_sync():
    """the expert-parallel layer with the whole-slab exchange (`ep_live_rows = False`, DeepSpeed's own wire
    format) never reads anything back to the host — routing, counts exchange, local experts (live rows only, counts read ON THE
    DEVICE as grouped-GEMM `m_valid`), exchange back, combine, and the whole backward run under
    `torch.cuda.set_sync_debug_mode("error")`.  The live-row form (default) pays exactly one read-back per layer and direction of
    the forward (the unequal split sizes of `all_to_all_single` are host integers — the same point where DeepSpeed's gate syncs
    `exp_counts`); which form wins at 8 GPUs is a bytes-vs-sync trade the multi-GPU bench decides (`LMOD_EP_FULL_SLABS=1`)."""
    from llavamod.model.language_model.qwen2_hip import Qwen2Config, Qwen2MLP, init_normal_
    from llavamod.model.moe_layer import MoE
    torch.manual_seed(0)
    mlp = init_normal_(Qwen2MLP(Qwen2Config(hidden_size=256, intermediate_size=512), "cuda"), std=0.05, seed=1)
    m = MoE(256, mlp, num_experts=8, k=2, capacity_factor=1.5, min_capacity=0)
    m.force_decomposed, m.ep_live_rows = True, False
    m.train()
    x = (torch.randn(1000, 256, device="cuda") * 0.5).to(torch.bfloat16)
    dout = torch.randn(1000, 256, device="cuda").to(torch.bfloat16)
    xi = x.clone().requires_grad_(True)
    out, l_aux, _ = m(xi)                                        # warm-up: lazy allocations / first-use setup may sync
    (out.float() * dout.float()).sum().backward()
    torch.cuda.synchronize()
    xi = x.clone().requires_grad_(True)
    torch.cuda.set_sync_debug_mode("error")
    try:
        out, l_aux, _ = m(xi)
        ((out.float() * dout.float()).sum() + 2.0 * l_aux).backward()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out.float()).all()) and bool(torch.isfinite(xi.grad.float()).all())
    m.ep_live_rows = True                                        # the live-row form does read the counts back: it must trip the detector
    with pytest.raises(RuntimeError):
        torch.cuda.set_sync_debug_mode("error")
        try:
            m(x.clone().requires_grad_(True))
        finally:
            torch.cuda.set_sync_debug_mode("default")
