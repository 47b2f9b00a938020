"""World size 2 on a ONE-GPU box: both ranks compute on cuda:0 with the REAL HIP kernels and exchange through gloo (gloo moves
CUDA tensors for every collective the step uses, in place included — tools/probe/gloo_cuda_probe.py).  What a 1-GPU box
cannot show is RCCL itself; what it can show is everything around it at world > 1 on hardware: ZeRO-2 shards through the HIP AdamW /
sum-of-squares / clip kernels, and the expert-parallel MoE layer — route, live-row exchange, local experts on
both ranks' rows, exchange back, combine, and the backward through all of it (config 5)."""
import os

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def test_zero2_sharded_hip_optimizer_two_ranks_share_the_gpu():
    """tests/test_dp_gloo.py's ZeRO-2 scenario (sharded == unsharded over 2 clipped steps, bias spans, bf16 exchange, accumulation
    window, exact resume of the sharded state) with the product kernels instead of the CPU stand-ins."""
    from test_dp_gloo import _spawn, _zero2_worker
    _spawn(_zero2_worker, "cuda", timeout=400)


def _ep_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), LMOD_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from llavamod.engine import init_distributed
    from llavamod.model.language_model.qwen2_hip import Qwen2Config, Qwen2MLP, init_normal_
    from llavamod.model.moe_layer import MoE
    init_distributed()
    dev, H, I, T = "cuda", 256, 512, 600
    cfg = Qwen2Config(hidden_size=H, intermediate_size=I)
    E = 4

    def make(ep):
        mlp = init_normal_(Qwen2MLP(cfg, dev), std=0.05, seed=1)
        m = MoE(H, mlp, num_experts=E, ep_size=ep, k=2, capacity_factor=1.5, min_capacity=0)
        g = torch.Generator(device="cpu"); g.manual_seed(7)
        with torch.no_grad():
            m.deepspeed_moe.gate.wg.weight.copy_(torch.randn(E, H, generator=g) * 0.5)
            n_local = E // ep
            for i, e in enumerate(m.deepspeed_moe.experts.deepspeed_experts):     # global expert id: experts differ
                gid = (rank * n_local + i) if ep > 1 else i
                for p in e.parameters():
                    p.mul_(1.0 + 0.1 * gid)
        m.train(); m.deterministic = True
        return m

    def tokens(r):
        g = torch.Generator(device="cpu"); g.manual_seed(100 + r)
        x = (torch.randn(T, H, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        d = torch.randn(T, H, generator=g).to(torch.bfloat16).to(dev)
        return x, d

    def run(m, x, d):
        xi = x.clone().requires_grad_(True)
        out, l_aux, counts = m(xi)
        (out.float() * d.float()).sum().backward(retain_graph=True)
        (l_aux * 2.0).backward()
        return out.detach(), l_aux.detach(), counts, xi.grad

    def grads(m):
        return {n: p.main_grad.clone() for n, p in m.named_parameters() if getattr(p, "main_grad", None) is not None}

    # E 4: two local experts per rank (capacity slabs on the receiving side); E 2: ONE local expert per rank — config 5's shape —
    # whose live-row form runs the dense fused block on the packed rows (no slabs, no row masks)
    for E, live in ((4, True), (4, False), (2, True), (2, False)):
        # reference: ALL experts in one process (fused block).  DeepSpeed routes every rank's tokens with that rank's own capacity, so
        # the reference for rank r's outputs is the full layer on rank r's tokens; an expert's weight gradient sums BOTH ranks' tokens.
        ref = make(1)
        x_me, d_me = tokens(rank)
        o_ref, l_ref, c_ref, gx_ref = run(ref, x_me, d_me)
        gate_ref = grads(ref)["deepspeed_moe.gate.wg.weight"].clone()
        x_ot, d_ot = tokens(1 - rank)
        run(ref, x_ot, d_ot)                                   # main_grad accumulates: expert gradients over both token sets
        g_ref = grads(ref)

        ep = make(2)
        ep.ep_live_rows = live
        o, l, c, gx = run(ep, x_me, d_me)
        g_ep = grads(ep)
        if live:
            pl = ep.last_ep_plan
            assert sum(pl.in_splits) == int(ep.last_state.slots_used.sum()) and len(pl.in_splits) == 2
        assert torch.equal(c, c_ref) and torch.equal(l, l_ref), "routing differs from the single-process layer"
        scale = o_ref.float().abs().max().item()
        assert (o.float() - o_ref.float()).abs().max().item() <= 2 ** -7 * scale, "expert-parallel output"
        assert (gx.float() - gx_ref.float()).abs().max().item() <= 2 ** -6 * gx_ref.float().abs().max().item(), "input gradient"
        assert len(g_ep) == 3 * (E // 2) + 1
        nl = E // 2
        for n, gv in g_ep.items():
            if "gate.wg" in n:                                 # replicated: this rank's tokens only (data-parallel all-reduce comes later)
                rv = gate_ref
            else:                                              # local expert i is global expert rank * 2 + i
                i = int(n.split("deepspeed_experts.")[1].split(".")[0])
                rv = g_ref[n.replace(f"deepspeed_experts.{i}.", f"deepspeed_experts.{rank * nl + i}.")]
            err = (gv - rv).abs().max().item()
            assert err <= 2e-3 * max(1e-6, rv.abs().max().item()), (live, n, err, rv.abs().max().item())
        if E == 2 and live:
            # the same layer with its round trip PIPELINED over 2 and 3 row chunks (exchange of chunk c+1 under the expert GEMMs of chunk
            # c, ops.chunked_expert_exchange): the expert block is row-wise, so outputs, aux loss and input gradients are bit-identical
            # to the unchunked exchange; the expert weight gradients accumulate chunk by chunk (fp32: equal to rounding of the sums)
            for nchunk in (2, 3):
                epc = make(2)
                epc.ep_live_rows, epc.ep_chunks = True, nchunk
                oc, lc, cc, gxc = run(epc, x_me, d_me)
                assert torch.equal(oc, o) and torch.equal(lc, l) and torch.equal(cc, c) and torch.equal(gxc, gx), nchunk
                for n, gv in grads(epc).items():
                    rv = g_ep[n]
                    assert (gv - rv).abs().max().item() <= 1e-4 * max(1e-6, rv.abs().max().item()), (nchunk, n)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


def test_expert_parallel_layer_two_ranks_share_the_gpu():
    """ep_size 2, 4 experts (2 per rank) and 2 experts (1 per rank: the packed single-expert form), top-2, capacity factor 1.5: forward output, aux loss, expert counts, input gradient, router
    gradient and the local experts' weight gradients against the single-process 4-expert layer — live-row exchange and full slabs."""
    from test_dp_gloo import _spawn
    _spawn(_ep_worker, timeout=400)
