"""Multi-process data-parallel path on CPU: 2 ranks over gloo (the GPU path uses the same code with the
"nccl" = RCCL backend).  Covers rank/world discovery from the launcher env, the flat gradient buffer
layout being identical on every rank, the bucketed SUM all-reduce, and rank-distinct data shards."""
import os
import socket
import sys

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "llava-mod_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), LMOD_DIST_BACKEND="gloo")      # CPU test even on a box that has a GPU
    from llavamod.engine import DataParallel, GradBuffer, init_distributed
    from llavamod.model.language_model.qwen2_hip import Qwen2Config, Qwen2DecoderLayer
    from llavamod.model.moe_layer import MoE
    r, l, w = init_distributed()
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    torch.manual_seed(0)
    cfg = Qwen2Config(vocab_size=64, hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=1)
    layer = Qwen2DecoderLayer(cfg, "cpu")
    layer.mlp = MoE(64, layer.mlp, num_experts=4, k=2, capacity_factor=1.5, min_capacity=0)
    for n, p in layer.named_parameters():
        p.requires_grad = ("mlp" in n)
    gb = GradBuffer(layer)
    # identical layout on every rank: spans are ordered by module traversal
    sizes = torch.tensor([n for _, _, n in gb.spans])
    gathered = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(gathered, sizes)
    assert all(torch.equal(g, sizes) for g in gathered)
    # dense (replicated) spans first — here just the router — then ONE contiguous span per stacked expert weight
    assert gb.numel == 4 * (2 * 128 * 64 + 64 * 128) + 4 * 64 and gb.n_dense == 4 * 64
    ex0 = layer.mlp.deepspeed_moe.experts.deepspeed_experts[0]
    assert ex0.gate_proj.weight.main_grad.data_ptr() == gb.flat[gb.n_dense:].data_ptr()
    gb.flat.copy_(torch.arange(gb.numel, dtype=torch.float32) * (rank + 1))
    dp = DataParallel(bucket_bytes=4096)                      # force several buckets
    assert dp.enabled and dp.world == world
    dp.all_reduce(gb.flat)
    expect = torch.arange(gb.numel, dtype=torch.float32) * sum(range(1, world + 1))
    assert torch.equal(gb.flat, expect)
    assert torch.equal(ex0.up_proj.weight.main_grad.reshape(-1), expect[gb.n_dense + 128 * 64:gb.n_dense + 2 * 128 * 64])
    # expert parallelism over the whole world: router grads are summed, expert grads stay rank-local
    gb.flat.copy_(torch.arange(gb.numel, dtype=torch.float32) * (rank + 1))
    dp.all_reduce(gb.flat, n_dense=gb.n_dense, ep_size=world)
    mine = torch.arange(gb.numel, dtype=torch.float32) * (rank + 1)
    assert torch.equal(gb.flat[:gb.n_dense], expect[:gb.n_dense]) and torch.equal(gb.flat[gb.n_dense:], mine[gb.n_dense:])
    # overlapped form: a weight's hook fires when its last wgrad contribution is in; finish() sweeps the rest
    gb.flat.copy_(torch.arange(gb.numel, dtype=torch.float32) * (rank + 1))
    dpo = DataParallel(bucket_bytes=4096).attach(gb)
    gu = layer.mlp._gu
    gu.note_use(); gu.note_use()
    gu.grad_done()
    assert not dpo._handles                                    # one contribution still pending
    gu.grad_done()
    assert dpo._handles and (id(gu), "w") in dpo._done
    dpo.finish()
    assert torch.equal(gb.flat, expect) and not dpo._handles and not dpo._done
    # all-to-all of capacity slabs (the EP exchange) and its autograd transpose
    from llavamod import ops
    from llavamod.engine import expert_parallel_group
    grp = expert_parallel_group(world)
    E_local, C, H = 2, 3, 4
    send = (torch.arange(world * E_local * C * H, dtype=torch.float32).view(world, E_local * C, H) + 1000 * rank)
    send.requires_grad_(True)
    recv = ops.AllToAll.apply(send, grp)
    for src in range(world):        # chunk `src` of what I hold came from rank src's chunk `rank`
        exp = torch.arange(world * E_local * C * H, dtype=torch.float32).view(world, E_local * C, H)[rank] + 1000 * src
        assert torch.equal(recv[src].detach(), exp)
    (recv * (rank + 1)).sum().backward()                       # grad chunk d returns from rank d scaled by (d+1)
    for dst in range(world):
        assert torch.all(send.grad[dst] == dst + 1)
    # live-row exchange: only the live rows of each capacity slab travel, through an unequal-split
    # all_to_all_single sized by the exchanged `slots_used`; it must deliver exactly what the full-slab exchange delivers
    # on every live row, forward and backward (gather kernel replaced by its torch meaning on CPU)
    import llavamod.kernels as K

    def gather_rows(src_a, src_b, idx, Hh):
        out = torch.zeros((idx.numel(), Hh), dtype=src_a.dtype)
        m = idx >= 0
        out[m] = src_a[idx[m].long()]
        return out
    K.gather_rows = gather_rows
    El, C, H = 2, 5, 4

    def live_case(starve):
        g = torch.Generator().manual_seed(100 + rank)
        used = torch.randint(0, C + 1, (world, El), generator=g, dtype=torch.int32)  # my live slots per (dest rank, expert)
        used[rank, 0] = 0                                                            # an empty slab
        if starve is not None:      # nobody routes anything to rank `starve`: its packed receive buffer has ZERO rows (ADVICE r03:
            used[starve] = 0        # a collapsed router, ep == num_experts, or a tiny eval batch) and it sends its rows as usual
        slab = torch.randn(world * El * C, H, generator=g)
        for r in range(world):
            for le in range(El):
                base = (r * El + le) * C
                slab[base + int(used[r, le]):base + C] = 0                               # empty slots are zero rows
        rr = torch.empty_like(used)
        dist.all_to_all_single(rr, used.contiguous(), group=grp)
        full_in = slab.clone().requires_grad_(True)
        full = ops.AllToAll.apply(full_in.view(world, El * C, H), grp).reshape(world * El * C, H)
        pl = ops.ep_live_row_plan(used.numpy(), rr.numpy(), C, "cpu")
        assert pl.in_splits == used.sum(1).tolist() and pl.out_splits == rr.sum(1).tolist()
        live_in = slab.clone().requires_grad_(True)
        packed = ops.RowGather.apply(live_in, pl.send_idx, pl.send_inv)
        assert packed.shape[0] == int(used.sum())
        recv_p = ops.AllToAllRows.apply(packed, pl.in_splits, pl.out_splits, grp)
        recv = ops.RowGather.apply(recv_p, pl.recv_slab, pl.recv_inv)
        assert torch.equal(recv.detach(), full.detach())                                 # dead slots: zero rows in both
        wgt = torch.randn(world * El * C, H, generator=torch.Generator().manual_seed(7 + rank))
        (full * wgt).sum().backward()
        (recv * wgt).sum().backward()
        livemask = (pl.send_inv >= 0)
        assert torch.equal(live_in.grad[livemask], full_in.grad[livemask]) and float(live_in.grad[~livemask].abs().max()) == 0.0
        # and the way back: expert outputs of the live rows return to their slots
        y = recv.detach() * 2.0 + 1.0
        y_p = ops.RowGather.apply(y, pl.recv_inv, pl.recv_slab)
        back_p = ops.AllToAllRows.apply(y_p, pl.out_splits, pl.in_splits, grp)
        back = ops.RowGather.apply(back_p, pl.send_inv, pl.send_idx)
        back_full = ops.AllToAll.apply(y.view(world, El * C, H), grp).reshape(world * El * C, H)
        assert torch.equal(back[livemask], back_full[livemask]) and torch.equal(back[livemask], slab[livemask] * 2.0 + 1.0)
        return int(rr.sum())
    live_case(None)
    got = live_case(1)
    assert (got == 0) == (rank == 1)
    # rank-distinct synthetic shards (bench.py seeds batches with the rank)
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ids = bench.synthetic_batch(1, 1000 * rank, text_len=32, response=8)["input_ids"]
    allids = [torch.zeros_like(ids) for _ in range(world)]
    dist.all_gather(allids, ids)
    assert not torch.equal(allids[0], allids[1])
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


def _cpu_kernels():
    """Torch stand-ins for the four HIP kernels the optimizer / exchange call, so that the ENGINE logic (span ownership,
    collectives, clipping, master copies) runs on CPU under gloo.  Test-only: the product path has no such fallback."""
    import llavamod.kernels as K

    def adamw_step(master, param, grad, m, v, lr, b1, b2, eps, wd, step, grad_scale=1.0, zero_grad=False, dev_scale=None):
        gs = grad_scale * (float(dev_scale[0]) if dev_scale is not None else 1.0)
        g = grad * gs
        if zero_grad:
            grad.zero_()
        master.mul_(1.0 - lr * wd)
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        master.addcdiv_(m, v.sqrt() / (bc2 ** 0.5) + eps, value=-lr / bc1)
        param.copy_(master.to(param.dtype))

    def sumsq(x, out, partials, accumulate=False):
        t = (x.double() ** 2).sum().float()
        out[0] = out[0] + t if accumulate else t
        return out

    def clip_coef(ss, norm_scale, max_norm, coef, norm_out=None):
        nrm = ss[0].sqrt() * norm_scale
        coef[0] = torch.clamp(max_norm / (nrm + 1e-6), max=1.0)
        if norm_out is not None:
            norm_out[0] = nrm
        return coef

    def cast_f32_bf16(src, dst):
        dst.copy_(src.to(dst.dtype))
        return dst

    K.adamw_step, K.sumsq, K.clip_coef, K.cast_f32_bf16 = adamw_step, sumsq, clip_coef, cast_f32_bf16


def _zero2_worker(rank, world, port, q, device="cpu"):
    """device "cpu": the four HIP kernels are replaced by torch stand-ins (this file's tests).  device "cuda": the REAL kernels, both
    ranks on the one GPU of the box, collectives through gloo (tests/test_two_ranks_one_gpu.py)."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), LMOD_DIST_BACKEND="gloo")      # CPU test even on a box that has a GPU
    from llavamod.engine import DataParallel, GradBuffer, HipAdamW, init_distributed
    from llavamod.model.language_model.qwen2_hip import Qwen2Config, Qwen2DecoderLayer, init_normal_
    from llavamod.model.moe_layer import MoE
    if device == "cpu":
        _cpu_kernels()
    init_distributed()

    def build():
        cfg = Qwen2Config(vocab_size=64, hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=1)
        layer = Qwen2DecoderLayer(cfg, device)
        layer.mlp = MoE(64, layer.mlp, num_experts=4, k=2, capacity_factor=1.5, min_capacity=0)
        init_normal_(layer, 0.02, 0)
        for n, p in layer.named_parameters():                 # experts + router + a BIASED fused weight (q/k/v)
            p.requires_grad = ("mlp" in n) or any(t in n for t in ("q_proj", "k_proj", "v_proj"))
        return layer

    def fake_grads(gb, step):                                  # rank-dependent, deterministic, O(1) magnitude
        i = torch.arange(gb.numel, dtype=torch.float32, device=gb.flat.device)
        gb.flat.copy_(torch.sin(i * 0.37 + step) * (0.5 + rank) + 0.1 * rank)

    def run(zero2, grad_dtype=torch.float32, clip=1.0, use_hooks=True):
        layer = build()
        gb = GradBuffer(layer)
        dp = DataParallel(bucket_bytes=4096, zero2=zero2, grad_dtype=grad_dtype, min_shard_numel=1).attach(gb)
        opt = HipAdamW(gb, lr=1e-2, weight_decay=0.01, dp=dp, max_grad_norm=clip)
        norms = []
        for step in range(2):
            gb.zero()
            fake_grads(gb, step)
            if use_hooks:                                      # the attention weight's hook fires during "backward": its
                qkv = layer.self_attn._qkv                     # BIAS span must be exchanged with it (it was not)
                qkv.note_use()
                qkv.grad_done()
                assert (id(qkv), "w") in dp._done and (id(qkv), "b") in dp._done
            dp.finish()
            opt.step(grad_scale=1.0 / world, clear_grads=True)
            norms.append(float(opt.grad_norm))
        return layer, gb, opt, norms

    ref_layer, ref_gb, ref_opt, ref_norms = run(False)
    # the all-reduced gradient of EVERY span (bias spans included) is the rank sum: redo one exchange and look at it
    fake_grads(ref_gb, 7)
    mine = ref_gb.flat.clone()
    d2 = DataParallel(bucket_bytes=4096).attach(ref_gb)
    q2 = ref_layer.self_attn._qkv
    q2.note_use(); q2.grad_done()
    d2.finish()
    other = torch.sin(torch.arange(ref_gb.numel, dtype=torch.float32, device=ref_gb.flat.device) * 0.37 + 7) * (0.5 + (1 - rank)) + 0.1 * (1 - rank)
    assert torch.allclose(ref_gb.flat, mine + other, atol=1e-6), "a span was not summed over the ranks"

    z_layer, z_gb, z_opt, z_norms = run(True)
    assert z_opt.n_state < 0.6 * ref_opt.n_state, (z_opt.n_state, ref_opt.n_state)       # ~1/2 + replicated crumbs
    for a, b in zip(ref_norms, z_norms):
        assert abs(a - b) <= 1e-5 * abs(a) and a > 1.0, (a, b)                          # clipping was active
    for (n, pa), (_, pb) in zip(ref_layer.named_parameters(), z_layer.named_parameters()):
        assert torch.allclose(pa.float(), pb.float(), rtol=0, atol=1e-6 if pa.dtype == torch.float32 else 0), n
    # parameters are views of the flat bf16 buffer and every rank holds the same weights after the all-gather
    w = z_layer.mlp._gu.w
    assert w.data_ptr() >= z_gb.pflat.data_ptr() and w.data_ptr() < z_gb.pflat.data_ptr() + 2 * z_gb.numel
    mine = z_gb.pflat.clone()
    both = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    assert torch.equal(both[0], both[1])
    assert float(z_gb.flat.abs().max()) == 0.0                                           # gradients cleared for the next step
    # bf16 gradient exchange: same result up to bf16 rounding of the gradients
    b_layer, _, _, b_norms = run(True, grad_dtype=torch.bfloat16)
    assert abs(b_norms[0] - ref_norms[0]) <= 1e-2 * ref_norms[0]
    for (n, pa), (_, pb) in zip(ref_layer.named_parameters(), b_layer.named_parameters()):
        # Adam's step is ~lr * sign-like: where the two ranks' gradients nearly cancel, bf16 rounding may flip the update
        far = ((pa.float() - pb.float()).abs() > 2e-3).float().mean().item()
        assert far < 0.02, (n, far)
    # gradient accumulation: nothing is exchanged while the window is open
    layer = build()
    gb = GradBuffer(layer)
    dp = DataParallel(zero2=True, min_shard_numel=1).attach(gb)
    dp.armed = False
    fake_grads(gb, 0)
    before = gb.flat.clone()
    qkv = layer.self_attn._qkv
    qkv.note_use(); qkv.grad_done()
    dp.finish()
    assert torch.equal(gb.flat, before) and not dp._handles
    # an optimizer built BEFORE dp.attach(gb) would keep full-span state and update it from reduce-scattered
    # gradients — it must refuse
    layer = build()
    gb = GradBuffer(layer)
    dp_late = DataParallel(zero2=True, min_shard_numel=1)
    try:
        HipAdamW(gb, dp=dp_late)
        raise AssertionError("HipAdamW accepted a DataParallel that was not attached to its GradBuffer")
    except RuntimeError:
        pass
    # exact resume of the SHARDED optimizer: per-rank state files, weights from the (replicated) bf16 buffer
    import tempfile
    from llavamod.checkpoint import load_optimizer, save_optimizer
    tmp = os.environ["LMOD_TEST_TMP"]
    a_layer, a_gb, a_opt, _ = run(True)
    weights = {n: p.detach().clone() for n, p in a_layer.named_parameters()}
    path = save_optimizer(a_opt, tmp)
    assert path.endswith(f"optimizer_rank{rank}_of{world}.pt")
    a_layer.mlp._calls = 5                                       # pretend the gate drew noise 5 times, then save again
    save_optimizer(a_opt, tmp)
    a_gb.zero(); fake_grads(a_gb, 2)
    a_opt.dp.finish(); a_opt.step(grad_scale=1.0 / world, clear_grads=True)
    b_layer = build()
    b_layer.mlp._layer_id = a_layer.mlp._layer_id                # a resumed PROCESS rebuilds the same layers in the same order
    with torch.no_grad():
        for n, p in b_layer.named_parameters():
            p.copy_(weights[n])
    b_gb = GradBuffer(b_layer)
    b_dp = DataParallel(bucket_bytes=4096, zero2=True, min_shard_numel=1).attach(b_gb)
    b_opt = HipAdamW(b_gb, lr=1e-2, weight_decay=0.01, dp=b_dp, max_grad_norm=1.0)
    load_optimizer(b_opt, tmp)
    assert b_opt.step_count == 2 and b_layer.mlp._calls == 5
    b_gb.zero(); fake_grads(b_gb, 2)
    b_dp.finish(); b_opt.step(grad_scale=1.0 / world, clear_grads=True)
    for (n, pa), (_, pb) in zip(a_layer.named_parameters(), b_layer.named_parameters()):
        assert torch.equal(pa, pb), n
    assert torch.equal(a_opt.m, b_opt.m) and torch.equal(a_opt.v, b_opt.v) and torch.equal(a_opt.master, b_opt.master)
    # a different shard layout cannot be resumed exactly: refuse instead of mis-assigning moments
    c_layer = build()
    c_gb = GradBuffer(c_layer)
    c_opt = HipAdamW(c_gb, dp=DataParallel(zero2=False).attach(c_gb))
    try:
        c_opt.load_state_dict(torch.load(path, weights_only=True))
        raise AssertionError("optimizer state of another shard layout was accepted")
    except ValueError:
        pass
    # without ZeRO-2 the layout is the same on every rank, so world / rank / zero2 and the MoE layer set are
    # compared too; and restoring masters BEFORE loading model weights is refused at the next step instead of silently
    # re-deriving the masters from bf16 weights
    st = c_opt.state_dict()
    for key, bad in (("rank", (rank + 1) % world), ("world", world + 1), ("zero2", True)):
        try:
            c_opt.load_state_dict({**st, key: bad})
            raise AssertionError(f"optimizer state with another {key} was accepted")
        except ValueError:
            pass
    try:
        c_opt.load_state_dict({**st, "moe_noise": st["moe_noise"] + [{"layer_id": 99, "calls": 0, "eval_calls": 0}]})
        raise AssertionError("optimizer state of another MoE layer set was accepted")
    except ValueError:
        pass
    if st["moe_noise"]:
        try:
            c_opt.load_state_dict({**st, "moe_noise": [{**st["moe_noise"][0], "layer_id": st["moe_noise"][0]["layer_id"] + 1}] + st["moe_noise"][1:]})
            raise AssertionError("gating-noise counters of another layer id were accepted")
        except ValueError:
            pass
    c_opt.load_state_dict(st)
    c_layer._weights_epoch = getattr(c_layer, "_weights_epoch", 0) + 1        # what checkpoint.load_checkpoint does
    try:
        c_opt.step()
        raise AssertionError("weights loaded after the optimizer state: the step went ahead from re-derived masters")
    except RuntimeError:
        pass
    c_opt.resync_master()                                                      # the explicit way out
    c_gb.zero()
    c_opt.step()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


def _guarded(target, rank, world, port, q, *extra):
    """Child entry point: run `target` and, if it raises, put THIS rank's own traceback on the queue before exiting non-zero.
    When one rank of a gloo world dies its peers fail too ("Connection closed by peer"), often first: the parent must be able to
    print the rank that failed for a reason of its own, not whichever traceback reached stderr first."""
    import traceback
    try:
        target(rank, world, port, q, *extra)
    except BaseException:
        q.put((rank, "error", traceback.format_exc()))
        raise


def _run_world(target, world, *extra, timeout=240):
    """`world` spawned processes of `target(rank, world, port, queue, *extra)`; every rank must put (rank, "ok").  On failure the
    assertion message carries each failed rank's own traceback, ranks whose error is not a peer-disconnect first."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_guarded, args=(target, r, world, port, q) + tuple(extra)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout)
    hung = [i for i, p in enumerate(procs) if p.exitcode is None]
    for i in hung:
        procs[i].terminate()                      # exactly the children started here
        procs[i].join(10)
    got = []
    while True:
        try:
            got.append(q.get(timeout=1 if got or hung else 5))
        except Exception:
            break
        if len(got) >= 2 * world:
            break
    errors = sorted((g for g in got if len(g) == 3 and g[1] == "error"),
                    key=lambda g: ("Connection closed" in g[2] or "Connection reset" in g[2] or "Broken pipe" in g[2], g[0]))
    if errors or hung or any(p.exitcode != 0 for p in procs):
        msg = [f"exit codes {[p.exitcode for p in procs]}, hung ranks {hung}"]
        msg += [f"---- rank {g[0]} ----\n{g[2]}" for g in errors]
        raise AssertionError("\n".join(msg))
    assert sorted(g for g in got if len(g) == 2) == [(r, "ok") for r in range(world)], got


def _spawn(target, *extra, timeout=180):
    import tempfile
    os.environ["LMOD_TEST_TMP"] = tempfile.mkdtemp(prefix="lmod_gloo_")
    _run_world(target, 2, *extra, timeout=timeout)


def test_two_rank_gloo_zero2_sharded_optimizer_equals_unsharded():
    """ZeRO-2 style path (reduce-scatter -> AdamW on the owned chunks -> all-gather of bf16 weights) against the
    all-reduce + full-optimizer path over 2 steps, with global-norm clipping active; bias spans ride with their weight's
    hook; bf16 gradient exchange; gradient-accumulation windows exchange nothing."""
    _spawn(_zero2_worker)


def test_two_rank_gloo_gradient_allreduce():
    _run_world(_worker, 2, timeout=120)


def _edp_worker(rank, world, port, q):
    """World 4, ep_size 2: expert-parallel pairs {0,1}, {2,3}; expert-data-parallel pairs {0,2}, {1,3}.  A dense span is summed over all
    four ranks, an expert span over the two ranks that hold the same experts; with ZeRO-2 the owned chunk is 1/4 resp. 1/2 of the span;
    the clipped AdamW step on the shards equals the unsharded one and equals a single-process AdamW on the summed gradients."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), LMOD_DIST_BACKEND="gloo")
    from llavamod.engine import DataParallel, GradBuffer, HipAdamW, init_distributed
    from llavamod.model.language_model.qwen2_hip import Qwen2Config, Qwen2DecoderLayer, init_normal_
    from llavamod.model.moe_layer import MoE
    _cpu_kernels()
    init_distributed()
    ep = 2

    def build():
        cfg = Qwen2Config(vocab_size=64, hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=1)
        layer = Qwen2DecoderLayer(cfg, "cpu")
        layer.mlp = MoE(64, layer.mlp, num_experts=4, ep_size=ep, k=2, capacity_factor=1.5, min_capacity=0)
        init_normal_(layer, 0.02, 0)
        for n, p in layer.named_parameters():
            p.requires_grad = ("mlp" in n) or any(t in n for t in ("q_proj", "k_proj", "v_proj"))
        return layer

    def grads_of(r, numel, step):                              # what rank r puts into its gradient buffer
        i = torch.arange(numel, dtype=torch.float32)
        return torch.sin(i * 0.37 + step) * (0.5 + r) + 0.1 * r

    def run(zero2):
        layer = build()
        gb = GradBuffer(layer)
        dp = DataParallel(bucket_bytes=4096, zero2=zero2, min_shard_numel=1).attach(gb, ep_size=ep)
        opt = HipAdamW(gb, lr=1e-2, weight_decay=0.01, dp=dp, max_grad_norm=1.0)
        seen = {}
        for step in range(2):
            gb.zero()
            gb.flat.copy_(grads_of(rank, gb.numel, step))
            dp.finish()
            if step == 0:                                      # the exchanged gradient of every span, on the chunk this rank owns
                for (kind, obj, n), off in zip(gb.spans, gb.offsets):
                    info = dp.plan[(id(obj), kind)]
                    peers = [r for r in range(world) if r % ep == rank % ep] if info["is_expert"] else list(range(world))
                    assert info["gsize"] == len(peers), (kind, info["is_expert"], info["gsize"])
                    lo, hi = info["lo"], info["hi"]
                    if zero2 and info["sharded"]:
                        assert hi - lo == n // len(peers) and lo == off + peers.index(rank) * (n // len(peers))
                    exp = sum(grads_of(r, gb.numel, 0)[lo:hi] for r in peers)
                    assert torch.allclose(gb.flat[lo:hi], exp, atol=1e-5), (kind, info["is_expert"], zero2)
                    seen[info["is_expert"]] = seen.get(info["is_expert"], 0) + 1
            opt.step(grad_scale=1.0 / world, clear_grads=True)
        assert seen.get(True, 0) == 2 and seen.get(False, 0) >= 2          # stacked expert gate/up + down; router, q/k/v weight + bias
        return layer, float(opt.grad_norm)

    a, na = run(False)
    b, nb = run(True)
    assert abs(na - nb) <= 1e-5 * abs(na) and na > 1.0                       # global-norm clipping was active and agrees
    # sums of FOUR fp32 terms depend on the order (gloo's all-reduce and reduce-scatter add in different orders), so the clip
    # coefficient and the masters differ in their last bits and a bf16 weight may land on the other side of a rounding boundary:
    # at most one bf16 ulp, on a handful of elements (world 2 — order-free sums — is checked for exact equality above)
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        dd = (pa.float() - pb.float()).abs()
        if pa.dtype == torch.float32:
            assert float(dd.max()) <= 1e-6, n
        else:
            assert float((dd > 0).float().mean()) <= 2e-3 and bool((dd <= pa.float().abs() * 2 ** -7 + 1e-12).all()), (n, float(dd.max()))
    # the ranks of an expert-data-parallel pair hold identical experts after the step; the two pairs hold different ones
    w = torch.cat([p.detach().float().reshape(-1) for n, p in b.named_parameters() if "deepspeed_experts" in n])
    allw = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(allw, w)
    assert torch.equal(allw[0], allw[2]) and torch.equal(allw[1], allw[3])
    d = torch.cat([p.detach().float().reshape(-1) for n, p in b.named_parameters() if "deepspeed_experts" not in n and p.requires_grad])
    alld = [torch.zeros_like(d) for _ in range(world)]
    dist.all_gather(alld, d)
    assert all(torch.equal(alld[0], x) for x in alld[1:])                    # dense weights: identical everywhere
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


def test_four_rank_gloo_expert_data_parallel_groups():
    _run_world(_edp_worker, 4, timeout=240)


def _starved_expert_worker(rank, world, port, q):
    """World 4 = expert-parallel size 2 x expert-data-parallel size 2, ONE local expert per rank.  Rank 2 receives
    no rows for its expert in this step, its expert-data-parallel peer (rank 0) does.  The peer's `MLPBlock.backward` reports
    down-then-gate/up as their weight gradients are enqueued, and `DataParallel._on_ready` issues the two spans' collectives
    over the pair's communicator right there; the starved rank runs `ops.EmptyExpertPass`, which must fire the same hooks in the
    same order — otherwise its spans would only go out in `finish()`, in buffer order (gate/up first), against the peer's
    down-first: two different sequences of differently sized collectives on one communicator."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), LMOD_DIST_BACKEND="gloo")
    from llavamod import ops
    from llavamod.engine import DataParallel, GradBuffer, init_distributed
    from llavamod.model.language_model.qwen2_hip import Qwen2Config, Qwen2DecoderLayer
    from llavamod.model.moe_layer import MoE
    from types import SimpleNamespace
    init_distributed()
    ep = 2
    cfg = Qwen2Config(vocab_size=64, hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=1)
    layer = Qwen2DecoderLayer(cfg, "cpu")
    layer.mlp = MoE(64, layer.mlp, num_experts=2, ep_size=ep, k=1, capacity_factor=1.0, min_capacity=0)
    assert layer.mlp.num_local_experts == 1
    for n, p in layer.named_parameters():
        p.requires_grad = "mlp" in n
    gb = GradBuffer(layer)
    dp = DataParallel(bucket_bytes=1 << 20).attach(gb, ep_size=ep)
    gu, down = layer.mlp._gu, layer.mlp._down
    gu.ensure(); down.ensure()
    sent = []
    real_send = dp._send
    dp._send = lambda obj, kind: (sent.append("gu" if obj is gu else "down" if obj is down else "other"), real_send(obj, kind))[1]
    spec = SimpleNamespace(gu=gu, down=down)

    class FedBlock(torch.autograd.Function):          # MLPBlock's contract with the engine, without its kernels
        @staticmethod
        def forward(ctx, x, sp, *params):
            ctx.sp = sp
            sp.gu.note_use(); sp.down.note_use()
            return x * 2.0

        @staticmethod
        def backward(ctx, dout):
            sp = ctx.sp
            sp.down.grad_buffer().add_(1.0 + rank); sp.down.grad_done()
            sp.gu.grad_buffer().add_(10.0 + rank); sp.gu.grad_done()
            return (dout * 2.0, None) + (None,) * (len(ctx.needs_input_grad) - 2)

    params = [p for p in layer.mlp.deepspeed_moe.experts.parameters() if p.requires_grad]
    starved = rank == 2
    gb.zero()
    x = torch.zeros((0 if starved else 3, 64), dtype=torch.bfloat16, requires_grad=True)
    y = (ops.EmptyExpertPass if starved else FedBlock).apply(x, spec, *params)
    assert y.shape == (x.shape[0], 64)
    y.float().sum().backward()
    assert sent == ["down", "gu"], (rank, sent)                # both spans went out DURING backward, in MLPBlock's order
    assert x.grad.shape == x.shape
    dp.finish()
    peers = [r for r in range(world) if r % ep == rank % ep]
    fed = [r for r in peers if r != 2]
    assert bool((down.grad_buffer() == sum(1.0 + r for r in fed)).all()), rank
    assert bool((gu.grad_buffer() == sum(10.0 + r for r in fed)).all()), rank
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


def test_four_rank_gloo_starved_expert_keeps_collective_order():
    _run_world(_starved_expert_worker, 4, timeout=240)


def _chunked_ep_worker(rank, world, port, q):
    """The pipelined expert-parallel round trip (ops.chunked_expert_exchange: exchange of chunk c+1 under the expert block of chunk c,
    both directions) against the unchunked one (AllToAllRows -> block -> AllToAllRows) over gloo: same outputs, same input gradients,
    same parameter gradient, for 1 (degenerate), 2, 3 and 5 chunks, unequal and ZERO split sizes, a rank that receives nothing."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), LMOD_DIST_BACKEND="gloo")
    from llavamod import ops
    from llavamod.engine import expert_parallel_group, init_distributed
    import llavamod.kernels as K
    init_distributed()
    grp = expert_parallel_group(world)

    def gather_rows(src_a, src_b, idx, Hh):
        out = torch.zeros((idx.numel(), Hh), dtype=src_a.dtype)
        m = idx >= 0
        out[m] = src_a[idx[m].long()]
        return out
    K.gather_rows = gather_rows
    H = 6
    cases = {"uneven": [[3, 7, 0][:world], [5, 1, 4][:world], [0, 2, 9][:world]], "starved": [[4, 0, 3][:world], [6, 0, 2][:world], [1, 0, 5][:world]]}
    for name, table in cases.items():                            # table[src][dst] rows
        in_splits = table[rank][:world]
        out_splits = [table[s][rank] for s in range(world)]
        g = torch.Generator().manual_seed(17 + rank)
        base = torch.randn(sum(in_splits), H, generator=g, dtype=torch.float64)
        w = torch.randn(H, H, generator=torch.Generator().manual_seed(5), dtype=torch.float64)       # the "expert": same on every rank here
        cot = torch.randn(sum(in_splits), H, generator=g, dtype=torch.float64)

        def run(nchunk):
            x = base.clone().requires_grad_(True)
            wp = w.clone().requires_grad_(True)
            block = lambda rows: torch.tanh(rows @ wp) * 1.5 + rows          # row-wise, like the expert FFN
            if nchunk == 0:
                recv = ops.AllToAllRows.apply(x, in_splits, out_splits, grp)
                y = ops.AllToAllRows.apply(block(recv), out_splits, in_splits, grp)
            else:
                y = ops.chunked_expert_exchange(x, in_splits, out_splits, grp, block, nchunk)
            (y * cot).sum().backward()
            return y.detach(), x.grad, wp.grad
        y0, gx0, gw0 = run(0)
        for nchunk in (1, 2, 3, 5):
            y, gx, gw = run(nchunk)
            # (a host BLAS may block a [rows, H] product differently for another row count: equal to fp64 rounding, not bitwise)
            assert torch.allclose(y, y0, rtol=1e-13, atol=1e-13) and torch.allclose(gx, gx0, rtol=1e-13, atol=1e-13), (name, nchunk)
            assert torch.allclose(gw, gw0, rtol=1e-12, atol=1e-12), (name, nchunk)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


def test_three_rank_gloo_chunked_expert_exchange_equals_unchunked():
    _run_world(_chunked_ep_worker, 3, timeout=240)


def _one_rank_fails_worker(rank, world, port, q):
    if rank == 1:
        raise ValueError("rank 1 failed for a reason of its own")
    q.put((rank, "ok"))


def test_a_failing_rank_reports_its_own_traceback():
    """The spawn helper surfaces the failing rank's OWN exception, not just exit codes."""
    with pytest.raises(AssertionError) as e:
        _run_world(_one_rank_fails_worker, 2, timeout=60)
    assert "rank 1 failed for a reason of its own" in str(e.value) and "---- rank 1 ----" in str(e.value)
