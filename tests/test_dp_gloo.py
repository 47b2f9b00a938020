"""Multi-process data-parallel path on CPU: 2 ranks over gloo (the GPU path uses the same code with the
"nccl" = RCCL backend).  Covers rank/world discovery from the launcher env, the flat gradient buffer
layout being identical on every rank, the bucketed SUM all-reduce, and rank-distinct data shards."""
import os
import socket
import sys

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
                      MASTER_PORT=str(port))
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
    assert dpo._handles and id(gu) in dpo._done
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


def test_two_rank_gloo_gradient_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, "ok"), (1, "ok")]
