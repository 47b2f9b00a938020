"""C-ABI collectives (csrc/comm.hip, include/lmod_hip.h) through real RCCL.  A 1-GPU box exercises the whole call path with a
world of one (unique id, communicator, in-place all-reduce / reduce-scatter / all-gather, grouped send/recv all-to-all); the
2-rank case runs when the box has two GPUs."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))


def test_native_collectives_world_of_one():
    from llavamod.comm import NativeComm, unique_id
    torch.cuda.set_device(0)
    uid = unique_id()
    assert len(uid) == 128 and any(b != 0 for b in uid)
    c = NativeComm(uid, 0, 1)
    g = torch.randn(1 << 20, device="cuda")
    ref = g.clone()
    c.allreduce_(g)
    gb = torch.randn(4096, device="cuda").to(torch.bfloat16)
    refb = gb.clone()
    c.allreduce_(gb)
    chunk = c.reduce_scatter_(g)
    c.allgather_(g)
    rows = (torch.randn(37, 256, device="cuda")).to(torch.bfloat16)
    back = c.moe_all_to_all(rows, [37], [37])
    empty = c.moe_all_to_all(rows[:0], [0], [0])
    torch.cuda.synchronize()
    assert torch.equal(g, ref) and torch.equal(gb, refb) and chunk.data_ptr() == g.data_ptr()
    assert torch.equal(back, rows) and empty.shape == (0, 256)
    c.close()


def _worker(rank, world, uid, q):
    from llavamod.comm import NativeComm
    torch.cuda.set_device(rank)
    c = NativeComm(uid, rank, world)
    g = torch.full((1024,), float(rank + 1), device="cuda")
    c.allreduce_(g)
    span = torch.arange(2048, device="cuda", dtype=torch.float32) * (rank + 1)
    mine = c.reduce_scatter_(span).clone()
    pb = torch.zeros(2048, device="cuda", dtype=torch.bfloat16)
    pb[rank * 1024:(rank + 1) * 1024] = rank + 1
    c.allgather_(pb)
    send_rows = [3, 5] if rank == 0 else [2, 4]               # rows for peer 0, peer 1
    recv_rows = [3, 2] if rank == 0 else [5, 4]
    send = (torch.arange(sum(send_rows) * 8, device="cuda").view(-1, 8) + 1000 * rank).to(torch.bfloat16)
    recv = c.moe_all_to_all(send, send_rows, recv_rows)
    torch.cuda.synchronize()
    ok = bool((g == 3.0).all()) and torch.equal(mine, torch.arange(2048, device="cuda")[rank * 1024:(rank + 1) * 1024] * 3.0)
    ok = ok and bool((pb[:1024] == 1).all()) and bool((pb[1024:] == 2).all())
    # what the peer sent me: its rows [0:2) (rank 0 receives rank 1's first block) or [3:8) (rank 1 receives rank 0's second)
    peer_rows = 6 if rank == 0 else 8
    peer = (torch.arange(peer_rows * 8, device="cuda").view(-1, 8) + 1000 * (1 - rank)).to(torch.bfloat16)
    exp = torch.cat([send[:3], peer[:2]]) if rank == 0 else torch.cat([peer[3:8], send[2:6]])
    ok = ok and torch.equal(recv, exp)
    q.put((rank, ok, tuple(recv.shape)))
    c.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_native_collectives_two_ranks():
    import torch.multiprocessing as mp
    from llavamod.comm import unique_id
    uid = unique_id()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, uid, q)) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(180)
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, True, (5, 8)), (1, True, (9, 8))], got
