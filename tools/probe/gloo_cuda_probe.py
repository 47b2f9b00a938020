"""Does gloo move CUDA tensors for the collectives the exchange uses (2 ranks on ONE GPU)?"""
import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp


def w(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    res = {}
    def tryit(name, f):
        try:
            f(); torch.cuda.synchronize(); res[name] = "ok"
        except Exception as e:
            res[name] = "FAIL " + repr(e)[:120]
    x = torch.arange(16, device=dev, dtype=torch.float32) + rank
    tryit("all_reduce", lambda: dist.all_reduce(x))
    flat = torch.arange(16, device=dev, dtype=torch.float32) + rank
    tryit("reduce_scatter_tensor in place", lambda: dist.reduce_scatter_tensor(flat[rank * 8:(rank + 1) * 8], flat))
    pb = torch.zeros(16, device=dev, dtype=torch.bfloat16); pb[rank * 8:(rank + 1) * 8] = rank + 1
    tryit("all_gather_into_tensor in place bf16", lambda: dist.all_gather_into_tensor(pb, pb[rank * 8:(rank + 1) * 8]))
    a = torch.full((6, 4), float(rank), device=dev, dtype=torch.bfloat16); o = torch.empty_like(a)
    tryit("all_to_all_single bf16", lambda: dist.all_to_all_single(o, a))
    h = None
    def asyncar():
        hh = dist.all_reduce(x, async_op=True); hh.wait()
    tryit("async all_reduce", asyncar)
    if rank == 0:
        print(res, flat.tolist()[:8], pb.float().tolist())
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(w, args=(2, 29611), nprocs=2)
