"""Single-rank RCCL plumbing check (the GPU box has one GPU): process-group init with the nccl backend,
all_reduce on views of a flat buffer with async handles, all_to_all_single, barrier."""
import os, sys, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
os.environ.setdefault("LMOD_FORCE_DIST", "1")
from llavamod.engine import init_distributed, expert_parallel_group
rank, local, world = init_distributed()
print("backend", dist.get_backend(), "world", world)
flat = torch.arange(1 << 20, device="cuda", dtype=torch.float32)
hs = [dist.all_reduce(flat[i:i + 4096], async_op=True) for i in range(0, 1 << 16, 4096)]
for h in hs: h.wait()
x = torch.randn(1, 8, 4, device="cuda").to(torch.bfloat16); y = torch.empty_like(x)
dist.all_to_all_single(y, x, group=expert_parallel_group(1)); assert torch.equal(x, y)
t = torch.tensor([1.5], device="cuda", dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier(); torch.cuda.synchronize(); dist.destroy_process_group(); print("rccl smoke ok")
