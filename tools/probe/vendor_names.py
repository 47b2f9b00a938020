"""Which library kernels PyTorch-ROCm's F.linear runs on the step's shapes (run under rocprofv3 --kernel-trace --stats)."""
import torch, torch.nn.functional as F
g = torch.Generator(device="cuda"); g.manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
with torch.no_grad():
    for M, N, K in [(32768, 12288, 4096), (32768, 4096, 11008), (32768, 6144, 2048), (32768, 2048, 5504), (8208, 151936, 2048)]:
        x, w = rnd(M, K), rnd(N, K)
        for _ in range(5):
            F.linear(x, w)
        torch.cuda.synchronize()
        del x, w
