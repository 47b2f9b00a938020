"""Sustained (seconds-long) GEMM throughput: does a kernel's advantage survive the chip's power management?"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llava-mod_amd"))
from llavamod import kernels as K
M, N, Kd = 32768, 12288, 4096
a = torch.randn(M, Kd, device="cuda").to(torch.bfloat16); b = torch.randn(N, Kd, device="cuda").to(torch.bfloat16)
o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
fl = 2.0 * M * N * Kd
for n in (10, 100, 800):
    K.gemm_nt(a, b, out=o); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): K.gemm_nt(a, b, out=o)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("waves", os.environ.get("LMOD_GEMM_WAVES", "4"), n, "launches", round(dt * 1e3, 1), "ms ->", round(fl * n / dt / 1e12), "TF", flush=True)
