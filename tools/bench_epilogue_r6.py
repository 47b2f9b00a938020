"""Round 6: the load-carrying epilogues at the step's shapes — fused SwiGLU backward (dense, grouped) and the residual-add projection
(teacher o_proj / down_proj, student o_proj) — one process per arm (LMOD_HIP_LIB variant library, LMOD_GEMM_STAGGER, ...).  JSON lines."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K  # noqa: E402

BF = torch.bfloat16


def t(fn, it=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3


arm = {"lib": os.path.basename(os.environ.get("LMOD_HIP_LIB", "default")), "stagger": os.environ.get("LMOD_GEMM_STAGGER", "0"),
       "sb4g": os.environ.get("LMOD_GEMM_SB4G", "1")}
T = 32768
gu = torch.randn(T, 11008, device="cuda").to(BF); dy = torch.randn(T, 2048, device="cuda").to(BF); wdt = torch.randn(5504, 2048, device="cuda").to(BF)
out = torch.empty_like(gu)
s = t(lambda: K.gemm_swiglu_bwd(dy, wdt, gu, out=out, K=2048))
print(json.dumps(dict(arm, op="swiglu_bwd dense [32768 x 5504 x 2048]", ms=round(s * 1e3, 4), tflops=round(2.0 * T * 5504 * 2048 / s / 1e12, 1))), flush=True)
E, C, H, I = 4, 24576, 2048, 5504
mv = torch.tensor([9000, 12000, 20000, 24536], dtype=torch.int32, device="cuda")       # 65536 live rows, uneven like a random-init router
dye = torch.randn(E, C, H, device="cuda").to(BF); wte = torch.randn(E, I, H, device="cuda").to(BF); gue = torch.randn(E, C, 2 * I, device="cuda").to(BF)
oute = torch.empty_like(gue)
s = t(lambda: K.gemm_swiglu_bwd(dye, wte, gue, out=oute, m_valid=mv, K=H))
print(json.dumps(dict(arm, op="swiglu_bwd grouped [65536 live x 5504 x 2048]", ms=round(s * 1e3, 4), tflops=round(2.0 * 65536 * I * H / s / 1e12, 1))), flush=True)
del gu, out, gue, oute, dye, wte
for M, N, Kd in ((32768, 4096, 11008), (32768, 4096, 4096), (32768, 2048, 5504), (32768, 2048, 2048)):
    x = torch.randn(M, Kd, device="cuda").to(BF); w = torch.randn(N, Kd, device="cuda").to(BF); r = torch.randn(M, N, device="cuda").to(BF)
    o = torch.empty(M, N, device="cuda", dtype=BF)
    s = t(lambda: K.gemm_nt_res(x, w, r, out=o))
    print(json.dumps(dict(arm, op=f"nt_res [{M} x {N} x {Kd}]", ms=round(s * 1e3, 4), tflops=round(2.0 * M * N * Kd / s / 1e12, 1))), flush=True)
    s = t(lambda: K.gemm_nt(x, w, out=o))
    print(json.dumps(dict(arm, op=f"nt (no residual) [{M} x {N} x {Kd}]", ms=round(s * 1e3, 4), tflops=round(2.0 * M * N * Kd / s / 1e12, 1))), flush=True)
