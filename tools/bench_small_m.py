"""Round 6: the GEMM launches of the reference's own launch regime (micro-batch 1: M = 2048 token rows) that under-fill the chip, as routed
today against a deterministic split-K into an fp32 image + cast (+ residual add).  JSON lines."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K  # noqa: E402

BF = torch.bfloat16


def t(fn, it=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3


for M, N, Kd, what in ((2048, 4096, 11008, "teacher down_proj"), (2048, 4096, 4096, "teacher o_proj"), (2048, 2048, 5504, "student down_proj"),
                       (2048, 2048, 2048, "student o_proj"), (2048, 12288, 4096, "teacher QKV (plain)"), (2048, 6144, 2048, "student QKV (plain)"),
                       (4096, 4096, 11008, "teacher down_proj, micro-batch 2"), (4096, 4096, 4096, "teacher o_proj, micro-batch 2")):
    x = torch.randn(M, Kd, device="cuda").to(BF); w = torch.randn(N, Kd, device="cuda").to(BF); r = torch.randn(M, N, device="cuda").to(BF)
    o = torch.empty(M, N, device="cuda", dtype=BF)
    acc = torch.empty(M, N, device="cuda", dtype=torch.float32)
    fl = 2.0 * M * N * Kd

    def today():
        if K.gemm_res_fusable(M, w, r):
            K.gemm_nt_res(x, w, r, out=o)
        else:
            K.gemm_nt(x, w, out=o)
            K.add(o, r, out=o)

    def splitk():
        acc.zero_()
        K.gemm_wgrad(x, w, acc)
        K.cast_f32_bf16(acc.view(-1), o.view(-1))
        K.add(o, r, out=o)

    a, b = t(today), t(splitk)
    print(json.dumps({"launch": what, "shape": [M, N, Kd], "tiles_256": ((M + 255) // 256) * ((N + 255) // 256), "fused_today": bool(K.gemm_res_fusable(M, w, r)),
                      "today_us": round(a * 1e6, 1), "today_tflops": round(fl / a / 1e12, 1), "splitk_fp32_cast_add_us": round(b * 1e6, 1),
                      "splitk_tflops": round(fl / b / 1e12, 1)}), flush=True)
