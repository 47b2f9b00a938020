"""Micro-benchmarks of the dominant kernels at config-2 shapes (MI355X).  Prints one JSON line per
kernel with achieved TFLOP/s (GEMM, attention) or GB/s (row kernels) from HIP-event timing."""
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def gemm(M, N, Kd, tag):
    a = torch.randn(M, Kd, device=DEV).to(BF)
    b = torch.randn(N, Kd, device=DEV).to(BF)
    out = torch.empty(M, N, device=DEV, dtype=BF)
    t = timeit(lambda: K.gemm_nt(a, b, out=out))
    print(json.dumps({"kernel": "gemm_nt", "tag": tag, "M": M, "N": N, "K": Kd, "ms": round(t * 1e3, 4),
                      "tflops": round(2.0 * M * N * Kd / t / 1e12, 1)}), flush=True)


def attn(B, S, nh, hd, causal, bwd):
    ld = 3 * nh * hd
    qkv = torch.randn(B * S, ld, device=DEV).to(BF)
    q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:2 * nh * hd], qkv[:, 2 * nh * hd:]
    sc = 1 / math.sqrt(hd)
    o, lse = K.attn_fwd(q, k, v, B, S, nh, nh, hd, sc, causal)
    fl = 4.0 * B * nh * S * S * hd * (0.5 if causal else 1.0)
    t = timeit(lambda: K.attn_fwd(q, k, v, B, S, nh, nh, hd, sc, causal))
    print(json.dumps({"kernel": "attn_fwd", "B": B, "S": S, "nh": nh, "hd": hd, "causal": causal,
                      "ms": round(t * 1e3, 4), "tflops": round(fl / t / 1e12, 1)}), flush=True)
    if bwd:
        do = torch.randn(B * S, nh * hd, device=DEV).to(BF)
        dqkv = torch.empty_like(qkv)
        f = lambda: K.attn_bwd(q, k, v, o, do, lse, dqkv[:, :nh * hd], dqkv[:, nh * hd:2 * nh * hd],
                               dqkv[:, 2 * nh * hd:], B, S, nh, nh, hd, sc, causal)
        t = timeit(f)
        print(json.dumps({"kernel": "attn_bwd", "B": B, "S": S, "nh": nh, "hd": hd, "causal": causal,
                          "ms": round(t * 1e3, 4), "tflops_algo(2.5x fwd)": round(2.5 * fl / t / 1e12, 1)}), flush=True)


def rows():
    T, H = 16384, 2048
    x = torch.randn(T, H, device=DEV).to(BF)
    w = torch.ones(H, device=DEV, dtype=BF)
    t = timeit(lambda: K.rmsnorm_fwd(x, w, 1e-6))
    print(json.dumps({"kernel": "rmsnorm_fwd", "T": T, "H": H, "ms": round(t * 1e3, 4),
                      "GBps": round(2 * T * H * 2 / t / 1e9, 1)}), flush=True)
    R, V = 4096, 151936
    s = torch.randn(R, V, device=DEV).to(BF)
    tt = torch.randn(R, V, device=DEV).to(BF)
    lab = torch.randint(0, V, (R,), device=DEV, dtype=torch.int32)
    t = timeit(lambda: K.rowloss_fwd(s, V, tt, V, lab), iters=5)
    print(json.dumps({"kernel": "rowloss_fwd", "R": R, "V": V, "ms": round(t * 1e3, 4),
                      "GBps": round(2 * R * V * 2 / t / 1e9, 1)}), flush=True)
    xt = torch.randn(16384, 5504, device=DEV).to(BF)
    t = timeit(lambda: K.transpose(xt))
    print(json.dumps({"kernel": "transpose", "R": 16384, "C": 5504, "ms": round(t * 1e3, 4),
                      "GBps": round(2 * 16384 * 5504 * 2 / t / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    T = 16384  # 8 samples x 2048 tokens
    gemm(4096, 4096, 4096, "square 4k")
    gemm(8192, 8192, 8192, "square 8k")
    gemm(T, 6144, 2048, "student QKV")
    gemm(T, 11008, 2048, "student gate+up")
    gemm(T, 2048, 5504, "student down")
    gemm(T, 12288, 4096, "teacher QKV")
    gemm(T, 22016, 4096, "teacher gate+up")
    gemm(T, 4096, 11008, "teacher down")
    gemm(4104, 151936, 2048, "student lm_head (loss rows)")
    gemm(4104, 151936, 4096, "teacher lm_head (loss rows)")
    gemm(8 * 577, 1024, 1024, "ViT proj")
    attn(8, 2048, 16, 128, True, True)
    attn(8, 2048, 32, 128, True, False)
    attn(8, 577, 16, 64, False, False)
    rows()
