"""Attention forward / backward throughput at the step's shapes (HIP-event timing, random data).
LMOD_ATTN_FWD=1 selects the round-1 forward kernel for hd 128 (A/B)."""
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K  # noqa: E402

from bench_kernels import timeit  # noqa: E402

BF = torch.bfloat16


def run(B, S, nh, nkv, hd, causal, bwd, ragged=False):
    ld = (nh + 2 * nkv) * hd
    qkv = torch.randn(B * S, ld, device="cuda").to(BF)
    q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
    sc = 1 / math.sqrt(hd)
    sl = None
    if ragged:
        sl = torch.randint(600 + 575, 1473 + 575 + 1, (B,), device="cuda", dtype=torch.int32)
    o, lse = K.attn_fwd(q, k, v, B, S, nh, nkv, hd, sc, causal, sl)
    fl = 4.0 * B * nh * S * S * hd * (0.5 if causal else 1.0)
    t = timeit(lambda: K.attn_fwd(q, k, v, B, S, nh, nkv, hd, sc, causal, sl))
    rec = {"kernel": "attn_fwd", "ver": os.environ.get("LMOD_ATTN_FWD", "2"), "B": B, "S": S, "nh": nh, "nkv": nkv, "hd": hd,
           "causal": causal, "ragged": ragged, "ms": round(t * 1e3, 4), "tflops": round(fl / t / 1e12, 1)}
    print(json.dumps(rec), flush=True)
    if bwd:
        do = torch.randn(B * S, nh * hd, device="cuda").to(BF)
        dqkv = torch.empty_like(qkv)
        f = lambda: K.attn_bwd(q, k, v, o, do, lse, dqkv[:, :nh * hd], dqkv[:, nh * hd:(nh + nkv) * hd],
                               dqkv[:, (nh + nkv) * hd:], B, S, nh, nkv, hd, sc, causal, sl)
        t = timeit(f)
        print(json.dumps({"kernel": "attn_bwd", "ver": os.environ.get("LMOD_ATTN_BWD", "2"),
                          "form": "dS spill + dQ GEMM (5 matmuls)" if K.attn_bwd_ds_fusable(B, S, nh, hd) else "dQ kernel + dK/dV kernel (7 matmuls)", "B": B, "S": S, "nh": nh, "hd": hd, "causal": causal, "ms": round(t * 1e3, 4),
                          "tflops_algo(2.5x fwd)": round(2.5 * fl / t / 1e12, 1)}), flush=True)


if __name__ == "__main__":
    bwd = "--bwd" in sys.argv
    if "--bwd-one" in sys.argv:
        run(16, 2048, 16, 16, 128, True, True)
        run(8, 2048, 16, 16, 128, False, True)
        sys.exit(0)
    if "--hd64" in sys.argv:                          # round 4: head dim 64 (Qwen2-0.5B student: 14 heads, 2 KV heads; CLIP tower: 16 heads, S 577)
        for causal, B, S, nh, nkv in ((True, 16, 2048, 14, 2), (True, 16, 2048, 16, 16), (False, 32, 577, 16, 16), (True, 4, 8192, 14, 2)):
            run(B, S, nh, nkv, 64, causal, True)
        sys.exit(0)
    if "--bwd-only" in sys.argv:                      # the backward at the step's shapes (LMOD_ATTN_BWD=1: generic kernels)
        run(16, 2048, 16, 16, 128, True, True)
        run(8, 2048, 16, 16, 128, False, True)
        run(4, 8192, 16, 16, 128, True, True)
        run(16, 2048, 16, 16, 128, True, True, ragged=True)
        sys.exit(0)
    run(8, 2048, 16, 16, 128, True, bwd)
    run(16, 2048, 16, 16, 128, True, False)
    run(16, 2048, 32, 32, 128, True, False)
    run(8, 2048, 16, 16, 128, False, False)
    run(4, 8192, 16, 16, 128, True, False)
    run(16, 2048, 16, 16, 128, True, False, ragged=True)
