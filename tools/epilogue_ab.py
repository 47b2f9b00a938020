"""Same-process A/B of two library builds on the GEMM launches whose EPILOGUES read memory: the fused SwiGLU backward
(MODE 4: reads [gate | up], writes [dgate | dup]), the fused QKV + bias + RoPE (MODE 5: bias, positions, cos/sin) and the fp32
read-modify-write of un-split weight gradients (MODE 6: lm_head, MoE experts).  Checks the outputs of both builds are
bit-identical, then times them alternately.

    python tools/epilogue_ab.py [--other alt_libs/liblmod_head.so] [--iters 20] [--rounds 3]
"""
import argparse, ctypes, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--other", default=os.path.join(ROOT, "alt_libs", "liblmod_head.so"))
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--rounds", type=int, default=3)
args = ap.parse_args()
P, I, Q = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
SIG = {"lmod_gemm_bf16_nt": [P, P, P, P, I, I, I, I, I, I, I, Q, Q, Q, P, P, I, I, I, P],
       "lmod_gemm_qkv_rope_bf16": [P, P, P, P, I, I, I, I, I, I, P, P, P, I, P],
       "lmod_gemm_swiglu_bwd_bf16": [P, P, P, P, I, I, I, I, I, I, I, I, Q, Q, Q, Q, P, P],
       "lmod_gemm_swiglu_bf16": [P, P, P, P, I, I, I, I, I, I, I, I, Q, Q, Q, Q, P, P]}
libs = {"new": ctypes.CDLL(os.path.join(ROOT, "llava-mod_amd", "llavamod", "_lib", "liblmod_hip.so")), "old": ctypes.CDLL(args.other)}
for lib in libs.values():
    for n, a in SIG.items():
        f = getattr(lib, n); f.restype = I; f.argtypes = a
dev = torch.device("cuda")
bf = torch.bfloat16
g = torch.Generator(device=dev); g.manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).to(bf)
ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None


def timed(fn):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(args.iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / args.iters


def case_swiglu_bwd(M, N, K, tag):
    A, Bt, gu = rnd(M, K), rnd(N, K), rnd(M, 2 * N)
    outs = {k: torch.empty(M, 2 * N, device=dev, dtype=bf) for k in libs}
    def run(k):
        rc = libs[k].lmod_gemm_swiglu_bwd_bf16(ptr(A), ptr(Bt), ptr(gu), ptr(outs[k]), M, N, K, K, K, 2 * N, 2 * N, 1, 0, 0, 0, 0, None, None)
        assert rc == 0, rc
    return tag, 2.0 * M * N * K, run, outs


def case_swiglu_fwd(M, N, K, tag, keep):
    A, W = rnd(M, K), rnd(2 * N, K)
    outs = {k: torch.empty(M, N + (2 * N if keep else 0), device=dev, dtype=bf) for k in libs}      # [act | gate | up]
    def run(k):
        act = outs[k][:, :N]
        gu = outs[k][:, N:] if keep else None
        rc = libs[k].lmod_gemm_swiglu_bf16(ptr(A), ptr(W), ptr(act), ptr(gu), M, N, K, K, K, outs[k].stride(0), outs[k].stride(0),
                                           1, 0, 0, 0, 0, None, None)
        assert rc == 0, rc
    return tag, 4.0 * M * N * K, run, outs


def case_qkv_rope(M, N, K, tag, S=2048):
    A, W, bias = rnd(M, K), rnd(N, K), rnd(N)
    cos, sin = rnd(S, 128), rnd(S, 128)
    pos = (torch.arange(M, device=dev, dtype=torch.int32) % S).contiguous()
    outs = {k: torch.empty(M, N, device=dev, dtype=bf) for k in libs}
    rope_cols = N * 2 // 3
    def run(k):
        rc = libs[k].lmod_gemm_qkv_rope_bf16(ptr(A), ptr(W), ptr(outs[k]), ptr(bias), M, N, K, K, K, N, ptr(cos), ptr(sin), ptr(pos), rope_cols, None)
        assert rc == 0, rc
    return tag, 2.0 * M * N * K, run, outs


def case_acc(M, N, K, tag, batch=1, kv=None):
    A, B = rnd(batch, M, K), rnd(batch, N, K)
    base = torch.rand(batch, M, N, device=dev, generator=g)
    outs = {k: base.clone() for k in libs}
    kvt = torch.tensor(kv, device=dev, dtype=torch.int32) if kv else None
    def run(k):
        rc = libs[k].lmod_gemm_bf16_nt(ptr(A), ptr(B), ptr(outs[k]), None, M, N, K, K, K, N, batch, M * K, N * K, M * N, None, ptr(kvt), 0, 1, 1, None)
        assert rc == 0, rc
    fl = 2.0 * M * N * (sum(kv) if kv else K * batch)
    return tag, fl, run, outs


cases = [lambda: case_swiglu_fwd(32768, 11008, 4096, "swiglu fwd teacher (act only) [32768 x 2*11008 x 4096]", False),
         lambda: case_swiglu_fwd(32768, 5504, 2048, "swiglu fwd student (act + [gate|up]) [32768 x 2*5504 x 2048]", True),
         lambda: case_swiglu_bwd(32768, 5504, 2048, "swiglu_bwd student dense [32768 x 5504 x 2048]"),
         lambda: case_swiglu_bwd(16384, 5504, 2048, "swiglu_bwd [16384 x 5504 x 2048]"),
         lambda: case_qkv_rope(32768, 6144, 2048, "qkv_rope student [32768 x 6144 x 2048]"),
         lambda: case_qkv_rope(32768, 12288, 4096, "qkv_rope teacher [32768 x 12288 x 4096]"),
         lambda: case_acc(151936, 2048, 8208, "fp32 accumulate: lm_head wgrad [151936 x 2048 x 8208]"),
         lambda: case_acc(11008, 2048, 24576, "fp32 accumulate: 4 experts gate+up wgrad, k_valid 9k..24k", batch=4, kv=[9000, 12000, 20000, 24576]),
         lambda: case_acc(2048, 5504, 24576, "fp32 accumulate: 4 experts down wgrad, k_valid 9k..24k", batch=4, kv=[9000, 12000, 20000, 24576])]
for mk in cases:
    tag, fl, run, outs = mk()
    for k in libs: run(k)
    torch.cuda.synchronize()
    same = bool(torch.equal(outs["new"], outs["old"]))
    if "accumulate" in tag:     # accumulating: equality after ONE call each from the same base is what was just compared
        pass
    ms = {k: [] for k in libs}
    for _ in range(args.rounds):
        for k in libs: ms[k].append(timed(lambda: run(k)))
    best = {k: min(v) for k, v in ms.items()}
    print(json.dumps({"case": tag, "bit_identical": same, "ms_new": round(best["new"], 4), "ms_old": round(best["old"], 4),
                      "tf_new": round(fl / best["new"] / 1e9, 1), "tf_old": round(fl / best["old"] / 1e9, 1),
                      "speedup": round(best["old"] / best["new"], 4)}), flush=True)
    del outs
    torch.cuda.empty_cache()
