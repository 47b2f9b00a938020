"""MoE expert weight-gradient GEMM (grouped NT, fp32 accumulate, k_valid) against dense references of the same work."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K
from bench_kernels import timeit
BF = torch.bfloat16
E, N, Kin, C, live = 4, 11008, 2048, 24576, 16384
for (N, Kin, tag) in ((11008, 2048, "gate_up"), (2048, 5504, "down")):
    at = torch.randn(E, N, C, device="cuda").to(BF) * 0.05
    bt = torch.randn(E, Kin, C, device="cuda").to(BF) * 0.05
    out = torch.zeros(E, N, Kin, device="cuda")
    rows = torch.full((E,), live, device="cuda", dtype=torch.int32)
    fl = 2.0 * E * N * Kin * live
    def rec(name, t, f=fl):
        print(json.dumps({"case": tag + ": " + name, "ms": round(t * 1e3, 3), "tflops": round(f / t / 1e12)}), flush=True)
    rec("grouped E=4, k_valid 16384 of C 24576 (the step's call)", timeit(lambda: K.gemm_nt(at, bt, out=out, out_f32=True, accumulate=True, k_valid=rows)))
    for name, rr in (("ascending [9000, 12000, 20000, 24576]", [9000, 12000, 20000, 24576]), ("descending", [24576, 20000, 12000, 9000]),
                     ("mild [15000, 17500, 16000, 17036]", [15000, 17500, 16000, 17036])):
        rv = torch.tensor(rr, device="cuda", dtype=torch.int32)
        rec("grouped E=4, imbalanced " + name, timeit(lambda: K.gemm_nt(at, bt, out=out, out_f32=True, accumulate=True, k_valid=rv)),
            2.0 * N * Kin * sum(rr))
    a2, b2 = at[:, :, :live].contiguous(), bt[:, :, :live].contiguous()
    rec("grouped E=4, dense K 16384 (row pitch 16384)", timeit(lambda: K.gemm_nt(a2, b2, out=out, out_f32=True, accumulate=True)))
    rec("grouped E=4, no accumulate", timeit(lambda: K.gemm_nt(a2, b2, out=out, out_f32=True, accumulate=False)))
    def four():
        for e in range(E):
            K.gemm_wgrad(a2[e], b2[e], out[e])
    rec("4 x gemm_wgrad (deterministic split-K by the heuristic)", timeit(four))
    def four_live():
        for e in range(E):
            K.gemm_wgrad(at[e, :, :live], bt[e, :, :live], out[e])
    rec("4 x gemm_wgrad on the slab views (pitch 24576)", timeit(four_live))
    a3 = torch.randn(N, 2 * live, device="cuda").to(BF) * 0.05
    b3 = torch.randn(Kin, 2 * live, device="cuda").to(BF) * 0.05
    rec("dense wgrad K 32768 (split-K heuristic)", timeit(lambda: K.gemm_wgrad(a3, b3, out[0])), 2.0 * N * Kin * 2 * live)
