import json, os, sys, torch
sys.path.insert(0, "/root/repo/llava-mod_amd")
from llavamod import kernels as K
BF = torch.bfloat16
def t(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3
for M, N, Kd, what in ((9232, 1024, 4096, "tower fc2"), (9232, 1024, 1024, "tower o_proj"), (9216, 2048, 1024, "projector 1"), (9216, 2048, 2048, "projector 2"),
                       (8208, 2048, 2048, "loss-row o_proj"), (8208, 2048, 5504, "loss-row down")):
    x = torch.randn(M, Kd, device="cuda").to(BF); w = torch.randn(N, Kd, device="cuda").to(BF); b = torch.randn(N, device="cuda").to(BF)
    o = torch.empty(M, N, device="cuda", dtype=BF)
    s = t(lambda: K.gemm_nt(x, w, bias=b, out=o))
    print(json.dumps({"tile": os.environ.get("LMOD_GEMM_TILE", "default"), "launch": what, "shape": [M, N, Kd], "tiles256": ((M+255)//256)*((N+255)//256), "us": round(s*1e6, 1), "tflops": round(2.0*M*N*Kd/s/1e12, 1)}), flush=True)
