"""A few launches of ONE GEMM kernel for the PMC passes (rocprofv3 --pmc ...): the kernel the bench line's `roofline` describes.
G1_KERNEL=swiglu (default since round 6: `gemm4_kernel<1>`, the fused SwiGLU forward at the teacher MLP shape, the top row of the committed
kernel summary) | nt (`gemm4_kernel<7>` at the teacher QKV shape, rounds 1-5).  G1_META=<file>: writes {kind, shape, output_cols, what, cmd}."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K
kind = os.environ.get("G1_KERNEL", "swiglu")
M, Kd = 32768, 4096
a = torch.randn(M, Kd, device="cuda").to(torch.bfloat16)
if kind == "swiglu":
    I = 11008
    b = torch.randn(2 * I, Kd, device="cuda").to(torch.bfloat16)
    o = torch.empty(M, I, device="cuda", dtype=torch.bfloat16)
    run = lambda: K.gemm_swiglu(a, b, act=o)
    meta = {"kind": kind, "shape": [M, 2 * I, Kd], "output_cols": I, "what": "the teacher MLP gate+up shape (fused SwiGLU forward, no pre-activation store)",
            "cmd": "G1_KERNEL=swiglu python tools/gemm_one.py"}
else:
    N = 12288       # teacher fused QKV projection at micro-batch 16
    b = torch.randn(N, Kd, device="cuda").to(torch.bfloat16)
    o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    run = lambda: K.gemm_nt(a, b, out=o)
    meta = {"kind": kind, "shape": [M, N, Kd], "what": "the teacher QKV shape", "cmd": "G1_KERNEL=nt python tools/gemm_one.py"}
for _ in range(4):
    run()
torch.cuda.synchronize()
if os.environ.get("G1_META"):
    json.dump(meta, open(os.environ["G1_META"], "w"))
