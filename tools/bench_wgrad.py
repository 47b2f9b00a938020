"""wgrad forms at the student's shapes: NT (both operands transposed copies) vs X read reduction-major (MODE 3) vs TN (8-wave kernel,
LMOD_GEMM_TN4=0) vs the 4-wave TN asm loop without / with deterministic split-K (tn4 / tn4_wgrad) and the MoE experts' batched k_valid form."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
os.environ.setdefault("LMOD_GEMM_ENV_DYNAMIC", "1")      # both arms of a routing switch in one process
from llavamod import kernels as K
T = 32768
def t(fn, n=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for tag, M, N in [("gate+up", 11008, 2048), ("qkv", 6144, 2048), ("down", 2048, 5504), ("o", 2048, 2048)]:
    dy = torch.randn(T, M, device="cuda").to(torch.bfloat16); x = torch.randn(T, N, device="cuda").to(torch.bfloat16)
    g = torch.zeros(M, N, device="cuda", dtype=torch.float32)
    fl = 2.0 * T * M * N
    dyt, xt = K.transpose(dy), K.transpose(x)
    res = {"T(dy)": t(lambda: K.transpose(dy)), "T(x)": t(lambda: K.transpose(x)),
           "nt": t(lambda: K.gemm_wgrad(dyt, xt, g)),
           "x_kmajor": t(lambda: K.gemm_wgrad(dyt, x, g, b_kmajor=True)),
           "tn4": t(lambda: K.gemm_tn(dy, x, out=g, accumulate=True)),
           "tn4_wgrad": t(lambda: K.gemm_wgrad(dy, x, g, a_kmajor=True))}
    os.environ["LMOD_GEMM_TN4"] = "0"
    res["tn8"] = t(lambda: K.gemm_tn(dy, x, out=g, accumulate=True))
    del os.environ["LMOD_GEMM_TN4"]
    res["nt+T"] = res["nt"] + res["T(dy)"] + res["T(x)"]
    print(tag, {k: f"{v:.3f}ms" + (f" {fl / v / 1e9:.0f}TF" if not k.startswith("T(") else "") for k, v in res.items()}, flush=True)
# MoE experts: 4 experts, capacity slabs, routed rows per expert as in the step (top-2 of 4: 65536 rows over 4 experts, uneven)
E, C = 4, 24576
rows = torch.tensor([9000, 12500, 20036, 24000], device="cuda", dtype=torch.int32)
kv = torch.clamp((rows + 63) & -64, max=C)
for tag, M, N in [("moe gate+up", 11008, 2048), ("moe down", 2048, 5504)]:
    dy = torch.randn(E, C, M, device="cuda").to(torch.bfloat16); x = torch.randn(E, C, N, device="cuda").to(torch.bfloat16)
    g = torch.zeros(E, M, N, device="cuda", dtype=torch.float32)
    fl = 2.0 * float(rows.sum()) * M * N
    dyt, xt = K.transpose(dy, r_valid=rows), K.transpose(x, r_valid=rows)
    res = {"T(dy)": t(lambda: K.transpose(dy, r_valid=rows)), "T(x)": t(lambda: K.transpose(x, r_valid=rows)),
           "nt": t(lambda: K.gemm_nt(dyt, xt, out=g, out_f32=True, accumulate=True, k_valid=kv)),
           "tn4": t(lambda: K.gemm_tn(dy, x, out=g, accumulate=True, k_valid=rows))}
    res["nt+T"] = res["nt"] + res["T(dy)"] + res["T(x)"]
    print(tag, {k: f"{v:.3f}ms" + (f" {fl / v / 1e9:.0f}TF" if not k.startswith("T(") else "") for k, v in res.items()}, flush=True)
