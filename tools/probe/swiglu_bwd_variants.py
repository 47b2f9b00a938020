"""Fused SwiGLU-backward GEMM at the student's dense shape: prints a hash of the result and the time.  Run once per routing
(default: 8-wave MODE 4;  LMOD_GEMM_WAVES=44: gemm4_kernel<4>, persistent when LMOD_GEMM_PERSIST != 0) and compare the hashes."""
import hashlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K
torch.manual_seed(3)
for (M, H, I) in [(32768, 2048, 5504), (32768, 4096, 11008), (8192 + 40, 2048, 5504)]:
    dy = torch.randn(M, H, device="cuda").to(torch.bfloat16)
    wt = (torch.randn(I, H, device="cuda") * 0.02).to(torch.bfloat16)
    gu = torch.randn(M, 2 * I, device="cuda").to(torch.bfloat16)
    out = K.gemm_swiglu_bwd(dy, wt, gu)
    torch.cuda.synchronize()
    h = hashlib.sha256(out.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(10): K.gemm_swiglu_bwd(dy, wt, gu, out=out)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print((M, H, I), h, f"{best:.4f} ms {2.0 * M * H * I / best / 1e9:.0f} TF", flush=True)
