import sys, torch
sys.path.insert(0, "llava-mod_amd")
from llavamod import kernels as K
BF = torch.bfloat16
def tables(maxpos, hd, theta=1e6):
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2).float() / hd)); fr = torch.outer(torch.arange(maxpos).float(), inv)
    emb = torch.cat((fr, fr), -1); return emb.cos().to(BF).cuda(), emb.sin().to(BF).cuda()
cos, sin = tables(4096, 128)
for name, T, Kd, nh, nkv in (("teacher", 32768, 4096, 32, 32), ("student", 32768, 2048, 16, 16)):
    N = (nh + 2 * nkv) * 128
    x = torch.randn(T, Kd, device="cuda").to(BF); w = (torch.randn(N, Kd, device="cuda") * 0.02).to(BF); b = torch.randn(N, device="cuda").to(BF)
    pos = (torch.arange(T, device="cuda") % 2048).to(torch.int32)
    out = torch.empty(T, N, device="cuda", dtype=BF)
    def unfused():
        K.gemm_nt(x, w, bias=b, out=out); K.rope_(out, cos, sin, pos, nh + nkv, 128)
    def fused():
        K.gemm_qkv_rope(x, w, b, cos, sin, pos, nh + nkv, out=out)
    for rnd in range(2):
        for tag, f in (("gemm + rope", unfused), ("fused", fused)):
            for _ in range(3): f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): f()
            e1.record(); torch.cuda.synchronize()
            print(f"{name} qkv [{T}x{N}x{Kd}] {tag}: {e0.elapsed_time(e1) / 30:.3f} ms", flush=True)
