"""Can an HBM-bound row kernel run under the MFMA-bound GEMM from a second stream?  serial vs two-stream wall time."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llava-mod_amd"))
from llavamod import kernels as K
M, N, Kd = 32768, 4096, 4096
a = torch.randn(M, Kd, device="cuda").to(torch.bfloat16); b = torch.randn(N, Kd, device="cuda").to(torch.bfloat16)
o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
x = torch.randn(32768, 5504, device="cuda").to(torch.bfloat16); xt = torch.empty(5504, 32768, device="cuda", dtype=torch.bfloat16)
h = torch.randn(32768, 2048, device="cuda").to(torch.bfloat16); w = torch.ones(2048, device="cuda", dtype=torch.bfloat16)
side = torch.cuda.Stream()
def gemms(n=8):
    for _ in range(n): K.gemm_nt(a, b, out=o)
def rows(n=8):
    for _ in range(n):
        K.transpose(x, out=xt); K.rmsnorm_fwd(h, w, 1e-6)
def wall(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
def both_serial(): gemms(); rows()
def both_overlap():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): rows()
    gemms()
    torch.cuda.current_stream().wait_stream(side)
def interleaved_overlap():
    side.wait_stream(torch.cuda.current_stream())
    for _ in range(8):
        K.gemm_nt(a, b, out=o)
        with torch.cuda.stream(side):
            K.transpose(x, out=xt); K.rmsnorm_fwd(h, w, 1e-6)
    torch.cuda.current_stream().wait_stream(side)
print("gemms only", wall(gemms), "rows only", wall(rows), "serial", wall(both_serial), "two streams", wall(both_overlap),
      "two streams interleaved", wall(interleaved_overlap))

def event_chained():
    # row kernels become ready exactly when a GEMM finishes and the next GEMM starts (the AdamW-behind-wgrad pattern)
    main = torch.cuda.current_stream()
    for _ in range(8):
        K.gemm_nt(a, b, out=o)
        ev = torch.cuda.Event(); ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            K.transpose(x, out=xt); K.rmsnorm_fwd(h, w, 1e-6)
    K.gemm_nt(a, b, out=o)
    main.wait_stream(side)
def nine_gemms(): gemms(9)
print("9 gemms", wall(nine_gemms), "9 gemms + event-chained rows", wall(event_chained))
