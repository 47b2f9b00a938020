"""Same-box, same-process A/B of two library builds on the dominant GEMM shape (plain ctypes: old builds lack newer symbols)."""
import ctypes, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = {"current": os.path.join(ROOT, "llava-mod_amd", "llavamod", "_lib", "liblmod_hip.so")}
for f in sorted(os.listdir(os.path.join(ROOT, "alt_libs"))) if os.path.isdir(os.path.join(ROOT, "alt_libs")) else []:
    if f.endswith(".so"): libs[f] = os.path.join(ROOT, "alt_libs", f)
M, N, Kd = 32768, 12288, 4096
a = torch.randn(M, Kd, device="cuda").to(torch.bfloat16); b = torch.randn(N, Kd, device="cuda").to(torch.bfloat16)
o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
fl = 2.0 * M * N * Kd
P, I, Q = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
fns = {}
for name, path in libs.items():
    f = ctypes.CDLL(path).lmod_gemm_bf16_nt
    f.restype = I
    f.argtypes = [P, P, P, P, I, I, I, I, I, I, I, Q, Q, Q, P, P, I, I, I, P]
    fns[name] = f
def run(f, n):
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(n):
        rc = f(a.data_ptr(), b.data_ptr(), o.data_ptr(), None, M, N, Kd, Kd, Kd, N, 1, 0, 0, 0, None, None, 0, 0, 0, s)
        assert rc == 0
for rnd in range(2):
    for name, f in fns.items():
        run(f, 2); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(f, 100); torch.cuda.synchronize()
        print(name, round(fl * 100 / (time.perf_counter() - t0) / 1e12), "TF", flush=True)
