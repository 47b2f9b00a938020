"""Per-wave cycle timeline of one K tile of gemm_256_kernel<0> (a -DG256_TRACE=1 build under alt_libs/, see gemm.hip):
stamps 0..7 of each phase = phase start | ds_reads issued | LDS-DMA issued | counted vmcnt wait done | (barrier 1) MFMA block
entered | lgkmcnt(0) done | 16 MFMAs issued | (barrier 2) phase end.   python tools/gemm_trace.py liblmod_trace.so [...]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M, N, Kd = 32768, 12288, 4096
a = torch.randn(M, Kd, device="cuda").to(torch.bfloat16); b = torch.randn(N, Kd, device="cuda").to(torch.bfloat16)
o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
P, I, Q = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
NAMES = ["start", "reads", "dma", "vmwait", "bar1", "lgkm", "mfma", "bar2"]
for libname in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.join(ROOT, "alt_libs", libname))
    f = lib.lmod_gemm_bf16_nt
    f.restype = I
    f.argtypes = [P, P, P, P, I, I, I, I, I, I, I, Q, Q, Q, P, P, I, I, I, P]
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(30):
        assert f(a.data_ptr(), b.data_ptr(), o.data_ptr(), None, M, N, Kd, Kd, Kd, N, 1, 0, 0, 0, None, None, 0, 0, 0, s) == 0
    torch.cuda.synchronize()
    buf = np.zeros(8 * 64, dtype=np.uint32)
    assert lib.lmod_debug_gemm_trace(buf.ctypes.data_as(P)) == 0
    st = buf.reshape(8, 64)[:, :32].astype(np.int64)
    base = st.min()
    st = (st - base) % (1 << 32)
    print(f"== {libname}: cycles relative to the earliest stamp of the traced K tile (one workgroup, 8 waves; wr = wave>>2)")
    print("wave | " + " | ".join(f"P{p + 1}.{NAMES[k]}" for p in range(4) for k in range(8)))
    for w in range(8):
        print(f"{w} (wr{w >> 2}) | " + " | ".join(str(int(st[w, i])) for i in range(32)))
    tile = st[:, 31].max() - st[:, 0].min()
    print(f"K tile span {int(tile)} cycles (2048 = MFMA-bound at 2 waves/SIMD x 64 MFMAs x 16 cycles)")
    for w in (0, 4):
        seg = []
        for p in range(4):
            r = st[w, p * 8:(p + 1) * 8]
            seg.append(f"P{p + 1}: reads {r[1] - r[0]}, dma {r[2] - r[1]}, vmwait {r[3] - r[2]}, bar1 {r[4] - r[3]}, lgkm {r[5] - r[4]}, mfma {r[6] - r[5]}, bar2 {r[7] - r[6]}")
        print(f"wave {w}: " + " || ".join(seg))
