"""Life of ONE attention-forward workgroup in time stamps (trace build: SRC=attn_fwd2 AGPRS=64 WPE=2 tools/build_b2_variants.sh
f2trace:"-DF2_TRACE=1"; run with LMOD_HIP_LIB=alt_libs/liblmod_f2trace.so).  s_memtime ticks are 10 ns.  Per pass of the causal
pair: start -> first tile staged -> every iteration's barrier -> drain done -> output stored -> inter-pass barrier.

    LMOD_HIP_LIB=$PWD/alt_libs/liblmod_f2trace.so python tools/attn_trace.py [B] [S] [nh]
"""
import ctypes, math, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
nh = int(sys.argv[3]) if len(sys.argv) > 3 else 16
hd = 128
lib = ctypes.CDLL(os.environ["LMOD_HIP_LIB"])
lib.lmod_debug_attn_trace.restype = ctypes.c_int
lib.lmod_debug_attn_trace.argtypes = [ctypes.c_void_p]
qkv = torch.randn(B * S, 3 * nh * hd, device="cuda").to(torch.bfloat16)
q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:2 * nh * hd], qkv[:, 2 * nh * hd:]
for _ in range(20):
    K.attn_fwd(q, k, v, B, S, nh, nh, hd, 1 / math.sqrt(hd), True)
torch.cuda.synchronize()
buf = np.zeros(8 * 128, dtype=np.uint64)
assert lib.lmod_debug_attn_trace(buf.ctypes.data) == 0
t = buf.reshape(8, 128).astype(np.int64)
t0 = t[:, 0].min()
us = lambda x: (x - t0) * 0.01
for w in (0, 3, 7):
    print(f"wave {w}")
    for ps in range(2):
        b = ps * 60
        if t[w, b] == 0:
            continue
        its = [j for j in range(2, 57) if t[w, b + j] > 0]
        st = [us(t[w, b + j]) for j in its]
        d = np.diff([us(t[w, b + 1])] + st)
        print(f"  pass {ps}: start {us(t[w, b]):8.2f}  staged {us(t[w, b + 1]):8.2f} (+{us(t[w, b + 1]) - us(t[w, b]):.2f})  "
              f"{len(its)} iterations, last barrier {st[-1]:8.2f}  drain {us(t[w, b + 57]):8.2f} (+{us(t[w, b + 57]) - st[-1]:.2f})  "
              f"stored {us(t[w, b + 58]):8.2f} (+{us(t[w, b + 58]) - us(t[w, b + 57]):.2f})  sync {us(t[w, b + 59]):8.2f} (+{us(t[w, b + 59]) - us(t[w, b + 58]):.2f})")
        print("    iteration us: " + " ".join(f"{x:.2f}" for x in d))
