#!/usr/bin/env python3
"""Index arithmetic of gemm4t_kernel's LDS image, replayed on the CPU: the LDS-DMA staging map (gemm.hip, gemm4t_kernel: voA / voB and the
M0 bases) followed by the transposing fragment reads of gemm4t_loop_asm.h must hand lane (g, li) of wave (wr, *) the MFMA operand
fragment  A[k = 32 kk + 8 g + e][m = 128 wr + 16 f + li], e = 0..7,  and no half-wave read may touch one LDS bank twice.
ds_read_b64_tr_b16 as gemm_256_kernel<2> uses it: inside a group of 16 lanes, lane i reads 8 bytes = row i >> 2, columns 4 (i & 3) .. +3 of
a 4 x 16 block and receives column i (4 values)."""
import re, os, sys
import numpy as np

SUB, PIECE, OPB, STAGE = 8448, 1056, 33792, 67584
K, M = 64, 256
A = np.arange(K * M, dtype=np.int64).reshape(K, M)          # element id = k * 256 + m
lds = np.full(STAGE // 2, -1, dtype=np.int64)
for wave in range(4):
    for j in range(8):
        for lane in range(64):
            lr, col = lane >> 3, wave * 64 + (lane & 7) * 8
            k = 8 * ((j >> 2) + 2 * (lr >> 1)) + (j & 3) + 4 * (lr & 1)
            dst = (wave * SUB + j * PIECE + lane * 16) // 2
            assert (lds[dst:dst + 8] == -1).all()
            lds[dst:dst + 8] = A[k, col:col + 8]
hdr = open(os.path.join(os.path.dirname(__file__), "..", "..", "llava-mod_amd", "csrc", "gemm4t_loop_asm.h")).read()
reads = re.findall(r"ds_read_b64_tr_b16 v\[(\d+):\d+\], %\[r([ab])([01])\] offset:(\d+)", hdr)
assert len(reads) == 32 + 64
seen = set()
for v, kind, reg, imm in reads:
    v, imm = int(v), int(imm)
    if kind != "a":
        base = 192
    else:
        base = 128
    ks, rest = divmod(v - base, 32)
    f, h = rest // 4, (rest % 4) // 2
    seen.add((kind, ks, f, h))
    for wr in range(2):
        addr = np.zeros(64, dtype=np.int64)
        for lane in range(64):
            g, li = lane >> 4, lane & 15
            lbase = ((li >> 2) + 4 * (g & 1)) * PIECE + (g >> 1) * 256 + (li & 3) * 8
            addr[lane] = (2 * wr) * SUB + lbase + imm
        for half in range(2):                       # bank check: 32 lanes x 8 bytes on 64 banks of 4 bytes
            banks = []
            for lane in range(half * 32, half * 32 + 32):
                banks += [(addr[lane] // 4) % 64, (addr[lane] // 4 + 1) % 64]
            assert len(set(banks)) == 64, (v, half)
        for lane in range(64):
            g, li = lane >> 4, lane & 15
            got = [lds[addr[(lane & 48) + rr * 4 + (li >> 2)] // 2 + (li & 3)] for rr in range(4)]
            want = [A[32 * ks + 8 * g + 4 * h + rr, 128 * wr + 16 * f + li] for rr in range(4)]
            assert got == want, (v, wr, lane, got, want)
assert len(seen) == 64, len(seen)
print("gemm4t layout: staging map, 96 transposing reads and bank spread check out")
