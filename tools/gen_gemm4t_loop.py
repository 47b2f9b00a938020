#!/usr/bin/env python3
"""Writes llava-mod_amd/csrc/gemm4t_loop_asm.h: the K loop of gemm4t_kernel — the weight-gradient form C += A^T B with BOTH operands
reduction-major as autograd holds them (A = dY [K, M], B = X [K, N], K = tokens), on the 4-wave 256x256x64 tile of gemm4_kernel.  No
transposed copies: the MFMA operand fragments (8 consecutive k for one m per lane) come out of a [k][m] LDS image through
ds_read_b64_tr_b16 (hardware transpose read).  Same placement rules as tools/gen_gemm4_loop.py (every instruction behind a named MFMA, at
most two per gap, waits by hand); what differs:

  LDS image   per stage and operand 4 sub-images of 64 columns; a sub-image is 8 LDS-DMA pieces of 8 k-rows x 128 bytes, 32 bytes of padding
              behind every piece (1056-byte pitch).  k-row 8q + u sits in piece (u & 3) + 4 (q & 1), row (u >> 2) + 2 (q >> 1): the 8 rows
              a half-wave's transposing read touches ({k..k+3} x {q, q+1}) are the same row of 8 different pieces, i.e. 8 x 32 bytes on
              64 distinct banks — conflict-free WITHOUT an XOR, so every fragment address is  lane base + immediate  (an XOR swizzle on
              the block index would need one address register per fragment).
  fragments   fixed physical registers (clobbered), because inline asm cannot name the halves of a 128-bit operand and a fragment is
              filled by two 64-bit reads: A k-step s fragment i = v[128 + 32 s + 4 i +: 4], B = v[192 + 32 s + 4 i +: 4].  The first
              fragments of the tile are read inside the statement.
  K advance   the descriptors walk: base += 64 rows, num_records -= the same (clamped at 0) once per K tile — a reduction-major operand's
              K edge is a ROW boundary, so any K (and a per-batch k_valid) is exact with no per-lane mask.
  toggles     stages are 67584 bytes apart (not a power of two): address registers flip by  x = (lo + hi) - x.

C++ side (gemm.hip, gemm4t_kernel) provides f32x4 acc[8][8]; uint32_t voA[8], voB[8], g4_ra0/ra1/rb0/rb1 (this lane's fragment base in
the stage of K tile 1 / 0), g4_sa / g4_sb (their sums), g4_nk, g4_dma (+ g4_dsum), g4_ksA / g4_ksB (bytes per 64 rows), g4_dA[4] / g4_dB[4]
(descriptors positioned at K tile 2).
"""
import argparse
import os

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="b2", help="b2: round 4's schedule (fragment reads one per MFMA in two bursts, barriers at 38 and 86); "
                                                "ob: ONE barrier per K tile (at MFMA 56) and the 64 transposing reads at two per three MFMAs instead of one per "
                                                "MFMA in two bursts (measured -3 ... -4 %: the bursts are not the limiter)")
ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llava-mod_amd", "csrc",
                                              "gemm4t_loop_asm.h"))
args = ap.parse_args()

SUB, PIECE, OPB = 8448, 1056, 33792       # sub-image, piece pitch, B operand offset inside a stage (stage = 67584)

sched = {i: [] for i in range(128)}


def mfma(i):
    ks, idx = i // 64, i % 64
    mt, nt = idx >> 3, idx & 7
    a = 128 + 32 * ks + 4 * mt
    b = 192 + 32 * ks + 4 * nt
    return f"v_mfma_f32_16x16x32_bf16 %[c{mt}_{nt}], v[{b}:{b + 3}], v[{a}:{a + 3}], %[c{mt}_{nt}]"


def rd(n, ks, reg):          # read n (0..31) of k-step ks: A fragments 0-7 (two halves each), then B; through address register set `reg`
    kind = "a" if n < 16 else "b"
    f, h = (n & 15) >> 1, n & 1
    v = (128 if kind == "a" else 192) + 32 * ks + 4 * f + 2 * h
    imm = (f >> 2) * SUB + (f & 3) * 32 + ks * 512 + h * 128
    return f"ds_read_b64_tr_b16 v[{v}:{v + 1}], %[r{kind}{reg}] offset:{imm}"


def piece(at, pc):
    j, isb = pc & 7, pc >= 8
    sched[at - 1].append(f"s_add_u32 m0, %[dma], {j * PIECE + (OPB if isb else 0)}")
    sched[at].append(f"buffer_load_dwordx4 %[v{'b' if isb else 'a'}{j}], {'s[88:91]' if isb else 's[84:87]'}, 0 offen lds")


def place(start, instrs, cap=2):          # in order, into the gaps from `start` on
    i = start
    for ins in instrs:
        while len(sched[i]) >= cap:
            i += 1
        sched[i].append(ins)
    return i


if args.variant == "ob":
    # B fragments first (all eight are needed by the first eight MFMAs of k-step 1), then A (fragment mt is needed at MFMA 64 + 8 mt)
    order1 = list(range(16, 32)) + list(range(0, 16))
    slots = [i for i in range(0, 48) if i % 3 != 2]
    assert len(slots) == 32
    for n, i in zip(order1, slots):
        sched[i].append(rd(n, 1, 1))
    place(48, ["v_sub_u32 %[ra1], %[sa], %[ra1]", "v_sub_u32 %[rb1], %[sb], %[rb1]"], cap=1)
    # ONE barrier: my reads of tile t are done (the last one was issued 9 MFMAs ago), all my pieces of tile t+1 (issued one tile ago)
    # have landed => for everyone: stage t is free, tile t+1 is complete
    sched[56] += ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier"]
    at = [58 + 4 * k for k in range(16)]                                  # 58 .. 118
    for pc, a in enumerate(at):
        piece(a, pc)
    # tile t+1's k-step-0 fragments (register set 0): A first (MFMA i of the next tile's k-step 0 needs A fragment i >> 3 and B fragment i & 7,
    # i.e. all of B within the first eight) — so B first here too, A behind
    order0 = list(range(16, 32)) + list(range(0, 16))
    slots0 = [i for i in range(59, 128) if len(sched[i]) == 0 and len(sched[i + 1] if i + 1 < 128 else []) < 2][:]
    k = 0
    for i in range(59, 122):
        if k == 32:
            break
        if len(sched[i]) >= 1:
            continue
        sched[i].append(rd(order0[k], 0, 0))
        k += 1
    assert k == 32, k
    tail = ["v_sub_u32 %[ra0], %[sa], %[ra0]", "v_sub_u32 %[rb0], %[sb], %[rb0]"]
    end = place(max(i for i in range(128) if any("%[ra0]" in x or "%[rb0]" in x for x in sched[i])) + 1, tail)
else:
    # k-step 0: this tile's k-step-1 fragments (stage of tile t: register set 1), one read per MFMA (four waves x 512 bytes per 16 cycles =
    # 128 of the 256 bytes per clock this read moves: half the LDS array's rate)
    for n in range(32):
        sched[n].append(rd(n, 1, 1))
    place(32, ["v_sub_u32 %[ra1], %[sa], %[ra1]", "v_sub_u32 %[rb1], %[sb], %[rb1]"], cap=1)
    sched[38] += ["s_waitcnt lgkmcnt(0)", "s_barrier"]                      # every wave holds all of tile t: its stage is free
    at = [39, 44, 49, 54, 59, 64, 69, 74, 79, 84] + [89, 94, 99, 104, 109, 114]
    for pc, a in enumerate(at):
        piece(a, pc)
    n_before = sum(1 for a in at if a <= 86)
    sched[86] += [f"s_waitcnt vmcnt({n_before})", "s_barrier"]             # all of tile t+1 has landed, for everyone
    for n in range(32):                                                     # tile t+1's k-step-0 fragments (register set 0)
        assert len(sched[87 + n]) <= 1
        sched[87 + n].append(rd(n, 0, 0))
    tail = ["v_sub_u32 %[ra0], %[sa], %[ra0]", "v_sub_u32 %[rb0], %[sb], %[rb0]"]
    end = place(119, tail)
# descriptors and the LDS-DMA base move on to K tile t+3 (after this tile's last piece)
walk = ["s_sub_u32 %[dma], %[dsum], %[dma]",
        "s_add_u32 s84, s84, %[ksA]", "s_addc_u32 s85, s85, 0", "s_sub_u32 s86, s86, %[ksA]", "s_cselect_b32 s86, 0, s86",
        "s_add_u32 s88, s88, %[ksB]", "s_addc_u32 s89, s89, 0", "s_sub_u32 s90, s90, %[ksB]", "s_cselect_b32 s90, 0, s90",
        "s_sub_u32 %[nk], %[nk], 1"]
end = place(max(at) + 1, walk)
assert end <= 125, end
sched[126] += ["s_cmp_lg_u32 %[nk], 0"]
sched[127] += ["s_waitcnt lgkmcnt(0)", "s_cbranch_scc1 1b"]
for i in range(128):
    assert len(sched[i]) <= 2 or any("s_barrier" in x for x in sched[i]), (i, sched[i])
    for ins in sched[i]:
        if ins.startswith("buffer_load"):
            assert any(x.startswith("s_add_u32 m0") for x in sched[i - 1]), i
# order-sensitive pairs stay in order with nothing that writes SCC between them
flat = [x for i in range(128) for x in sched[i]]
for a, b in (("s_add_u32 s84", "s_addc_u32 s85"), ("s_sub_u32 s86", "s_cselect_b32 s86"), ("s_add_u32 s88", "s_addc_u32 s89"),
             ("s_sub_u32 s90", "s_cselect_b32 s90")):
    ia = next(k for k, x in enumerate(flat) if x.startswith(a))
    ib = next(k for k, x in enumerate(flat) if x.startswith(b))
    assert ib > ia and not any(x.startswith(("s_add", "s_sub", "s_cmp", "s_addc")) for x in flat[ia + 1:ib]), (a, b)
assert flat.index(walk[1]) > max(k for k, x in enumerate(flat) if x.startswith("buffer_load"))

lines = ["s_mov_b32 s84, %[dA0]", "s_mov_b32 s85, %[dA1]", "s_mov_b32 s86, %[dA2]", "s_mov_b32 s87, %[dA3]",
         "s_mov_b32 s88, %[dB0]", "s_mov_b32 s89, %[dB1]", "s_mov_b32 s90, %[dB2]", "s_mov_b32 s91, %[dB3]"]
lines += [rd(n, 0, 1) for n in range(32)]                              # tile 0, k-step 0 (its stage is register set 1's)
lines += ["s_waitcnt lgkmcnt(0)", "1:"]
n_head = len(lines)
for i in range(128):
    lines.append(mfma(i))
    lines += sched[i]
outs, ins = [], []
for mt in range(8):
    for nt in range(8):
        outs.append(f'[c{mt}_{nt}] "+a"(acc[{mt}][{nt}])')
for r in ("ra0", "ra1", "rb0", "rb1"):
    outs.append(f'[{r}] "+v"(g4_{r})')
for r in ("nk", "dma"):
    outs.append(f'[{r}] "+s"(g4_{r})')
for i in range(8):
    ins.append(f'[va{i}] "v"(voA[{i}])')
for i in range(8):
    ins.append(f'[vb{i}] "v"(voB[{i}])')
ins += ['[sa] "v"(g4_sa)', '[sb] "v"(g4_sb)', '[dsum] "s"(g4_dsum)', '[ksA] "s"(g4_ksA)', '[ksB] "s"(g4_ksB)']
for i in range(4):
    ins.append(f'[dA{i}] "s"(g4_dA[{i}])')
for i in range(4):
    ins.append(f'[dB{i}] "s"(g4_dB[{i}])')
clob = ['"memory"', '"scc"'] + [f'"s{i}"' for i in range(84, 92)] + [f'"v{i}"' for i in range(128, 256)]
body = lines[n_head:]
body = lines[n_head:]
n_dma = sum(1 for l in body if "buffer_load" in l)
n_rd = sum(1 for l in body if "ds_read" in l)
n_mf = sum(1 for l in body if "v_mfma" in l)
assert (n_dma, n_rd, n_mf) == (16, 64, 128), (n_dma, n_rd, n_mf)
H = ["// GENERATED by tools/gen_gemm4t_loop.py — do not edit.",
     "// The K loop of gemm4t_kernel (C += A^T B, both operands reduction-major) as one inline-asm statement.", "#pragma once",
     f"// G4T_ASM_LOOP: per K tile {n_mf} MFMAs, {n_rd} ds_read_b64_tr_b16, {n_dma} LDS-DMA pieces, 2 barriers, "
     f"{len(body) - n_mf - n_rd - n_dma - 2} other instructions",
     f"#define G4T_ASM_LOOP{'_OB' if args.variant == 'ob' else ''}() asm volatile( \\"]
for l in lines:
    H.append(f'    "{l}\\n\\t" \\')
H.append("    : " + ", ".join(outs) + " \\")
H.append("    : " + ", ".join(ins) + " \\")
H.append("    : " + ", ".join(clob) + ")")
open(args.out, "w").write("\n".join(H) + "\n")
print("wrote", args.out)
