#!/usr/bin/env python3
"""Writes llava-mod_amd/csrc/gemm4_loop_asm.h: the K loop of gemm4_kernel (256x256x64 tile, 4 waves of 128x128, one wave per SIMD) as
ONE hand-placed inline-asm statement — VERDICT r03 next #1.  hipcc schedules nothing inside it: every instruction sits behind the
MFMA this file names, at most two cheap instructions in any gap, no s_nop, no select, waits placed by hand.

The statement is generated from a SCHEDULE: for each of the 128 MFMAs of a K tile (k-step 0: 0..63, k-step 1: 64..127) the list of
instructions issued right after it.  `--variant` picks the schedule:
  b2    the round-3 loop's order (two barriers per tile), cleaned: M0 set one MFMA ahead of its LDS-DMA (the MFMA is the wait state),
        address toggles and scalar bookkeeping in MFMA shadows
  b3    three barriers (A half and B half of the stage freed separately, LDS-DMA of A starts at MFMA 23)
  even  b2 with the 16 LDS-DMA pieces at a constant stride over the free span
Registers: the 64 accumulator tiles, the 32 operand fragments and the LDS / global offsets are asm OPERANDS (hipcc allocates
them; the statement is the whole loop, so nothing else is live but what the epilogue needs); the two buffer descriptors are
assembled into fixed SGPRs s[84:91] (clobbered) because inline asm cannot name a sub-register of a 128-bit operand and the
"tile past the end" switch is a scalar write of num_records.

C++ side (gemm.hip) provides, in scope of G4_ASM_LOOP():
  f32x4 acc[8][8]; bf16x8 fa[2][8], fb[2][8]; uint32_t voA[8], voB[8];
  uint32_t g4_ra0, g4_ra1, g4_rb0, g4_rb1   LDS byte addresses of this lane's fragment rows: *1 = stage of tile 0, k-half 1;
                                            *0 = stage of tile 1, k-half 0 (XORed with the stage size every tile)
  uint32_t g4_nk (tiles left), g4_koff (byte offset of tile t+2 along K), g4_klim (K bytes), g4_dma (LDS byte address of this
  wave's first piece in the stage of tile t+2), g4_dA[4], g4_dB[4] (descriptor words)
"""
import argparse
import os

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="b2")
ap.add_argument("--order", default="mn", help="mn: 8 consecutive MFMAs share the A-side fragment (src1), round 3's order; nm: they share the "
                                          "B-side fragment (src0), the order of hipBLASLt's kernel")
ap.add_argument("--split-barrier", type=int, default=0, help="1: the wait and its s_barrier one MFMA apart")
ap.add_argument("--peel", type=int, default=0, help="1: the first K tile is a copy of the loop body whose first k-step multiplies into C = 0 (no 256 "
                                                    "v_accvgpr_write per output tile).  NOT shipped: round 4 found no way to tell hipcc — as write-only "
                                                    "'=&a' operands, or '+a' over uninitialised values, the accumulators lose their home registers across "
                                                    "the persistent tile loop (104 ... 858 spilled VGPRs)")
ap.add_argument("--persist", type=int, default=0, help="1: also emit G4_ASM_LOOP_P(): when the look-ahead LDS-DMA runs off the end of the K "
                                                       "range it SWITCHES to the next output tile's operand windows (descriptor words dAn / dBn, offset 0) "
                                                       "instead of fetching nothing: a persistent workgroup's operand stream never stops")
ap.add_argument("--suffix", default="", help="appended to the macro names (a second schedule beside the default one)")
ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llava-mod_amd", "csrc",
                                              "gemm4_loop_asm.h"))
args = ap.parse_args()
V = args.variant
STAGE = 0x10000            # XOR layout: 64 KiB per stage, A rows then B rows (+32768), fragment i of an operand 2048 bytes on

sched = {i: [] for i in range(128)}      # MFMA index -> instructions behind it


def mfma(i):
    ks, idx = i // 64, i % 64
    mt, nt = (idx >> 3, idx & 7) if args.order == "mn" else (idx & 7, idx >> 3)
    return f"v_mfma_f32_16x16x32_bf16 %[c{mt}_{nt}], %[b{ks}_{nt}], %[a{ks}_{mt}], %[c{mt}_{nt}]"


def rd(frag, kind, ks):       # fragment `frag` of operand kind ('a'/'b') for k-step ks: k-step 1 reads THIS tile's stage, 0 the next's
    return f"ds_read_b128 %[{kind}{ks}_{frag}], %[r{kind}{ks}] offset:{frag * 2048}"


def piece(at, pc):            # LDS-DMA piece pc (0-7 A, 8-15 B) of tile t+2 behind MFMA `at`; M0 one MFMA earlier
    j = pc & 7
    isb = pc >= 8
    sched[at - 1].append(f"s_add_u32 m0, %[dma], {j * 4096 + (32768 if isb else 0)}")
    sched[at].append(f"buffer_load_dwordx4 %[v{'b' if isb else 'a'}{j}], {'s[88:91]' if isb else 's[84:87]'}, %[koff] offen lds")


if V in ("b2", "even"):
    for i in range(0, 16, 2):
        sched[i].append(rd(i // 2, "a", 1))
    for i in range(16, 32, 2):
        sched[i].append(rd((i - 16) // 2, "b", 1))
    sched[38] += ["s_waitcnt lgkmcnt(0)", "s_barrier"]                 # every wave holds all of tile t: its stage is free
    if V == "b2":
        at = [39, 44, 49, 54, 59, 64, 69, 74, 79, 84] + [89, 95, 101, 107, 113, 119]
    else:
        at = [39 + round(k * 5.4) for k in range(16)]                   # 39 .. 120
        at = [a + 1 if a == 86 else a for a in at]                      # keep the barrier's own gap clear
    for pc, a in enumerate(at):
        piece(a, pc)
    n_before = sum(1 for a in at if a <= 86)
    sched[86] += [f"s_waitcnt vmcnt({n_before})", "s_barrier"]          # all of tile t+1 has landed, for everyone
    f = 0
    for i in range(87, 111):
        if (i - 87) % 3 != 2:
            sched[i].append(rd(f, "a", 0) if f < 8 else rd(f - 8, "b", 0))
            f += 1
    assert f == 16
elif V == "b3":
    # A fragments of k-step 1 behind MFMAs 0..14; barrier 1 at 21 frees the A half; A pieces from 23 every 3; B fragments
    # interleaved (24..45); barrier 2 at 51 frees the B half; B pieces; vmcnt + barrier 3 at 91; next tile's k-step-0 fragments
    for i in range(0, 16, 2):
        sched[i].append(rd(i // 2, "a", 1))
    sched[21] += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    for k in range(5):
        piece(23 + 3 * k, k)                                            # 23 26 29 32 35
    for k in range(8):
        sched[24 + 3 * k if k < 5 else 38 + 2 * (k - 5)].append(rd(k, "b", 1))     # 24 27 30 33 36 38 40 42
    sched[51] += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    for k, a in enumerate([53, 56, 59]):
        piece(a, 5 + k)                                                 # A 5-7
    for k, a in enumerate([62, 65, 86, 88, 90]):
        piece(a, 8 + k)                                                 # B 0-4
    sched[91] += ["s_waitcnt vmcnt(13)", "s_barrier"]
    f = 0
    for i in [93, 94, 95, 97, 98, 100, 101, 102, 104, 105, 107, 109, 111, 113, 115, 117]:
        sched[i].append(rd(f, "a", 0) if f < 8 else rd(f - 8, "b", 0))
        f += 1
    for k, a in enumerate([96, 103, 112]):
        piece(a, 13 + k)                                                # B 5-7
else:
    raise SystemExit("unknown variant")

# address toggles (after the last read through each register) and scalar bookkeeping (after the last piece)
last_piece = max(i for i in range(128) if any("buffer_load" in s for s in sched[i]))
last_r1 = max(i for i in range(128) if any("%[ra1]" in s or "%[rb1]" in s for s in sched[i]))
last_r0 = max(i for i in range(128) if any("%[ra0]" in s or "%[rb0]" in s for s in sched[i]))
assert last_piece <= 121 and last_r0 <= 118


def free_slot(start, n):
    out, i = [], start
    while len(out) < n:
        if len(sched[i]) == 0:
            out.append(i)
        i += 1
        assert i < 128
    return out


s1 = free_slot(last_r1 + 1, 2)
sched[s1[0]].append(f"v_xor_b32 %[ra1], {STAGE}, %[ra1]")
sched[s1[1]].append(f"v_xor_b32 %[rb1], {STAGE}, %[rb1]")
s0 = free_slot(last_r0 + 1, 2)
sched[s0[0]].append(f"v_xor_b32 %[ra0], {STAGE}, %[ra0]")
sched[s0[1]].append(f"v_xor_b32 %[rb0], {STAGE}, %[rb0]")
import copy
base_sched = copy.deepcopy(sched)


def build(persist):
    global sched
    sched = copy.deepcopy(base_sched)
    if not persist:
        tail = free_slot(max(last_piece + 1, s0[1] + 1), 3)
        sched[tail[0]] += [f"s_xor_b32 %[dma], %[dma], {STAGE}", "s_add_u32 %[koff], %[koff], 128"]
        sched[tail[1]] += ["s_cmp_lt_u32 %[koff], %[klim]", "s_cselect_b32 s86, %[dA2], 0"]    # tile t+3 past the end: zero records
        sched[tail[2]] += ["s_cselect_b32 s90, %[dB2], 0", "s_sub_u32 %[nk], %[nk], 1"]
        assert tail[2] <= 126
        entry = ["s_cmp_lt_u32 %[koff], %[klim]", "s_cselect_b32 s86, %[dA2], 0", "s_cselect_b32 s90, %[dB2], 0"]
    else:
        # look-ahead past the end of K: the DMA window moves to the NEXT tile (offset 0).  SCC = "still inside this tile"
        sw = ["s_cmp_lt_u32 %[koff], %[klim]", "s_cselect_b32 %[koff], %[koff], 0",
              "s_cselect_b32 s84, s84, %[dAn0]", "s_cselect_b32 s85, s85, %[dAn1]", "s_cselect_b32 s86, s86, %[dAn2]",
              "s_cselect_b32 s88, s88, %[dBn0]", "s_cselect_b32 s89, s89, %[dBn1]", "s_cselect_b32 s90, s90, %[dBn2]"]
        body = [f"s_xor_b32 %[dma], %[dma], {STAGE}", "s_sub_u32 %[nk], %[nk], 1", "s_add_u32 %[koff], %[koff], 128"] + sw
        tail = free_slot(max(last_piece + 1, s0[1] + 1), (len(body) + 1) // 2)
        assert tail[-1] <= 125, tail
        for k, ins in enumerate(body):
            sched[tail[k // 2]].append(ins)
        entry = sw
    sched[126] += ["s_cmp_lg_u32 %[nk], 0"]
    sched[127] += ["s_waitcnt lgkmcnt(0)", "s_cbranch_scc1 1b"]
    for i in range(128):
        assert len(sched[i]) <= 2 or any("s_barrier" in x for x in sched[i]), (i, sched[i])
    if args.split_barrier:
        for i in [j for j in range(127) if "s_barrier" in sched[j]]:
            k = sched[i].index("s_barrier")
            sched[i] = sched[i][:k] + sched[i][k + 1:]        # (an M0 write behind it stays: one MFMA ahead of its LDS-DMA)
            sched[i + 1] = ["s_barrier"] + sched[i + 1]
    for i in range(128):                          # an M0 write must sit exactly one MFMA ahead of its LDS-DMA
        for ins in sched[i]:
            if ins.startswith("buffer_load"):
                assert any(x.startswith("s_add_u32 m0") for x in sched[i - 1]), i
    lines = ["s_mov_b32 s84, %[dA0]", "s_mov_b32 s85, %[dA1]", "s_mov_b32 s86, %[dA2]", "s_mov_b32 s87, %[dA3]",
             "s_mov_b32 s88, %[dB0]", "s_mov_b32 s89, %[dB1]", "s_mov_b32 s90, %[dB2]", "s_mov_b32 s91, %[dB3]"] + entry + \
            ["s_waitcnt lgkmcnt(0)"]
    if args.peel:
        # the first K tile, peeled: same placement, k-step 0 accumulates into the constant 0, the closing branch leaves the
        # statement when it was the only tile and falls into the loop otherwise
        for i in range(128):
            m = mfma(i)
            if i < 64:
                m = m[:m.rindex(",")] + ", 0"
            lines.append(m)
            for x in sched[i]:
                lines.append("s_cbranch_scc0 2f" if x == "s_cbranch_scc1 1b" else x)
    lines.append("1:")
    n_head = len(lines) - (128 if args.peel else 0)
    for i in range(128):
        lines.append(mfma(i))
        lines += sched[i]
    if args.peel:
        lines.append("2:")
    outs, ins = [], []
    for mt in range(8):
        for nt in range(8):
            # "+a" also with the peel (whose first k-step ignores the incoming value): as write-only "=&a" operands hipcc loses the
            # accumulators' home registers across a persistent tile loop and spills > 100 VGPRs; the C++ side simply leaves acc
            # uninitialised on this path, so no v_accvgpr_write is emitted either way
            outs.append(f'[c{mt}_{nt}] "+a"(acc[{mt}][{nt}])')
    for i in range(8):
        outs.append(f'[a0_{i}] "+v"(fa[0][{i}])')
    for i in range(8):
        outs.append(f'[b0_{i}] "+v"(fb[0][{i}])')
    for i in range(8):
        outs.append(f'[a1_{i}] "=&v"(fa[1][{i}])')
    for i in range(8):
        outs.append(f'[b1_{i}] "=&v"(fb[1][{i}])')
    for r in ("ra0", "ra1", "rb0", "rb1"):
        outs.append(f'[{r}] "+v"(g4_{r})')
    for r in ("nk", "koff", "dma"):
        outs.append(f'[{r}] "+s"(g4_{r})')
    for i in range(8):
        ins.append(f'[va{i}] "v"(voA[{i}])')
    for i in range(8):
        ins.append(f'[vb{i}] "v"(voB[{i}])')
    ins.append('[klim] "s"(g4_klim)')
    for i in range(4):
        ins.append(f'[dA{i}] "s"(g4_dA[{i}])')
    for i in range(4):
        ins.append(f'[dB{i}] "s"(g4_dB[{i}])')
    if persist:
        for i in range(3):
            ins.append(f'[dAn{i}] "s"(g4_dAn[{i}])')
        for i in range(3):
            ins.append(f'[dBn{i}] "s"(g4_dBn[{i}])')
    clob = ['"memory"', '"scc"'] + [f'"s{i}"' for i in range(84, 92)]
    n_dma = sum(1 for l in lines if "buffer_load" in l)
    n_rd = sum(1 for l in lines if "ds_read" in l)
    n_mf = sum(1 for l in lines if "v_mfma" in l)
    mul = 2 if args.peel else 1
    assert (n_dma, n_rd, n_mf) == (16 * mul, 32 * mul, 128 * mul), (n_dma, n_rd, n_mf)
    n_dma, n_rd, n_mf = n_dma // mul, n_rd // mul, n_mf // mul
    name = ("G4_ASM_LOOP_P" if persist else "G4_ASM_LOOP") + args.suffix
    H = [f"// {name}: per K tile {n_mf} MFMAs, {n_rd} ds_read_b128, {n_dma} LDS-DMA pieces, {sum(1 for l in lines if 's_barrier' in l) // mul} barriers, "
         f"{(len(lines) - n_head - n_mf * mul) // mul} other instructions" + (", first K tile peeled (C = 0)" if args.peel else "") + (" (look-ahead switches to the next tile's operand windows)" if persist else ""),
         f"#define {name}() asm volatile( \\"]
    for l in lines:
        H.append(f'    "{l}\\n\\t" \\')
    H.append("    : " + ", ".join(outs) + " \\")
    H.append("    : " + ", ".join(ins) + " \\")
    H.append("    : " + ", ".join(clob) + ")")
    return H


H = [f"// GENERATED by tools/gen_gemm4_loop.py --variant {V} --order {args.order} --split-barrier {args.split_barrier} --peel {args.peel} --persist {args.persist} — do not edit.",
     "// The K loop of gemm4_kernel as one inline-asm statement.", "#pragma once"] + \
    ([f'#define G4_ASM_VARIANT "{V}"', f"#define G4_ASM_PEEL {args.peel}"] if not args.suffix else [f"#define G4_ASM_HAVE{args.suffix} 1"])
H += build(False)
if args.persist:
    H += build(True)
open(args.out, "w").write("\n".join(H) + "\n")
print("wrote", args.out, "variant", V)
