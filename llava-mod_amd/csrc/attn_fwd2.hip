// Flash attention forward for head dim 128 on gfx950 — the decoder's attention core
// (qwen2/modeling_qwen2.py:700-708 SDPA path, :290-309 eager path; causal + right-padding key mask, GQA).
//
// One workgroup = 8 waves = 256 queries of one (batch, head); K/V tiles of 64 keys; 32 queries per wave.
// MFMA 32x32x16 (operands: index = lane&31, 8 consecutive reduction slots at (lane>>5)*8; D: column = lane&31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5)):
//   S^T tile [32 keys x 32 queries] = mfma(A = K rows, B = Q rows)  -> a lane owns ONE query (lane&31) and, per 64-key
//   tile, 32 of its keys; the other 32 sit in lane^32.  Row max = in-lane max3 chain + ONE v_permlane32_swap; the row
//   sum stays a per-lane partial until the epilogue (both lanes of a row always share max and rescale factor).
//   An MFMA's reduction-slot order is free when both operands agree, so the exponentiated S^T accumulators, packed to
//   bf16 in place, ARE the B operand of O^T = mfma(A = V^T rows (features), B = P^T): no LDS round trip, no cross-lane
//   movement.  V^T is never materialised: the row-major V tile in LDS is read with ds_read_b64_tr_b16 (lanes 4r..4r+3
//   of a 16-lane group address key r's 16 features, lane i receives feature i of the 4 keys), addresses chosen so that
//   the MFMA row index maps to a feature permutation under which every lane ends with 16 CONTIGUOUS features of its
//   query per 32-feature strip (32-byte epilogue stores).
// Q fragments live in registers (loaded once per query block).  K tile image: 256-byte rows, 16-byte chunks XOR (key&15)
// -> conflict-free ds_read_b128 operand reads.  V tile image: chunks XOR ((key&3)<<2): the 32 lanes of a transposing
// read cover 4 keys x 64 contiguous bytes in 4 different 64-byte groups -> conflict-free.
// Pipeline (per wave, software-pipelined by one tile; one workgroup barrier per tile):
//   iteration j:  phase A  QK^T(j)          16 MFMAs, K fragments read two MFMAs ahead
//                 phase B  P(j-1)·V(j-1)    16 MFMAs interleaved with the online softmax of tile j (max, deferred
//                                           rescale decision, 32 exp2, bf16 packing): the matrix pipe works on the previous
//                                           tile while the VALU turns this tile's scores into probabilities
//   staging: the registers holding K(j+1), V(j) are written to LDS at the TOP of iteration j (right after the barrier) and the
//   loads of K(j+2), V(j+1) re-issued at once (one register set per tensor).
// Rescale is deferred (threshold F2_THR in log2 units): the running max moves only when some row of the wave grows by
// more than the threshold, so the O-wide multiply is rare; probabilities are then bounded by 2^F2_THR instead of 1.
#include "attn_common.h"
#include <type_traits>

#define F2_TB 16384
#ifndef F2_TRACE
#define F2_TRACE 0                 // debug build (tools/attn_trace.py): s_memtime stamps of ONE workgroup's life into f2_trace_buf
#endif
#if F2_TRACE
#ifndef F2_TRACE_Y
#define F2_TRACE_Y 1
#endif
__device__ unsigned long long f2_trace_buf[8 * 128];
#define F2_STAMP(k) do { if (trace_on && (threadIdx.x & 63) == 0) { unsigned long long t_; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory"); f2_trace_buf[(threadIdx.x >> 6) * 128 + (k)] = t_; } } while (0)
#else
#define F2_STAMP(k) do { } while (0)
#endif
#ifndef F2_THR
#define F2_THR 6.0f
#endif
#ifndef F2_DEPTH
#define F2_DEPTH 3                 // operand fragments are read from LDS this many MFMAs ahead
#endif
#ifndef F2_SCHED
#define F2_SCHED 1
#endif
#ifndef F2_PK
#define F2_PK 0                     // 1: packed fp32 math in the softmax slices (v_pk_fma_f32 / v_pk_add_f32: 224 each instead of 448 scalar
#endif                              //    FMAs / adds).  Measured round 4: 752 -> 505 TF (B16 S2048 causal) — the even-aligned register pairs cost
#if 0                               //    hipcc 34-39 spilled VGPRs and compiler moves through a[0:63] (the ISA audit in tests/test_abi.py fails)
#endif
#ifndef F2_INPIN
#define F2_INPIN 0                  // 1: a VALU slice's input is pinned behind the MFMA it follows in the source (see attn_fwd3.hip F3_INPIN)
#endif
#ifndef F2_DMA
#define F2_DMA 1                    // round 5: K / V tiles go global -> LDS by LDS-DMA (buffer_load ... lds, 1 KiB per wave-instruction): no
#endif                              //   staging registers, no ds_write issue slots, no in-loop wait on a register load (0: the round-2 path —
                                    //   global -> 16 VGPRs -> ds_write_b128 one iteration later; the timing ablation "no staging" read +18 %
                                    //   at B16 S2048 causal, profiles/r05_attn_ablation.md)
#ifndef F2_RING
#define F2_RING 2                   // K / V tile buffers per tensor.  3 (LDS-DMA only): a tile is requested TWO iterations before its first
#endif                              //   read and the end-of-iteration wait leaves the newest request in flight (counted vmcnt) — measured
                                    //   -1 ... -3.5 % against 2 (one iteration ahead, vmcnt(0) in front of the barrier): the requests are not
                                    //   late (profiles/r05_attn_fwd_variants.jsonl), the 96 KiB ring and its address stepping only cost
#define F2_VBASE (F2_RING * F2_TB)  // V buffers follow the K buffers
#ifndef F2_PRIO
#define F2_PRIO 0                   // 1: s_setprio 1 for waves 4-7 (the later-dispatched wave of every SIMD) for the whole block
#endif
#ifndef F2_TOUCH
#define F2_TOUCH 0                  // > 0: L2 touch of the K / V tiles this many iterations ahead of their register loads
#endif

// The statements between two MFMAs are ordinary (movable) code: pinning one result of each slice with an empty volatile asm
// keeps the slice between its two neighbouring asm-volatile MFMAs (hipcc otherwise sank all exponentials below the last one).
#define F2_PIN(x) asm volatile("" : "+v"(x))
#ifndef F2_ABL
#define F2_ABL 0                   // timing ablations (WRONG RESULTS): 1 no barrier, 2 no exponentials, 3 no P·V MFMAs, 4 no QK^T MFMAs,
#endif                             //   5 no global->LDS staging, 6 no operand reads from LDS, 7 no row sum
#define F2_SB() do { if (F2_SCHED) __builtin_amdgcn_sched_barrier(0); } while (0)

// O^T accumulators: a[0:63], OWNED BY INLINE ASM for the lifetime of a query block (strip dt = a[16dt : 16dt+15]).  They are
// never a C++ value: as a variable (builtin MFMA, or an asm "+a" operand) hipcc kept two copies of the 64 registers alive
// across the loop's control-flow merges, shuttled them through v_accvgpr moves and spilled the Q fragments.  Every
// statement that touches them names them as clobbers (which also makes the kernel descriptor allocate them); the file is
// built with -mllvm -amdgpu-spill-vgpr-to-agpr=0 so the compiler has no use of its own for accumulator registers, and
// tests/test_abi.py audits the ISA (no spills, no compiler v_accvgpr_* outside the asm statements).  hipcc does not see
// inside the statements: the hazards MFMA-result -> v_accvgpr_read and v_accvgpr_write -> MFMA are padded by hand.
#define F2_CLOB0 "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15"
#define F2_CLOB1 "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31"
#define F2_CLOB2 "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47"
#define F2_CLOB3 "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63"
#define F2_CLOB_ALL F2_CLOB0, F2_CLOB1, F2_CLOB2, F2_CLOB3
// S^T accumulators stay ordinary VGPR values (the softmax reads them), but their MFMAs are asm statements too, with the
// accumulator registers in the clobber list: no statement-free stretch of the main loop is left in which hipcc could
// decide to park a value of its own in a[0:63].  FIRST: C = 0.
template <bool FIRST>
__device__ __forceinline__ void qk_mfma(f32x16& acc, const bf16x8 a, const bf16x8 b) {
  if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b) : F2_CLOB_ALL);
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : F2_CLOB_ALL);
}
template <int DT>
__device__ __forceinline__ void pv_mfma(const bf16x8 a, const bf16x8 b) {
  if constexpr (DT == 0) asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" ::"v"(a), "v"(b) : F2_CLOB_ALL);
  else if constexpr (DT == 1) asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" ::"v"(a), "v"(b) : F2_CLOB_ALL);
  else if constexpr (DT == 2) asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, a[32:47]" ::"v"(a), "v"(b) : F2_CLOB_ALL);
  else asm volatile("v_mfma_f32_32x32x16_bf16 a[48:63], %0, %1, a[48:63]" ::"v"(a), "v"(b) : F2_CLOB_ALL);
}
__device__ __forceinline__ void acc_zero() {
  asm volatile(
      "v_accvgpr_write_b32 a0, 0\n\t"
      "v_accvgpr_write_b32 a1, 0\n\t"
      "v_accvgpr_write_b32 a2, 0\n\t"
      "v_accvgpr_write_b32 a3, 0\n\t"
      "v_accvgpr_write_b32 a4, 0\n\t"
      "v_accvgpr_write_b32 a5, 0\n\t"
      "v_accvgpr_write_b32 a6, 0\n\t"
      "v_accvgpr_write_b32 a7, 0\n\t"
      "v_accvgpr_write_b32 a8, 0\n\t"
      "v_accvgpr_write_b32 a9, 0\n\t"
      "v_accvgpr_write_b32 a10, 0\n\t"
      "v_accvgpr_write_b32 a11, 0\n\t"
      "v_accvgpr_write_b32 a12, 0\n\t"
      "v_accvgpr_write_b32 a13, 0\n\t"
      "v_accvgpr_write_b32 a14, 0\n\t"
      "v_accvgpr_write_b32 a15, 0\n\t"
      "v_accvgpr_write_b32 a16, 0\n\t"
      "v_accvgpr_write_b32 a17, 0\n\t"
      "v_accvgpr_write_b32 a18, 0\n\t"
      "v_accvgpr_write_b32 a19, 0\n\t"
      "v_accvgpr_write_b32 a20, 0\n\t"
      "v_accvgpr_write_b32 a21, 0\n\t"
      "v_accvgpr_write_b32 a22, 0\n\t"
      "v_accvgpr_write_b32 a23, 0\n\t"
      "v_accvgpr_write_b32 a24, 0\n\t"
      "v_accvgpr_write_b32 a25, 0\n\t"
      "v_accvgpr_write_b32 a26, 0\n\t"
      "v_accvgpr_write_b32 a27, 0\n\t"
      "v_accvgpr_write_b32 a28, 0\n\t"
      "v_accvgpr_write_b32 a29, 0\n\t"
      "v_accvgpr_write_b32 a30, 0\n\t"
      "v_accvgpr_write_b32 a31, 0\n\t"
      "v_accvgpr_write_b32 a32, 0\n\t"
      "v_accvgpr_write_b32 a33, 0\n\t"
      "v_accvgpr_write_b32 a34, 0\n\t"
      "v_accvgpr_write_b32 a35, 0\n\t"
      "v_accvgpr_write_b32 a36, 0\n\t"
      "v_accvgpr_write_b32 a37, 0\n\t"
      "v_accvgpr_write_b32 a38, 0\n\t"
      "v_accvgpr_write_b32 a39, 0\n\t"
      "v_accvgpr_write_b32 a40, 0\n\t"
      "v_accvgpr_write_b32 a41, 0\n\t"
      "v_accvgpr_write_b32 a42, 0\n\t"
      "v_accvgpr_write_b32 a43, 0\n\t"
      "v_accvgpr_write_b32 a44, 0\n\t"
      "v_accvgpr_write_b32 a45, 0\n\t"
      "v_accvgpr_write_b32 a46, 0\n\t"
      "v_accvgpr_write_b32 a47, 0\n\t"
      "v_accvgpr_write_b32 a48, 0\n\t"
      "v_accvgpr_write_b32 a49, 0\n\t"
      "v_accvgpr_write_b32 a50, 0\n\t"
      "v_accvgpr_write_b32 a51, 0\n\t"
      "v_accvgpr_write_b32 a52, 0\n\t"
      "v_accvgpr_write_b32 a53, 0\n\t"
      "v_accvgpr_write_b32 a54, 0\n\t"
      "v_accvgpr_write_b32 a55, 0\n\t"
      "v_accvgpr_write_b32 a56, 0\n\t"
      "v_accvgpr_write_b32 a57, 0\n\t"
      "v_accvgpr_write_b32 a58, 0\n\t"
      "v_accvgpr_write_b32 a59, 0\n\t"
      "v_accvgpr_write_b32 a60, 0\n\t"
      "v_accvgpr_write_b32 a61, 0\n\t"
      "v_accvgpr_write_b32 a62, 0\n\t"
      "v_accvgpr_write_b32 a63, 0\n\t"
      "s_nop 3" ::: F2_CLOB0, F2_CLOB1, F2_CLOB2, F2_CLOB3);
}
// acc *= alpha (per lane).  Entered only after every MFMA of the previous tile has been issued; the leading wait states
// let the last of them write its accumulators back, the trailing ones cover v_accvgpr_write -> MFMA.
__device__ __forceinline__ void acc_scale(const float alpha) {
  float t0, t1;
  asm volatile(
      "s_nop 15\n\ts_nop 15\n\t"
      "v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %1\n\t"
      "v_accvgpr_read_b32 %0, a2\n\tv_accvgpr_read_b32 %1, a3\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a2, %0\n\tv_accvgpr_write_b32 a3, %1\n\t"
      "v_accvgpr_read_b32 %0, a4\n\tv_accvgpr_read_b32 %1, a5\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a4, %0\n\tv_accvgpr_write_b32 a5, %1\n\t"
      "v_accvgpr_read_b32 %0, a6\n\tv_accvgpr_read_b32 %1, a7\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a6, %0\n\tv_accvgpr_write_b32 a7, %1\n\t"
      "v_accvgpr_read_b32 %0, a8\n\tv_accvgpr_read_b32 %1, a9\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a8, %0\n\tv_accvgpr_write_b32 a9, %1\n\t"
      "v_accvgpr_read_b32 %0, a10\n\tv_accvgpr_read_b32 %1, a11\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a10, %0\n\tv_accvgpr_write_b32 a11, %1\n\t"
      "v_accvgpr_read_b32 %0, a12\n\tv_accvgpr_read_b32 %1, a13\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a12, %0\n\tv_accvgpr_write_b32 a13, %1\n\t"
      "v_accvgpr_read_b32 %0, a14\n\tv_accvgpr_read_b32 %1, a15\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a14, %0\n\tv_accvgpr_write_b32 a15, %1\n\t"
      "v_accvgpr_read_b32 %0, a16\n\tv_accvgpr_read_b32 %1, a17\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a16, %0\n\tv_accvgpr_write_b32 a17, %1\n\t"
      "v_accvgpr_read_b32 %0, a18\n\tv_accvgpr_read_b32 %1, a19\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a18, %0\n\tv_accvgpr_write_b32 a19, %1\n\t"
      "v_accvgpr_read_b32 %0, a20\n\tv_accvgpr_read_b32 %1, a21\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a20, %0\n\tv_accvgpr_write_b32 a21, %1\n\t"
      "v_accvgpr_read_b32 %0, a22\n\tv_accvgpr_read_b32 %1, a23\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a22, %0\n\tv_accvgpr_write_b32 a23, %1\n\t"
      "v_accvgpr_read_b32 %0, a24\n\tv_accvgpr_read_b32 %1, a25\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a24, %0\n\tv_accvgpr_write_b32 a25, %1\n\t"
      "v_accvgpr_read_b32 %0, a26\n\tv_accvgpr_read_b32 %1, a27\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a26, %0\n\tv_accvgpr_write_b32 a27, %1\n\t"
      "v_accvgpr_read_b32 %0, a28\n\tv_accvgpr_read_b32 %1, a29\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a28, %0\n\tv_accvgpr_write_b32 a29, %1\n\t"
      "v_accvgpr_read_b32 %0, a30\n\tv_accvgpr_read_b32 %1, a31\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a30, %0\n\tv_accvgpr_write_b32 a31, %1\n\t"
      "v_accvgpr_read_b32 %0, a32\n\tv_accvgpr_read_b32 %1, a33\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a32, %0\n\tv_accvgpr_write_b32 a33, %1\n\t"
      "v_accvgpr_read_b32 %0, a34\n\tv_accvgpr_read_b32 %1, a35\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a34, %0\n\tv_accvgpr_write_b32 a35, %1\n\t"
      "v_accvgpr_read_b32 %0, a36\n\tv_accvgpr_read_b32 %1, a37\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a36, %0\n\tv_accvgpr_write_b32 a37, %1\n\t"
      "v_accvgpr_read_b32 %0, a38\n\tv_accvgpr_read_b32 %1, a39\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a38, %0\n\tv_accvgpr_write_b32 a39, %1\n\t"
      "v_accvgpr_read_b32 %0, a40\n\tv_accvgpr_read_b32 %1, a41\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a40, %0\n\tv_accvgpr_write_b32 a41, %1\n\t"
      "v_accvgpr_read_b32 %0, a42\n\tv_accvgpr_read_b32 %1, a43\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a42, %0\n\tv_accvgpr_write_b32 a43, %1\n\t"
      "v_accvgpr_read_b32 %0, a44\n\tv_accvgpr_read_b32 %1, a45\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a44, %0\n\tv_accvgpr_write_b32 a45, %1\n\t"
      "v_accvgpr_read_b32 %0, a46\n\tv_accvgpr_read_b32 %1, a47\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a46, %0\n\tv_accvgpr_write_b32 a47, %1\n\t"
      "v_accvgpr_read_b32 %0, a48\n\tv_accvgpr_read_b32 %1, a49\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a48, %0\n\tv_accvgpr_write_b32 a49, %1\n\t"
      "v_accvgpr_read_b32 %0, a50\n\tv_accvgpr_read_b32 %1, a51\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a50, %0\n\tv_accvgpr_write_b32 a51, %1\n\t"
      "v_accvgpr_read_b32 %0, a52\n\tv_accvgpr_read_b32 %1, a53\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a52, %0\n\tv_accvgpr_write_b32 a53, %1\n\t"
      "v_accvgpr_read_b32 %0, a54\n\tv_accvgpr_read_b32 %1, a55\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a54, %0\n\tv_accvgpr_write_b32 a55, %1\n\t"
      "v_accvgpr_read_b32 %0, a56\n\tv_accvgpr_read_b32 %1, a57\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a56, %0\n\tv_accvgpr_write_b32 a57, %1\n\t"
      "v_accvgpr_read_b32 %0, a58\n\tv_accvgpr_read_b32 %1, a59\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a58, %0\n\tv_accvgpr_write_b32 a59, %1\n\t"
      "v_accvgpr_read_b32 %0, a60\n\tv_accvgpr_read_b32 %1, a61\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a60, %0\n\tv_accvgpr_write_b32 a61, %1\n\t"
      "v_accvgpr_read_b32 %0, a62\n\tv_accvgpr_read_b32 %1, a63\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\t"
      "v_accvgpr_write_b32 a62, %0\n\tv_accvgpr_write_b32 a63, %1\n\t"
      "s_nop 7"
      : "=&v"(t0), "=&v"(t1) : "v"(alpha) : F2_CLOB0, F2_CLOB1, F2_CLOB2, F2_CLOB3);
}
template <int DT>
__device__ __forceinline__ void acc_read(float (&v)[16]) {      // caller has waited for the last MFMA (s_nop below)
  if constexpr (DT == 0) {
    asm volatile("v_accvgpr_read_b32 %0, a0\n\t" "v_accvgpr_read_b32 %1, a1\n\t" "v_accvgpr_read_b32 %2, a2\n\t" "v_accvgpr_read_b32 %3, a3\n\t" "v_accvgpr_read_b32 %4, a4\n\t" "v_accvgpr_read_b32 %5, a5\n\t" "v_accvgpr_read_b32 %6, a6\n\t" "v_accvgpr_read_b32 %7, a7\n\t" "v_accvgpr_read_b32 %8, a8\n\t" "v_accvgpr_read_b32 %9, a9\n\t" "v_accvgpr_read_b32 %10, a10\n\t" "v_accvgpr_read_b32 %11, a11\n\t" "v_accvgpr_read_b32 %12, a12\n\t" "v_accvgpr_read_b32 %13, a13\n\t" "v_accvgpr_read_b32 %14, a14\n\t" "v_accvgpr_read_b32 %15, a15\n\t" 
                 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]), "=v"(v[8]), "=v"(v[9]), "=v"(v[10]), "=v"(v[11]), "=v"(v[12]), "=v"(v[13]), "=v"(v[14]), "=v"(v[15]));
  } else if constexpr (DT == 1) {
    asm volatile("v_accvgpr_read_b32 %0, a16\n\t" "v_accvgpr_read_b32 %1, a17\n\t" "v_accvgpr_read_b32 %2, a18\n\t" "v_accvgpr_read_b32 %3, a19\n\t" "v_accvgpr_read_b32 %4, a20\n\t" "v_accvgpr_read_b32 %5, a21\n\t" "v_accvgpr_read_b32 %6, a22\n\t" "v_accvgpr_read_b32 %7, a23\n\t" "v_accvgpr_read_b32 %8, a24\n\t" "v_accvgpr_read_b32 %9, a25\n\t" "v_accvgpr_read_b32 %10, a26\n\t" "v_accvgpr_read_b32 %11, a27\n\t" "v_accvgpr_read_b32 %12, a28\n\t" "v_accvgpr_read_b32 %13, a29\n\t" "v_accvgpr_read_b32 %14, a30\n\t" "v_accvgpr_read_b32 %15, a31\n\t" 
                 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]), "=v"(v[8]), "=v"(v[9]), "=v"(v[10]), "=v"(v[11]), "=v"(v[12]), "=v"(v[13]), "=v"(v[14]), "=v"(v[15]));
  } else if constexpr (DT == 2) {
    asm volatile("v_accvgpr_read_b32 %0, a32\n\t" "v_accvgpr_read_b32 %1, a33\n\t" "v_accvgpr_read_b32 %2, a34\n\t" "v_accvgpr_read_b32 %3, a35\n\t" "v_accvgpr_read_b32 %4, a36\n\t" "v_accvgpr_read_b32 %5, a37\n\t" "v_accvgpr_read_b32 %6, a38\n\t" "v_accvgpr_read_b32 %7, a39\n\t" "v_accvgpr_read_b32 %8, a40\n\t" "v_accvgpr_read_b32 %9, a41\n\t" "v_accvgpr_read_b32 %10, a42\n\t" "v_accvgpr_read_b32 %11, a43\n\t" "v_accvgpr_read_b32 %12, a44\n\t" "v_accvgpr_read_b32 %13, a45\n\t" "v_accvgpr_read_b32 %14, a46\n\t" "v_accvgpr_read_b32 %15, a47\n\t" 
                 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]), "=v"(v[8]), "=v"(v[9]), "=v"(v[10]), "=v"(v[11]), "=v"(v[12]), "=v"(v[13]), "=v"(v[14]), "=v"(v[15]));
  } else {
    asm volatile("v_accvgpr_read_b32 %0, a48\n\t" "v_accvgpr_read_b32 %1, a49\n\t" "v_accvgpr_read_b32 %2, a50\n\t" "v_accvgpr_read_b32 %3, a51\n\t" "v_accvgpr_read_b32 %4, a52\n\t" "v_accvgpr_read_b32 %5, a53\n\t" "v_accvgpr_read_b32 %6, a54\n\t" "v_accvgpr_read_b32 %7, a55\n\t" "v_accvgpr_read_b32 %8, a56\n\t" "v_accvgpr_read_b32 %9, a57\n\t" "v_accvgpr_read_b32 %10, a58\n\t" "v_accvgpr_read_b32 %11, a59\n\t" "v_accvgpr_read_b32 %12, a60\n\t" "v_accvgpr_read_b32 %13, a61\n\t" "v_accvgpr_read_b32 %14, a62\n\t" "v_accvgpr_read_b32 %15, a63\n\t" 
                 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]), "=v"(v[8]), "=v"(v[9]), "=v"(v[10]), "=v"(v[11]), "=v"(v[12]), "=v"(v[13]), "=v"(v[14]), "=v"(v[15]));
  }
}

__device__ __forceinline__ bf16x8 lds_tr2(const char* p0) {      // two transposing reads: reduction slots 0-3 | 4-7
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 2048));
  return (bf16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

__device__ __forceinline__ float half_swap_max(float v) {          // max over lane and lane^32
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2v;
  const unsigned int u = __float_as_uint(v);
  const u32x2v r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_swap_sum(float v) {
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2v;
  const unsigned int u = __float_as_uint(v);
  const u32x2v r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// HD 64 (round 4: the Qwen2-0.5B student of the reference's shells, 14 heads of 64): the SAME tile images, step schedule and register
// layout with the upper 64 features absent — a K / V row fills the first 8 of its 16 swizzled chunk positions' worth of data, the
// QK^T k-steps 4-7 and the P·V feature strips 2-3 are not issued (8 + 8 MFMAs per tile instead of 16 + 16), O^T lives in a[0:31].
// The softmax work per tile is unchanged, so per flop it doubles: this instantiation is VALU-bound by construction.
template <bool CAUSAL, int HD = 128>
__device__ __forceinline__ void fwd2_block(const AttnP& p, char* smem, int qb, int h, int b, const int tbase = 0, const bool trace_on = false) {
  static_assert(HD == 128 || HD == 64, "head dim 128 or 64");
  constexpr int QB = 256, NKS = HD / 16, NDT = HD / 32;
  F2_STAMP(tbase + 0);
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));     // per-lane addresses are re-derived per pass, not hoisted (and spilled) across passes
  const int lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hk = h / p.group;
  const int S = p.cu ? (p.cu[b + 1] - p.cu[b]) : p.S;             // varlen: this sample's own length, tokens packed
  const int len = p.seqlens ? min(p.seqlens[b], S) : S;
  const int q0 = qb * QB, qw0 = q0 + wave * 32;
  if (q0 >= S) return;                                           // (varlen) query block past the end of a short sample
  const int q = qw0 + l31;
  const long long tok0 = p.cu ? (long long)p.cu[b] : (long long)b * S;
  const float c = p.scale * 1.4426950408889634f;

  const int kv_end = CAUSAL ? min(q0 + QB, len) : len;
  const int ntiles = (kv_end + 63) >> 6;                         // tiles the workgroup stages
  int ntw = CAUSAL ? min(ntiles, ((qw0 + 31) >> 6) + 1) : ntiles;   // tiles this wave computes
  if (qw0 >= S) ntw = 0;

  // ---- Q fragments (B operand of S^T): query l31, features ks*16 + hi*8 .. +7
  bf16x8 qf[8];
  {
    const bf16_t* qp = p.Q + (tok0 + min(q, S - 1)) * p.ldq + h * HD + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks < NKS) qf[ks] = *(const bf16x8*)(qp + ks * 16);
      if (q >= S || ks >= NKS) qf[ks] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  acc_zero();
  if (F2_PRIO && wave >= 4) __builtin_amdgcn_s_setprio(1);
  float mrun = -INFINITY, lrun = 0.f;

  // ---- staging: 2 x 16-byte chunks of K and of V per thread and tile
  const bf16_t* Kb = p.K + tok0 * p.ldk + hk * HD;
  const bf16_t* Vb = p.V + tok0 * p.ldv + hk * HD;
  const int srow = tid >> 4, sch = tid & 15;                    // + 32 rows for the second chunk
  const bool no_chunk = (HD == 64) && sch >= 8;                 // HD 64: a row has 8 chunks; the other lanes' loads are out of range (no traffic)
  const uint32_t kgo = no_chunk ? 0x80000000u : (uint32_t)(srow * p.ldk + sch * 8) * 2u, vgo = no_chunk ? 0x80000000u : (uint32_t)(srow * p.ldv + sch * 8) * 2u;
  const uint32_t kgo2 = no_chunk ? 0x80000000u : kgo + (uint32_t)(32 * p.ldk) * 2u, vgo2 = no_chunk ? 0x80000000u : vgo + (uint32_t)(32 * p.ldv) * 2u;
  const int kwo = srow * 256 + ((sch ^ (srow & 15)) << 4);      // rows srow and srow+32 share (row & 15) and (row & 3)
  const int vwo = srow * 256 + ((sch ^ ((srow & 3) << 2)) << 4);
  // buffer loads: rows past the end of the sequence are out of range of the descriptor and read as zeros (the tile advance
  // sits in the vector offset: a raw buffer's range check does not see the scalar offset)
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, (int)(((long long)(S - 1) * p.ldk + HD) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, (int)(((long long)(S - 1) * p.ldv + HD) * 2), 0x00020000);
#if F2_DMA
  // LDS-DMA: one wave-instruction fills 1 KiB = 4 consecutive key rows of a tile image; lane l writes bytes [16 l, 16 l + 16) of the
  // piece, i.e. row (l >> 4), physical chunk (l & 15), so it FETCHES the logical chunk that the image's XOR puts there.  Wave w takes
  // pieces w and w + 8 (keys 4w + r and 32 + 4w + r: the same (key & 15) and (key & 3), hence ONE per-lane offset per tensor; the
  // second piece and the tile advance are added to the VECTOR offset — a raw buffer's range check does not see the scalar offset, and
  // rows past the end of the sequence must read as zeros).
  const int dr = lane >> 4, dpc = lane & 15, dkey = 4 * wave + dr;
  const int klc = dpc ^ (dkey & 15), vlc = dpc ^ ((dkey & 3) << 2);
  const uint32_t kdo = (HD == 64 && klc >= 8) ? 0x80000000u : (uint32_t)(dkey * p.ldk + klc * 8) * 2u;
  const uint32_t vdo = (HD == 64 && vlc >= 8) ? 0x80000000u : (uint32_t)(dkey * p.ldv + vlc * 8) * 2u;
  auto dma_k = [&](const int t, const int buf) {
    const uint32_t adv = (uint32_t)(t * 64 * p.ldk) * 2u;
    char* dst = smem + buf * F2_TB + wave * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, LDS_PTR(dst), 16, kdo + adv, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, LDS_PTR(dst + 8192), 16, kdo + adv + (uint32_t)(32 * p.ldk) * 2u, 0, 0, 0);
  };
  auto dma_v = [&](const int t, const int buf) {
    const uint32_t adv = (uint32_t)(t * 64 * p.ldv) * 2u;
    char* dst = smem + F2_VBASE + buf * F2_TB + wave * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, LDS_PTR(dst), 16, vdo + adv, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, LDS_PTR(dst + 8192), 16, vdo + adv + (uint32_t)(32 * p.ldv) * 2u, 0, 0, 0);
  };
#endif
  u32x4 kr0, kr1, vr0, vr1;
#if F2_TOUCH
  uint32_t tk0 = 0, tv0 = 0;
#endif
  auto gload_k = [&](int t) {
    const uint32_t adv = (uint32_t)(t * 64 * p.ldk) * 2u;
    kr0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, kgo + adv, 0, 0));
    kr1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, kgo2 + adv, 0, 0));
  };
  auto gload_v = [&](int t) {
    const uint32_t adv = (uint32_t)(t * 64 * p.ldv) * 2u;
    vr0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, vgo + adv, 0, 0));
    vr1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, vgo2 + adv, 0, 0));
  };
  auto write_k = [&](int buf) {
    *(u32x4*)(smem + buf * F2_TB + kwo) = kr0;
    *(u32x4*)(smem + buf * F2_TB + kwo + 8192) = kr1;
  };
  auto write_v = [&](int buf) {
    *(u32x4*)(smem + F2_VBASE + buf * F2_TB + vwo) = vr0;
    *(u32x4*)(smem + F2_VBASE + buf * F2_TB + vwo + 8192) = vr1;
  };

  // ---- operand read addresses
  int kaddr[8];                                                  // K rows: key l31 (+32 per key tile), chunk 2ks + hi
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) kaddr[ks] = l31 * 256 + (((2 * ks + hi) ^ (l31 & 15)) << 4);
  int vaddr[4];                                                  // V^T rows of feature strip dt (see header)
  {
    const int i = lane & 15, gi = (lane >> 4) & 1, r = i >> 2, cc = i & 3;
    const int vb = (4 * hi + r) * 256 + (2 * (cc & 1) + gi) * 16 + 8 * (cc >> 1);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vaddr[dt] = vb + ((dt ^ r) << 6);
  }

  f32x16 s[2];
  bf16x8 pk[4];                                                  // P of the previous tile, B operands of its P·V
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) pk[kk] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};

  float nmc = 0.f, rs = 0.f;           // -max*c of the tile being exponentiated; row-sum partial of the running iteration
#if F2_PK
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 rs2 = {0.f, 0.f};
#endif
  float pm0 = -INFINITY, pm1 = -INFINITY;
  // exp2 of 2 scores of key half KT (elements e0, e0+1 of s[KT]) against nmc; packs a finished group of 8 into pk[2*KT + g]
  auto exp_pair = [&](auto kt_t, const int e0) {
    constexpr int KT = decltype(kt_t)::value;
    if (F2_INPIN) F2_PIN(s[KT]);
#if F2_PK
    {   // packed fp32 (v_pk_fma_f32 / v_pk_add_f32: two lanes' worth of work per issue slot): the scale-and-shift of both scores in
        // one instruction, the row sum as two partials (rs2) in one
      const f32x2 t = __builtin_elementwise_fma((f32x2){s[KT][e0], s[KT][e0 + 1]}, (f32x2){c, c}, (f32x2){nmc, nmc});
      f32x2 pv = {(F2_ABL == 2) ? s[KT][e0] : __builtin_amdgcn_exp2f(t[0]), (F2_ABL == 2) ? s[KT][e0 + 1] : __builtin_amdgcn_exp2f(t[1])};
      s[KT][e0] = pv[0]; s[KT][e0 + 1] = pv[1];
      if (F2_ABL != 7) rs2 += pv;
    }
    F2_PIN(rs2);
#else
#pragma unroll
    for (int e = e0; e < e0 + 2; ++e) {
      const float pv = (F2_ABL == 2) ? s[KT][e] : __builtin_amdgcn_exp2f(__builtin_fmaf(s[KT][e], c, nmc));
      s[KT][e] = pv;
      if (F2_ABL != 7) rs += pv;                               // ablation 7: no row sum
    }
    F2_PIN(rs);
#endif
    if ((e0 & 7) == 6) {
      const int rb = e0 - 6;
      u32x4 w = {pack2bf(s[KT][rb], s[KT][rb + 1]), pack2bf(s[KT][rb + 2], s[KT][rb + 3]),
                 pack2bf(s[KT][rb + 4], s[KT][rb + 5]), pack2bf(s[KT][rb + 6], s[KT][rb + 7])};
      pk[2 * KT + (rb >> 3)] = __builtin_bit_cast(bf16x8, w);
      F2_PIN(pk[2 * KT + (rb >> 3)]);
    }
  };
  // masked max over elements r0 .. r0+7 of s[KT] (masking writes -inf back into the scores)
  auto max8 = [&](auto masked_t, auto kt_t, const int r0, const int mthr) {
    constexpr bool MASKED = decltype(masked_t)::value;
    constexpr int KT = decltype(kt_t)::value;
    float m = -INFINITY;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int r = r0 + e;
      float v = s[KT][r];
      if constexpr (MASKED) {
        if (KT * 32 + (r & 3) + 8 * (r >> 2) > mthr) v = -INFINITY;
        s[KT][r] = v;
      }
      m = fmaxf(m, v);
    }
    F2_PIN(m);
    return m;
  };
  using KT0 = std::integral_constant<int, 0>;
  using KT1 = std::integral_constant<int, 1>;

  // -------------------------------------------------------------------------------- phase A
  // 16 MFMAs of S^T(j) = K(j) Q^T, key half 0 first (steps 0-7), then key half 1 (steps 8-15).  Under the first eight, the
  // VALU exponentiates key half 1 of tile j-1 (its registers are rewritten by steps 8-15); under the last eight it takes
  // the (masked) maximum of key half 0 of tile j.  MFMA: this wave computes tile j.  EXPS: a previous tile exists.
  auto phase_a = [&](auto mfma_t, auto exps_t, auto masked_t, const int mthr) {
    constexpr bool MFMA = decltype(mfma_t)::value, EXPS = decltype(exps_t)::value;
    const char* kb = smem;                                       // kaddr points into the K buffer of tile j
    bf16x8 kf[F2_DEPTH + 1];
    auto kread = [&](const int st) { return *(const bf16x8*)(kb + kaddr[st & 7] + (st >> 3) * 8192); };
    if constexpr (MFMA) {
#pragma unroll
      for (int st = 0; st < F2_DEPTH; ++st) if ((st & 7) < NKS) kf[st] = kread(st);
    }
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      const int kt = st >> 3, ks = st & 7, nx = st + F2_DEPTH;
      if constexpr (MFMA) {
        if (F2_ABL != 6) { if (nx < 16 && (nx & 7) < NKS) kf[nx % (F2_DEPTH + 1)] = kread(nx); }
        if (F2_ABL == 4) { if (ks == 0) { for (int r = 0; r < 16; ++r) s[kt][r] = (float)(r + kt); F2_PIN(s[kt]); } }
        else if (ks >= NKS) { }                                  // HD 64: features 64-127 do not exist
        else if (kt == 0) { if (ks == 0) qk_mfma<true>(s[0], kf[st % (F2_DEPTH + 1)], qf[ks]); else qk_mfma<false>(s[0], kf[st % (F2_DEPTH + 1)], qf[ks]); }
        else { if (ks == 0) qk_mfma<true>(s[1], kf[st % (F2_DEPTH + 1)], qf[ks]); else qk_mfma<false>(s[1], kf[st % (F2_DEPTH + 1)], qf[ks]); }
      }
      if constexpr (EXPS) { if (st < 8) exp_pair(KT1{}, 2 * st); }
      if constexpr (MFMA) {
        if (st == 11) {
          asm volatile("s_nop 7" ::: F2_CLOB_ALL);               // S^T half 0: last MFMA (step 7) -> first VALU read, padded by hand
          pm0 = max8(masked_t, KT0{}, 0, mthr);
        }
        if (st == 12) pm1 = max8(masked_t, KT0{}, 8, mthr);
      }
      F2_SB();
    }
    if constexpr (MFMA) asm volatile("s_nop 15" ::: F2_CLOB_ALL);   // S^T half 1: last MFMA -> VALU reads in phase B
  };

  // -------------------------------------------------------------------------------- phase B
  // 16 MFMAs of O^T += V(j-1)^T P(j-1)^T; under them the VALU finishes the row maximum of tile j (key half 1), takes the
  // deferred-rescale decision and exponentiates key half 0 of tile j (key half 1 follows under the next phase A).
  // PV: a previous tile exists.  SM: this wave computes tile j.
  auto phase_b = [&](auto masked_t, auto pv_t, auto sm_t, const int mthr) {   // vaddr points into the V buffer of tile j-1
    constexpr bool PV = decltype(pv_t)::value, SM = decltype(sm_t)::value;
    const char* vbp = smem + F2_VBASE;
    bf16x8 vf[F2_DEPTH + 1];
    if constexpr (PV) {
#pragma unroll
      for (int st = 0; st < F2_DEPTH; ++st) if ((st & 3) < NDT) vf[st] = lds_tr2(vbp + vaddr[st & 3] + (st >> 2) * 4096);
    }
    float pm2 = -INFINITY, pm3 = -INFINITY, alpha = 1.f;
    bool resc = false;
#pragma unroll
    for (int st = 0; st < 16; ++st) {                           // st = kk*4 + dt
      if constexpr (PV) {
        const int kk = st >> 2, dt = st & 3, nx = st + F2_DEPTH;
        if (F2_ABL != 6) { if (nx < 16 && (nx & 3) < NDT) vf[nx % (F2_DEPTH + 1)] = lds_tr2(vbp + vaddr[nx & 3] + (nx >> 2) * 4096); }
        if (F2_ABL == 3) { F2_PIN(vf[st % (F2_DEPTH + 1)]); }
        else if (dt >= NDT) { }                                  // HD 64: feature strips 2, 3 do not exist
        else if (dt == 0) pv_mfma<0>(vf[st % (F2_DEPTH + 1)], pk[kk]);
        else if (dt == 1) pv_mfma<1>(vf[st % (F2_DEPTH + 1)], pk[kk]);
        else if (dt == 2) pv_mfma<2>(vf[st % (F2_DEPTH + 1)], pk[kk]);
        else pv_mfma<3>(vf[st % (F2_DEPTH + 1)], pk[kk]);
      }
      if constexpr (SM) {
        if (st == 2) pm2 = max8(masked_t, KT1{}, 0, mthr);
        if (st == 3) pm3 = max8(masked_t, KT1{}, 8, mthr);
        if (st == 4) {                                          // row max, deferred-rescale decision (no control flow here)
          float mx = fmaxf(fmaxf(pm0, pm1), fmaxf(pm2, pm3));
          mx = half_swap_max(mx);
          resc = __builtin_amdgcn_ballot_w64((mx - mrun) * c > F2_THR) != 0;       // wave-uniform; NaN compares false
          const float mnew = resc ? fmaxf(mrun, mx) : mrun;
          const float a0 = __builtin_amdgcn_exp2f((mrun - mnew) * c);              // NaN only when both are -inf
          alpha = (mnew == mrun) ? 1.f : a0;
#if F2_PK
          rs = rs2[0] + rs2[1]; rs2 = (f32x2){0.f, 0.f};
#endif
          lrun = (lrun + rs) * alpha;                            // rs: every probability of the tiles before j
          rs = 0.f;
          mrun = mnew;
          nmc = (mnew == -INFINITY) ? 0.f : -mnew * c;
          F2_PIN(nmc);
        }
        if (st >= 5 && st < 13) exp_pair(KT0{}, 2 * (st - 5));  // pk[0] at step 8, pk[1] at step 12: their old values fed steps 0-7
      }
      F2_SB();
    }
    if constexpr (SM) {
      if (resc) acc_scale(alpha);                               // rare: every P·V MFMA of the previous tile is issued above
    }
  };

  // ---- prologue: K(0) -> LDS; K(1) and V(0) in flight towards the staging registers (F2_DMA: issued at the top of iteration 0)
#if F2_DMA
  if (ntiles > 0) dma_k(0, 0);
  if (F2_RING == 3) {                 // K(1) and V(0) too: iteration 0 requests K(2), V(1)
    if (ntiles > 1) dma_k(1, 1);
    if (ntiles > 0) dma_v(0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
  if (ntiles > 0) {
    gload_k(0); write_k(0);
    if (ntiles > 1) gload_k(1);
    gload_v(0);
  }
#endif
  __syncthreads();
  F2_STAMP(tbase + 1);

  // Staging (one register set per tensor): the registers hold K(j+1) and V(j), loaded during iteration j-1.  They are written
  // to LDS right AFTER the barrier that ended iteration j-1 (their buffers' last readers are done) and the loads of K(j+2),
  // V(j+1) are re-issued at once: the loads get a whole iteration to land, the ds_writes overlap this iteration's MFMAs, and
  // only the barrier follows the last MFMA (written before the barrier they sat serialised behind the MFMAs).
#if F2_RING == 3
  int kring0 = 0, kring2 = 2, vring0 = 0, vring1 = 1;          // slots of K(j), K(j+2), V(j-1) (from j = 1), V(j+1)
#endif
  for (int j = 0; j <= ntiles; ++j) {
#if F2_DMA
    // K(j+1) and V(j) go straight into the buffers whose last readers finished before the barrier that ended iteration j-1; they are
    // first read after the barrier that ends THIS iteration (vmcnt(0) in front of it): a whole iteration to land, nothing to wait for
    // inside it
    if (F2_ABL != 5) {
#if F2_RING == 3
      // K(j+2) -> the slot K(j-1) left, V(j+1) -> the slot V(j-2) left (their last readers finished before the barrier that ended
      // iteration j-1); first read two barriers from now
      if (j + 2 < ntiles) dma_k(j + 2, kring2);
      if (j + 1 < ntiles) dma_v(j + 1, vring1);
#else
      if (j + 1 < ntiles) dma_k(j + 1, (j + 1) & 1);
      if (j < ntiles) dma_v(j, j & 1);
#endif
    }
#endif
    if (F2_ABL != 5 && !F2_DMA) {
      if (j + 1 < ntiles) write_k((j + 1) & 1);
      if (j < ntiles) write_v(j & 1);
#if F2_TOUCH
      asm volatile("" :: "v"(tk0), "v"(tv0));   // last iteration's touches: older than anything in flight now
#endif
      if (j + 2 < ntiles) gload_k(j + 2);
      if (j + 1 < ntiles) gload_v(j + 1);
#if F2_TOUCH
      // L2 touch: one dword per 32 bytes of the tiles whose register loads are issued F2_TOUCH iterations from now (thread -> row
      // tid >> 3, byte (tid & 7) * 32; tiles past the end are out of the descriptor's range: no traffic).  The register loads have
      // ONE iteration (~4.9k cycles) to land, which is the loaded HBM latency: the first workgroup pass over a (batch, head)'s
      // K / V ran at 49 hundred-cycles per tile, a pass over L2-hot tiles at 44.8 (tools/attn_trace.py).
      {
        const uint32_t tr = (uint32_t)(tid >> 3), tc = (uint32_t)(tid & 7) * 32u;
        tk0 = __builtin_amdgcn_raw_buffer_load_b32(rk, (((uint32_t)(j + 2 + F2_TOUCH) * 64u + tr) * (uint32_t)p.ldk) * 2u + tc, 0, 0);
        tv0 = __builtin_amdgcn_raw_buffer_load_b32(rv, (((uint32_t)(j + 1 + F2_TOUCH) * 64u + tr) * (uint32_t)p.ldv) * 2u + tc, 0, 0);
      }
#endif
    }
    const bool do_a = (j < ntw), do_pv = (j >= 1 && j <= ntw);
    // masked when the key's position inside the tile exceeds mthr (covers the causal diagonal and the key length)
    const int mthr = (CAUSAL ? min(q, len - 1) : len - 1) - j * 64 - 4 * hi;
    const bool need_mask = (j * 64 + 64 > len) || (CAUSAL && j * 64 + 63 > qw0);
    using T = BoolTag<true>;
    using F = BoolTag<false>;
    if (do_a) {
      if (do_pv) {
        if (need_mask) { phase_a(T{}, T{}, T{}, mthr); phase_b(T{}, T{}, T{}, mthr); }
        else { phase_a(T{}, T{}, F{}, mthr); phase_b(F{}, T{}, T{}, mthr); }
      } else {
        if (need_mask) { phase_a(T{}, F{}, T{}, mthr); phase_b(T{}, F{}, T{}, mthr); }
        else { phase_a(T{}, F{}, F{}, mthr); phase_b(F{}, F{}, T{}, mthr); }
      }
    } else if (do_pv) {
      phase_a(F{}, T{}, F{}, mthr);
      phase_b(F{}, T{}, F{}, mthr);
    }
    if (j == ntiles) break;
#if F2_RING == 3
    // K(j) sat in slot j % 3, V(j-1) in slot (j-1) % 3: step the read bases round the ring (kstep / vstep: +F2_TB, or -2 F2_TB at the wrap)
    {
      const int kstep = (kring0 == 2) ? -2 * F2_TB : F2_TB;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) kaddr[ks] += kstep;
      kring0 = (kring0 == 2) ? 0 : kring0 + 1;
      kring2 = (kring2 == 2) ? 0 : kring2 + 1;
      if (j >= 1) {
        const int vstep = (vring0 == 2) ? -2 * F2_TB : F2_TB;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vaddr[dt] += vstep;
        vring0 = (vring0 == 2) ? 0 : vring0 + 1;
      }
      vring1 = (vring1 == 2) ? 0 : vring1 + 1;
    }
    // everything requested BEFORE this iteration has landed; what this iteration requested (0, 2 or 4 wave-instructions) stays in flight
    {
      const int mine = ((F2_ABL != 5 && j + 2 < ntiles) ? 2 : 0) + ((F2_ABL != 5 && j + 1 < ntiles) ? 2 : 0);
      if (mine == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else if (mine == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
#else
    // K(j) sat in buffer j&1, K(j+1) sits in the other one; V(j-1) sat in (j-1)&1, V(j) sits in j&1: flip the bases
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kaddr[ks] ^= F2_TB;
    if (j >= 1) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) vaddr[dt] ^= F2_TB;
    }
#endif
#if F2_RING == 3
#elif F2_DMA
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // my pieces of K(j+1), V(j) have landed; my reads of K(j), V(j-1) are done
#else
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    if (F2_ABL != 1) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    F2_STAMP(tbase + 2 + j);
  }
  F2_STAMP(tbase + 57);

  // ---- epilogue: lane (query l31, half hi) holds features dt*32 + hi*16 + r of its query
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");             // last asm MFMA -> accumulator reads
  if (q < S) {                                                   // no visible key at all: zeros, lse = -inf
#if F2_PK
    rs = rs2[0] + rs2[1];
#endif
    const float lt = half_swap_sum(lrun + rs);
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    bf16_t* op = p.O + (tok0 + q) * p.ldo + h * HD + hi * 16;
    auto store_strip = [&](auto dt_t) {
      constexpr int dt = decltype(dt_t)::value;
      float v[16];
      acc_read<dt>(v);
      u32x4 w0, w1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        w0[k] = pack2bf(v[2 * k] * inv, v[2 * k + 1] * inv);
        w1[k] = pack2bf(v[8 + 2 * k] * inv, v[8 + 2 * k + 1] * inv);
      }
      *(u32x4*)(op + dt * 32) = w0;
      *(u32x4*)(op + dt * 32 + 8) = w1;
    };
    store_strip(std::integral_constant<int, 0>{}); store_strip(std::integral_constant<int, 1>{});
    if constexpr (NDT == 4) { store_strip(std::integral_constant<int, 2>{}); store_strip(std::integral_constant<int, 3>{}); }
    if (hi == 0 && p.LSE)
      p.LSE[((long long)b * p.nh + h) * p.S + q] =
          (lt > 0.f) ? mrun * p.scale + __builtin_amdgcn_logf(lt) * 0.6931471805599453f : -INFINITY;
  }
  F2_STAMP(tbase + 58);
}

// Causal work per query block grows linearly with its index: every workgroup takes the pair (nqb-1-x, x), so all
// workgroups carry the same number of K/V tiles and the launch has no ragged tail.
// Grid (heads, query blocks, batch): the dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs, so with the
// head as the FASTEST index every workgroup of head h — all query blocks, all samples — lands on XCD h % 8 and the query
// blocks of one (batch, head), which read the same K / V rows, share one private L2 (+3 % over the query block fastest).
template <bool CAUSAL, int HD = 128>
__global__ __launch_bounds__(512, 2) void attn_fwd2_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#if F2_TRACE
  const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
#else
  int bx, by, bz;
  xcd_work_id(p.xcd_remap, bx, by, bz);
#endif
  if constexpr (CAUSAL) {
    const int nqb = (p.S + 255) / 256, x = by;
    const int npass = (2 * x + 1 < nqb) ? 2 : 1;
#pragma nounroll
    for (int pass = 0; pass < npass; ++pass) {
#if F2_TRACE
      const bool trace_on = blockIdx.x == 3 && blockIdx.y == F2_TRACE_Y && blockIdx.z == 5;
      fwd2_block<true, HD>(p, smem, pass ? x : nqb - 1 - x, blockIdx.x, blockIdx.z, pass * 60, trace_on);
      __syncthreads();
      F2_STAMP(pass * 60 + 59);
#else
      fwd2_block<true, HD>(p, smem, pass ? x : nqb - 1 - x, bx, bz);
      __syncthreads();
#endif
    }
  } else {
    fwd2_block<false, HD>(p, smem, by, bx, bz);
  }
}

void lmod_launch_attn_fwd2(const AttnP& p, int causal, hipStream_t stream, int hd) {
  static bool attr = false;
  const int lds = 2 * F2_RING * F2_TB;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_fwd2_kernel<true, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)attn_fwd2_kernel<false, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)attn_fwd2_kernel<true, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)attn_fwd2_kernel<false, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  const int nqb = (p.S + 255) / 256;
  const dim3 grid(p.nh, causal ? (nqb + 1) / 2 : nqb, p.B);
  if (hd == 64) {
    if (causal) hipLaunchKernelGGL((attn_fwd2_kernel<true, 64>), grid, dim3(512), lds, stream, p);
    else hipLaunchKernelGGL((attn_fwd2_kernel<false, 64>), grid, dim3(512), lds, stream, p);
  } else {
    if (causal) hipLaunchKernelGGL((attn_fwd2_kernel<true, 128>), grid, dim3(512), lds, stream, p);
    else hipLaunchKernelGGL((attn_fwd2_kernel<false, 128>), grid, dim3(512), lds, stream, p);
  }
}

#if F2_TRACE
extern "C" int lmod_debug_attn_trace(void* host_dst) {      // trace builds only (not part of the C-ABI)
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(f2_trace_buf), sizeof(unsigned long long) * 8 * 128);
}
#endif
