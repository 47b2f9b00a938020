// Flash attention forward for head dim 128 on gfx950, ONE WAVE PER SIMD (round 5) — the decoder's attention core
// (qwen2/modeling_qwen2.py:700-708 SDPA path, :290-309 eager path; causal + right-padding key mask, GQA).
//
// attn_fwd2.hip runs 8 waves of 32 queries: every wave reads the whole 64-key K and V tile from LDS for 32 MFMAs — 48 read instructions
// and their waits per wave and tile, which the timing ablations (profiles/r05_attn_ablation.md) price at 10-16 % of the kernel (the LDS array
// itself, 256 B/clk for these reads, is ~20 % busy).  Here a workgroup is 4 waves = one wave per
// SIMD with the whole 512-register file, a wave owns 64 queries as TWO independent 32-query halves, and every K / V fragment read
// from LDS feeds two MFMAs (one per half): half the LDS traffic per flop, no second wave competing for the SIMD's VALU issue, and
// the softmax of one half is independent work beside the MFMAs of the other.
//
//   registers   O^T accumulators: a[0:63] half 0, a[64:127] half 1 (strip 4 half + dt); Q fragments: a[128:191] (fragment 8 half + ks,
//               the MFMA's B operand straight from the accumulator file) — all owned by inline asm (attn_acc256.h, built by hipcc_agpr.sh
//               with "amdgpu-agpr-alloc"="256").  S^T accumulators, packed P, K / V fragments, softmax state: ordinary VGPRs.
//   MFMA 32x32x16, S^T = mfma(A = K rows, B = Q rows), O^T += mfma(A = V^T rows via ds_read_b64_tr_b16, B = P^T packed in place):
//               lane layouts, LDS tile images (K: 256-byte rows, chunk XOR key&15; V: chunk XOR (key&3)<<2) and the feature
//               permutation of the epilogue are attn_fwd2.hip's.
//   staging     K / V tiles global -> LDS by LDS-DMA (buffer_load ... lds, 1 KiB = 4 key rows per wave-instruction, 4 + 4 per wave and
//               tile), one iteration ahead, two buffers per tensor.
//   schedule    per 64-key tile and wave 64 MFMAs in two phases, software-pipelined by one tile, one workgroup barrier per tile:
//                 phase A  S^T(j) for both halves (32 MFMAs; a K fragment is read once and used twice)
//                          || exp2 + bf16 packing of key half 1 of tile j-1 (both halves), then the masked row maxima of key half 0
//                 phase B  O^T += V(j-1)^T P(j-1)^T for both halves (32 MFMAs; a V^T fragment is read once and used twice)
//                          || row maxima of key half 1, deferred-rescale decision, exp2 + packing of key half 0 of tile j
//               The VALU work per MFMA is attn_fwd2's (the same slices, once per half); what changes is who competes for the issue slots.
#include "attn_common.h"
#include "attn_acc256.h"
#include <type_traits>

#define F3_TB 16384
#ifndef F3_THR
#define F3_THR 6.0f
#endif
#ifndef F3_DEPTH
#define F3_DEPTH 3                 // operand fragments are read from LDS this many steps ahead (a step = 2 MFMAs)
#endif
#ifndef F3_ABL
#define F3_ABL 0                   // timing ablations (WRONG RESULTS): 1 no barrier, 2 no exponentials, 3 no P·V MFMAs, 4 no QK^T MFMAs, 5 no LDS-DMA
#endif                             //   in the loop, 6 no operand reads from LDS, 7 no softmax VALU at all (max / decide / exp / pack), 8 no hazard pads
#ifndef F3_ILV
#define F3_ILV 1                   // 1: the VALU slice of half 0 between the step's two MFMAs, the slice of half 1 behind the second one
#endif
#ifndef F3_STAG
#define F3_STAG 0                  // 1: the staggered schedule (halves half an iteration apart, one MFMA per fragment, three V buffers); 0: paired.
                                   //    Measured (profiles/r05_attn_fwd3_staggered.jsonl, parity green): 528 / 600 / 813 / 791 TF against the paired
                                   //    schedule's 600 / 708 / 907 / 951 — balancing the exponentials over the gaps does not repay reading every
                                   //    K / V fragment twice; kept as the documented arm
#endif
#ifndef F3_RS2
#define F3_RS2 1
#endif
#ifndef F3_INPIN
#define F3_INPIN 1                 // 1: a VALU slice's INPUT is pinned behind the MFMA it follows in the source (an empty asm volatile, ordered with
#endif                             //   the asm MFMAs), its output in front of the next one: the slice executes in that MFMA's shadow.  Without the
                                   //   input pin hipcc is free to hoist the slice above the MFMA — where two MFMAs then meet back to back, the
                                   //   second stalls the (only) wave for the first one's 32 cycles and the slice ran exposed
#define F3_PIN(x) asm volatile("" : "+v"(x))
#define F3_SB() __builtin_amdgcn_sched_barrier(0)

namespace {

__device__ __forceinline__ bf16x8 f3_lds_tr2(const char* p0) {      // two transposing reads: reduction slots 0-3 | 4-7
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 2048));
  return (bf16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
__device__ __forceinline__ float f3_swap_max(float v) {             // max over lane and lane^32
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2v;
  const unsigned int u = __float_as_uint(v);
  const u32x2v r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float f3_swap_sum(float v) {
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2v;
  const unsigned int u = __float_as_uint(v);
  const u32x2v r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// the 16 Q fragments / 8 O^T strips are named by template arguments: dispatch a (compile-time after unrolling) index
template <bool FIRST> __device__ __forceinline__ void f3_qk(const int idx, f32x16& acc, const bf16x8 kf) {
  switch (idx) {
    case 0: qk_mfma_q<0, FIRST>(acc, kf); break;   case 1: qk_mfma_q<1, FIRST>(acc, kf); break;
    case 2: qk_mfma_q<2, FIRST>(acc, kf); break;   case 3: qk_mfma_q<3, FIRST>(acc, kf); break;
    case 4: qk_mfma_q<4, FIRST>(acc, kf); break;   case 5: qk_mfma_q<5, FIRST>(acc, kf); break;
    case 6: qk_mfma_q<6, FIRST>(acc, kf); break;   case 7: qk_mfma_q<7, FIRST>(acc, kf); break;
    case 8: qk_mfma_q<8, FIRST>(acc, kf); break;   case 9: qk_mfma_q<9, FIRST>(acc, kf); break;
    case 10: qk_mfma_q<10, FIRST>(acc, kf); break; case 11: qk_mfma_q<11, FIRST>(acc, kf); break;
    case 12: qk_mfma_q<12, FIRST>(acc, kf); break; case 13: qk_mfma_q<13, FIRST>(acc, kf); break;
    case 14: qk_mfma_q<14, FIRST>(acc, kf); break; default: qk_mfma_q<15, FIRST>(acc, kf); break;
  }
}
__device__ __forceinline__ void f3_pv(const int strip, const bf16x8 vf, const bf16x8 pk) {
  switch (strip) {
    case 0: acc_mfma<0>(vf, pk); break; case 1: acc_mfma<1>(vf, pk); break; case 2: acc_mfma<2>(vf, pk); break; case 3: acc_mfma<3>(vf, pk); break;
    case 4: acc_mfma<4>(vf, pk); break; case 5: acc_mfma<5>(vf, pk); break; case 6: acc_mfma<6>(vf, pk); break; default: acc_mfma<7>(vf, pk); break;
  }
}
__device__ __forceinline__ void f3_set_q(const int idx, const bf16x8 q) {
  const u32x4 w = __builtin_bit_cast(u32x4, q);
  switch (idx) {
    case 0: acc_set_q<0>(w[0], w[1], w[2], w[3]); break;   case 1: acc_set_q<1>(w[0], w[1], w[2], w[3]); break;
    case 2: acc_set_q<2>(w[0], w[1], w[2], w[3]); break;   case 3: acc_set_q<3>(w[0], w[1], w[2], w[3]); break;
    case 4: acc_set_q<4>(w[0], w[1], w[2], w[3]); break;   case 5: acc_set_q<5>(w[0], w[1], w[2], w[3]); break;
    case 6: acc_set_q<6>(w[0], w[1], w[2], w[3]); break;   case 7: acc_set_q<7>(w[0], w[1], w[2], w[3]); break;
    case 8: acc_set_q<8>(w[0], w[1], w[2], w[3]); break;   case 9: acc_set_q<9>(w[0], w[1], w[2], w[3]); break;
    case 10: acc_set_q<10>(w[0], w[1], w[2], w[3]); break; case 11: acc_set_q<11>(w[0], w[1], w[2], w[3]); break;
    case 12: acc_set_q<12>(w[0], w[1], w[2], w[3]); break; case 13: acc_set_q<13>(w[0], w[1], w[2], w[3]); break;
    case 14: acc_set_q<14>(w[0], w[1], w[2], w[3]); break; default: acc_set_q<15>(w[0], w[1], w[2], w[3]); break;
  }
}
template <int HF> __device__ __forceinline__ void f3_scale_half(const float alpha) {
  acc_scale16<4 * HF + 0>(alpha); acc_scale16<4 * HF + 1>(alpha); acc_scale16<4 * HF + 2>(alpha); acc_scale16<4 * HF + 3>(alpha);
}
}  // namespace

template <int I> using F3IC = std::integral_constant<int, I>;

template <bool CAUSAL>
__device__ __forceinline__ void fwd3_block(const AttnP& p, char* smem, int qb, int h, int b) {
  constexpr int QB = 256;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));     // per-lane addresses are re-derived per pass, not hoisted (and spilled) across passes
  const int lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // 0..3
  const int hk = h / p.group;
  const int S = p.cu ? (p.cu[b + 1] - p.cu[b]) : p.S;
  const int len = p.seqlens ? min(p.seqlens[b], S) : S;
  const int q0 = qb * QB, qw0 = q0 + wave * 64;
  if (q0 >= S) return;
  const long long tok0 = p.cu ? (long long)p.cu[b] : (long long)b * S;
  const float c = p.scale * 1.4426950408889634f;

  const int kv_end = CAUSAL ? min(q0 + QB, len) : len;
  const int ntiles = (kv_end + 63) >> 6;                         // tiles the workgroup stages
  // a wave's 64 rows start on a 64-row boundary, so both halves end on the same key tile
  int ntw = CAUSAL ? min(ntiles, ((qw0 + 63) >> 6) + 1) : ntiles;
  if (qw0 >= S) ntw = 0;

  // ---- accumulators: O^T strips zeroed; Q fragments (B operand of S^T) -> a[128:191]: half hf, query qw0 + 32 hf + l31, features
  // ks*16 + hi*8 .. +7
  acc_zero_all();
  int qrow[2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    qrow[hf] = qw0 + 32 * hf + l31;
    const bf16_t* qp = p.Q + (tok0 + min(qrow[hf], S - 1)) * p.ldq + h * 128 + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      bf16x8 qf = *(const bf16x8*)(qp + ks * 16);
      if (qrow[hf] >= S) qf = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      f3_set_q(hf * 8 + ks, qf);
    }
  }
  float mrun[2] = {-INFINITY, -INFINITY}, lrun[2] = {0.f, 0.f};

  // ---- staging by LDS-DMA: one wave-instruction fills 1 KiB = 4 consecutive key rows of a tile image; lane l writes bytes [16 l, +16) of
  // the piece — row (l >> 4), physical chunk (l & 15) — so it FETCHES the logical chunk the image's XOR puts there.  Wave w takes pieces
  // w, w + 4, w + 8, w + 12 (keys 4w + r + 16 i: the same (key & 15) and (key & 3), hence ONE per-lane offset per tensor); the piece and tile
  // advances are added to the VECTOR offset (a raw buffer's range check does not see the scalar offset; rows past the end read as zeros).
  const bf16_t* Kb = p.K + tok0 * p.ldk + hk * 128;
  const bf16_t* Vb = p.V + tok0 * p.ldv + hk * 128;
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, (int)(((long long)(S - 1) * p.ldk + 128) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, (int)(((long long)(S - 1) * p.ldv + 128) * 2), 0x00020000);
  const int dr = lane >> 4, dpc = lane & 15, dkey = 4 * wave + dr;
  const uint32_t kdo = (uint32_t)(dkey * p.ldk + (dpc ^ (dkey & 15)) * 8) * 2u;
  const uint32_t vdo = (uint32_t)(dkey * p.ldv + (dpc ^ ((dkey & 3) << 2)) * 8) * 2u;
  auto dma_k = [&](const int t, const int buf) {
    const uint32_t adv = (uint32_t)(t * 64 * p.ldk) * 2u, step = (uint32_t)(16 * p.ldk) * 2u;
    char* dst = smem + buf * F3_TB + wave * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, LDS_PTR(dst + i * 4096), 16, kdo + adv + i * step, 0, 0, 0);
  };
  auto dma_v = [&](const int t, const int buf) {
    const uint32_t adv = (uint32_t)(t * 64 * p.ldv) * 2u, step = (uint32_t)(16 * p.ldv) * 2u;
    char* dst = smem + 2 * F3_TB + buf * F3_TB + wave * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, LDS_PTR(dst + i * 4096), 16, vdo + adv + i * step, 0, 0, 0);
  };

  // ---- operand read addresses (attn_fwd2.hip's)
  int kaddr[8];                                                  // K rows: key l31 (+32 per key half), chunk 2ks + hi
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) kaddr[ks] = l31 * 256 + (((2 * ks + hi) ^ (l31 & 15)) << 4);
  int vaddr[4];                                                  // V^T rows of feature strip dt
  {
    const int i = lane & 15, gi = (lane >> 4) & 1, r = i >> 2, cc = i & 3;
    const int vb = (4 * hi + r) * 256 + (2 * (cc & 1) + gi) * 16 + 8 * (cc >> 1);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vaddr[dt] = vb + ((dt ^ r) << 6);
  }

  f32x16 s[2][2];                                                // [half][key half]
  bf16x8 pk[2][4];                                               // P of the previous tile, B operands of its P·V
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) pk[hf][kk] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
  float nmc[2] = {0.f, 0.f}, rs[2] = {0.f, 0.f}, rs2[2] = {0.f, 0.f};   // -max*c of the tile being exponentiated; row-sum partials (two
                                                                 // per half: consecutive adds do not wait for each other's result)
  float pm0[2] = {-INFINITY, -INFINITY}, pm1[2] = {-INFINITY, -INFINITY};

  // exp2 of 2 scores of half HF, key half KT (elements e0, e0+1) against nmc; packs a finished group of 8 into pk[HF][2*KT + g]
  auto exp_pair = [&](auto hf_t, auto kt_t, const int e0) {
    constexpr int HF = decltype(hf_t)::value, KT = decltype(kt_t)::value;
    if (F3_ABL == 7) return;
    if (F3_INPIN) F3_PIN(s[HF][KT]);
#pragma unroll
    for (int e = e0; e < e0 + 2; ++e) {
      const float pv = (F3_ABL == 2) ? s[HF][KT][e] : __builtin_amdgcn_exp2f(__builtin_fmaf(s[HF][KT][e], c, nmc[HF]));
      s[HF][KT][e] = pv;
      if (F3_RS2 && (e & 1)) rs2[HF] += pv; else rs[HF] += pv;
    }
    F3_PIN(rs[HF]);
    if (F3_RS2) F3_PIN(rs2[HF]);
    if ((e0 & 7) == 6) {
      const int rb = e0 - 6;
      u32x4 w = {pack2bf(s[HF][KT][rb], s[HF][KT][rb + 1]), pack2bf(s[HF][KT][rb + 2], s[HF][KT][rb + 3]),
                 pack2bf(s[HF][KT][rb + 4], s[HF][KT][rb + 5]), pack2bf(s[HF][KT][rb + 6], s[HF][KT][rb + 7])};
      pk[HF][2 * KT + (rb >> 3)] = __builtin_bit_cast(bf16x8, w);
      F3_PIN(pk[HF][2 * KT + (rb >> 3)]);
    }
  };
  // masked max over elements r0 .. r0+7 of s[HF][KT] (masking writes -inf back into the scores)
  auto max8 = [&](auto masked_t, auto hf_t, auto kt_t, const int r0, const int mthr) {
    constexpr bool MASKED = decltype(masked_t)::value;
    constexpr int HF = decltype(hf_t)::value, KT = decltype(kt_t)::value;
    float m = -INFINITY;
    if (F3_ABL == 7) return m;
    if (F3_INPIN) F3_PIN(s[HF][KT]);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int r = r0 + e;
      float v = s[HF][KT][r];
      if constexpr (MASKED) {
        if (KT * 32 + (r & 3) + 8 * (r >> 2) > mthr) v = -INFINITY;
        s[HF][KT][r] = v;
      }
      m = fmaxf(m, v);
    }
    F3_PIN(m);
    return m;
  };
  using H0 = F3IC<0>; using H1 = F3IC<1>; using KT0 = F3IC<0>; using KT1 = F3IC<1>;

  // -------------------------------------------------------------------------------- phase A
  // 16 steps of 2 MFMAs: S^T(j) = K(j) Q^T for both halves, key half 0 (steps 0-7) then key half 1 (steps 8-15).  Under the first eight the
  // VALU exponentiates key half 1 of tile j-1 (its registers are rewritten by steps 8-15); under the last eight it takes the (masked) maxima
  // of key half 0 of tile j.  MFMA: this wave computes tile j.  EXPS: a previous tile exists.
  auto phase_a = [&](auto mfma_t, auto exps_t, auto masked_t, const int mthr0, const int mthr1) {
    constexpr bool MFMA = decltype(mfma_t)::value, EXPS = decltype(exps_t)::value;
    const char* kb = smem;                                       // kaddr points into the K buffer of tile j
    bf16x8 kf[F3_DEPTH + 1];
    auto kread = [&](const int st) { if (F3_ABL == 6) return pk[0][st & 3]; return *(const bf16x8*)(kb + kaddr[st & 7] + (st >> 3) * 8192); };
    if constexpr (MFMA) {
#pragma unroll
      for (int st = 0; st < F3_DEPTH; ++st) kf[st] = kread(st);
    }
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      const int kt = st >> 3, ks = st & 7, nx = st + F3_DEPTH;
      if constexpr (MFMA) {
        if (nx < 16) kf[nx % (F3_DEPTH + 1)] = kread(nx);
        if (F3_ABL == 4) { if (ks == 0) { for (int r = 0; r < 16; ++r) s[0][kt][r] = (float)(r + kt); F3_PIN(s[0][kt]); } }
        else if (ks == 0) f3_qk<true>(ks, s[0][kt], kf[st % (F3_DEPTH + 1)]); else f3_qk<false>(ks, s[0][kt], kf[st % (F3_DEPTH + 1)]);
      }
      if constexpr (EXPS && F3_ILV) { if (st < 8) { exp_pair(H0{}, KT1{}, 2 * st); F3_SB(); } }
      if constexpr (MFMA) {
        if (F3_ABL == 4) { if (ks == 0) { for (int r = 0; r < 16; ++r) s[1][kt][r] = (float)(r - kt); F3_PIN(s[1][kt]); } }
        else if (ks == 0) f3_qk<true>(8 + ks, s[1][kt], kf[st % (F3_DEPTH + 1)]); else f3_qk<false>(8 + ks, s[1][kt], kf[st % (F3_DEPTH + 1)]);
      }
      if constexpr (EXPS) { if (st < 8) { if (!F3_ILV) exp_pair(H0{}, KT1{}, 2 * st); exp_pair(H1{}, KT1{}, 2 * st); } }
      if constexpr (MFMA) {
        if (st == 10) {
          if (F3_ABL != 8) asm volatile("s_nop 7" ::: B2_CLOB_ALL);   // S^T key half 0: last MFMAs (step 7) -> first VALU read, padded by hand
          pm0[0] = max8(masked_t, H0{}, KT0{}, 0, mthr0);
        }
        if (st == 11) pm1[0] = max8(masked_t, H0{}, KT0{}, 8, mthr0);
        if (st == 12) pm0[1] = max8(masked_t, H1{}, KT0{}, 0, mthr1);
        if (st == 13) pm1[1] = max8(masked_t, H1{}, KT0{}, 8, mthr1);
      }
      F3_SB();
    }
    if constexpr (MFMA) { if (F3_ABL != 8) asm volatile("s_nop 15" ::: B2_CLOB_ALL); }   // S^T key half 1: last MFMAs -> VALU reads in phase B
  };

  // -------------------------------------------------------------------------------- phase B
  // 16 steps of 2 MFMAs: O^T += V(j-1)^T P(j-1)^T for both halves; under them the VALU finishes the row maxima of tile j (key half 1),
  // takes the deferred-rescale decisions and exponentiates key half 0 of tile j (key half 1 follows under the next phase A).
  auto phase_b = [&](auto masked_t, auto pv_t, auto sm_t, const int mthr0, const int mthr1) {
    constexpr bool PV = decltype(pv_t)::value, SM = decltype(sm_t)::value;
    const char* vbp = smem + 2 * F3_TB;                          // vaddr points into the V buffer of tile j-1
    bf16x8 vf[F3_DEPTH + 1];
    if constexpr (PV) {
#pragma unroll
      for (int st = 0; st < F3_DEPTH; ++st) vf[st] = (F3_ABL == 6) ? pk[1][st & 3] : f3_lds_tr2(vbp + vaddr[st & 3] + (st >> 2) * 4096);
    }
    float pm2[2] = {-INFINITY, -INFINITY}, pm3[2] = {-INFINITY, -INFINITY}, alpha[2] = {1.f, 1.f};
    bool resc[2] = {false, false};
    auto decide = [&](auto hf_t) {                               // row max, deferred-rescale decision (no control flow here)
      constexpr int HF = decltype(hf_t)::value;
      float mx = fmaxf(fmaxf(pm0[HF], pm1[HF]), fmaxf(pm2[HF], pm3[HF]));
      mx = f3_swap_max(mx);
      resc[HF] = __builtin_amdgcn_ballot_w64((mx - mrun[HF]) * c > F3_THR) != 0;      // wave-uniform; NaN compares false
      const float mnew = resc[HF] ? fmaxf(mrun[HF], mx) : mrun[HF];
      if (F3_ABL == 7) return;
      const float a0 = __builtin_amdgcn_exp2f((mrun[HF] - mnew) * c);                 // NaN only when both are -inf
      alpha[HF] = (mnew == mrun[HF]) ? 1.f : a0;
      lrun[HF] = (lrun[HF] + rs[HF] + rs2[HF]) * alpha[HF];      // rs: every probability of the tiles before j
      rs[HF] = 0.f; rs2[HF] = 0.f;
      mrun[HF] = mnew;
      nmc[HF] = (mnew == -INFINITY) ? 0.f : -mnew * c;
      F3_PIN(nmc[HF]);
    };
#pragma unroll
    for (int st = 0; st < 16; ++st) {                            // st = kk*4 + dt
      if constexpr (PV) {
        const int kk = st >> 2, dt = st & 3, nx = st + F3_DEPTH;
        if (nx < 16) vf[nx % (F3_DEPTH + 1)] = (F3_ABL == 6) ? pk[1][nx & 3] : f3_lds_tr2(vbp + vaddr[nx & 3] + (nx >> 2) * 4096);
        if (F3_ABL == 3) { F3_PIN(vf[st % (F3_DEPTH + 1)]); } else f3_pv(dt, vf[st % (F3_DEPTH + 1)], pk[0][kk]);
      }
      if constexpr (SM && F3_ILV) { if (st >= 7 && st < 15) { exp_pair(H0{}, KT0{}, 2 * (st - 7)); F3_SB(); } }
      if constexpr (PV) {
        const int kk = st >> 2, dt = st & 3;
        if (F3_ABL != 3) f3_pv(4 + dt, vf[st % (F3_DEPTH + 1)], pk[1][kk]);
      }
      if constexpr (SM) {
        if (st == 1) pm2[0] = max8(masked_t, H0{}, KT1{}, 0, mthr0);
        if (st == 2) pm3[0] = max8(masked_t, H0{}, KT1{}, 8, mthr0);
        if (st == 3) pm2[1] = max8(masked_t, H1{}, KT1{}, 0, mthr1);
        if (st == 4) pm3[1] = max8(masked_t, H1{}, KT1{}, 8, mthr1);
        if (st == 5) decide(H0{});
        if (st == 6) decide(H1{});
        // pk[.][0] is packed at step 10, pk[.][1] at step 14: their old values fed steps 0-7
        if (st >= 7 && st < 15) { if (!F3_ILV) exp_pair(H0{}, KT0{}, 2 * (st - 7)); exp_pair(H1{}, KT0{}, 2 * (st - 7)); }
      }
      F3_SB();
    }
    if constexpr (SM) {
      if (resc[0] || resc[1]) {                                  // rare: every P·V MFMA of the previous tile is issued above
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: B2_CLOB_ALL);   // the last MFMAs' results have landed
        if (resc[0]) f3_scale_half<0>(alpha[0]);
        if (resc[1]) f3_scale_half<1>(alpha[1]);
      }
    }
  };

#if F3_STAG
  // ==================================================================================== staggered schedule (F3_STAG)
  // The paired schedule above shares every K / V fragment between the halves, which makes both halves' scores ready at the same moment:
  // all 64 exponentials of a tile then crowd into the 34 MFMA gaps between the row-max decision and the next tile's S^T MFMAs (two per gap:
  // ~40 clocks of VALU issue in a 32-clock shadow) while the other 30 gaps stay empty, and with ONE wave on the SIMD nothing else fills
  // either (profiles/r05_attn_ablation.md §3: the softmax costs 36-43 %).  Here the halves run HALF AN ITERATION APART:
  //     block 1  S0^T(j)   = K(j) Q0^T          16 MFMAs   ||  half 1: exponentials of tile j-1, second part
  //     block 2  O1^T     += V(j-1)^T P1(j-1)   16 MFMAs   ||  half 0: row maxima, decision, exponentials of tile j, first part
  //     block 3  S1^T(j)   = K(j) Q1^T          16 MFMAs   ||  half 0: exponentials, second part
  //     block 4  O0^T     += V(j)^T P0(j)       16 MFMAs   ||  half 1: row maxima, decision, exponentials of tile j, first part
  // so at any moment exactly ONE half is exponentiating: ~1.2 exponentials per gap in 54 of the 64 gaps instead of 2 per gap in 34.  The
  // price: a fragment feeds one MFMA again (the LDS read traffic of the 8-wave kernel), and V(j) is read in the iteration that also reads
  // V(j-1): three V buffers (a tile is requested one iteration before its first read).
  const bf16x8 zero8 = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
  (void)zero8;
  float alpha_h[2] = {1.f, 1.f};
  bool resc_h[2] = {false, false};
  float pm2s[2] = {-INFINITY, -INFINITY}, pm3s[2] = {-INFINITY, -INFINITY};
  auto decide_s = [&](auto hf_t) {
    constexpr int HF = decltype(hf_t)::value;
    float mx = fmaxf(fmaxf(pm0[HF], pm1[HF]), fmaxf(pm2s[HF], pm3s[HF]));
    mx = f3_swap_max(mx);
    resc_h[HF] = __builtin_amdgcn_ballot_w64((mx - mrun[HF]) * c > F3_THR) != 0;
    const float mnew = resc_h[HF] ? fmaxf(mrun[HF], mx) : mrun[HF];
    const float a0 = __builtin_amdgcn_exp2f((mrun[HF] - mnew) * c);
    alpha_h[HF] = (mnew == mrun[HF]) ? 1.f : a0;
    lrun[HF] = (lrun[HF] + rs[HF] + rs2[HF]) * alpha_h[HF];
    rs[HF] = 0.f; rs2[HF] = 0.f;
    mrun[HF] = mnew;
    nmc[HF] = (mnew == -INFINITY) ? 0.f : -mnew * c;
    F3_PIN(nmc[HF]);
  };
  // exp pair number pr (0..15) of half HF: key half pr >> 3, elements 2 (pr & 7), +1
  auto exp_nr = [&](auto hf_t, const int pr) {
    if (pr < 8) exp_pair(hf_t, KT0{}, 2 * pr); else exp_pair(hf_t, KT1{}, 2 * (pr - 8));
  };
  // VALU slices by gap (st = 0..15 inside a block).  HEAD: the block right after the half's S^T block; TAIL: the block after that.
  auto valu_head = [&](auto masked_t, auto hf_t, const int st, const int mthr) {
    constexpr int HF = decltype(hf_t)::value;
    if (st == 2) { if (F3_ABL != 8) asm volatile("s_nop 7" ::: B2_CLOB_ALL); pm0[HF] = max8(masked_t, hf_t, KT0{}, 0, mthr); }
    if (st == 3) pm1[HF] = max8(masked_t, hf_t, KT0{}, 8, mthr);
    if (st == 4) pm2s[HF] = max8(masked_t, hf_t, KT1{}, 0, mthr);
    if (st == 5) pm3s[HF] = max8(masked_t, hf_t, KT1{}, 8, mthr);
    if (st == 6) decide_s(hf_t);
    if (st == 7) exp_nr(hf_t, 0);
    if (st == 8) exp_nr(hf_t, 1);
    if (st == 10) exp_nr(hf_t, 2);
    if (st == 11) exp_nr(hf_t, 3);
    if (st == 13) exp_nr(hf_t, 4);
    if (st == 14) exp_nr(hf_t, 5);
    (void)HF;
  };
  auto valu_tail = [&](auto hf_t, const int st) {
    // pairs 6..15 in gaps 0 1 3 4 6 7 9 10 12 13
    const int q = st / 3, r = st % 3;
    if (st <= 13 && r != 2) exp_nr(hf_t, 6 + 2 * q + r);
  };
  auto qk_block = [&](auto hf_t, const char* kb, auto&& valu) {
    constexpr int HF = decltype(hf_t)::value;
    bf16x8 kf[F3_DEPTH + 1];
    auto kread = [&](const int st) { if (F3_ABL == 6) return pk[0][st & 3]; return *(const bf16x8*)(kb + kaddr[st & 7] + (st >> 3) * 8192); };
#pragma unroll
    for (int st = 0; st < F3_DEPTH; ++st) kf[st] = kread(st);
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      const int kt = st >> 3, ks = st & 7, nx = st + F3_DEPTH;
      if (nx < 16) kf[nx % (F3_DEPTH + 1)] = kread(nx);
      if (F3_ABL == 4) { if (ks == 0) { for (int r = 0; r < 16; ++r) s[HF][kt][r] = (float)(r + kt); F3_PIN(s[HF][kt]); } }
      else if (ks == 0) f3_qk<true>(HF * 8 + ks, s[HF][kt], kf[st % (F3_DEPTH + 1)]); else f3_qk<false>(HF * 8 + ks, s[HF][kt], kf[st % (F3_DEPTH + 1)]);
      valu(st);
      F3_SB();
    }
  };
  auto pv_block = [&](auto hf_t, const char* vb, auto&& valu) {
    constexpr int HF = decltype(hf_t)::value;
    bf16x8 vf[F3_DEPTH + 1];
    auto vread = [&](const int st) { if (F3_ABL == 6) return pk[1][st & 3]; return f3_lds_tr2(vb + vaddr[st & 3] + (st >> 2) * 4096); };
#pragma unroll
    for (int st = 0; st < F3_DEPTH; ++st) vf[st] = vread(st);
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      const int kk = st >> 2, dt = st & 3, nx = st + F3_DEPTH;
      if (nx < 16) vf[nx % (F3_DEPTH + 1)] = vread(nx);
      if (F3_ABL == 3) { F3_PIN(vf[st % (F3_DEPTH + 1)]); } else f3_pv(4 * HF + dt, vf[st % (F3_DEPTH + 1)], pk[HF][kk]);
      valu(st);
      F3_SB();
    }
  };
  auto valu_only = [&](auto&& valu) {                           // a block with no MFMAs to issue (drain): its VALU slices alone
#pragma unroll
    for (int st = 0; st < 16; ++st) { valu(st); F3_SB(); }
  };
  auto none = [&](const int) {};
  auto dma_v3 = [&](const int t, const int slot) {
    const uint32_t adv = (uint32_t)(t * 64 * p.ldv) * 2u, step = (uint32_t)(16 * p.ldv) * 2u;
    char* dst = smem + 2 * F3_TB + slot * F3_TB + wave * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, LDS_PTR(dst + i * 4096), 16, vdo + adv + i * step, 0, 0, 0);
  };

  // ---- prologue: K(0), V(0) -> LDS
  if (ntiles > 0) { dma_k(0, 0); dma_v3(0, 0); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int vs_prev = 2, vs_cur = 0, vs_next = 1;                      // ring slots of V(j-1), V(j), V(j+1)
  for (int j = 0; j <= ntiles; ++j) {
    if (F3_ABL != 5 && j + 1 < ntiles) { dma_k(j + 1, (j + 1) & 1); dma_v3(j + 1, vs_next); }
    const bool do_q = (j < ntw), do_pv1 = (j >= 1 && j <= ntw);
    const int mthr0 = (CAUSAL ? min(qrow[0], len - 1) : len - 1) - j * 64 - 4 * hi;
    const int mthr1 = (CAUSAL ? min(qrow[1], len - 1) : len - 1) - j * 64 - 4 * hi;
    const bool need_mask = (j * 64 + 64 > len) || (CAUSAL && j * 64 + 63 > qw0);
    const char* kb = smem + (j & 1) * F3_TB;
    const char* vprev = smem + 2 * F3_TB + vs_prev * F3_TB;
    const char* vcur = smem + 2 * F3_TB + vs_cur * F3_TB;
    using T = BoolTag<true>;
    using F = BoolTag<false>;
    // block 1: S0^T(j) || half 1's exponentials of tile j-1, second part
    if (do_q) {
      if (do_pv1) qk_block(H0{}, kb, [&](const int st) { valu_tail(H1{}, st); });
      else qk_block(H0{}, kb, none);
      asm volatile("" ::: "memory");
    } else if (do_pv1) {
      valu_only([&](const int st) { valu_tail(H1{}, st); });
    }
    // block 2: O1^T += V(j-1)^T P1(j-1) || half 0: maxima, decision, first exponentials of tile j
    if (do_pv1) {
      if (do_q) {
        if (need_mask) pv_block(H1{}, vprev, [&](const int st) { valu_head(T{}, H0{}, st, mthr0); });
        else pv_block(H1{}, vprev, [&](const int st) { valu_head(F{}, H0{}, st, mthr0); });
      } else pv_block(H1{}, vprev, none);
    } else if (do_q) {
      if (need_mask) valu_only([&](const int st) { valu_head(T{}, H0{}, st, mthr0); });
      else valu_only([&](const int st) { valu_head(F{}, H0{}, st, mthr0); });
    }
    if (do_q && resc_h[0]) {                                      // rare; no MFMA into half 0's strips is in flight (its last: block 4 of j-1)
      asm volatile("s_nop 15\n\ts_nop 15" ::: B2_CLOB_ALL);
      f3_scale_half<0>(alpha_h[0]);
      resc_h[0] = false;
    }
    // block 3: S1^T(j) || half 0's exponentials, second part; block 4: O0^T += V(j)^T P0(j) || half 1: maxima, decision, first exponentials
    if (do_q) {
      qk_block(H1{}, kb, [&](const int st) { valu_tail(H0{}, st); });
      if (need_mask) pv_block(H0{}, vcur, [&](const int st) { valu_head(T{}, H1{}, st, mthr1); });
      else pv_block(H0{}, vcur, [&](const int st) { valu_head(F{}, H1{}, st, mthr1); });
      if (resc_h[1]) {                                            // half 1's strips: last MFMA in block 2 of this iteration
        asm volatile("s_nop 15\n\ts_nop 15" ::: B2_CLOB_ALL);
        f3_scale_half<1>(alpha_h[1]);
        resc_h[1] = false;
      }
    }
    if (j == ntiles) break;
    { const int t = vs_prev; vs_prev = vs_cur; vs_cur = vs_next; vs_next = t; }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (F3_ABL != 1) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
#else
  // ---- prologue: K(0) -> LDS
  if (ntiles > 0) dma_k(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int j = 0; j <= ntiles; ++j) {
    // K(j+1) and V(j) go straight into the buffers whose last readers finished before the barrier that ended iteration j-1; they are first
    // read after the barrier that ends THIS iteration (vmcnt(0) in front of it)
    if (F3_ABL != 5) {
      if (j + 1 < ntiles) dma_k(j + 1, (j + 1) & 1);
      if (j < ntiles) dma_v(j, j & 1);
    }
    const bool do_a = (j < ntw), do_pv = (j >= 1 && j <= ntw);
    // masked when the key's position inside the tile exceeds mthr (covers the causal diagonal and the key length)
    const int mthr0 = (CAUSAL ? min(qrow[0], len - 1) : len - 1) - j * 64 - 4 * hi;
    const int mthr1 = (CAUSAL ? min(qrow[1], len - 1) : len - 1) - j * 64 - 4 * hi;
    const bool need_mask = (j * 64 + 64 > len) || (CAUSAL && j * 64 + 63 > qw0);
    using T = BoolTag<true>;
    using F = BoolTag<false>;
    if (do_a) {
      if (do_pv) {
        if (need_mask) { phase_a(T{}, T{}, T{}, mthr0, mthr1); phase_b(T{}, T{}, T{}, mthr0, mthr1); }
        else { phase_a(T{}, T{}, F{}, mthr0, mthr1); phase_b(F{}, T{}, T{}, mthr0, mthr1); }
      } else {
        if (need_mask) { phase_a(T{}, F{}, T{}, mthr0, mthr1); phase_b(T{}, F{}, T{}, mthr0, mthr1); }
        else { phase_a(T{}, F{}, F{}, mthr0, mthr1); phase_b(F{}, F{}, T{}, mthr0, mthr1); }
      }
    } else if (do_pv) {
      phase_a(F{}, T{}, F{}, mthr0, mthr1);
      phase_b(F{}, T{}, F{}, mthr0, mthr1);
    }
    if (j == ntiles) break;
    // K(j) sat in buffer j&1, K(j+1) sits in the other one; V(j-1) sat in (j-1)&1, V(j) sits in j&1: flip the bases
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kaddr[ks] ^= F3_TB;
    if (j >= 1) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) vaddr[dt] ^= F3_TB;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // my pieces of K(j+1), V(j) have landed; my reads of K(j), V(j-1) are done
    if (F3_ABL != 1) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

#endif
  // ---- epilogue: lane (query l31 of half hf, feature half hi) holds features dt*32 + hi*16 + r of its query
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: B2_CLOB_ALL);    // last asm MFMAs -> accumulator reads
  auto store_half = [&](auto hf_t) {
    constexpr int HF = decltype(hf_t)::value;
    const int q = qrow[HF];
    if (q >= S) return;
    const float lt = f3_swap_sum(lrun[HF] + rs[HF] + rs2[HF]);
    const float inv = lt > 0.f ? 1.f / lt : 0.f;                 // no visible key at all: zeros, lse = -inf
    bf16_t* op = p.O + (tok0 + q) * p.ldo + h * 128 + hi * 16;
    auto store_strip = [&](auto dt_t) {
      constexpr int dt = decltype(dt_t)::value;
      float v[16];
      acc_read16<4 * HF + dt>(v);
      u32x4 w0, w1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        w0[k] = pack2bf(v[2 * k] * inv, v[2 * k + 1] * inv);
        w1[k] = pack2bf(v[8 + 2 * k] * inv, v[8 + 2 * k + 1] * inv);
      }
      *(u32x4*)(op + dt * 32) = w0;
      *(u32x4*)(op + dt * 32 + 8) = w1;
    };
    store_strip(F3IC<0>{}); store_strip(F3IC<1>{}); store_strip(F3IC<2>{}); store_strip(F3IC<3>{});
    if (hi == 0 && p.LSE)
      p.LSE[((long long)b * p.nh + h) * p.S + q] =
          (lt > 0.f) ? mrun[HF] * p.scale + __builtin_amdgcn_logf(lt) * 0.6931471805599453f : -INFINITY;
  };
  store_half(H0{});
  store_half(H1{});
}

// Grid and causal pairing: attn_fwd2.hip's (heads fastest: one head's workgroups share an XCD's L2; every workgroup takes the query
// blocks (nqb-1-x, x), so all carry the same number of K/V tiles).
template <bool CAUSAL>
__global__ __launch_bounds__(256, 1) void attn_fwd3_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if constexpr (CAUSAL) {
    const int nqb = (p.S + 255) / 256, x = blockIdx.y;
    const int npass = (2 * x + 1 < nqb) ? 2 : 1;
#pragma nounroll
    for (int pass = 0; pass < npass; ++pass) {
      fwd3_block<true>(p, smem, pass ? x : nqb - 1 - x, blockIdx.x, blockIdx.z);
      __syncthreads();
    }
  } else {
    fwd3_block<false>(p, smem, blockIdx.y, blockIdx.x, blockIdx.z);
  }
}

void lmod_launch_attn_fwd3(const AttnP& p, int causal, hipStream_t stream) {
  static bool attr = false;
  const int lds = (F3_STAG ? 5 : 4) * F3_TB;
  if (!__atomic_load_n(&attr, __ATOMIC_ACQUIRE)) {
    (void)hipFuncSetAttribute((const void*)attn_fwd3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)attn_fwd3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    __atomic_store_n(&attr, true, __ATOMIC_RELEASE);
  }
  const int nqb = (p.S + 255) / 256;
  const dim3 grid(p.nh, causal ? (nqb + 1) / 2 : nqb, p.B);
  if (causal) hipLaunchKernelGGL((attn_fwd3_kernel<true>), grid, dim3(256), lds, stream, p);
  else hipLaunchKernelGGL((attn_fwd3_kernel<false>), grid, dim3(256), lds, stream, p);
}
