// LAB ARM — not part of the default build (`make LAB=1` adds it with -DLMOD_LAB=1 into liblmod_hip_lab.so).
// The round-1 generic attention kernels (MFMA 16x16x32, 8 waves x 32 queries, register-staged tiles) for head dims 64 and 128:
// the baseline the shipped kernels (attn_fwd2.hip, attn_bwd2.hip) were measured against (LMOD_ATTN_FWD=1 / LMOD_ATTN_BWD=1 /
// LMOD_ATTN_BWD64=1).  The product library does not contain them; asking for these arms there returns LMOD_EUNSUPPORTED.
//
// Flash-style attention for gfx950, forward + backward, bf16 in / fp32 accumulate.
// Replaces F.scaled_dot_product_attention / flash_attn_func / the eager path of the reference
// decoder (qwen2/modeling_qwen2.py:700-708, :535-581, :290-309) and HF CLIP's encoder attention
// (call site multimodal_encoder/clip_encoder.py:54).  Causal + right-padding key mask
// (keys >= seqlens[b] are masked, like the 4-D mask of :1019-1027); GQA via `group`.
//
// Geometry: one workgroup = NWAVE waves.  Forward / dQ: 32 queries per wave against K/V tiles of 64 keys;
// dK/dV: 16 keys per wave against Q/dO tiles of 64 queries.  With 8 waves (256 queries or 128 keys per
// workgroup) the bytes staged into LDS per flop are half those of a 4-wave workgroup, which ran at the
// per-CU L2->LDS bandwidth bound.
//
// Lane algebra (MFMA 16x16x32; first/second operand share one register layout:
// index = lane&15, reduction slots = (lane>>4)*8 + j; D[row=(lane>>4)*4+r][col=lane&15]):
//   S^T tile = mfma(first = K rows, second = Q rows)  -> lane holds 4 keys x ONE query (lane&15)
//   => row max / row sum are in-lane + two xor-shuffles; alpha, m, l are lane-local.
//   The reduction-slot order of an MFMA is free as long as both operands agree, so the S^T
//   accumulators ARE the second operand of the next MFMA (slot j<4 <- key tile 2s, j>=4 <- tile
//   2s+1): O^T = mfma(first = V^T rows (d), second = P).  No LDS round trip, no cross-lane movement.
//   V^T (and K^T, Q^T, dO^T in backward) are never materialised: the row-major V/K/Q/dO tiles
//   already in LDS are read with ds_read_b64_tr_b16 (hardware transpose read, see read_tr), the
//   per-lane addresses chosen so each lane ends up with 16 contiguous d values of one token row
//   -> 16-byte epilogue stores.
// Staging: tiles go global -> registers -> LDS (loads issued before a tile's MFMAs, ds_writes after them,
// double-buffered LDS, one barrier per tile).  LDS-DMA is not used here: hipcc drains vmcnt(0) before any
// ds_read while an LDS-DMA is in flight, which serialises load and compute.
#include "attn_common.h"
#include <stdlib.h>

// ---- LDS tile image: [rows = tokens][HD bf16], 16-byte chunks XOR-swizzled per row.  Swizzles for HD 128:
//  SW 0  key = row & 15                      conflict-free for ds_read_b128 operand reads (rows = lane&15);
//                                            4-way conflicts under tr reads
//  SW 1  key = bit-permutation of row & 15   conflict-free for b128, 2-way for tr (tiles read both ways)
//  SW 2  key = (row&1) | ((row>>1)&1)<<3, the two 8-byte halves of a chunk swapped when (row>>2)&1:
//                                            conflict-free for tr reads (tiles read ONLY through read_tr)
// (tr pattern per half-wave: 8 consecutive tokens x 4 lanes reading 8 bytes of chunks {b, b+2, b+4, b+6}.)
// HD 64 tiles (ViT) always use key = row & 7.
template <int HD, int SW>
__device__ __forceinline__ int swz_key(int row) {
  if constexpr (HD == 128 && SW == 1)
    return (row & 1) | (((row >> 2) & 1) << 1) | (((row >> 3) & 1) << 2) | (((row >> 1) & 1) << 3);
  else if constexpr (HD == 128 && SW == 2)
    return (row & 1) | (((row >> 1) & 1) << 3);
  else
    return row & (HD / 8 - 1);
}
template <int HD, int SW>
__device__ __forceinline__ int swz_half(int row) {
  if constexpr (HD == 128 && SW == 2) return (row >> 2) & 1;
  else return 0;
}

// One MFMA operand (8 bf16 of row `row`, logical chunk `chunk`): ds_read_b128.
template <int HD, int SW>
__device__ __forceinline__ bf16x8 read_rows(const char* tile, int row, int chunk) {
  static_assert(!(HD == 128 && SW == 2), "SW 2 tiles are tr-read only");
  return *(const bf16x8*)(tile + row * (HD * 2) + ((chunk ^ swz_key<HD, SW>(row)) << 4));
}

// "Transposed" MFMA operand straight from the row-major tile: the operand wants index = feature d
// (lane&15) and reduction slots = tokens.  ds_read_b64_tr_b16 transposes at read time: within each 16-lane
// group, lanes 4r..4r+3 each fetch 4 contiguous bf16 of "row r" from THEIR OWN address and lane i receives
// column i (one element per row) [semantics probed on hardware: tools/probe/tr_probe.hip].  Rows r = 0..3
// are pointed at tokens 32st + 4g + r (second read: +16), exactly the slot order in which the S^T / dS
// accumulators hold their tokens; the 4 lanes of a row are pointed at features c4*16 + dt*4 + (0..3) of a
// 64-wide strip, so lane i ends up with feature (i>>2)*16 + dt*4 + (i&3): every lane finishes with 16
// contiguous output features.
template <int HD, int SW>
__device__ __forceinline__ bf16x8 read_tr(const char* tile, int dtile, int st, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int tok = st * 32 + g * 4 + (i >> 2);
  const int chunk = (dtile >> 2) * 8 + (i & 3) * 2 + ((dtile & 3) >> 1);
  const char* p = tile + tok * (HD * 2) + ((chunk ^ swz_key<HD, SW>(tok)) << 4) +
                  (((dtile & 1) ^ swz_half<HD, SW>(tok)) << 3);
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  // token + 16 has the same swizzle key / half as token (they only look at row bits 0..3)
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 16 * HD * 2));
  return (bf16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// [ROWS x HD] row-major tile through registers, split so the global loads can be issued before a compute
// phase and the LDS writes after it.  Offsets are 32-bit against a wave-uniform base.
template <int HD, int SW, int ROWS = 64>
struct RStage {
  static constexpr int NC = HD / 8, NL = ROWS * NC / NTHR;     // 16-byte chunks per thread
  static_assert(NL >= 1 && NL * NTHR == ROWS * NC, "tile does not divide over the workgroup");
  u32x4 vr[NL];
  __device__ __forceinline__ void load(const bf16_t* base, int ld, int rows_valid, int tid) {
#pragma unroll
    for (int a = 0; a < NL; ++a) {
      const int id = tid + a * NTHR, row = id / NC, ch = id % NC;
      if (row < rows_valid) vr[a] = *(const u32x4*)((const char*)base + (uint32_t)(row * ld + ch * 8) * 2u);
      else vr[a] = (u32x4){0u, 0u, 0u, 0u};
    }
  }
  __device__ __forceinline__ void write(char* tile, int tid) const {
#pragma unroll
    for (int a = 0; a < NL; ++a) {
      const int id = tid + a * NTHR, row = id / NC, ch = id % NC;
      u32x4 v = vr[a];
      if (swz_half<HD, SW>(row)) v = (u32x4){v[2], v[3], v[0], v[1]};
      *(u32x4*)(tile + row * (HD * 2) + ((ch ^ swz_key<HD, SW>(row)) << 4)) = v;
    }
  }
};

__device__ __forceinline__ bf16x8 pack_frag(const f32x4& lo, const f32x4& hi) {
  u32x4 r = {pack2bf(lo[0], lo[1]), pack2bf(lo[2], lo[3]), pack2bf(hi[0], hi[1]), pack2bf(hi[2], hi[3])};
  return __builtin_bit_cast(bf16x8, r);
}

__device__ __forceinline__ float xmax16(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float xsum16(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

// Epilogue: lane (index li, group g) owns, per 64-wide strip, the 16 contiguous features g*16 + dt*4 + r.
template <int HD>
__device__ __forceinline__ void store_rows16(bf16_t* dst, const f32x4 (&acc)[HD / 16], float mul) {
#pragma unroll
  for (int sp = 0; sp < HD / 64; ++sp) {
    u32x4 w0, w1;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      w0[k * 2] = pack2bf(acc[sp * 4 + k][0] * mul, acc[sp * 4 + k][1] * mul);
      w0[k * 2 + 1] = pack2bf(acc[sp * 4 + k][2] * mul, acc[sp * 4 + k][3] * mul);
      w1[k * 2] = pack2bf(acc[sp * 4 + 2 + k][0] * mul, acc[sp * 4 + 2 + k][1] * mul);
      w1[k * 2 + 1] = pack2bf(acc[sp * 4 + 2 + k][2] * mul, acc[sp * 4 + 2 + k][3] * mul);
    }
    *(u32x4*)(dst + sp * 64) = w0;
    *(u32x4*)(dst + sp * 64 + 8) = w1;
  }
}

// ============================================================================ forward
// Schedule: the 8 waves form two groups (waves w and w+4 share a SIMD).  A K/V tile is processed in two
// barrier intervals, X = [QK^T MFMAs | softmax of query tile 0] and Y = [softmax of query tile 1 | PV MFMAs];
// group 1 runs one interval behind group 0 (it passes one extra barrier up front), so on every SIMD one wave
// is in its MFMA half while the other is in its VALU half instead of both fighting for the same pipe.
// Tile j+1 is loaded to registers by everybody in interval 2j and written to LDS at the end of interval 2j+1.
// raw barrier: waits for this wave's LDS traffic only — the register prefetch (vmcnt) stays in flight across it
#define ATT_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
template <int HD, bool CAUSAL>
__device__ __forceinline__ void attn_fwd_block(const AttnP& p, char* smem, int qb, int h, int b) {
  constexpr int KS = HD / 32, DT = HD / 16, QB = NWAVE * 32;
  constexpr int TB = 64 * HD * 2;                       // tile bytes; layout: K[2], V[2], Q[QB rows]
  char* sQ = smem + 4 * TB;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));     // per-lane addresses are re-derived per pass, not hoisted (and spilled) across passes
  const int lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = (NWAVE == 8) ? (wave >> 2) : 0;
  const int hk = h / p.group;
  const int S = p.S;
  const int len = p.seqlens ? min(p.seqlens[b], S) : S;
  const int q0 = qb * QB, qw0 = q0 + wave * 32;
  const long long tok0 = (long long)b * S;

  f32x4 o[2][DT];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int d = 0; d < DT; ++d) o[qt][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float mrun[2] = {-INFINITY, -INFINITY}, lrun[2] = {0.f, 0.f};
  f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  asm volatile("" : "+v"(zero4));          // keep ONE zero tile live instead of re-materialising zeros per K/V tile

  const bf16_t* Kb = p.K + tok0 * p.ldk + hk * HD;
  const bf16_t* Vb = p.V + tok0 * p.ldv + hk * HD;
  const float c = p.scale * 1.4426950408889634f;
  const int kv_end = CAUSAL ? min(q0 + QB, len) : len;
  const int ntiles = (kv_end + 63) >> 6;

  RStage<HD, 0> kst;        // K: b128 reads only
  RStage<HD, 2> vst;        // V: tr reads only
  auto prefetch = [&](int t) {
    kst.load(Kb + (long long)t * 64 * p.ldk, p.ldk, S - t * 64, tid);
    vst.load(Vb + (long long)t * 64 * p.ldv, p.ldv, S - t * 64, tid);
  };
  auto commit = [&](int t) {
    kst.write(smem + (t & 1) * TB, tid);
    vst.write(smem + 2 * TB + (t & 1) * TB, tid);
  };
  {                         // this workgroup's Q rows, once (operand fragments are re-read per K tile)
    RStage<HD, 0, QB> qst;
    qst.load(p.Q + (tok0 + q0) * p.ldq + h * HD, p.ldq, S - q0, tid);
    qst.write(sQ, tid);
  }
  if (ntiles > 0) { prefetch(0); commit(0); }
  __syncthreads();
  if (grp == 1) {           // interval 0 of group 0: group 1 only fetches tile 1
    if (1 < ntiles) prefetch(1);
    ATT_BARRIER();
  }

  for (int j = 0; j < ntiles; ++j) {
    const int kv0 = j * 64;
    const char* sK = smem + (j & 1) * TB;
    const char* sV = smem + 2 * TB + (j & 1) * TB;
    const bool active = !(CAUSAL && kv0 > qw0 + 31);      // wave-uniform: something visible to this wave
    const bool need_mask = (kv0 + 64 > len) || (CAUSAL && kv0 + 63 > qw0);
    f32x4 s[2][4];
    bf16x8 pf[2][2];
    // online softmax of one 16-query tile; the masked flavour only runs on diagonal / padded tiles so the
    // steady state carries no compares or selects
    auto softmax = [&](auto masked, const int qt) {
      constexpr bool MASKED = decltype(masked)::value;
      const int q = qw0 + qt * 16 + li;
      float mx = -INFINITY;                       // running max is kept in RAW score units; c > 0
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = s[qt][nt][r];
          if constexpr (MASKED) {
            const int key = kv0 + nt * 16 + g * 4 + r;
            if (key >= len || (CAUSAL && key > q)) v = -INFINITY;
            s[qt][nt][r] = v;
          }
          mx = fmaxf(mx, v);
        }
      mx = xmax16(mx);
      const float mnew = fmaxf(mrun[qt], mx);
      const float msafe = (mnew == -INFINITY) ? 0.f : mnew;
      const float alpha = (mnew == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f((mrun[qt] - mnew) * c);
      const float nmc = -msafe * c;
      float rs = 0.f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qt][nt][r], c, nmc));
          s[qt][nt][r] = pv;
          rs += pv;
        }
      rs = xsum16(rs);
      lrun[qt] = lrun[qt] * alpha + rs;
      mrun[qt] = mnew;
      if (!__all(alpha == 1.f)) {                  // wave-uniform: most tiles do not move any row max
#pragma unroll
        for (int d = 0; d < DT; ++d) o[qt][d] *= alpha;
      }
      pf[qt][0] = pack_frag(s[qt][0], s[qt][1]);
      pf[qt][1] = pack_frag(s[qt][2], s[qt][3]);
    };

    // ---------------- interval X
    if (grp == 0 && j + 1 < ntiles) prefetch(j + 1);
    if (active) {
      if (ATT_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 q0f = read_rows<HD, 0>(sQ, wave * 32 + li, ks * 4 + g);
        const bf16x8 q1f = read_rows<HD, 0>(sQ, wave * 32 + 16 + li, ks * 4 + g);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const bf16x8 kf = read_rows<HD, 0>(sK, nt * 16 + li, ks * 4 + g);
          // first k-step accumulates onto a loop-invariant zero (no per-tile clears)
          s[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, q0f, ks == 0 ? zero4 : s[0][nt], 0, 0, 0);
          s[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, q1f, ks == 0 ? zero4 : s[1][nt], 0, 0, 0);
        }
        if (ATT_SCHED) __builtin_amdgcn_sched_barrier(0);      // keep hipcc from hoisting every operand read of the tile (spills)
      }
      if (ATT_PRIO) __builtin_amdgcn_s_setprio(0);
      if (need_mask) softmax(BoolTag<true>{}, 0); else softmax(BoolTag<false>{}, 0);
    }
    if (grp == 1 && j + 1 < ntiles) commit(j + 1);
    ATT_BARRIER();
    // ---------------- interval Y
    if (grp == 1 && j + 2 < ntiles) prefetch(j + 2);
    if (active) {
      if (need_mask) softmax(BoolTag<true>{}, 1); else softmax(BoolTag<false>{}, 1);
      if (ATT_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const bf16x8 vf = read_tr<HD, 2>(sV, d, st, lane);
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) o[qt][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qt][st], o[qt][d], 0, 0, 0);
          if (ATT_SCHED && (d & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
      if (ATT_PRIO) __builtin_amdgcn_s_setprio(0);
    }
    if (grp == 0 && j + 1 < ntiles) commit(j + 1);
    ATT_BARRIER();
  }
  if (grp == 0 && NWAVE == 8) ATT_BARRIER();      // group 0 matches group 1's extra barrier

#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = qw0 + qt * 16 + li;
    if (q >= S) continue;
    const float inv = lrun[qt] > 0.f ? 1.f / lrun[qt] : 0.f;
    store_rows16<HD>(p.O + (tok0 + q) * p.ldo + h * HD + g * 16, o[qt], inv);
    if (g == 0 && p.LSE)
      p.LSE[((long long)b * p.nh + h) * S + q] =
          (lrun[qt] > 0.f) ? mrun[qt] * p.scale + __builtin_amdgcn_logf(lrun[qt]) * 0.6931471805599453f : -INFINITY;
  }
}

// Causal work per query block grows linearly with its index: every workgroup takes the pair
// (nqb-1-x, x), so all workgroups carry the same number of K/V tiles and the launch has no ragged tail.
template <int HD, bool CAUSAL>
__global__ __launch_bounds__(NTHR, 2) void attn_fwd_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if constexpr (CAUSAL) {
    const int nqb = (p.S + NWAVE * 32 - 1) / (NWAVE * 32), x = blockIdx.x;
    const int npass = (2 * x + 1 < nqb) ? 2 : 1;
#pragma nounroll
    for (int pass = 0; pass < npass; ++pass) {
      attn_fwd_block<HD, true>(p, smem, pass ? x : nqb - 1 - x, blockIdx.y, blockIdx.z);
      __syncthreads();                    // the Q rows of a block are read until its last tile
    }
  } else {
    attn_fwd_block<HD, false>(p, smem, blockIdx.x, blockIdx.y, blockIdx.z);
  }
}

// ============================================================================ backward: dQ
// One workgroup = NWAVE*32 queries of one (b, head); loops over K/V tiles of 64 keys (same range as forward).
// Q fragments live in registers; the workgroup's dO rows are staged ONCE in LDS (fragments re-read per tile
// with ds_read_b128), so the only global loads inside the loop are the register-staged K/V prefetch.
template <int HD, bool CAUSAL>
__device__ __forceinline__ void attn_bwd_dq_block(const AttnP& p, char* smem, int qb, int h, int b) {
  constexpr int KS = HD / 32, DT = HD / 16, QB = NWAVE * 32;
  constexpr int TB = 64 * HD * 2;                       // layout: K[2], V[2], dO[QB rows]
  char* sdO = smem + 4 * TB;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));     // per-lane addresses are re-derived per pass, not hoisted (and spilled) across passes
  const int lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hk = h / p.group;
  const int S = p.cu ? (p.cu[b + 1] - p.cu[b]) : p.S;
  const int len = p.seqlens ? min(p.seqlens[b], S) : S;
  const int q0 = qb * QB, qw0 = q0 + wave * 32;
  if (q0 >= S) return;
  const long long tok0 = p.cu ? (long long)p.cu[b] : (long long)b * S;
  const float c = p.scale * 1.4426950408889634f;

  bf16x8 qf[2][KS];
  float lse2[2], dl[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = qw0 + qt * 16 + li;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (q < S) qf[qt][ks] = *(const bf16x8*)(p.Q + (tok0 + q) * p.ldq + h * HD + ks * 32 + g * 8);
      else qf[qt][ks] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
    const long long si = ((long long)b * p.nh + h) * p.S + min(q, S - 1);
    lse2[qt] = p.LSE[si] * 1.4426950408889634f;
    dl[qt] = p.Delta[si];
  }
  f32x4 dq[2][DT];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int d = 0; d < DT; ++d) dq[qt][d] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bf16_t* Kb = p.K + tok0 * p.ldk + hk * HD;
  const bf16_t* Vb = p.V + tok0 * p.ldv + hk * HD;
  const int kv_end = CAUSAL ? min(q0 + QB, len) : len;
  const int ntiles = (kv_end + 63) >> 6;

  RStage<HD, 1> kst;        // K: b128 (S^T) and tr (dQ^T) reads
  RStage<HD, 0> vst;        // V: b128 only
  {
    RStage<HD, 0, QB> dst;  // dO rows of this workgroup, once
    dst.load(p.dO + (tok0 + q0) * p.lddo + h * HD, p.lddo, S - q0, tid);
    dst.write(sdO, tid);
  }
  if (ntiles > 0) {
    kst.load(Kb, p.ldk, S, tid);
    vst.load(Vb, p.ldv, S, tid);
    kst.write(smem, tid);
    vst.write(smem + 2 * TB, tid);
  }
  __syncthreads();
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[qt][ks]));
    asm volatile("" : "+v"(lse2[qt]), "+v"(dl[qt]));
  }

  for (int j = 0; j < ntiles; ++j) {
    const int kv0 = j * 64;
    const char* sK = smem + (j & 1) * TB;
    const char* sV = smem + 2 * TB + (j & 1) * TB;
    const bool more = (j + 1 < ntiles);
    if (more) {
      kst.load(Kb + (long long)(kv0 + 64) * p.ldk, p.ldk, S - kv0 - 64, tid);
      vst.load(Vb + (long long)(kv0 + 64) * p.ldv, p.ldv, S - kv0 - 64, tid);
    }
    if (!(CAUSAL && kv0 > qw0 + 31)) {
      // q-tiles are processed one after the other to keep S / dP live ranges at 32 registers
      bf16x8 dsf[2][2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        f32x4 s[4], dp[4];
        const int q = qw0 + qt * 16 + li;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { s[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8 dofr = read_rows<HD, 0>(sdO, wave * 32 + qt * 16 + li, ks * 4 + g);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const bf16x8 kf = read_rows<HD, 1>(sK, nt * 16 + li, ks * 4 + g);
            const bf16x8 vf = read_rows<HD, 0>(sV, nt * 16 + li, ks * 4 + g);
            s[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][ks], s[nt], 0, 0, 0);
            dp[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dofr, dp[nt], 0, 0, 0);
          }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kv0 + nt * 16 + g * 4 + r;
            float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[nt][r], c, -lse2[qt]));
            if (key >= len || (CAUSAL && key > q) || q >= S) pv = 0.f;
            s[nt][r] = pv * (dp[nt][r] - dl[qt]);
          }
        dsf[qt][0] = pack_frag(s[0], s[1]);
        dsf[qt][1] = pack_frag(s[2], s[3]);
      }
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const bf16x8 ktf = read_tr<HD, 1>(sK, d, st, lane);
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) dq[qt][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf[qt][st], dq[qt][d], 0, 0, 0);
        }
    }
    if (more) {
      kst.write(smem + ((j + 1) & 1) * TB, tid);
      vst.write(smem + 2 * TB + ((j + 1) & 1) * TB, tid);
    }
    __syncthreads();
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = qw0 + qt * 16 + li;
    if (q >= S) continue;
    store_rows16<HD>(p.dQ + (tok0 + q) * p.lddq + h * HD + g * 16, dq[qt], p.scale);
  }
}

template <int HD, bool CAUSAL>
__global__ __launch_bounds__(NTHR, 2) void attn_bwd_dq_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if constexpr (CAUSAL) {                 // balanced pairs, as in forward
    const int nqb = (p.S + NWAVE * 32 - 1) / (NWAVE * 32), x = blockIdx.x;
    const int npass = (2 * x + 1 < nqb) ? 2 : 1;
#pragma nounroll
    for (int pass = 0; pass < npass; ++pass) {
      attn_bwd_dq_block<HD, true>(p, smem, pass ? x : nqb - 1 - x, blockIdx.y, blockIdx.z);
      __syncthreads();                    // the dO rows of a block are read until its last tile
    }
  } else {
    attn_bwd_dq_block<HD, false>(p, smem, blockIdx.x, blockIdx.y, blockIdx.z);
  }
}

// ============================================================================ backward: dK, dV
// One workgroup = NWAVE*16 keys of one (b, kv head); loops over the q heads of the GQA group and over
// Q / dO tiles of 64 queries (register-staged, double-buffered, together with their lse / delta rows).
template <int HD, bool CAUSAL>
__device__ __forceinline__ void attn_bwd_dkv_block(const AttnP& p, char* smem, int kb, int hk, int b) {
  constexpr int KS = HD / 32, DT = HD / 16, KBLK = NWAVE * 16;
  constexpr int TB = 64 * HD * 2;                       // layout: Q[2], dO[2], {lse[64], delta[64]}[2]
  float* sLD = (float*)(smem + 4 * TB);
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = p.cu ? (p.cu[b + 1] - p.cu[b]) : p.S;
  const int len = p.seqlens ? min(p.seqlens[b], S) : S;
  const int k0 = kb * KBLK, kw0 = k0 + wave * 16;
  if (k0 >= S) return;
  const int key = kw0 + li;
  const long long tok0 = p.cu ? (long long)p.cu[b] : (long long)b * S;
  const float c = p.scale * 1.4426950408889634f;

  bf16x8 kf[KS], vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (key < S) {
      kf[ks] = *(const bf16x8*)(p.K + (tok0 + key) * p.ldk + hk * HD + ks * 32 + g * 8);
      vf[ks] = *(const bf16x8*)(p.V + (tok0 + key) * p.ldv + hk * HD + ks * 32 + g * 8);
    } else {
      kf[ks] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      vf[ks] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  f32x4 dk[DT], dv[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) { dk[d] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[d] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  const int nq = (S + 63) >> 6;
  const int qt_first = CAUSAL ? (k0 >> 6) : 0;
  const int per_head = nq - qt_first;
  const int total = (k0 < len) ? per_head * p.group : 0;       // (head-in-group, q tile) pairs, flattened

  RStage<HD, 1> qst, ost;    // Q and dO tiles: b128 (S, dP) and tr (dK^T, dV^T) reads
  float ld_reg = 0.f;        // one lse (threads 0..63) or delta (64..127) value of the tile being staged
  auto prefetch = [&](int it) {
    const int hh = it / per_head, j = qt_first + it - hh * per_head;
    const int h = hk * p.group + hh, q0 = j * 64;
    qst.load(p.Q + (tok0 + q0) * p.ldq + h * HD, p.ldq, S - q0, tid);
    ost.load(p.dO + (tok0 + q0) * p.lddo + h * HD, p.lddo, S - q0, tid);
    const long long si = ((long long)b * p.nh + h) * p.S;
    if (tid < 64) ld_reg = (q0 + tid < S) ? p.LSE[si + q0 + tid] * 1.4426950408889634f : 0.f;
    else if (tid < 128) ld_reg = (q0 + tid - 64 < S) ? p.Delta[si + q0 + tid - 64] : 0.f;
  };
  auto commit = [&](int buf) {
    qst.write(smem + buf * TB, tid);
    ost.write(smem + 2 * TB + buf * TB, tid);
    if (tid < 128) sLD[buf * 128 + tid] = ld_reg;
  };
  if (total > 0) { prefetch(0); commit(0); }
  __syncthreads();
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(kf[ks]), "+v"(vf[ks]));

  for (int it = 0; it < total; ++it) {
    const int hh = it / per_head, j = qt_first + it - hh * per_head;
    const int q0 = j * 64;
    const char* sQ = smem + (it & 1) * TB;
    const char* sdO = smem + 2 * TB + (it & 1) * TB;
    const float* sLse = sLD + (it & 1) * 128;
    const float* sDl = sLse + 64;
    const bool more = (it + 1 < total);
    if (more) prefetch(it + 1);
    if (!(CAUSAL && q0 + 63 < kw0)) {          // not every query of the tile precedes this wave's keys
      f32x4 s[4], dp[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) { s[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const bf16x8 qfr = read_rows<HD, 1>(sQ, t * 16 + li, ks * 4 + g);
          const bf16x8 dofr = read_rows<HD, 1>(sdO, t * 16 + li, ks * 4 + g);
          s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[ks], s[t], 0, 0, 0);
          dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dofr, vf[ks], dp[t], 0, 0, 0);
        }
      // lane holds queries q0 + t*16 + g*4 + r (r=0..3) for ONE key (li)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 l4 = *(const f32x4*)(sLse + t * 16 + g * 4);
        const f32x4 d4 = *(const f32x4*)(sDl + t * 16 + g * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = q0 + t * 16 + g * 4 + r;
          float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r], c, -l4[r]));
          if (q >= S || key >= len || (CAUSAL && key > q)) pv = 0.f;
          s[t][r] = pv;
          dp[t][r] = pv * (dp[t][r] - d4[r]);
        }
      }
      const bf16x8 pf0 = pack_frag(s[0], s[1]), pf1 = pack_frag(s[2], s[3]);
      const bf16x8 ds0 = pack_frag(dp[0], dp[1]), ds1 = pack_frag(dp[2], dp[3]);
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        const bf16x8 o0 = read_tr<HD, 1>(sdO, d, 0, lane), o1 = read_tr<HD, 1>(sdO, d, 1, lane);
        dv[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(o0, pf0, dv[d], 0, 0, 0);
        dv[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(o1, pf1, dv[d], 0, 0, 0);
        const bf16x8 t0 = read_tr<HD, 1>(sQ, d, 0, lane), t1 = read_tr<HD, 1>(sQ, d, 1, lane);
        dk[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(t0, ds0, dk[d], 0, 0, 0);
        dk[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(t1, ds1, dk[d], 0, 0, 0);
      }
    }
    if (more) commit((it + 1) & 1);
    __syncthreads();
  }
  if (key < S) {
    store_rows16<HD>(p.dK + (tok0 + key) * p.lddk + hk * HD + g * 16, dk, p.scale);
    store_rows16<HD>(p.dV + (tok0 + key) * p.lddv + hk * HD + g * 16, dv, 1.f);
  }
}

template <int HD, bool CAUSAL>
__global__ __launch_bounds__(NTHR, 2) void attn_bwd_dkv_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if constexpr (CAUSAL) {                 // early key blocks see every query tile, late ones almost none: pair them
    const int nkb = (p.S + NWAVE * 16 - 1) / (NWAVE * 16), x = blockIdx.x;
    const int npass = (2 * x + 1 < nkb) ? 2 : 1;
#pragma nounroll
    for (int pass = 0; pass < npass; ++pass)
      attn_bwd_dkv_block<HD, true>(p, smem, pass ? nkb - 1 - x : x, blockIdx.y, blockIdx.z);
  } else {
    attn_bwd_dkv_block<HD, false>(p, smem, blockIdx.x, blockIdx.y, blockIdx.z);
  }
}

template <typename KT>
static void set_lds_lab(KT kern, int bytes) {
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

void lmod_launch_attn_fwd_generic(const AttnP& p, int causal, hipStream_t stream, int hd) {
  constexpr int QB = NWAVE * 32;
  const int S = p.S, nqb = (S + QB - 1) / QB;
  const dim3 grid(causal ? (nqb + 1) / 2 : nqb, p.nh, p.B);
  const int lds = 4 * 64 * hd * 2 + QB * hd * 2;
  if (hd == 128 && causal) { set_lds_lab(attn_fwd_kernel<128, true>, lds); hipLaunchKernelGGL((attn_fwd_kernel<128, true>), grid, dim3(NTHR), lds, stream, p); }
  else if (hd == 128) { set_lds_lab(attn_fwd_kernel<128, false>, lds); hipLaunchKernelGGL((attn_fwd_kernel<128, false>), grid, dim3(NTHR), lds, stream, p); }
  else if (causal) { set_lds_lab(attn_fwd_kernel<64, true>, lds); hipLaunchKernelGGL((attn_fwd_kernel<64, true>), grid, dim3(NTHR), lds, stream, p); }
  else { set_lds_lab(attn_fwd_kernel<64, false>, lds); hipLaunchKernelGGL((attn_fwd_kernel<64, false>), grid, dim3(NTHR), lds, stream, p); }
}

void lmod_launch_attn_bwd_generic(const AttnP& p, int causal, hipStream_t stream, int hd) {
  constexpr int QB = NWAVE * 32, KBLK = NWAVE * 16;
  const int S = p.S, nkv = p.nh / p.group;
  const int nqb = (S + QB - 1) / QB, nkb = (S + KBLK - 1) / KBLK;
  const dim3 gq(causal ? (nqb + 1) / 2 : nqb, p.nh, p.B), gk(causal ? (nkb + 1) / 2 : nkb, nkv, p.B);
  const int lds_q = 4 * 64 * hd * 2 + QB * hd * 2;
  const int lds_k = 4 * 64 * hd * 2 + 1024;
#define LAUNCH_BWD(HDV, CZ)                                                                                        \
  do {                                                                                                             \
    set_lds_lab(attn_bwd_dq_kernel<HDV, CZ>, lds_q);                                                               \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<HDV, CZ>), gq, dim3(NTHR), lds_q, stream, p);                           \
    set_lds_lab(attn_bwd_dkv_kernel<HDV, CZ>, lds_k);                                                              \
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<HDV, CZ>), gk, dim3(NTHR), lds_k, stream, p);                          \
  } while (0)
  if (hd == 128 && causal) LAUNCH_BWD(128, true);
  else if (hd == 128) LAUNCH_BWD(128, false);
  else if (causal) LAUNCH_BWD(64, true);
  else LAUNCH_BWD(64, false);
#undef LAUNCH_BWD
}
