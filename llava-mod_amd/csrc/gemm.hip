// bf16 MFMA GEMM for gfx950:  C[M,N] (+)= act(A[M,K] · B[N,K]^T + bias)
//
// "NT" form only: both operands are K-contiguous.  That is nn.Linear's forward as stored
// (reference: qwen2/modeling_qwen2.py:262-264,320 QKV/O; :186-187 SwiGLU; :1163 lm_head;
// multimodal_projector/builder.py:57-61; HF CLIP linears).  dgrad (dX = dY·W) and wgrad
// (dW = dY^T·X) reach this kernel through lmod_transpose_bf16 (weights are transposed once per
// optimizer step, activations per use) — see DESIGN.md "GEMM forms".
//
// Batched / grouped: `batch` problems with element strides; optional per-batch m_valid[] /
// k_valid[] row / reduction extents make it the MoE expert grouped GEMM over [E, C, H] capacity
// slabs (tiles past an expert's live rows exit immediately) — replaces DeepSpeed's per-expert
// python loop (deepspeed.moe.experts, call site llava_qwen2_moe.py:536-546).
//
// Tile: 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 tiles.
// Staging: buffer_load_dwordx4 ... lds (LDS-DMA), 2 LDS buffers, one barrier per K tile.
// LDS image: [128 rows][8 chunks of 16 B], chunk XOR-swizzled by (row & 7); because LDS-DMA
// writes lane-linear, the swizzle is applied to the per-lane SOURCE address and again on the
// ds_read side (same involution).  B rows are additionally permuted at staging so that, with
// the MFMA operands swapped (D = B_frag x A_frag = C^T tile), every lane ends up holding 16
// CONTIGUOUS output columns of one output row -> 16-byte coalesced epilogue stores.
#include "common.h"
#include <stdlib.h>

struct GemmP {
  const bf16_t* A; const bf16_t* B; void* C; const bf16_t* bias;
  int M, N, K, lda, ldb, ldc;
  int batch; long long sA, sB, sC;
  const int* m_valid; const int* k_valid;
  int act;         // 0 none, 1 exact GELU, 2 quick_gelu
  int out_f32;     // 0: bf16 C, 1: f32 C
  int accumulate;  // C += result (read-modify-write)
  int vec_ok;      // C pointer / ldc allow 16-byte vector stores
  int tiles_m, tiles_n;
  void* C2; int ldc2; long long sC2;   // gemm_swiglu_256 only: optional [M, 2N] gate|up pre-activations
  int splitk, kchunk;                  // MODE 0, batch 1, fp32 accumulate: deterministic split-K (lmod_gemm_wgrad_bf16_nt)
  float* ws; int* counters;            //   ws [splitk, M, N] partial tiles, counters [tiles] arrival semaphores (self-resetting)
  // MODE 5 (lmod_gemm_qkv_rope_bf16): rotary embedding of head-dim-128 heads in the epilogue
  const bf16_t* rope_cos; const bf16_t* rope_sin; const int* rope_pos; int rope_cols;
  int ptotal;                          // persistent gemm4 launches: tiles in all (the grid is min(ptotal, CUs))
  // gemm4t_kernel<2> (attention dQ from the spilled dS^T, lmod_launch_attn_dq_gemm): A = dS^T [B * nh][S keys][S queries],
  // B = the K rows of the QKV buffer, C = dQ; rope_* above as in MODE 5
  int at_nh, at_group, at_S, at_nqb, at_causal; float at_scale; const int* at_seqlens;
};

#define GEMM_OOB 0x80000000u
#ifndef G256_NT_C
#define G256_NT_C 0                // 1: the 256-tile kernels store their output tiles non-temporally (streamed past the L2's LRU)
#endif
__device__ __forceinline__ void st_c(void* p, const u32x4 v) {
#if G256_NT_C
  __builtin_nontemporal_store(v, (u32x4*)p);
#else
  *(u32x4*)p = v;
#endif
}

// Fused SwiGLU epilogue (gemm_swiglu_256): v[0..7] are 8 gate pre-activations, v[8..15] the matching 8 up values.
// Same roundings as GEMM + the separate kernel: bf16 gate/up, bf16 silu, bf16 product (qwen2/modeling_qwen2.py:186-187).
// bf16 roundings go through v_cvt_pk_bf16_f32 on PAIRS (one convert + two unpacks per pair; rounding one value at a time was
// a convert + a shift each): gp / up are the packed bf16 pre-activations, the very words a training forward also stores.
__device__ __forceinline__ u32x4 swiglu_pairs(const float (&v)[16], u32x4& gp, u32x4& up) {
  u32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    gp[k] = pack2bf(v[2 * k], v[2 * k + 1]);
    up[k] = pack2bf(v[8 + 2 * k], v[8 + 2 * k + 1]);
    const uint32_t sw = pack2bf(fast_silu(bflo(gp[k])), fast_silu(bfhi(gp[k])));
    o[k] = pack2bf(bflo(sw) * bflo(up[k]), bfhi(sw) * bfhi(up[k]));
  }
  return o;
}
__device__ __forceinline__ u32x4 swiglu_pairs(const float (&v)[16]) {
  u32x4 gp, up;
  return swiglu_pairs(v, gp, up);
}

// SwiGLU backward on 8 (dact, gate, up) triples packed as bf16 pairs — the arithmetic of swiglu_bwd_kernel (rowops.hip).
__device__ __forceinline__ void swiglu_bwd8(const float (&d)[8], const u32x4 g, const u32x4 u, u32x4& og, u32x4& ou) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float gg[2] = {bflo(g[k]), bfhi(g[k])}, uu[2] = {bflo(u[k]), bfhi(u[k])};
    const uint32_t dw = pack2bf(d[2 * k], d[2 * k + 1]);      // the standalone path rounds dact to bf16 in HBM
    const float dd2[2] = {bflo(dw), bfhi(dw)};
    float rg[2], ru[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float dd = dd2[e];
      const float sg = fast_sigmoid(gg[e]);
      rg[e] = dd * uu[e] * (sg * (1.f + gg[e] * (1.f - sg)));
      ru[e] = dd * (gg[e] * sg);
    }
    og[k] = pack2bf(rg[0], rg[1]);
    ou[k] = pack2bf(ru[0], ru[1]);
  }
}

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == 1) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
  if (act == 2) return v * fast_sigmoid(1.702f * v);
  return v;
}


// Grouped launch with per-batch live row counts: map the linear block id onto the LIVE (batch, row-tile,
// col-tile) space only, so the launch behaves like one dense GEMM over the concatenated live rows (same
// XCD-contiguous, M-grouped walk, no dead tiles inside any XCD's chunk).  Blocks past the live count exit.
// Returns false if this block has no work.  batch <= GEMM_MAX_GROUPS.
#define GEMM_MAX_GROUPS 16
template <int T, int GROUP_M>
__device__ __forceinline__ bool grouped_tile(const GemmP& p, int& bz, int& tm, int& tn) {
  int live[GEMM_MAX_GROUPS];
  int total = 0;
#pragma unroll
  for (int e = 0; e < GEMM_MAX_GROUPS; ++e) {
    live[e] = 0;
    if (e < p.batch) { live[e] = (min(p.m_valid[e], p.M) + T - 1) / T; total += live[e]; }
  }
  const int nlive = total * p.tiles_n;
  if ((int)blockIdx.x >= nlive) return false;
  const int id = xcd_remap(blockIdx.x, nlive);
  const int grp = id / (GROUP_M * p.tiles_n);
  const int first_m = grp * GROUP_M;
  const int gsz = min(total - first_m, GROUP_M);
  const int rr = id - grp * GROUP_M * p.tiles_n;
  int vr = first_m + rr % gsz;                      // virtual row tile over the concatenated live rows
  tn = rr / gsz;
  bz = 0;
#pragma unroll
  for (int e = 0; e < GEMM_MAX_GROUPS; ++e) {
    if (e < p.batch && bz == e && vr >= live[e]) { vr -= live[e]; bz = e + 1; }
  }
  tm = vr;
  return true;
}

// The same mapping for a PERSISTENT workgroup (gemm4_kernel<MODE, true, true>): the per-batch live row-tile counts were read once
// at kernel entry — before any store of the launch, so they are scalar loads and every derived value stays in SGPRs — and the
// tile id is the walk's own (`bid`), not blockIdx.x.  `nlive` = total * tiles_n.
template <int GROUP_M>
__device__ __forceinline__ bool grouped_tile_pre(const GemmP& p, const int (&rows)[GEMM_MAX_GROUPS], const int total, const int nlive,
                                                 const int bid, int& bz, int& tm, int& tn) {
  if (bid >= nlive) return false;
  const int id = xcd_remap(bid, nlive);
  const int grp = id / (GROUP_M * p.tiles_n);
  const int first_m = grp * GROUP_M;
  const int gsz = min(total - first_m, GROUP_M);
  const int rr = id - grp * GROUP_M * p.tiles_n;
  int vr = first_m + rr % gsz;
  tn = rr / gsz;
  bz = 0;
#pragma unroll
  for (int e = 0; e < GEMM_MAX_GROUPS; ++e) {
    const int live = (rows[e] + 255) >> 8;          // (only the live ROWS are carried: 16 scalars instead of 32)
    if (e < p.batch && bz == e && vr >= live) { vr -= live; bz = e + 1; }
  }
  tm = vr;
  return true;
}

__global__ __launch_bounds__(256, 2) void gemm_nt_128(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x (A 16 KiB + B 16 KiB)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- tile coordinates (XCD-contiguous, grouped along M for L2 panel reuse) ----
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int tpb = p.tiles_m * p.tiles_n;
  int bz = id / tpb;
  const int r = id - bz * tpb;
  const int GROUP_M = 8;
  const int grp = r / (GROUP_M * p.tiles_n);
  const int first_m = grp * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int rr = r - grp * GROUP_M * p.tiles_n;
  int tm = first_m + rr % gsz, tn = rr / gsz;
  if (p.m_valid) {
    if (p.batch <= GEMM_MAX_GROUPS) {
      if (!grouped_tile<128, 8>(p, bz, tm, tn)) return;
    } else {      // many groups: (group, row tile) fastest, N slowest keeps every XCD chunk equally live
      const int per_col = p.batch * p.tiles_m;
      tn = id / per_col;
      const int rem = id - tn * per_col;
      bz = rem / p.tiles_m;
      tm = rem - bz * p.tiles_m;
    }
  }
  int Mv = p.m_valid ? min(p.m_valid[bz], p.M) : p.M;
  int Kv = p.k_valid ? min(p.k_valid[bz], p.K) : p.K;
  const int row0 = tm * 128, col0 = tn * 128;
  if (row0 >= Mv) return;

  const bf16_t* Ab = p.A + (long long)bz * p.sA + (long long)row0 * p.lda;
  const bf16_t* Bb = p.B + (long long)bz * p.sB + (long long)col0 * p.ldb;
  const int rowsA = min(128, Mv - row0), rowsB = min(128, p.N - col0);
  const int Kv8 = (Kv + 7) & ~7;   // chunks are 8 elements; K % 8 == 0 so Kv8 <= K <= ld
  const uint32_t bytesA = Kv > 0 ? (uint32_t)(((long long)(rowsA - 1) * p.lda + Kv8) * 2) : 0u;
  const uint32_t bytesB = Kv > 0 ? (uint32_t)(((long long)(rowsB - 1) * p.ldb + Kv8) * 2) : 0u;
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)bytesA, 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, (int)bytesB, 0x00020000);

  // ---- per-lane staging offsets (constant over the K loop; K advances through soffset) ----
  // wave-load j of this wave fills LDS rows wave*32 + j*8 + (lane>>3), physical chunk lane&7.
  const int cchunk = (lane & 7) ^ (lane >> 3);          // logical K chunk this lane fetches
  uint32_t voA[4], voB[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ra = wave * 32 + j * 8 + (lane >> 3);      // LDS row == tile row for A
    voA[j] = (ra < rowsA) ? (uint32_t)((ra * p.lda + cchunk * 8) * 2) : GEMM_OOB;
    const int rl = ra & 63, strip = ra >> 6;             // LDS row -> global column (permuted)
    const int ii = rl & 15, ntl = rl >> 4;
    const int nloc = strip * 64 + (ii >> 2) * 16 + ntl * 4 + (ii & 3);
    voB[j] = (nloc < rowsB) ? (uint32_t)((nloc * p.ldb + cchunk * 8) * 2) : GEMM_OOB;
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nkt = (Kv + 63) >> 6;

  auto stage = [&](int t, int buf) {
    const int k0 = t * 64;
    char* sa = smem + buf * 32768 + wave * 4096;        // 32 rows x 128 B per wave
    char* sb = sa + 16384;
    const bool tail = (k0 + 64 > Kv);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t va = voA[j], vb = voB[j];
      if (tail && (k0 + cchunk * 8 >= Kv)) { va = GEMM_OOB; vb = GEMM_OOB; }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(sa + j * 1024), 16, va, k0 * 2, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(sb + j * 1024), 16, vb, k0 * 2, 0, 0);
    }
  };

  // per-lane ds_read offsets: row (lane&15), logical chunk kk*4 + (lane>>4), swizzled by row&7.
  const int rdrow = (lane & 15) * 128;
  const int ph0 = (((lane >> 4)) ^ (lane & 7)) * 16;
  const int ph1 = ((4 + (lane >> 4)) ^ (lane & 7)) * 16;

  if (nkt > 0) stage(0, 0);
  for (int t = 0; t < nkt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < nkt) stage(t + 1, (t + 1) & 1);
    const char* sa = smem + (t & 1) * 32768 + wm * 8192 + rdrow;
    const char* sb = smem + (t & 1) * 32768 + 16384 + wn * 8192 + rdrow;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ph = kk ? ph1 : ph0;
      bf16x8 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8*)(sa + i * 2048 + ph);
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = *(const bf16x8*)(sb + i * 2048 + ph);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[nt], a[mt], acc[mt][nt], 0, 0, 0);
    }
  }

  // ---- epilogue: lane holds, for each mt, row (lane&15) and 16 contiguous columns ----
  const int g = lane >> 4;
  const int cb = col0 + wn * 64 + g * 16;
  float bia[16];
#pragma unroll
  for (int x = 0; x < 16; ++x) bia[x] = 0.f;
  if (p.bias) {   // clamped index + select: no per-element branches
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      const float bv = bf2f(p.bias[min(cb + x, p.N - 1)]);
      bia[x] = (cb + x < p.N) ? bv : 0.f;
    }
  }
  char* Cb = (char*)p.C + (long long)bz * p.sC * (p.out_f32 ? 4 : 2);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int row = row0 + wm * 64 + mt * 16 + (lane & 15);
    if (row >= Mv) continue;
    float v[16];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) v[nt * 4 + q] = acc[mt][nt][q] + bia[nt * 4 + q];
    if (p.act) {
#pragma unroll
      for (int x = 0; x < 16; ++x) v[x] = act_apply(bfround(v[x]), p.act);
    }
    const bool full = (cb + 16 <= p.N) && p.vec_ok;
    if (p.out_f32) {
      float* cp = (float*)Cb + (long long)row * p.ldc + cb;
      if (full) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          f32x4 o = {v[4 * x], v[4 * x + 1], v[4 * x + 2], v[4 * x + 3]};
          if (p.accumulate) { f32x4 old = *(f32x4*)(cp + 4 * x); o += old; }
          *(f32x4*)(cp + 4 * x) = o;
        }
      } else {
#pragma unroll
        for (int x = 0; x < 16; ++x) if (cb + x < p.N) cp[x] = p.accumulate ? cp[x] + v[x] : v[x];
      }
    } else {
      bf16_t* cp = (bf16_t*)Cb + (long long)row * p.ldc + cb;
      if (full && !p.accumulate) {
        u32x4 o0 = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
        u32x4 o1 = {pack2bf(v[8], v[9]), pack2bf(v[10], v[11]), pack2bf(v[12], v[13]), pack2bf(v[14], v[15])};
        *(u32x4*)(cp) = o0;
        *(u32x4*)(cp + 8) = o1;
      } else {
#pragma unroll
        for (int x = 0; x < 16; ++x)
          if (cb + x < p.N) cp[x] = f2bf(p.accumulate ? bf2f(cp[x]) + v[x] : v[x]);
      }
    }
  }
}


// =============================================================================================
// 256 x 256 x 64 tile, 8 waves (2 along M x 4 along N, 128 x 64 per wave), phase-pipelined.
//
// LDS: 8 half-tile slots of 16 KiB = 2 (double buffer) x {A(mh0), A(mh1), B(nh0), B(nh1)}; a half-tile is
// 128 rows x 128 B.  A(mh) holds, for both wave rows, the 64 tile rows of M-half mh; B(nh) holds, for the
// four wave columns, the 32 (permuted) tile columns of N-half nh.  Each K tile is consumed in 4 phases, one
// 64x32 output quadrant of every wave per phase (16 MFMAs):
//     P1 (mh0,nh0): read A(mh0) 8 + B(nh0) 4      prefetch A(mh0) of tile t+1
//     P2 (mh0,nh1): read B(nh1) 4                 prefetch B(nh0)
//     P3 (mh1,nh1): read A(mh1) 8                 prefetch B(nh1)
//     P4 (mh1,nh0): read B(nh0) 4                 prefetch A(mh1)
// so every half-tile is issued >= 3 phases before its first read and 3 half-tiles (6 LDS-DMA loads per
// thread) stay in flight across the barriers: waits are counted (s_waitcnt vmcnt(4)), never 0, and are
// placed one phase BEFORE the read they guard (LDS-DMA data is ordered for other waves' ds_reads only by
// the issuing wave's vmcnt followed by a barrier the reader has passed).  Barriers are raw s_barrier
// (no fence => no implicit vmcnt(0)).  The two wave rows run staggered by one barrier so that on every
// SIMD one wave is in its MFMA cluster (s_setprio 1) while the other issues ds_reads / LDS-DMA.
// The prefetch is issued for tile t+1 even past the end of K (all lanes out of bounds -> zero fill, no
// memory traffic), which keeps the vmcnt bookkeeping uniform.
// =============================================================================================
#define G256_SLOT 16384
#ifndef G256_GROUP_M
#define G256_GROUP_M 4
#endif
#ifndef G256_PRIO
#define G256_PRIO 1
#endif
#ifndef G256_STAGGER
#define G256_STAGGER 1
#endif
// Main-loop experiment knobs (tools/build_gemm_variants.sh builds alt_libs/ with them; tools/gemm_variants_ab.py times the
// builds in ONE process beside a power / clock trace).  Defaults = the shipped schedule.
//   G256_STAGE_POS  where a phase issues its 2 LDS-DMA prefetch pieces: 0 before the phase's first barrier (with the
//                   ds_reads), 1 right after that barrier (head of the MFMA interval), 2 after 8 of the 16 MFMAs,
//                   3 one piece after 4 and one after 12 MFMAs, 4 one piece with the ds_reads and one after the 16th
//                   MFMA (the MFMA wave's issue slots are free there; it is about to park at the barrier).  (1-3: one
//                   stage fewer is outstanding at the counted waits, so they wait vmcnt(2) where 0 waits vmcnt(4); 4
//                   waits vmcnt(3).)
//   G256_M32        TIMING ONLY (wrong results): the 16 v_mfma_f32_16x16x32_bf16 of a phase replaced by 8
//                   v_mfma_f32_32x32x16_bf16 over the SAME fragment registers / LDS reads / staging — what the
//                   32x32x16 shape does to power, clock and throughput inside this very loop.
//   G256_ABL        TIMING ONLY ablation bit mask: 1 no LDS-DMA in the loop, 2 no ds_reads in the loop, 4 no counted
//                   vmcnt waits, 8 no MFMAs, 16 no second barrier of a phase, 32 no first barrier of a phase.
//   G256_DMA_AUX    cache-policy bits of the LDS-DMA loads (buffer_load ... lds aux): 0 default, 1 sc0, 2 nt, 16 sc1, 17 sc0 sc1
#ifndef G256_DMA_AUX
#define G256_DMA_AUX 0
#endif
//   G256_DEEP       1: deep-prefetch schedule.  Both B half-tile fragment sets stay in registers (+16 VGPRs), so B(nh0) is
//                   read once per tile and EVERY half-tile slot is free two phases after its tile's P1..P3 read; it is then
//                   refilled at once with the half-tile of tile t+2 (the two LDS buffers of a kind alternate as before):
//                       P1: read A(mh0), B(nh0) | prefetch B(nh1) of t+1      P3: read A(mh1) | prefetch A(mh0) of t+2
//                       P2: read B(nh1)         | prefetch A(mh1) of t+1      P4: no reads    | prefetch B(nh0) of t+2
//                   Issue order = consumption order (vmcnt retires in order), 4 half-tiles (8 loads per thread) stay in
//                   flight across the counted waits (vmcnt(8)) instead of 2.
//                   2: deep + the ds_reads of the NEXT phase issued under this phase's MFMAs (after the 4th of 16) into a second A
//                   fragment set (+32 VGPRs): B(nh1) under P1, A(mh1) under P2, A(mh0) of tile t+1 under P3; only B(nh0) is still
//                   read in a reading interval.  Data read under the MFMAs of phase p must have been waited for by EVERY wave one
//                   barrier earlier than before (the two wave rows run one barrier apart): waits are vmcnt(6), 3 half-tiles in flight.
#ifndef G256_DEEP
#define G256_DEEP 1
#endif
//   G256_EARLYBAR   n in {0, 4, 8}: the phase's SECOND barrier is executed after 16-n of the 16 MFMAs have been issued
//                   instead of after all of them.  MFMAs touch no LDS, so every hazard the barrier orders is unchanged; the
//                   other wave of the SIMD (parked at that barrier, operands ready) is released while n MFMAs of this wave
//                   are still queued, which closes the matrix-pipe bubble at every hand-over (8 per K tile).
#ifndef G256_EARLYBAR
#define G256_EARLYBAR 0
#endif
//   G256_DMA_SPLIT  (deep schedule) 1: the odd wave columns issue their LDS-DMA pieces BEFORE their ds_reads, the even ones after
//                   (as before): the four waves of a row then reach the CU's one vector-memory issue port at different times
//                   instead of queueing behind each other (a piece costs its wave 75-100 cycles of issue in the cycle trace).
#ifndef G256_DMA_SPLIT
#define G256_DMA_SPLIT 0
#endif
//   G256_TRACE      instrumentation build: workgroup G256_TRACE_WG stamps the shader cycle counter (s_memtime + lgkmcnt wait: perturbing) at 7 points of each of the 4 phases of K tile G256_TRACE_T into one VGPR
//                   (v_writelane) and dumps it; tools/gemm_trace.py prints the per-wave timeline.
#ifndef G256_TRACE
#define G256_TRACE 0
#endif
#ifndef G256_TRACE_T
#define G256_TRACE_T 24
#endif
#ifndef G256_TRACE_WG
#define G256_TRACE_WG 1000
#endif
#if G256_TRACE
__device__ uint32_t g256_trace_buf[8 * 64];
extern "C" int lmod_debug_gemm_trace(uint32_t* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g256_trace_buf), sizeof(uint32_t) * 8 * 64) == hipSuccess ? 0 : -1;
}
#endif
#ifndef G256_STAGE_POS
#define G256_STAGE_POS 0
#endif
#ifndef G256_M32
#define G256_M32 0
#endif
#ifndef G256_ABL
#define G256_ABL 0
#endif

// MODE 0: plain NT GEMM.  MODE 1 (gemm_swiglu_256): B is the [2N, K] gate-over-up weight; an N tile is 128 output
// columns fed by 128 gate rows (B half-tile nh0) and the matching 128 up rows (nh1), staged so that every lane
// ends up with 8 contiguous gate columns and the same 8 up columns -> C = silu(gate) * up leaves as one 16-byte
// store and the [M, 2N] pre-activations are written only if the backward needs them (C2).
// MODE 2 (wgrad, "TN"): C[M,N] (+)= A^T B with BOTH operands reduction-major as autograd hands them over
// (A = dY [K, M], B = X [K, N], K = tokens): no transposed copies.  A half-tile is a [64 k][128 m] image (256-byte
// rows, 32-byte blocks XOR-swizzled by (k&3 | (k>>3&1)<<2)) filled by LDS-DMA; the MFMA operand fragments (8
// consecutive k for one m per lane) come out of it through ds_read_b64_tr_b16 (hardware transpose read),
// conflict-free.  Columns are in natural order, so a lane owns 4 groups of 4 contiguous output columns.
// MODE 3 (the step's wgrad): A = dY^T K-contiguous as in MODE 0, B = X reduction-major as in MODE 2 — only dY needs
// a transposed copy, and two thirds of the operand fragments keep the full-rate ds_read_b128 path.
// MODE 5 (fused QKV projection + RoPE, lmod_gemm_qkv_rope_bf16): MODE 0's loop; a 256-column tile is two whole heads of
// 128 and a wave column owns 64 of them, so the rotate_half partner (feature f +- 64) of every value sits in the NEIGHBOURING
// wave column at the same lane / register position.  The epilogue rounds acc + bias to bf16 (what the unfused path stores),
// swaps those images between wave columns 2c <-> 2c+1 through the (now idle) 128 KiB of LDS, and applies
// q*cos + rotate_half(q)*sin with the roundings of rope_kernel (rowops.hip) — bit-identical to GEMM + lmod_rope, one pass
// over the QKV buffer less.  Tiles at or past rope_cols (the V heads) are stored as they are.
// K64: the host knows that every reduction length this launch sees is a multiple of 64 (K % 64 == 0, no k_valid): a K tile is wholly
// live or wholly past the end, so "past the end" is a SCALAR choice of descriptor (zero records: no traffic) instead of one
// v_cndmask per LDS-DMA in the MFMA gaps (the 4-wave kernel gained 2 % from the same change).
template <int MODE, bool K64 = false>
__global__ __launch_bounds__(512, 2) void gemm_256_kernel(GemmP p) {
  constexpr int TN = (MODE == 1) ? 128 : 256;
  constexpr bool AK = (MODE == 2), BK = (MODE == 2 || MODE == 3);      // operand stored reduction-major?
  constexpr bool PLAIN = (MODE == 0 || MODE == 6);   // 6 = 0 with the fp32 read-modify-write epilogue in two batches (below)
  constexpr bool SPLIT_OK = (PLAIN || MODE == 3);
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 8 x 16 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  const int tpb = p.tiles_m * p.tiles_n;
  int id = xcd_remap(blockIdx.x, gridDim.x);
  int bz = id / tpb, split = 0;
  if (SPLIT_OK && p.splitk > 1) {      // split-major: an XCD's contiguous chunk is many tiles of ONE K split (same L2 reuse)
    split = bz; bz = 0; id -= split * tpb;
  }
  int r = id - bz * tpb;
  if (PLAIN && p.k_valid && !p.m_valid && p.batch > 1 && p.batch <= GEMM_MAX_GROUPS && p.splitk <= 1 && !(tpb & 7)) {
    // Batched weight gradients with a different live reduction length per batch (MoE experts: k_valid = routed rows).
    // The generic mapping hands each XCD ONE contiguous chunk of (batch, tile) ids, i.e. whole experts: with routed-row
    // counts of 9k / 12k / 20k / 24k the XCDs holding the long experts ran 2.7x longer than the others (880 TF against
    // 1240 TF for the same flops, balanced).  Here every XCD takes 1/8 of EVERY expert's tiles (still a contiguous run of
    // tiles per expert: same L2 reuse), longest reduction first so that the short tiles fill the tail.
    const int tpb8 = tpb >> 3, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int rnk = k / tpb8;
    int sel = rnk;
    for (int e = 0; e < p.batch; ++e) {
      const int ke = p.k_valid[e];
      int rank = 0;
      for (int f = 0; f < p.batch; ++f) {
        const int kf = p.k_valid[f];
        rank += (kf > ke || (kf == ke && f < e)) ? 1 : 0;
      }
      if (rank == rnk) sel = e;
    }
    bz = sel;
    r = xcd * tpb8 + (k - rnk * tpb8);
    id = bz * tpb + r;
  }
  const int GROUP_M = G256_GROUP_M;
  const int grp = r / (GROUP_M * p.tiles_n);
  const int first_m = grp * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int rr = r - grp * GROUP_M * p.tiles_n;
  int tm = first_m + rr % gsz, tn = rr / gsz;
  if (p.m_valid) {   // grouped launch: see gemm_nt_128
    if (p.batch <= GEMM_MAX_GROUPS) {
      if (!grouped_tile<256, G256_GROUP_M>(p, bz, tm, tn)) return;    // (row tiles are 256 in both modes)
    } else {
      const int per_col = p.batch * p.tiles_m;
      tn = id / per_col;
      const int rem = id - tn * per_col;
      bz = rem / p.tiles_m;
      tm = rem - bz * p.tiles_m;
    }
  }
  int Mv = p.m_valid ? min(p.m_valid[bz], p.M) : p.M;
  int Kv = p.k_valid ? min(p.k_valid[bz], p.K) : p.K;
  const int row0 = tm * 256, col0 = tn * TN;
  if (row0 >= Mv) return;
  int kbeg = 0;
  if (SPLIT_OK && p.splitk > 1) {      // long-K, few-tile problems (wgrad): the batch index is the K split
    kbeg = split * p.kchunk;
    Kv = max(0, min(Kv - kbeg, p.kchunk));      // an empty split still arrives at the semaphore with a zero tile
  }

  const bf16_t* Ab = p.A + (long long)bz * p.sA + (AK ? (long long)row0 + (long long)kbeg * p.lda : (long long)row0 * p.lda + kbeg);
  const bf16_t* Bb = p.B + (long long)bz * p.sB + (BK ? (long long)col0 + (long long)kbeg * p.ldb : (long long)col0 * p.ldb + kbeg);
  const int rowsA = min(256, Mv - row0), rowsB = min(TN, p.N - col0);
  const int Kv8 = (Kv + 7) & ~7;
  const uint32_t bytesA = Kv > 0 ? (uint32_t)(((long long)(rowsA - 1) * p.lda + Kv8) * 2) : 0u;
  const uint32_t bytesB = Kv > 0 ? (uint32_t)(((long long)((MODE == 1 ? p.N : 0) + rowsB - 1) * p.ldb + Kv8) * 2) : 0u;
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)bytesA, 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, (int)bytesB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsAz = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, 0, 0x00020000);      // K64: tiles past the end
  const __amdgpu_buffer_rsrc_t rsBz = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, 0, 0x00020000);

  // ---- staging offsets: this wave fills half-tile rows 16*wave + 8*j + (lane>>3), physical chunk lane&7 ----
  const int cchunk = (lane & 7) ^ (lane >> 3);
  uint32_t voA[2][2], voB[2][2];      // [half][j]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      // reduction-major image: this wave fills k rows 8*wave + 4*j + (lane>>4), physical 16-byte chunk lane&15
      const int krow = wave * 8 + j * 4 + (lane >> 4), pc = lane & 15;
      const int key = (krow & 3) | (((krow >> 3) & 1) << 2);
      const int mp = ((((pc >> 1) ^ key) << 1) | (pc & 1)) * 8;        // logical position 0..127 in the half-tile
      // K-contiguous image: this wave fills half-tile rows 16*wave + 8*j + (lane>>3), physical chunk lane&7
      const int hr = wave * 16 + j * 8 + (lane >> 3);                 // half-tile row 0..127
      if constexpr (AK) {
        const int ra = (mp >> 6) * 128 + h * 64 + (mp & 63);           // A: tile row (both wave rows' M-half h)
        voA[h][j] = (ra < rowsA) ? (uint32_t)((krow * p.lda + ra) * 2) : GEMM_OOB;
      } else {
        const int ra = (hr >> 6) * 128 + h * 64 + (hr & 63);           // A: tile row
        voA[h][j] = (ra < rowsA) ? (uint32_t)((ra * p.lda + cchunk * 8) * 2) : GEMM_OOB;
      }
      const int wcs = hr >> 5, rl = hr & 31, ntl = rl >> 4, ii = rl & 15;
      if constexpr (BK) {
        const int nb = (mp >> 5) * 64 + h * 32 + (mp & 31);            // B: tile column (natural order)
        voB[h][j] = (nb < rowsB) ? (uint32_t)((krow * p.ldb + nb) * 2) : GEMM_OOB;
      } else if (PLAIN || MODE == 4 || MODE == 5) {
        const int nloc = wcs * 64 + (ii >> 2) * 16 + (h * 2 + ntl) * 4 + (ii & 3);   // B: permuted tile column
        voB[h][j] = (nloc < rowsB) ? (uint32_t)((nloc * p.ldb + cchunk * 8) * 2) : GEMM_OOB;
      } else {       // half h = gate (0) / up (1) rows of the same 128 output columns
        const int nloc = wcs * 32 + (ii >> 2) * 8 + ntl * 4 + (ii & 3);
        voB[h][j] = (nloc < rowsB) ? (uint32_t)((((long long)h * p.N + nloc) * p.ldb + cchunk * 8) * 2) : GEMM_OOB;
      }
    }

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nkt = (Kv + 63) >> 6;

  // kind: 0 A(mh0), 1 A(mh1), 2 B(nh0), 3 B(nh1); tile index t (may be >= nkt: fully out of bounds)
  // pieces jlo..jhi-1 (of 2) of half-tile `kind` of tile t
  auto stage_j = [&](int kind, int t, int jlo, int jhi) {
    const int k0 = t * 64;
    char* dst = smem + (t & 1) * (4 * G256_SLOT) + kind * G256_SLOT + wave * 2048;
    const bool kmaj = (kind < 2) ? AK : BK;
    if (kmaj) {
      // the K advance goes into the (64-bit) base: byte offsets of a token-major operand overflow 32 bits
      const long long adv = (long long)k0 * (kind < 2 ? p.lda : p.ldb);
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((kind < 2 ? Ab : Bb) + adv), 0,
                                                                    (int)0x80000000u, 0x00020000);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (j < jlo || j >= jhi) continue;
        uint32_t v = (kind < 2) ? voA[kind & 1][j] : voB[kind & 1][j];
        if (k0 + wave * 8 + j * 4 + (lane >> 4) >= Kv) v = GEMM_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(dst + j * 1024), 16, v, 0, 0, G256_DMA_AUX);
      }
    } else if constexpr (K64) {
      const bool past = (k0 >= Kv);                         // wave-uniform
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (j < jlo || j >= jhi) continue;
        const uint32_t v = (kind < 2) ? voA[kind & 1][j] : voB[kind & 1][j];
        if (kind < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(past ? rsAz : rsA, LDS_PTR(dst + j * 1024), 16, v, k0 * 2, 0, G256_DMA_AUX);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(past ? rsBz : rsB, LDS_PTR(dst + j * 1024), 16, v, k0 * 2, 0, G256_DMA_AUX);
      }
    } else {
      const bool dead = (k0 + cchunk * 8 >= Kv);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (j < jlo || j >= jhi) continue;
        uint32_t v = (kind < 2) ? voA[kind & 1][j] : voB[kind & 1][j];
        if (dead) v = GEMM_OOB;
        if (kind < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(dst + j * 1024), 16, v, k0 * 2, 0, G256_DMA_AUX);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(dst + j * 1024), 16, v, k0 * 2, 0, G256_DMA_AUX);
      }
    }
  };
  auto stage = [&](int kind, int t) { stage_j(kind, t, 0, 2); };

  const int li = lane & 15;
  const int rdrow = li * 128;
  const int ph0 = ((lane >> 4) ^ (lane & 7)) * 16;
  const int ph1 = ((4 + (lane >> 4)) ^ (lane & 7)) * 16;

  bf16x8 af2[2][4][2], bfr2[2][2][2];  // bfr2[nh] / af2[mh]: fragments of B half-tile nh / A half-tile mh (classic: set 0 only; deep: both B sets; deep 2: both A sets too)
#define BFR(NH) bfr2[G256_DEEP ? (NH) : 0]
#define AFR(MH) af2[G256_DEEP == 2 ? (MH) : 0]
  // MODE 2: per-lane pieces of the transposing read (see the kernel header): k row g*8 + (i>>2) (+4 for the second
  // half of the fragment, +32 per k-step), 8 bytes at (i&3)*8 inside the 32-byte block (m-tile index ^ key)
  const int trk = ((lane >> 4) * 8 + (li >> 2)) * 256 + (li & 3) * 8;
  const int trkey = ((li >> 2) & 3) | (((lane >> 4) & 1) << 2);
  auto read_tr16 = [&](const char* slot, int blk, int kk) -> bf16x8 {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    const char* q = slot + kk * 8192 + trk + ((blk ^ trkey) << 5);
    const s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)q);
    const s16x4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(q + 1024));
    return (bf16x8){x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
  };
  auto readA = [&](int t, int mh) {
    if constexpr (AK) {
      const char* sl = smem + (t & 1) * (4 * G256_SLOT) + mh * G256_SLOT;
#pragma unroll
      for (int ml = 0; ml < 4; ++ml) {
        AFR(mh)[ml][0] = read_tr16(sl, wr * 4 + ml, 0);
        AFR(mh)[ml][1] = read_tr16(sl, wr * 4 + ml, 1);
      }
      return;
    }
    const char* s = smem + (t & 1) * (4 * G256_SLOT) + mh * G256_SLOT + wr * 8192 + rdrow;
#pragma unroll
    for (int ml = 0; ml < 4; ++ml) {
      AFR(mh)[ml][0] = *(const bf16x8*)(s + ml * 2048 + ph0);
      AFR(mh)[ml][1] = *(const bf16x8*)(s + ml * 2048 + ph1);
    }
  };
  auto readB = [&](int t, int nh) {
    if constexpr (BK) {
      const char* sl = smem + (t & 1) * (4 * G256_SLOT) + (2 + nh) * G256_SLOT;
#pragma unroll
      for (int nl = 0; nl < 2; ++nl) {
        BFR(nh)[nl][0] = read_tr16(sl, wc * 2 + nl, 0);
        BFR(nh)[nl][1] = read_tr16(sl, wc * 2 + nl, 1);
      }
      return;
    }
    const char* s = smem + (t & 1) * (4 * G256_SLOT) + (2 + nh) * G256_SLOT + wc * 4096 + rdrow;
#pragma unroll
    for (int nl = 0; nl < 2; ++nl) {
      BFR(nh)[nl][0] = *(const bf16x8*)(s + nl * 2048 + ph0);
      BFR(nh)[nl][1] = *(const bf16x8*)(s + nl * 2048 + ph1);
    }
  };
#define G256_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#if G256_TRACE
  uint32_t vtrace = 0;
  // gfx950 has no SHADER_CYCLES hwreg: s_memtime + its lgkmcnt wait (which also drains the wave's pending ds_reads: the stamps
  // perturb the schedule they measure by that much)
#define G256_STAMP(IDX) do { if (t == G256_TRACE_T) { unsigned long long c_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c_) :: "memory"); \
                                                      asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(vtrace) : "s"((uint32_t)c_), "n"(IDX)); } } while (0)
#else
#define G256_STAMP(IDX) do { } while (0)
#endif
#define G256_BARRIER2() do { if (!(G256_ABL & 16)) G256_BARRIER(); } while (0)
#define G256_BARRIER1() do { if (!(G256_ABL & 32)) G256_BARRIER(); } while (0)
  // loop-side staging by position (see the knob list above); LOOPSTAGE(pos, ...) issues only if this build stages at `pos`
#define G256_LOOPSTAGE(POS, KIND, T, JLO, JHI) do { if (G256_STAGE_POS == (POS) && !(G256_ABL & 1)) stage_j(KIND, T, JLO, JHI); } while (0)
#define G256_VMWAIT() do { if (!(G256_ABL & 4)) { if (G256_STAGE_POS == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  \
                                                  else if (G256_STAGE_POS == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); \
                                                  else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); } } while (0)
#if G256_M32
  typedef __attribute__((ext_vector_type(16))) float f32x16_t;
  f32x16_t acc32[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc32[i][j][q] = 0.f;
#define G256_MFMA_BODY(MH, NH, KK, MLLO, MLHI)                                                              \
      _Pragma("unroll") for (int ml = (MLLO); ml < (MLHI); ++ml)                                            \
        acc32[(MH) * 2 + (ml & 1)][NH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                           \
            BFR(NH)[ml >> 1][KK], AFR(MH)[ml][KK], acc32[(MH) * 2 + (ml & 1)][NH], 0, 0, 0);
#else
#define G256_MFMA_BODY(MH, NH, KK, MLLO, MLHI)                                                              \
      _Pragma("unroll") for (int ml = (MLLO); ml < (MLHI); ++ml)                                            \
        _Pragma("unroll") for (int nl = 0; nl < 2; ++nl)                                                    \
          acc[(MH) * 4 + ml][(NH) * 2 + nl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                      \
              BFR(NH)[nl][KK], AFR(MH)[ml][KK], acc[(MH) * 4 + ml][(NH) * 2 + nl], 0, 0, 0);
#endif
#define G256_MFMA(MH, NH, KIND, T)                                                                          \
  do {                                                                                                      \
    G256_STAMP(ph_ * 8 + 4);                                                                                \
    G256_LOOPSTAGE(1, KIND, T, 0, 2);                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
    G256_STAMP(ph_ * 8 + 5);                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (G256_PRIO) __builtin_amdgcn_s_setprio(1);                                                           \
    if (!(G256_ABL & 8)) {                                                                                  \
      G256_MFMA_BODY(MH, NH, 0, 0, 2)                                                                       \
      if (G256_DEEP == 2) { __builtin_amdgcn_sched_barrier(0); G256_UNDER_MFMA(MH, NH); __builtin_amdgcn_sched_barrier(0); } \
      if (G256_STAGE_POS == 3) { __builtin_amdgcn_sched_barrier(0); G256_LOOPSTAGE(3, KIND, T, 0, 1); __builtin_amdgcn_sched_barrier(0); } \
      G256_MFMA_BODY(MH, NH, 0, 2, 4)                                                                       \
      if (G256_STAGE_POS == 2) { __builtin_amdgcn_sched_barrier(0); G256_LOOPSTAGE(2, KIND, T, 0, 2); __builtin_amdgcn_sched_barrier(0); } \
      if (G256_EARLYBAR == 8) { __builtin_amdgcn_sched_barrier(0); G256_BARRIER2(); __builtin_amdgcn_sched_barrier(0); } \
      G256_MFMA_BODY(MH, NH, 1, 0, 2)                                                                       \
      if (G256_STAGE_POS == 3) { __builtin_amdgcn_sched_barrier(0); G256_LOOPSTAGE(3, KIND, T, 1, 2); __builtin_amdgcn_sched_barrier(0); } \
      if (G256_EARLYBAR == 4) { __builtin_amdgcn_sched_barrier(0); G256_BARRIER2(); __builtin_amdgcn_sched_barrier(0); } \
      G256_MFMA_BODY(MH, NH, 1, 2, 4)                                                                       \
    } else {                                                                                                \
      G256_LOOPSTAGE(2, KIND, T, 0, 2); G256_LOOPSTAGE(3, KIND, T, 0, 2);                                   \
    }                                                                                                       \
    if (G256_PRIO) __builtin_amdgcn_s_setprio(0);                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    G256_STAMP(ph_ * 8 + 6);                                                                                \
    G256_LOOPSTAGE(4, KIND, T, 1, 2);                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (G256_EARLYBAR == 0 || (G256_ABL & 8)) G256_BARRIER2();                                              \
  } while (0)
#define G256_RA(T, MH) do { if (!(G256_ABL & 2)) readA(T, MH); } while (0)
#define G256_RB(T, NH) do { if (!(G256_ABL & 2)) readB(T, NH); } while (0)

#if G256_DEEP == 2
  stage(0, 0); stage(2, 0); stage(3, 0); stage(1, 0); stage(0, 1); stage(2, 1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");        // all of tile 0 landed
  G256_BARRIER();
  readA(0, 0);                                            // A(mh0)_0: in the steady state it is read under P3 of the previous tile
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  G256_BARRIER();
  if (G256_STAGGER && wr == 1) G256_BARRIER();
#define G256_DSTAGE(KIND, T) do { if (!(G256_ABL & 1)) stage(KIND, T); } while (0)
#define G256_DWAIT() do { if (!(G256_ABL & 4)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); } while (0)
  for (int t = 0; t < nkt; ++t) {
    // ---- P1: quadrant (mh0, nh0); under its MFMAs: read B(nh1)_t ----
    G256_RB(t, 0);
    G256_DSTAGE(3, t + 1);
    G256_DWAIT();                                         // retires A(mh1)_t (read under P2's MFMAs)
    G256_BARRIER1();
#define G256_UNDER_MFMA(MH, NH) G256_RB(t, 1)
    G256_MFMA(0, 0, 0, 0);
#undef G256_UNDER_MFMA
    // ---- P2: quadrant (mh0, nh1); under its MFMAs: read A(mh1)_t ----
    G256_DSTAGE(1, t + 1);
    G256_DWAIT();                                         // retires A(mh0)_{t+1} (read under P3's MFMAs)
    G256_BARRIER1();
#define G256_UNDER_MFMA(MH, NH) G256_RA(t, 1)
    G256_MFMA(0, 1, 0, 0);
#undef G256_UNDER_MFMA
    // ---- P3: quadrant (mh1, nh1); under its MFMAs: read A(mh0)_{t+1} ----
    G256_DSTAGE(0, t + 2);
    G256_BARRIER1();
#define G256_UNDER_MFMA(MH, NH) G256_RA(t + 1, 0)
    G256_MFMA(1, 1, 0, 0);
#undef G256_UNDER_MFMA
    // ---- P4: quadrant (mh1, nh0) ----
    G256_DSTAGE(2, t + 2);
    G256_DWAIT();                                         // retires B(nh0)_{t+1} (read in the next P1) and B(nh1)_{t+1} (read under its MFMAs)
    G256_BARRIER1();
#define G256_UNDER_MFMA(MH, NH) do { } while (0)
    G256_MFMA(1, 0, 0, 0);
#undef G256_UNDER_MFMA
  }
#undef G256_DSTAGE
#undef G256_DWAIT
#elif G256_DEEP
#define G256_UNDER_MFMA(MH, NH) do { } while (0)
  // deep-prefetch schedule (see the knob list): prologue = what the steady state would have issued before P1 of tile 0
  stage(0, 0); stage(2, 0); stage(3, 0); stage(1, 0); stage(0, 1); stage(2, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // A(mh0)_0, B(nh0)_0 landed
  G256_BARRIER();
  if (G256_STAGGER && wr == 1) G256_BARRIER();
#define G256_DSTAGE(KIND, T) do { if (!(G256_ABL & 1) && !(G256_DMA_SPLIT && (wc & 1))) stage(KIND, T); } while (0)
#define G256_DSTAGE_EARLY(KIND, T) do { if (!(G256_ABL & 1) && G256_DMA_SPLIT && (wc & 1)) stage(KIND, T); } while (0)
#define G256_DWAIT() do { if (!(G256_ABL & 4)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); } while (0)
  for (int t = 0; t < nkt; ++t) {
    // ---- P1: quadrant (mh0, nh0) ----
    { constexpr int ph_ = 0; G256_STAMP(ph_ * 8 + 0);
    G256_DSTAGE_EARLY(3, t + 1);
    G256_RA(t, 0); G256_RB(t, 0);
    G256_STAMP(ph_ * 8 + 1);
    G256_DSTAGE(3, t + 1);                                // slot last read in P2 of tile t-1
    G256_STAMP(ph_ * 8 + 2);
    G256_DWAIT();                                         // retires B(nh1)_t for P2
    G256_STAMP(ph_ * 8 + 3);
    G256_BARRIER1();
    G256_MFMA(0, 0, 0, 0);
    G256_STAMP(ph_ * 8 + 7); }
    // ---- P2: quadrant (mh0, nh1) ----
    { constexpr int ph_ = 1; G256_STAMP(ph_ * 8 + 0);
    G256_DSTAGE_EARLY(1, t + 1);
    G256_RB(t, 1);
    G256_STAMP(ph_ * 8 + 1);
    G256_DSTAGE(1, t + 1);                                // slot last read in P3 of tile t-1
    G256_STAMP(ph_ * 8 + 2);
    G256_DWAIT();                                         // retires A(mh1)_t for P3
    G256_STAMP(ph_ * 8 + 3);
    G256_BARRIER1();
    G256_MFMA(0, 1, 0, 0);
    G256_STAMP(ph_ * 8 + 7); }
    // ---- P3: quadrant (mh1, nh1) ----
    { constexpr int ph_ = 2; G256_STAMP(ph_ * 8 + 0);
    G256_DSTAGE_EARLY(0, t + 2);
    G256_RA(t, 1);
    G256_STAMP(ph_ * 8 + 1);
    G256_DSTAGE(0, t + 2);                                // slot last read in P1 of this tile
    G256_STAMP(ph_ * 8 + 2);
    G256_STAMP(ph_ * 8 + 3);
    G256_BARRIER1();
    G256_MFMA(1, 1, 0, 0);
    G256_STAMP(ph_ * 8 + 7); }
    // ---- P4: quadrant (mh1, nh0): B(nh0) fragments are still in registers ----
    { constexpr int ph_ = 3; G256_STAMP(ph_ * 8 + 0);
    G256_DSTAGE_EARLY(2, t + 2);
    G256_STAMP(ph_ * 8 + 1);
    G256_DSTAGE(2, t + 2);                                // slot last read in P1 of this tile
    G256_STAMP(ph_ * 8 + 2);
    G256_DWAIT();                                         // retires A(mh0)_{t+1}, B(nh0)_{t+1} for the next P1
    G256_STAMP(ph_ * 8 + 3);
    G256_BARRIER1();
    G256_MFMA(1, 0, 0, 0);
    G256_STAMP(ph_ * 8 + 7); }
  }
#undef G256_DSTAGE
#undef G256_DSTAGE_EARLY
#undef G256_DWAIT
#else
#define G256_UNDER_MFMA(MH, NH) do { } while (0)
  // prologue: tile 0 in issue order A(mh0), B(nh0), B(nh1), A(mh1); first reads need the first two
  stage(0, 0); stage(2, 0); stage(3, 0); stage(1, 0);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  G256_BARRIER();
  if (G256_ABL & 2) {                 // ablation: fragments are read once, here (after the whole first tile has landed)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G256_BARRIER();
    readA(0, 0); readB(0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (G256_STAGGER && wr == 1) G256_BARRIER();   // stagger the second wave row by one barrier

  for (int t = 0; t < nkt; ++t) {
    // ---- P1: quadrant (mh0, nh0) ----
    { constexpr int ph_ = 0; G256_STAMP(ph_ * 8 + 0);
    G256_RA(t, 0); G256_RB(t, 0);
    G256_STAMP(ph_ * 8 + 1);
    G256_LOOPSTAGE(0, 0, t + 1, 0, 2); G256_LOOPSTAGE(4, 0, t + 1, 0, 1);
    G256_STAMP(ph_ * 8 + 2);
    G256_VMWAIT();                                        // retires B(nh1)_t for P2
    G256_STAMP(ph_ * 8 + 3);
    G256_BARRIER1();
    G256_MFMA(0, 0, 0, t + 1);
    G256_STAMP(ph_ * 8 + 7); }
    // ---- P2: quadrant (mh0, nh1) ----
    { constexpr int ph_ = 1; G256_STAMP(ph_ * 8 + 0);
    G256_RB(t, 1);
    G256_STAMP(ph_ * 8 + 1);
    G256_LOOPSTAGE(0, 2, t + 1, 0, 2); G256_LOOPSTAGE(4, 2, t + 1, 0, 1);
    G256_STAMP(ph_ * 8 + 2);
    G256_VMWAIT();                                        // retires A(mh1)_t for P3
    G256_STAMP(ph_ * 8 + 3);
    G256_BARRIER1();
    G256_MFMA(0, 1, 2, t + 1);
    G256_STAMP(ph_ * 8 + 7); }
    // ---- P3: quadrant (mh1, nh1) ----
    { constexpr int ph_ = 2; G256_STAMP(ph_ * 8 + 0);
    G256_RA(t, 1);
    G256_STAMP(ph_ * 8 + 1);
    G256_LOOPSTAGE(0, 3, t + 1, 0, 2); G256_LOOPSTAGE(4, 3, t + 1, 0, 1);
    G256_STAMP(ph_ * 8 + 2);
    G256_STAMP(ph_ * 8 + 3);
    G256_BARRIER1();
    G256_MFMA(1, 1, 3, t + 1);
    G256_STAMP(ph_ * 8 + 7); }
    // ---- P4: quadrant (mh1, nh0) ----
    { constexpr int ph_ = 3; G256_STAMP(ph_ * 8 + 0);
    G256_RB(t, 0);
    G256_STAMP(ph_ * 8 + 1);
    G256_LOOPSTAGE(0, 1, t + 1, 0, 2); G256_LOOPSTAGE(4, 1, t + 1, 0, 1);
    G256_STAMP(ph_ * 8 + 2);
    G256_VMWAIT();                                        // retires A(mh0)_{t+1}, B(nh0)_{t+1} for next P1
    G256_STAMP(ph_ * 8 + 3);
    G256_BARRIER1();
    G256_MFMA(1, 0, 1, t + 1);
    G256_STAMP(ph_ * 8 + 7); }
  }
#endif
  if (G256_STAGGER && wr == 0) G256_BARRIER();   // re-balance the stagger
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // drain the (out-of-bounds) tail prefetch
#if G256_TRACE
  if (blockIdx.x == G256_TRACE_WG) g256_trace_buf[wave * 64 + lane] = vtrace;
#endif
#if G256_M32
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[i][j][q] = acc32[i >> 1][j >> 1][((i & 1) * 2 + (j & 1)) * 4 + q];
#endif
#undef G256_MFMA_BODY
#undef G256_LOOPSTAGE
#undef G256_VMWAIT
#undef G256_BARRIER2
#undef G256_BARRIER1
#undef G256_RA
#undef G256_RB
#undef G256_MFMA
#undef G256_BARRIER
#undef BFR
#undef AFR
#if G256_DEEP != 2
#undef G256_UNDER_MFMA
#endif
#undef G256_STAMP

  // ---- epilogue: lane holds, for each mt, row (lane&15) and 16 contiguous columns ----
  const int g = lane >> 4;
  if constexpr (MODE == 1) {      // 8 gate + 8 up columns per lane
    const int cs = col0 + wc * 32 + g * 8;
    bf16_t* Cact = (bf16_t*)p.C + (long long)bz * p.sC;
    bf16_t* Cgu = p.C2 ? (bf16_t*)p.C2 + (long long)bz * p.sC2 : nullptr;
    const int Mz = min((Mv + 7) & ~7, p.M);      // rows Mv..Mz-1 are zero-filled: a k_valid wgrad reads whole 8-row chunks
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const int row = row0 + wr * 128 + mt * 16 + li;
      if (row >= Mz || cs >= p.N) continue;
      if (row >= Mv) { *(u32x4*)(Cact + (long long)row * p.ldc + cs) = (u32x4){0u, 0u, 0u, 0u}; continue; }
      float v[16];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[nt * 4 + q] = acc[mt][nt][q];
      u32x4 o0, o1;                   // packed bf16 [gate], [up]: what the activation is computed from AND what training keeps
      st_c(Cact + (long long)row * p.ldc + cs, swiglu_pairs(v, o0, o1));
      if (Cgu) {
        st_c(Cgu + (long long)row * p.ldc2 + cs, o0);
        st_c(Cgu + (long long)row * p.ldc2 + p.N + cs, o1);
      }
    }
    return;
  }
  if constexpr (BK) {      // natural column order: lane owns columns cw + nt*16 + (0..3), nt = 0..3
    const int cw = col0 + wc * 64 + g * 4;
    char* Cb2 = (char*)p.C + (long long)bz * p.sC * (p.out_f32 ? 4 : 2);
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const int row = row0 + wr * 128 + mt * 16 + li;
      if (SPLIT_OK && p.splitk > 1) {      // partial tile -> workspace, lane-linear (reduced below by the last split)
        float* wp = p.ws + (((long long)split * tpb + id) * 32 + mt * 4) * 2048 + tid * 4;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(wp + nt * 2048), "v"(acc[mt][nt]) : "memory");
        continue;
      }
      if (row >= Mv) continue;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int col = cw + nt * 16;
        if (col >= p.N) continue;            // N % 4 == 0 (host-checked): groups of 4 are all in or all out
        f32x4 o = acc[mt][nt];
        if (p.out_f32) {
          float* cp = (float*)Cb2 + (long long)row * p.ldc + col;
          if (p.accumulate) o += *(f32x4*)cp;
          *(f32x4*)cp = o;
        } else {
          bf16_t* cp = (bf16_t*)Cb2 + (long long)row * p.ldc + col;
          if (p.accumulate) {
            const u32x2 old = *(u32x2*)cp;
            o += (f32x4){bflo(old[0]), bfhi(old[0]), bflo(old[1]), bfhi(old[1])};
          }
          *(u32x2*)cp = (u32x2){pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
        }
      }
    }
    if (!(SPLIT_OK && p.splitk > 1)) return;
  }
  if constexpr (MODE == 5) {
    const int cb = col0 + wc * 64 + g * 16;
    float bia[16];
#pragma unroll
    for (int x = 0; x < 16; ++x) bia[x] = 0.f;
    if (p.bias) {           // clamped, unconditional loads: 16 in flight, ONE wait (a per-element condition made each its own round trip)
#pragma unroll
      for (int x = 0; x < 16; ++x) {
        const float bv = bf2f(p.bias[min(cb + x, p.N - 1)]);
        bia[x] = (cb + x < p.N) ? bv : 0.f;
      }
    }
    u32x4 own[8][2];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      float v[16];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[nt * 4 + q] = acc[mt][nt][q] + bia[nt * 4 + q];
      own[mt][0] = (u32x4){pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
      own[mt][1] = (u32x4){pack2bf(v[8], v[9]), pack2bf(v[10], v[11]), pack2bf(v[12], v[13]), pack2bf(v[14], v[15])};
    }
    const bool rot = col0 < p.rope_cols;            // workgroup-uniform: rope_cols is a multiple of 256 (host-checked)
    bf16_t* Cb5 = (bf16_t*)p.C + (long long)bz * p.sC;
    if (rot) {
      __syncthreads();                              // every wave is done with the operand tiles in LDS
#pragma unroll
      for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) *(u32x4*)(smem + ((wave * 8 + mt) * 2 + hx) * 1024 + lane * 16) = own[mt][hx];
      __syncthreads();
      const int fh = (wc & 1) * 64 + g * 16;         // feature of v[0] inside its head
      if (cb >= p.N) return;                         // (no barrier below)
      // Memory order as in the MODE 4 epilogue: row tile by row tile (position -> cos/sin -> store) every tile was two
      // exposed round trips behind the previous tile's stores.  Here: all 8 positions, then cos/sin of row tiles 0-3,
      // compute, cos/sin of 4-7, THEN the stores of 0-3, compute, stores of 4-7.  Rows past Mv read a clamped row.
      auto rowof = [&](int mt) { return row0 + wr * 128 + mt * 16 + li; };
      int ps[8];
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) ps[mt] = p.rope_pos[(long long)bz * p.M + min(rowof(mt), p.M - 1)];
      u32x4 CC[2][4][2], SS[2][4][2];
      auto ld = [&](const int b4) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const bf16_t* cp = p.rope_cos + (long long)ps[b4 * 4 + m] * 128 + fh;
          const bf16_t* sp = p.rope_sin + (long long)ps[b4 * 4 + m] * 128 + fh;
#pragma unroll
          for (int hx = 0; hx < 2; ++hx) { CC[b4][m][hx] = *(const u32x4*)(cp + hx * 8); SS[b4][m][hx] = *(const u32x4*)(sp + hx * 8); }
        }
      };
      auto cmp = [&](const int b4) {                 // result in CC
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int mt = b4 * 4 + m;
#pragma unroll
          for (int hx = 0; hx < 2; ++hx) {
            const u32x4 pr = *(const u32x4*)(smem + (((wave ^ 1) * 8 + mt) * 2 + hx) * 1024 + lane * 16);
            const u32x4 cc = CC[b4][m][hx], ss = SS[b4][m][hx];
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float x0 = bflo(own[mt][hx][k]), x1 = bfhi(own[mt][hx][k]);
              float y0 = bflo(pr[k]), y1 = bfhi(pr[k]);
              if (!(wc & 1)) { y0 = -y0; y1 = -y1; }    // first half: x1*cos + (-x2)*sin ; second half: x2*cos + x1*sin
              o[k] = pack2bf(bfround(x0 * bflo(cc[k])) + bfround(y0 * bflo(ss[k])),
                             bfround(x1 * bfhi(cc[k])) + bfround(y1 * bfhi(ss[k])));
            }
            CC[b4][m][hx] = o;
          }
        }
      };
      auto pin = [&](const int b4) {
#pragma unroll
        for (int m = 0; m < 4; ++m) asm volatile("" : "+v"(CC[b4][m][0]), "+v"(CC[b4][m][1]) : : "memory");
      };
      auto st = [&](const int b4) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int row = rowof(b4 * 4 + m);
          if (row >= Mv) continue;
          bf16_t* op = Cb5 + (long long)row * p.ldc + cb;
          st_c(op, CC[b4][m][0]);
          st_c(op + 8, CC[b4][m][1]);
        }
      };
      ld(0); cmp(0); pin(0); ld(1); st(0); cmp(1); st(1);
    } else {
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        const int row = row0 + wr * 128 + mt * 16 + li;
        if (row >= Mv || cb >= p.N) continue;
        bf16_t* op = Cb5 + (long long)row * p.ldc + cb;
        st_c(op, own[mt][0]);
        st_c(op + 8, own[mt][1]);
      }
    }
    return;
  }
  if constexpr (MODE == 4) {
    // fused SwiGLU backward (lmod_gemm_swiglu_bwd_bf16) as its OWN instantiation: the accumulators are d(act); with the
    // saved [gate | up] pre-activations (C2) the epilogue writes [dgate | dup].  (Inside the MODE 0 epilogue this block
    // cost the plain GEMM 22 %.)  N % 16 == 0 (host-checked).
    // Memory order: written row tile by row tile (load gate/up, compute, store) hipcc cannot lift the next loads over the
    // previous stores (C and C2 may alias) and gfx9's one vmcnt counts loads and stores that retire out of order, so every
    // row tile became load -> vmcnt(0) -> store -> load ...: 16 exposed round trips per tile with ONE workgroup on the CU
    // (26 us beside a 53 us main loop at K 2048).  Here: loads of row tiles 0-3, compute, loads of 4-7, THEN the stores of
    // 0-3, compute, stores of 4-7 — two exposed round trips.  Rows past Mv are loaded from a clamped (allocated) row and
    // never stored.
    const int cb = col0 + wc * 64 + g * 16;
    const int Mz = min((Mv + 7) & ~7, p.M);
    if (cb >= p.N) return;
    bf16_t* obase = (bf16_t*)p.C + (long long)bz * p.sC + cb;
    const bf16_t* gbase = (const bf16_t*)p.C2 + (long long)bz * p.sC2 + cb;
    u32x4 G[2][4][2], U[2][4][2];
    auto rowof = [&](int mt) { return row0 + wr * 128 + mt * 16 + li; };
    auto ld = [&](const int b4) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const bf16_t* gp = gbase + (long long)min(rowof(b4 * 4 + m), p.M - 1) * p.ldc2;
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) {
          G[b4][m][hx] = *(const u32x4*)(gp + hx * 8);
          U[b4][m][hx] = *(const u32x4*)(gp + p.N + hx * 8);
        }
      }
    };
    auto cmp = [&](const int b4) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) {
          float d8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) d8[e] = acc[b4 * 4 + m][hx * 2 + (e >> 2)][e & 3];
          u32x4 og, ou;
          swiglu_bwd8(d8, G[b4][m][hx], U[b4][m][hx], og, ou);
          G[b4][m][hx] = og;
          U[b4][m][hx] = ou;
        }
    };
    auto st = [&](const int b4) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int row = rowof(b4 * 4 + m);
        if (row >= Mv) continue;
        bf16_t* op = obase + (long long)row * p.ldc;
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) {
          *(u32x4*)(op + hx * 8) = G[b4][m][hx];          // (not st_c: non-temporal stores cost this epilogue 3.5-7 %)
          *(u32x4*)(op + p.N + hx * 8) = U[b4][m][hx];
        }
      }
    };
    auto zero_tail = [&]() {           // rows Mv .. roundup8(Mv)-1 are zeroed: a k_valid wgrad reads whole 8-row chunks
      if (Mz == Mv) return;            // (its own pass: merged into st() hipcc turned the branch into one select per output register)
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        const int row = rowof(mt);
        if (row < Mv || row >= Mz) continue;
        bf16_t* op = obase + (long long)row * p.ldc;
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) { *(u32x4*)(op + hx * 8) = (u32x4){0u, 0u, 0u, 0u}; *(u32x4*)(op + p.N + hx * 8) = (u32x4){0u, 0u, 0u, 0u}; }
      }
    };
    // the pin ties batch 0's results to a point AHEAD of batch 1's loads: without it hipcc issues all 32 loads up front
    // (128 registers beside the 128 accumulators -> scratch spills); with it batch 1 lands in the dead accumulators
    auto pin = [&](const int b4) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) asm volatile("" : "+v"(G[b4][m][hx]), "+v"(U[b4][m][hx]) : : "memory");
    };
    ld(0); cmp(0); pin(0); ld(1); st(0); cmp(1); st(1);
    zero_tail();
    return;
  }
  if constexpr (!BK) {
  const int cb = col0 + wc * 64 + g * 16;
  float bia[16];
#pragma unroll
  for (int x = 0; x < 16; ++x) bia[x] = 0.f;
  if (p.bias) {
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      const float bv = bf2f(p.bias[min(cb + x, p.N - 1)]);
      bia[x] = (cb + x < p.N) ? bv : 0.f;
    }
  }
  char* Cb = (char*)p.C + (long long)bz * p.sC * (p.out_f32 ? 4 : 2);
  const bool full = (cb + 16 <= p.N) && p.vec_ok;
  if constexpr (MODE == 6) {
    // C (fp32) += tile without split-K: the weight gradients of the lm_head and of the MoE experts (k_valid batches).  The
    // generic loop below compiles to load -> vmcnt(0) -> add -> store per 16 bytes (the next load cannot be lifted over a
    // store that may alias): 32 exposed round trips per tile with one workgroup on the CU.  Here, as in the MODE 4 epilogue:
    // old values of row tiles 0-3, add, old values of 4-7, THEN the stores of 0-3, add, stores of 4-7.
    if (p.out_f32 && p.accumulate && p.splitk <= 1 && full && !p.act) {
      float* Cf = (float*)Cb;
      auto rowof = [&](int mt) { return row0 + wr * 128 + mt * 16 + li; };
      f32x4 R[2][4][4];
      auto ld = [&](const int b4) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const float* cp = Cf + (long long)min(rowof(b4 * 4 + m), p.M - 1) * p.ldc + cb;
#pragma unroll
          for (int x = 0; x < 4; ++x) R[b4][m][x] = *(const f32x4*)(cp + 4 * x);
        }
      };
      auto add = [&](const int b4) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          float v[16];
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[nt * 4 + q] = acc[b4 * 4 + m][nt][q] + bia[nt * 4 + q];
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            f32x4 o = {v[4 * x], v[4 * x + 1], v[4 * x + 2], v[4 * x + 3]};
            o += R[b4][m][x];
            R[b4][m][x] = o;
          }
        }
      };
      auto pin = [&](const int b4) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
          asm volatile("" : "+v"(R[b4][m][0]), "+v"(R[b4][m][1]), "+v"(R[b4][m][2]), "+v"(R[b4][m][3]) : : "memory");
      };
      auto st = [&](const int b4) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int row = rowof(b4 * 4 + m);
          if (row >= Mv) continue;
          float* cp = Cf + (long long)row * p.ldc + cb;
#pragma unroll
          for (int x = 0; x < 4; ++x) *(f32x4*)(cp + 4 * x) = R[b4][m][x];
        }
      };
      ld(0); add(0); pin(0); ld(1); st(0); add(1); st(1);
      return;
    }
  }
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    const int row = row0 + wr * 128 + mt * 16 + li;
    if (row >= Mv) continue;
    float v[16];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) v[nt * 4 + q] = acc[mt][nt][q] + bia[nt * 4 + q];
    if (p.act) {
#pragma unroll
      for (int x = 0; x < 16; ++x) v[x] = act_apply(bfround(v[x]), p.act);
    }
    if (p.out_f32) {
      float* cp = (float*)Cb + (long long)row * p.ldc + cb;
      if (p.splitk > 1) {        // partial tile -> workspace (lane-linear, 1 KiB per store); the last split reduces (below)
        float* wp = p.ws + (((long long)split * tpb + id) * 32 + mt * 4) * 2048 + tid * 4;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const f32x4 o = {v[4 * x], v[4 * x + 1], v[4 * x + 2], v[4 * x + 3]};
          asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(wp + x * 2048), "v"(o) : "memory");
        }
      } else if (full) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          f32x4 o = {v[4 * x], v[4 * x + 1], v[4 * x + 2], v[4 * x + 3]};
          if (p.accumulate) { f32x4 old = *(f32x4*)(cp + 4 * x); o += old; }
          *(f32x4*)(cp + 4 * x) = o;
        }
      } else {
#pragma unroll
        for (int x = 0; x < 16; ++x) if (cb + x < p.N) cp[x] = p.accumulate ? cp[x] + v[x] : v[x];
      }
    } else {
      bf16_t* cp = (bf16_t*)Cb + (long long)row * p.ldc + cb;
      if (full && !p.accumulate) {
        u32x4 o0 = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
        u32x4 o1 = {pack2bf(v[8], v[9]), pack2bf(v[10], v[11]), pack2bf(v[12], v[13]), pack2bf(v[14], v[15])};
        st_c(cp, o0);
        st_c(cp + 8, o1);
      } else {
#pragma unroll
        for (int x = 0; x < 16; ++x)
          if (cb + x < p.N) cp[x] = f2bf(p.accumulate ? bf2f(cp[x]) + v[x] : v[x]);
      }
    }
  }
  }   // !BK epilogue
  if (SPLIT_OK && p.splitk > 1) {
    // Deterministic split-K reduction: every split publishes its partial tile, the LAST one to arrive adds all of
    // them in split order (its own included, re-read) plus the old C.  Arrival order never changes the result.
    // The splits of a tile run on different XCDs (private L2s): the partials move with agent-scope (sc1) stores and
    // loads, ordered by vmcnt(0) + the semaphore atomic — no cache-wide writeback / invalidate fences.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = (int*)smem;                         // the LDS tiles are dead by now
    if (tid == 0)
      *flag = (__hip_atomic_fetch_add(p.counters + id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p.splitk - 1);
    __syncthreads();
    if (!*flag) return;
    float* Cf = (float*)p.C;
    for (int mt = 0; mt < 8; ++mt) {
      const int row = row0 + wr * 128 + mt * 16 + li;
      // lane-linear partial x of row-tile mt holds 4 contiguous columns starting at colx(x)
      auto colx = [&](int x) { return BK ? col0 + wc * 64 + (lane >> 4) * 4 + 16 * x : col0 + wc * 64 + (lane >> 4) * 16 + 4 * x; };
      f32x4 o[4];
      bool live[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        live[x] = (row < Mv) && (colx(x) < p.N);      // N % 4 == 0: a group of 4 columns is all in or all out
        o[x] = live[x] ? *(f32x4*)(Cf + (long long)row * p.ldc + colx(x)) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      for (int s0 = 0; s0 < p.splitk; s0 += 4) {      // up to 16 agent-scope loads in flight
        f32x4 part[4][4];
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
          const float* wp = p.ws + (((long long)min(s0 + ds, p.splitk - 1) * tpb + id) * 32 + mt * 4) * 2048 + tid * 4;
#pragma unroll
          for (int x = 0; x < 4; ++x)
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(part[ds][x]) : "v"(wp + x * 2048) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int ds = 0; ds < 4; ++ds)
          if (s0 + ds < p.splitk) {
#pragma unroll
            for (int x = 0; x < 4; ++x) {
              asm volatile("" : "+v"(part[ds][x]));   // value is defined only after the wait above
              o[x] += part[ds][x];
            }
          }
      }
#pragma unroll
      for (int x = 0; x < 4; ++x)
        if (live[x]) *(f32x4*)(Cf + (long long)row * p.ldc + colx(x)) = o[x];
    }
    if (tid == 0) __hip_atomic_store(p.counters + id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch
  }
}

// =============================================================================================
// 256 x 256 x 64 tile, FOUR waves (2 x 2), 128 x 128 output per wave, ONE wave per SIMD.  By default the plain bf16 GEMM launches
// (MODE 7) and the fused SwiGLU forward (MODE 1) run on it; see launch_256.
//
// Why this shape: with 128x128 per wave every LDS operand fragment feeds 8 MFMAs (LDS reads 128 KB per K tile instead of the 192 KB
// of the 8-wave 128x64 layout) and half as many waves issue.  It is also the shape of hipBLASLt's hand-written gfx950 kernel, whose
// loop structure the main loop below follows (profiles/r03_vendor_ab.md).
// Registers: the 64 accumulator tiles (256 registers) are pinned to the AGPR half of the unified file through inline-asm MFMAs with
// "+a" operands (left alone, hipcc shuttles them through a[0:3] with ~1000 v_accvgpr moves per K tile); operand fragments are
// double-buffered in VGPRs (2 x 64).  With a single wave per SIMD anything that is not in an MFMA shadow is lost time: every
// non-MFMA instruction of the loop sits behind a named MFMA (see the schedule at the loop).
// LDS: 2 stages x (A [256][64] + B [256][64]) bf16 = 128 KiB, rows XOR-swizzled as in the 128 kernel, B rows permuted so a lane ends
// with 16 contiguous output columns per 64-column group.
// MODE 0: C = act(A B^T + b) with every option of lmod_gemm_bf16_nt (grouped, k_valid, f32 / accumulate, split-K, fused SwiGLU
//         backward) behind run-time branches — 109 spilled VGPRs in that epilogue, so the hot option sets have their own instantiations:
// MODE 7: bf16 C = act(acc + bias).  MODE 4: fused SwiGLU backward.  MODE 6: fp32 C += acc.  MODE 5: fused q/k/v + bias + RoPE.
// MODE 1: fused SwiGLU forward (see gemm_256_kernel<1>): an N tile is 128 output columns; wave column wc takes 64 of them, its first
//         64 LDS B rows are the gate rows and the next 64 the matching up rows, so a lane owns 16 gate and the same 16 up columns.
// =============================================================================================
#ifndef G4_PAD
#define G4_PAD 0                    // 1: hipBLASLt's LDS image — source addresses LINEAR inside a 128-byte row (no XOR on the
#endif                              //    LDS-DMA source), 16 bytes of padding behind every 1 KiB piece, 2-way conflicts on the fragment reads
#define G4_PIECE (G4_PAD ? 1040 : 1024)
#define G4_STAGE (64 * G4_PIECE)
#ifndef G4_ASM
#define G4_ASM (!G4_PAD)              // the K loop as one hand-placed asm statement (XOR layout only)
#endif
#if G4_ASM
#ifndef G4_ASM_HEADER
#define G4_ASM_HEADER "gemm4_loop_asm.h"
#endif
#include G4_ASM_HEADER
#ifndef G4_ASM_NO_B3
#include "gemm4_loop_asm_b3.h"      // the three-barrier schedule: +0.8 ... 0.9 % from K = 5504 up, -0.4 % at K = 4096 (profiles/r04_gemm_loop.md)
#endif
#endif
#ifndef G4_ASM_PEEL
#define G4_ASM_PEEL 0
#endif
#define G4_BOFF (32 * G4_PIECE)
#define G4_FSTR (G4_PAD ? 128 : 2048)
// Which tile column (inside a 64-column group) the LDS B row (ntl * 16 + ii) holds; ii = 4 * g + e is the MFMA row a lane group g
// ends up owning.  Default: 16 contiguous columns per lane (g * 16 + ntl * 4 + e: two 16-byte stores 32 bytes apart from their
// neighbours').  MODE 7 and MODE 1 (gate and up alike) with G4_SPLIT_COLS: two runs of 8 (columns g * 8 + .. and 32 + g * 8 + ..), so that ONE store instruction writes
// 64 contiguous, 64-byte aligned bytes per row (4 lanes x 16 bytes) instead of 4 pieces of 16 bytes at a 32-byte stride.
#ifndef G4_SPLIT_COLS
#define G4_SPLIT_COLS 1
#endif
template <int MODE>
__device__ __forceinline__ int g4_bcol(const int ntl, const int ii) {
  if ((MODE == 7 || MODE == 8 || MODE == 1 || MODE == 5) && G4_SPLIT_COLS) return (ntl >> 1) * 32 + (ii >> 2) * 8 + (ntl & 1) * 4 + (ii & 3);
  return (ii >> 2) * 16 + ntl * 4 + (ii & 3);
}
struct G4Tile {      // one output tile: coordinates, operand windows, this lane's staging offsets
  int id, bz, split, Mv, Kv, row0, col0, rowsA, rowsB, nkt;
  const bf16_t* Ab; const bf16_t* Bb;
  uint32_t bytesA, bytesB, voA[8], voB[8];
};
// PERSIST (round 4): one workgroup per CU walks the tile space (tile = blockIdx.x + i * gridDim.x, XCD-remapped as before) and its
// operand stream never stops: the K loop's look-ahead (LDS-DMA of K tile t+2, fragment reads of t+1) runs on INTO THE NEXT
// OUTPUT TILE (G4_ASM_LOOP_P switches the two descriptors when it passes the end of K), so a tile's epilogue starts with the
// next tile's first fragments in registers and its second K tile in flight, and its C stores drain under the next tile's first
// MFMAs — no workgroup turn-around, no prologue latency, no store drain on the critical path.  Plain launches with K % 64 == 0,
// K >= 256 (the launcher decides); results are those of the one-tile-per-workgroup form, bit for bit.
// PG (round 5): the persistent form for GROUPED launches (MoE capacity slabs, `m_valid` live rows per batch, batch <=
// GEMM_MAX_GROUPS): the walk runs over the LIVE tiles only (their count is computed in the kernel from m_valid; dead tiles are
// never visited, so there is nothing to skip), with the same continuous operand stream across tiles.
template <int MODE, bool PERSIST = false, bool PG = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm4_kernel(GemmP p) {
  static_assert(!PG || PERSIST, "PG is a persistent form");
  constexpr int TN = (MODE == 1) ? 128 : 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x 64 KiB
  // (not const: the persistent form re-launders them once per tile, see G4_LAUNDER)
  int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int wr = wave >> 1, wc = wave & 1;

  const int tpb = p.tiles_m * p.tiles_n;
  // PG: live rows and live row tiles of every batch, read ONCE here (scalar loads: nothing of this launch has been stored yet)
  int pg_rows[GEMM_MAX_GROUPS], pg_total = 0;
  if constexpr (PG) {
#pragma unroll
    for (int e = 0; e < GEMM_MAX_GROUPS; ++e) {
      pg_rows[e] = 0;
      if (e < p.batch) {
        pg_rows[e] = __builtin_amdgcn_readfirstlane(min(p.m_valid[e], p.M));
        pg_total += (pg_rows[e] + 255) >> 8;
      }
    }
  }
  const int pg_nlive = pg_total * p.tiles_n;
  // staging: piece j of wave w fills LDS rows (j*4 + w)*8 + (lane>>3), physical chunk lane&7 (logical chunk ^ row&7)
  int cchunk = G4_PAD ? (lane & 7) : ((lane & 7) ^ (lane >> 3));
  auto setup = [&](const int bid, const int nwg, G4Tile& T) -> bool {
    // persistent launches carry no k_valid, and no m_valid unless PG (launcher): without the `PERSIST ? nullptr` below their loads —
    // re-issued after the previous tile's stores, so not scalarisable — drag the whole tile arithmetic into VGPRs (PG reads its
    // counts from the pg_* registers filled at kernel entry instead)
    const int* const mvp = PERSIST ? nullptr : p.m_valid;
    const int* const kvp = PERSIST ? nullptr : p.k_valid;
    int id = xcd_remap(bid, nwg);
    int bz = id / tpb, split = 0;
    if (MODE == 0 && p.splitk > 1) { split = bz; bz = 0; id -= split * tpb; }
    int r = id - bz * tpb;
    if ((MODE == 6 || MODE == 0) && kvp && !mvp && p.batch > 1 && p.batch <= GEMM_MAX_GROUPS && p.splitk <= 1 && !(tpb & 7)) {
      // batched weight gradients with a live reduction length per batch (MoE experts): every XCD takes 1/8 of EVERY expert's tiles,
      // longest reduction first (see gemm_256_kernel: whole experts per XCD ran 2.7x apart with uneven routers)
      const int tpb8 = tpb >> 3, xcd = bid & 7, k = bid >> 3;
      const int rnk = k / tpb8;
      int sel = rnk;
      for (int e = 0; e < p.batch; ++e) {
        const int ke = kvp[e];
        int rank = 0;
        for (int f = 0; f < p.batch; ++f) {
          const int kf = kvp[f];
          rank += (kf > ke || (kf == ke && f < e)) ? 1 : 0;
        }
        if (rank == rnk) sel = e;
      }
      bz = sel;
      r = xcd * tpb8 + (k - rnk * tpb8);
      id = bz * tpb + r;
    }
    const int GROUP_M = G256_GROUP_M;
    const int grp = r / (GROUP_M * p.tiles_n);
    const int first_m = grp * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    const int rr = r - grp * GROUP_M * p.tiles_n;
    int tm = first_m + rr % gsz, tn = rr / gsz;
    if (mvp) {
      if (p.batch <= GEMM_MAX_GROUPS) {
        if (!grouped_tile<256, G256_GROUP_M>(p, bz, tm, tn)) return false;
      } else {
        const int per_col = p.batch * p.tiles_m;
        tn = id / per_col;
        const int rem = id - tn * per_col;
        bz = rem / p.tiles_m;
        tm = rem - bz * p.tiles_m;
      }
    }
    int Mv = mvp ? min(mvp[bz], p.M) : p.M;
    if constexpr (PG) {
      if (!grouped_tile_pre<G256_GROUP_M>(p, pg_rows, pg_total, pg_nlive, bid, bz, tm, tn)) return false;
      Mv = 0;
#pragma unroll
      for (int e = 0; e < GEMM_MAX_GROUPS; ++e) Mv = (e == bz) ? pg_rows[e] : Mv;
      id = bz * tpb + tm * p.tiles_n + tn;
    }
    int Kv = kvp ? min(kvp[bz], p.K) : p.K;
    const int row0 = tm * 256, col0 = tn * TN;
    if (row0 >= Mv) return false;
    int kbeg = 0;
    if (MODE == 0 && p.splitk > 1) {
      kbeg = split * p.kchunk;
      Kv = max(0, min(Kv - kbeg, p.kchunk));
    }
    T.id = id; T.bz = bz; T.split = split; T.Mv = Mv; T.Kv = Kv; T.row0 = row0; T.col0 = col0;
    T.Ab = p.A + (long long)bz * p.sA + (long long)row0 * p.lda + kbeg;
    T.Bb = p.B + (long long)bz * p.sB + (long long)col0 * p.ldb + kbeg;
    const int rowsA = min(256, Mv - row0), rowsB = min(TN, p.N - col0);
    T.rowsA = rowsA; T.rowsB = rowsB;
    const int Kv8 = (Kv + 7) & ~7;
    T.bytesA = Kv > 0 ? (uint32_t)(((long long)(rowsA - 1) * p.lda + Kv8) * 2) : 0u;
    // (MODE 1: the window spans the gate rows AND the up rows, p.N weight rows apart.  In the persistent forms a ragged N edge is cut by
    // this byte count alone — the per-lane offsets there do not depend on the tile — so gate-side rows between rowsB and the tile's 128
    // read LIVE weight rows (the up rows' window covers them); the epilogue masks those columns (`cs >= p.N`), nothing of them is stored)
    T.bytesB = Kv > 0 ? (uint32_t)(((long long)((MODE == 1 ? p.N : 0) + rowsB - 1) * p.ldb + Kv8) * 2) : 0u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#if G4_PAD   // physical row p of a 128-row group holds logical row (p & 7) * 16 + (p >> 3): a lane's 8 fragments sit 128 bytes apart
      const int lrp = (j * 4 + wave) * 8 + (lane >> 3), p128 = lrp & 127;
      const int lr = (lrp & ~127) + (p128 & 7) * 16 + (p128 >> 3);
#else
      const int lr = (j * 4 + wave) * 8 + (lane >> 3);                      // LDS row 0..255
#endif
      T.voA[j] = (lr < rowsA) ? (uint32_t)((lr * p.lda + cchunk * 8) * 2) : GEMM_OOB;
      const int grp64 = lr >> 6, rl = lr & 63, ntl = rl >> 4, ii = rl & 15;
      if (MODE != 1) {
        const int nloc = grp64 * 64 + g4_bcol<MODE>(ntl, ii);                       // permuted tile column
        T.voB[j] = (nloc < rowsB) ? (uint32_t)((nloc * p.ldb + cchunk * 8) * 2) : GEMM_OOB;
      } else {       // groups 2wc / 2wc+1 = gate / up rows of output columns wc*64 .. +63
        const int nloc = (grp64 >> 1) * 64 + g4_bcol<MODE>(ntl, ii);
        T.voB[j] = (nloc < rowsB) ? (uint32_t)((((long long)(grp64 & 1) * p.N + nloc) * p.ldb + cchunk * 8) * 2) : GEMM_OOB;
      }
    }
    T.nkt = (Kv + 63) >> 6;
    return true;
  };
  // descriptor of an operand window, from wave-uniform words the compiler can SEE are uniform (inside the tile loop it would
  // otherwise wrap every LDS-DMA in a waterfall loop: guide T20)
  auto rsrc_of = [&](const bf16_t* base, const uint32_t bytes) {
    const unsigned long long q = (unsigned long long)base;
    const unsigned long long qu = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(q >> 32)) << 32) |
                                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)q);
    return __builtin_amdgcn_make_buffer_rsrc((void*)qu, 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
  };
  auto stage_tiles01 = [&](const G4Tile& T) {              // the first two K tiles of a tile (32 pieces per wave) -> both LDS stages
    const __amdgpu_buffer_rsrc_t ra = rsrc_of(T.Ab, T.bytesA), rb = rsrc_of(T.Bb, T.bytesB);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k0 = t * 64;
        char* dst = smem + t * G4_STAGE + wave * G4_PIECE;
        const bool dead = (k0 + cchunk * 8 >= T.Kv);          // tiles past the end are fully out of bounds: zero fill, no traffic
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(dst + j * (4 * G4_PIECE)), 16, dead ? GEMM_OOB : T.voA[j], k0 * 2, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(dst + G4_BOFF + j * (4 * G4_PIECE)), 16, dead ? GEMM_OOB : T.voB[j], k0 * 2, 0, 0);
      }
  };
  G4Tile T;
  int pbid = blockIdx.x;
  const int ptotal = PG ? pg_nlive : (PERSIST ? p.ptotal : (int)gridDim.x);
  if (!setup(pbid, ptotal, T)) return;                     // (persistent walks see live tiles only; PG: a workgroup past the live count)
  stage_tiles01(T);
  bf16x8 fa[2][8], fb[2][8];
  // PERSIST: what the K loop carries from one tile to the next — the LDS stage parity (fragment addresses, LDS-DMA base) and the
  // first fragments of the next tile (its last pass already read them) — and staging offsets that do not depend on the tile (rows
  // past a ragged edge are cut by the descriptor's byte count instead of per-lane out-of-range markers)
  uint32_t g4_ra0 = 0, g4_ra1 = 0, g4_rb0 = 0, g4_rb1 = 0, g4_dma = 0;
  bool first = true;
  for (;;) {                                               // ---- one output tile per pass (exactly one pass unless PERSIST)
  const int id = T.id, bz = T.bz, split = T.split, Mv = T.Mv, Kv = T.Kv, row0 = T.row0, col0 = T.col0, nkt = T.nkt;
  const bf16_t* Ab = T.Ab; const bf16_t* Bb = T.Bb;
  const uint32_t bytesA = T.bytesA, bytesB = T.bytesB;
  uint32_t voA[8], voB[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if constexpr (!PERSIST) { voA[j] = T.voA[j]; voB[j] = T.voB[j]; }
    else {   // offsets that do not depend on the tile (a ragged edge is cut by the descriptor's byte count), recomputed per tile from
             // the laundered lane id so that they are not carried through the epilogue
      const int lr = (j * 4 + wave) * 8 + (lane >> 3);
      voA[j] = (uint32_t)((lr * p.lda + cchunk * 8) * 2);
      const int grp64 = lr >> 6, rl = lr & 63, ntl = rl >> 4, ii = rl & 15;
      if (MODE != 1) voB[j] = (uint32_t)(((grp64 * 64 + g4_bcol<MODE>(ntl, ii)) * p.ldb + cchunk * 8) * 2);
      else voB[j] = (uint32_t)((((long long)(grp64 & 1) * p.N + (grp64 >> 1) * 64 + g4_bcol<MODE>(ntl, ii)) * p.ldb + cchunk * 8) * 2);
    }
  }
  __amdgpu_buffer_rsrc_t rsA = rsrc_of(Ab, bytesA);
  __amdgpu_buffer_rsrc_t rsB = rsrc_of(Bb, bytesB);
  (void)id; (void)split;

  f32x4 acc[8][8];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  // (the asm loop's peeled first K tile multiplies into C = 0 and WRITES the accumulators: no 256 v_accvgpr_write per tile there)
  if (!(G4_ASM && G4_ASM_PEEL) || (Kv & 63) != 0 || nkt == 0) zero_acc();

#if G4_PAD
  const int abase = (wr * 16 + li) * G4_PIECE, bbase = G4_BOFF + (wc * 16 + li) * G4_PIECE;
  const int ph[2] = {g * 16, (4 + g) * 16};
#else
  const int abase = (wr * 128 + li) * 128, bbase = 32768 + (wc * 128 + li) * 128;
  const int ph[2] = {((g) ^ (li & 7)) * 16, ((4 + g) ^ (li & 7)) * 16};
#endif
#ifndef G4_DEEP
#define G4_DEEP 1
#endif
#ifndef G4_B2
#define G4_B2 1                     // two barriers per K tile (0: three, the vendor kernel's arrangement: -0.5 ... -0.9 %)
#endif
#ifndef G4_ABL
#define G4_ABL 0                    // timing ablations of the deep schedule (WRONG RESULTS): 1 no barriers, 2 no LDS-DMA in the loop, 4 no fragment reads, 8 no vmcnt wait, 16 every LDS-DMA out of range (no traffic)
#endif
#define G4_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#if !G4_DEEP
  // one k-step: 64 MFMAs on (fa[cur], fb[cur]); in their shadow the fragments of the next k-step (stage ts, k-half kn) are
  // read into (fa[cur^1], fb[cur^1]) and, if DMA, the 16 LDS-DMA pieces of tile td are issued
  auto kstep = [&](const int cur, const int ts, const int kn, const bool dma, const int td) {
    const char* s = smem + (ts & 1) * G4_STAGE + ph[kn];
    const int k0 = td * 64;
    char* dst = smem + (td & 1) * G4_STAGE + wave * G4_PIECE;
    const bool dead = (k0 + cchunk * 8 >= Kv);
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[mt][nt]) : "v"(fb[cur][nt]), "v"(fa[cur][mt]));
        if (nt == 1) { fa[cur ^ 1][mt] = *(const bf16x8*)(s + abase + mt * G4_FSTR); __builtin_amdgcn_sched_barrier(0); }
        if (nt == 3) { fb[cur ^ 1][mt] = *(const bf16x8*)(s + bbase + mt * G4_FSTR); __builtin_amdgcn_sched_barrier(0); }
        if (nt == 5 && dma) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(dst + mt * (4 * G4_PIECE)), 16, dead ? GEMM_OOB : voA[mt], k0 * 2, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (nt == 7 && dma) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(dst + G4_BOFF + mt * (4 * G4_PIECE)), 16, dead ? GEMM_OOB : voB[mt], k0 * 2, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };

  // the first two K tiles were put in flight by stage_tiles01; a persistent workgroup's LATER tiles arrive primed: the previous
  // tile's K loop fetched their first two K tiles with its look-ahead and read the first fragments in its last pass
  if (!PERSIST || first) {
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");     // tile 0 landed (tile 1's 16 pieces may still fly)
    G4_BARRIER();
    const char* s = smem + ph[0];
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[0][i] = *(const bf16x8*)(s + abase + i * G4_FSTR); fb[0][i] = *(const bf16x8*)(s + bbase + i * G4_FSTR); }
  } else {
    // primed by the previous tile's K loop: K tile 0 of THIS tile sits in the stage g4_ra1 / g4_rb1 point at (k-half 1: the k-half-0
    // chunk is the address with bit 6 flipped), landed and barrier-ed.  Re-reading 16 fragments costs less than carrying 64 registers
    // through the epilogue (hipcc spills them through scratch)
    typedef __attribute__((address_space(3))) const bf16x8* lds_frag_t;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      fa[0][i] = *(lds_frag_t)(uintptr_t)((g4_ra1 ^ 64u) + i * G4_FSTR);
      fb[0][i] = *(lds_frag_t)(uintptr_t)((g4_rb1 ^ 64u) + i * G4_FSTR);
    }
  }
  for (int t = 0; t < nkt; ++t) {
    kstep(0, t, 1, false, 0);                              // k-step 0; reads this tile's k-step-1 operands
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // my reads of tile t's stage are done
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // my pieces of tile t+1 (issued one tile ago) have landed
    G4_BARRIER();                                          // => stage t&1 is free for everyone, tile t+1 is complete
    kstep(1, t + 1, 0, true, t + 2);                       // k-step 1; reads tile t+1's k-step-0 operands, stages tile t+2
    __builtin_amdgcn_sched_barrier(0);
  }
#else
  // Schedule (G4_DEEP, the default; G4_DEEP=0 is the round-1 loop: one barrier at mid tile behind vmcnt(0), 0.5-0.75 tiles of cover for
  // the LDS-DMA, 1270 TF).  It follows hipBLASLt's hand-written gfx950 kernel of the same shape (its main loop read from the library's
  // code object with llvm-objdump: 1587 TF on the teacher QKV shape).  Per K tile t — stage X = t & 1 holds it, stage Y holds t + 1,
  // the k-step-0 fragments are in registers — with G4_B2 (two barriers, the default):
  //   k-step 0  MFMAs 0-30   both operands' k-step-1 fragments are read (one ds_read behind every second MFMA)
  //             MFMA 38      lgkmcnt(0) + barrier 1: every wave holds ALL of tile t in registers => stage X is free
  //             MFMAs 39-59  LDS-DMA pieces 0-4 of tile t+2 into X, one every 5 MFMAs (four waves share one address unit)
  //   k-step 1  MFMAs 0-20   pieces 5-9
  //             MFMA 22      vmcnt(10) — everything older than this tile's 10 pieces so far, i.e. all of tile t+1 — + barrier 2
  //             MFMAs 23-46  the k-step-0 fragments of tile t+1 are read from Y (two behind every three MFMAs, the last one 12 MFMAs
  //                          ahead of the loop end); pieces 10-15 in the free slots (25, 31, ..., 55)
  // A piece has 1.0-1.4 tiles to land.  G4_B2=0 keeps the vendor kernel's three barriers (A half and B half freed separately).
  // K64 (the reduction length is a multiple of 64: every shape of the step but the MoE experts' routed-row counts): a tile is wholly
  // live or wholly past the end, so "past the end" is a SCALAR choice of descriptor (zero records: every lane out of range, no
  // traffic) instead of one v_cndmask per LDS-DMA in the MFMA gaps.
  const __amdgpu_buffer_rsrc_t rsAz = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsBz = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, 0, 0x00020000);
  auto dma_a = [&](auto k64_t, const int td, const int j) {
    const int k0 = td * 64;
    char* dst = smem + (td & 1) * G4_STAGE + wave * G4_PIECE;
    if constexpr (decltype(k64_t)::value) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds((k0 >= Kv || (G4_ABL & 16)) ? rsAz : rsA, LDS_PTR(dst + j * (4 * G4_PIECE)), 16, voA[j], k0 * 2, 0, 0);
    } else {
      const bool dead = (G4_ABL & 16) || (k0 + cchunk * 8 >= Kv);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(dst + j * (4 * G4_PIECE)), 16, dead ? GEMM_OOB : voA[j], k0 * 2, 0, 0);
    }
  };
  auto dma_b = [&](auto k64_t, const int td, const int j) {
    const int k0 = td * 64;
    char* dst = smem + (td & 1) * G4_STAGE + wave * G4_PIECE;
    if constexpr (decltype(k64_t)::value) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds((k0 >= Kv || (G4_ABL & 16)) ? rsBz : rsB, LDS_PTR(dst + G4_BOFF + j * (4 * G4_PIECE)), 16, voB[j], k0 * 2, 0, 0);
    } else {
      const bool dead = (G4_ABL & 16) || (k0 + cchunk * 8 >= Kv);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(dst + G4_BOFF + j * (4 * G4_PIECE)), 16, dead ? GEMM_OOB : voB[j], k0 * 2, 0, 0);
    }
  };
#define G4_MFMA(cur, idx) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[(idx) >> 3][(idx) & 7]) : "v"(fb[cur][(idx) & 7]), "v"(fa[cur][(idx) >> 3]))
#define G4_SB() __builtin_amdgcn_sched_barrier(0)

  // the first two K tiles were put in flight by stage_tiles01; a persistent workgroup's LATER tiles arrive primed: the previous
  // tile's K loop fetched their first two K tiles with its look-ahead and read the first fragments in its last pass
  if (!PERSIST || first) {
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");     // tile 0 landed (tile 1's 16 pieces may still fly)
    G4_BARRIER();
    const char* s = smem + ph[0];
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[0][i] = *(const bf16x8*)(s + abase + i * G4_FSTR); fb[0][i] = *(const bf16x8*)(s + bbase + i * G4_FSTR); }
  } else {
    // primed by the previous tile's K loop: K tile 0 of THIS tile sits in the stage g4_ra1 / g4_rb1 point at (k-half 1: the k-half-0
    // chunk is the address with bit 6 flipped), landed and barrier-ed.  Re-reading 16 fragments costs less than carrying 64 registers
    // through the epilogue (hipcc spills them through scratch)
    typedef __attribute__((address_space(3))) const bf16x8* lds_frag_t;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      fa[0][i] = *(lds_frag_t)(uintptr_t)((g4_ra1 ^ 64u) + i * G4_FSTR);
      fb[0][i] = *(lds_frag_t)(uintptr_t)((g4_rb1 ^ 64u) + i * G4_FSTR);
    }
  }
  auto run_loop = [&](auto k64_t) {
  for (int t = 0; t < nkt; ++t) {
    const char* sx = smem + (t & 1) * G4_STAGE + ph[1];              // this tile, k-half 1
    const char* sy = smem + ((t + 1) & 1) * G4_STAGE + ph[0];        // next tile, k-half 0
#if G4_B2
    // two barriers per tile: both operands' k-step-1 fragments first (MFMAs 0-30), ONE barrier frees all of X, then the 16 pieces
    // of tile t+2 every 5 MFMAs (A 0-7, B 0-7), barrier 3 with vmcnt(10) in between
    auto piece = [&](const int pc) { if (pc < 8) dma_a(k64_t, t + 2, pc); else dma_b(k64_t, t + 2, pc - 8); };
#pragma unroll
    for (int idx = 0; idx < 64; ++idx) {                             // ---- k-step 0
      G4_MFMA(0, idx);
      if (idx < 16 && !(idx & 1)) { fa[1][idx >> 1] = *(const bf16x8*)(sx + abase + (idx >> 1) * G4_FSTR); G4_SB(); }
      if (idx >= 16 && idx < 32 && !(idx & 1)) { fb[1][(idx - 16) >> 1] = *(const bf16x8*)(sx + bbase + ((idx - 16) >> 1) * G4_FSTR); G4_SB(); }
      if (idx == 38) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); G4_BARRIER(); G4_SB(); }
      if (idx >= 39 && idx <= 59 && (idx - 39) % 5 == 0) { piece((idx - 39) / 5); G4_SB(); }                       // pieces 0-4
    }
    G4_SB();
#pragma unroll
    for (int idx = 0; idx < 64; ++idx) {                             // ---- k-step 1
      G4_MFMA(1, idx);
      if (idx <= 20 && idx % 5 == 0) { piece(5 + idx / 5); G4_SB(); }                                              // pieces 5-9
      if (idx == 22) { asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); G4_BARRIER(); G4_SB(); }
      if (idx >= 25 && idx <= 55 && (idx - 25) % 6 == 0) { piece(10 + (idx - 25) / 6); G4_SB(); }                  // pieces 10-15
      if (idx >= 23 && idx <= 46 && (idx - 23) % 3 != 2) {
        const int f = ((idx - 23) / 3) * 2 + (idx - 23) % 3;
        if (f < 8) fa[0][f] = *(const bf16x8*)(sy + abase + f * G4_FSTR);
        else fb[0][f - 8] = *(const bf16x8*)(sy + bbase + (f - 8) * G4_FSTR);
        G4_SB();
      }
    }
#else
#pragma unroll
    for (int idx = 0; idx < 64; ++idx) {                             // ---- k-step 0
      G4_MFMA(0, idx);
      // A fragments of k-step 1: behind MFMAs 0, 2, ..., 14
      if (!(G4_ABL & 4) && idx < 16 && !(idx & 1)) { fa[1][idx >> 1] = *(const bf16x8*)(sx + abase + (idx >> 1) * G4_FSTR); G4_SB(); }
      if (idx == 21) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (!(G4_ABL & 1)) G4_BARRIER(); G4_SB(); }   // barrier 1
      // A(t+2): behind MFMAs 22, 27, ..., 57 (the four waves share ONE address unit, 16 cycles per 1 KiB piece: pieces issued every 3
      // MFMAs by all four waves oversubscribe it and the in-order waves stall behind their own VMEM issue);  B fragments of k-step 1:
      // behind 24, 27, ..., 45
      if (!(G4_ABL & 2) && idx >= 22 && idx <= 57 && (idx - 22) % 5 == 0) { dma_a(k64_t, t + 2, (idx - 22) / 5); G4_SB(); }
      if (!(G4_ABL & 4) && idx >= 24 && idx <= 45 && (idx - 24) % 3 == 0) { fb[1][(idx - 24) / 3] = *(const bf16x8*)(sx + bbase + ((idx - 24) / 3) * G4_FSTR); G4_SB(); }
      if (idx == 54) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (!(G4_ABL & 1)) G4_BARRIER(); G4_SB(); }   // barrier 2
      if (!(G4_ABL & 2) && idx == 62) { dma_b(k64_t, t + 2, 0); G4_SB(); }                                                       // B(t+2) piece 0
    }
    G4_SB();
#pragma unroll
    for (int idx = 0; idx < 64; ++idx) {                             // ---- k-step 1
      G4_MFMA(1, idx);
      if (!(G4_ABL & 2) && (idx == 5 || idx == 13)) { dma_b(k64_t, t + 2, idx == 5 ? 1 : 2); G4_SB(); }                          // pieces 1, 2
      // barrier 3: everything older than this tile's 11 pieces so far (A 0-7, B 0-2) has landed, i.e. all of tile t+1
      if (idx == 22) { if (!(G4_ABL & 8)) asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); if (!(G4_ABL & 1)) G4_BARRIER(); G4_SB(); }
      // B(t+2) pieces 3-7 in the free slots of the fragment reads below, the last one 11 MFMAs ahead of the loop end
      if (!(G4_ABL & 2) && (idx == 25 || idx == 31 || idx == 37 || idx == 43 || idx == 52)) { dma_b(k64_t, t + 2, idx == 52 ? 7 : 3 + (idx - 25) / 6); G4_SB(); }
      // k-step-0 fragments of tile t+1: two behind every three MFMAs, the last one 12 MFMAs ahead of the loop end
      if (!(G4_ABL & 4) && idx >= 23 && idx <= 46 && (idx - 23) % 3 != 2) {
        const int f = ((idx - 23) / 3) * 2 + (idx - 23) % 3;         // 0..15: A fragments 0-7, then B fragments 0-7
        if (f < 8) fa[0][f] = *(const bf16x8*)(sy + abase + f * G4_FSTR);
        else fb[0][f - 8] = *(const bf16x8*)(sy + bbase + (f - 8) * G4_FSTR);
        G4_SB();
      }
    }
#endif
    G4_SB();
  }
  };
#if G4_ASM
  // K % 64 == 0 (every launch of the step on this kernel): the loop as ONE hand-placed asm statement (gemm4_loop_asm.h, written
  // by tools/gen_gemm4_loop.py); the hipcc-scheduled loop above stays for ragged reduction lengths and as the A/B arm (G4_ASM=0)
  G4Tile Nx;
  bool more = false;
  if ((Kv & 63) == 0) {
    if (nkt > 0) {
      const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
      if (!PERSIST || first) {
        g4_ra1 = lds0 + ph[1] + abase; g4_rb1 = lds0 + ph[1] + bbase;                         // tile 0: stage 0, k-half 1
        g4_ra0 = lds0 + G4_STAGE + ph[0] + abase; g4_rb0 = lds0 + G4_STAGE + ph[0] + bbase;   // tile 1: stage 1, k-half 0
        g4_dma = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);
      }
      uint32_t g4_nk = __builtin_amdgcn_readfirstlane(nkt), g4_koff = 256u;
      const uint32_t g4_klim = __builtin_amdgcn_readfirstlane(Kv * 2);
      const unsigned long long pa = (unsigned long long)Ab, pb = (unsigned long long)Bb;
      const uint32_t g4_dA[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pa), (uint32_t)__builtin_amdgcn_readfirstlane((int)((pa >> 32) & 0xffffu)),
                                 (uint32_t)__builtin_amdgcn_readfirstlane((int)bytesA), 0x00020000u};
      const uint32_t g4_dB[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pb), (uint32_t)__builtin_amdgcn_readfirstlane((int)((pb >> 32) & 0xffffu)),
                                 (uint32_t)__builtin_amdgcn_readfirstlane((int)bytesB), 0x00020000u};
      if constexpr (PERSIST) {
        // the NEXT tile's operand windows: the loop's look-ahead moves onto them when it runs off the end of this tile's K range
        // (no next tile: zero records, nothing is fetched)
        pbid += (int)gridDim.x;
        more = pbid < ptotal;
        uint32_t g4_dAn[3] = {g4_dA[0], g4_dA[1], 0u}, g4_dBn[3] = {g4_dB[0], g4_dB[1], 0u};
        if (more) {
          setup(pbid, ptotal, Nx);
          const unsigned long long na = (unsigned long long)Nx.Ab, nb = (unsigned long long)Nx.Bb;
          g4_dAn[0] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)na); g4_dAn[1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)((na >> 32) & 0xffffu));
          g4_dAn[2] = (uint32_t)__builtin_amdgcn_readfirstlane((int)Nx.bytesA);
          g4_dBn[0] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)nb); g4_dBn[1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)((nb >> 32) & 0xffffu));
          g4_dBn[2] = (uint32_t)__builtin_amdgcn_readfirstlane((int)Nx.bytesB);
        }
        G4_ASM_LOOP_P();
      } else {
#ifdef G4_ASM_HAVE_B3
        if (nkt >= 80) G4_ASM_LOOP_B3();
        else
#endif
        G4_ASM_LOOP();
      }
    }
  } else if constexpr (!PERSIST) run_loop(std::false_type{});       // (persistent launches: K % 64 == 0, launcher-checked)
#else
  if ((Kv & 63) == 0) run_loop(std::true_type{}); else run_loop(std::false_type{});
#endif
#undef G4_MFMA
#undef G4_SB
#endif
  // the asm MFMAs are invisible to the hazard recognizer: let the last accumulator writes retire before reading them
  if constexpr (PERSIST) asm volatile("s_nop 15\n s_nop 15" ::: "memory");          // (the next tile's K tile 1 stays in flight)
  else asm volatile("s_waitcnt vmcnt(0)\n s_nop 15\n s_nop 15" ::: "memory");

  // PERSIST: the epilogue's addresses are lane / wave arithmetic that does not change from tile to tile, and hipcc hoists all of
  // it out of the tile loop — i.e. ACROSS the K loop, where 404 of the 512 registers are taken: ~100 spilled VGPRs, reloaded
  // through scratch behind the LDS-DMA queue (measured: -6 ... -12 %).  Laundering the ids here makes the derived values per tile.
  if constexpr (PERSIST) {
    asm volatile("" : "+v"(tid), "+s"(wave));
    lane = tid & 63; li = lane & 15; g = lane >> 4;
    cchunk = G4_PAD ? (lane & 7) : ((lane & 7) ^ (lane >> 3));
    wr = wave >> 1; wc = wave & 1;
  }
#undef G4_BARRIER

  // ---------------------------------------------------------------- epilogue (a lambda: `return` = this tile is done)
  auto epilogue = [&]() {
  if constexpr (MODE == 1) {      // lane: 16 gate columns (acc[mt][0..3]) and the same 16 up columns (acc[mt][4..7])
    constexpr int H2 = G4_SPLIT_COLS ? 32 : 8;                  // distance between the lane's two 8-column runs (see g4_bcol)
    const int cs = col0 + wc * 64 + g * (G4_SPLIT_COLS ? 8 : 16);
    bf16_t* Cact = (bf16_t*)p.C + (long long)bz * p.sC;
    bf16_t* Cgu = p.C2 ? (bf16_t*)p.C2 + (long long)bz * p.sC2 : nullptr;
    const int Mz = min((Mv + 7) & ~7, p.M);      // rows Mv..Mz-1 are zero-filled: a k_valid wgrad reads whole 8-row chunks
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const int row = row0 + wr * 128 + mt * 16 + li;
      if (row >= Mz || cs >= p.N) continue;        // N % 8 == 0; a lane's 16 columns may straddle N (second half dropped)
      bf16_t* ap = Cact + (long long)row * p.ldc + cs;
      const bool hi = (cs + H2 < p.N);
      if (row >= Mv) { *(u32x4*)ap = (u32x4){0u, 0u, 0u, 0u}; if (hi) *(u32x4*)(ap + H2) = (u32x4){0u, 0u, 0u, 0u}; continue; }
#pragma unroll
      for (int hx = 0; hx < 2; ++hx) {
        if (hx && !hi) continue;
        float v[16];
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] = acc[mt][hx * 2 + (e >> 2)][e & 3]; v[8 + e] = acc[mt][4 + hx * 2 + (e >> 2)][e & 3]; }
        *(u32x4*)(ap + hx * H2) = swiglu_pairs(v);
        if (Cgu) {
          bf16_t* gp = Cgu + (long long)row * p.ldc2 + cs + hx * H2;
          *(u32x4*)gp = (u32x4){pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
          *(u32x4*)(gp + p.N) = (u32x4){pack2bf(v[8], v[9]), pack2bf(v[10], v[11]), pack2bf(v[12], v[13]), pack2bf(v[14], v[15])};
        }
      }
    }
    return;
  }
  char* Cb = (char*)p.C + (long long)bz * p.sC * (p.out_f32 ? 4 : 2);
  // Epilogue variants.  Each is a lambda over ONE 16-row x 64-column piece, called 16 times with literal (gq, mt) so
  // every accumulator index is a constant; the variant is chosen ONCE, outside, so a tile only ever fetches the
  // instructions of the variant it runs (interleaving all variants in 16 copies cost ~17 us of instruction-cache
  // misses per tile).
#define G4_FOR_ALL_TILES(F)                                                                         \
  do {                                                                                              \
    F(0, 0); F(0, 1); F(0, 2); F(0, 3); F(0, 4); F(0, 5); F(0, 6); F(0, 7);                         \
    F(1, 0); F(1, 1); F(1, 2); F(1, 3); F(1, 4); F(1, 5); F(1, 6); F(1, 7);                         \
  } while (0)
  auto epi_swiglu_bwd = [&](const int gq, const int mt) {    // lmod_gemm_swiglu_bwd_bf16, N % 16 == 0
    const int cb = col0 + wc * 128 + gq * 64 + g * 16;
    const int row = row0 + wr * 128 + mt * 16 + li;
    if (cb >= p.N || row >= min((Mv + 7) & ~7, p.M)) return;
    bf16_t* op = (bf16_t*)p.C + (long long)bz * p.sC + (long long)row * p.ldc + cb;
    if (row >= Mv) {
#pragma unroll
      for (int hx = 0; hx < 2; ++hx) { *(u32x4*)(op + hx * 8) = (u32x4){0u, 0u, 0u, 0u}; *(u32x4*)(op + p.N + hx * 8) = (u32x4){0u, 0u, 0u, 0u}; }
      return;
    }
    const bf16_t* gp = (const bf16_t*)p.C2 + (long long)bz * p.sC2 + (long long)row * p.ldc2 + cb;
#pragma unroll
    for (int hx = 0; hx < 2; ++hx) {
      float d8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) d8[e] = acc[mt][gq * 4 + hx * 2 + (e >> 2)][e & 3];
      u32x4 og, ou;
      swiglu_bwd8(d8, *(const u32x4*)(gp + hx * 8), *(const u32x4*)(gp + p.N + hx * 8), og, ou);
      *(u32x4*)(op + hx * 8) = og;
      *(u32x4*)(op + p.N + hx * 8) = ou;
    }
  };
  auto epi_partial = [&](const int gq, const int mt) {       // split-K: partial tile -> workspace, lane-linear
    float* wp = p.ws + (((long long)split * tpb + id) * 64 + (mt * 2 + gq) * 4) * 1024 + tid * 4;
#pragma unroll
    for (int x = 0; x < 4; ++x)
      asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(wp + x * 1024), "v"(acc[mt][gq * 4 + x]) : "memory");
  };
  // bias of this lane's 2 x 16 columns, loaded once (not per 16-row piece)
  float biav[2][16];
#pragma unroll
  for (int gq = 0; gq < 2; ++gq)
#pragma unroll
    for (int x = 0; x < 16; ++x) biav[gq][x] = 0.f;
  if (MODE != 8 && p.bias) {       // (MODE 8: no bias — the decoder's o / down projections have none; host-checked)             // clamped, unconditional loads: 32 in flight, one wait (a per-element condition made each its own round trip)
#pragma unroll
    for (int gq = 0; gq < 2; ++gq)
#pragma unroll
      for (int x = 0; x < 16; ++x) {
        const int c = ((MODE == 7 || MODE == 8 || MODE == 5) && G4_SPLIT_COLS) ? col0 + wc * 128 + gq * 64 + (x >> 3) * 32 + g * 8 + (x & 7)
                                                   : col0 + wc * 128 + gq * 64 + g * 16 + x;
        const float bv = bf2f(p.bias[min(c, p.N - 1)]);
        biav[gq][x] = (c < p.N) ? bv : 0.f;
      }
  }
  if constexpr (MODE == 5) {
    // fused q/k/v projection + bias + RoPE (lmod_gemm_qkv_rope_bf16, hd 128).  A wave column is 128 output columns = ONE head, and a
    // lane holds columns g*16 .. +15 of both 64-column halves: the rotate_half partner of feature f (f + 64 / f - 64) sits in the
    // SAME lane (acc[mt][nt] and acc[mt][nt + 4]) — no exchange through LDS and no barrier, unlike the 8-wave kernel's epilogue.
    // Same roundings as GEMM + bias -> bf16, then rope_kernel.  N % 128 == 0 (host-checked).  Memory order as in the other
    // epilogues: positions first, then cos / sin of two row tiles at a time, issued ahead of the previous two's stores.
    bf16_t* Cb5 = (bf16_t*)p.C + (long long)bz * p.sC;
    constexpr int GW = G4_SPLIT_COLS ? 8 : 16, H2 = G4_SPLIT_COLS ? 32 : 8;   // the lane's two 8-feature runs: g*GW + 0..7 and + H2 (see g4_bcol)
    const int cb = col0 + wc * 128 + g * GW;
    if (cb >= p.N) return;
    auto rowof = [&](int mt) { return row0 + wr * 128 + mt * 16 + li; };
    auto own = [&](const int mt, const int gq, u32x4& lo, u32x4& hi) {      // bf16(acc + bias): features gq*64 + g*16 + 0..7 | 8..15
      float v[16];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[nt * 4 + q] = acc[mt][gq * 4 + nt][q] + biav[gq][nt * 4 + q];
      lo = (u32x4){pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
      hi = (u32x4){pack2bf(v[8], v[9]), pack2bf(v[10], v[11]), pack2bf(v[12], v[13]), pack2bf(v[14], v[15])};
    };
    if (col0 >= p.rope_cols) {                        // V columns: bias only (workgroup-uniform: rope_cols is a multiple of 256)
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        const int row = rowof(mt);
        if (row >= Mv) continue;
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
          u32x4 lo, hi;
          own(mt, gq, lo, hi);
          bf16_t* op = Cb5 + (long long)row * p.ldc + cb + gq * 64;
          st_c(op, lo);
          st_c(op + H2, hi);
        }
      }
      return;
    }
    int ps[8];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) ps[mt] = p.rope_pos[(long long)bz * p.M + min(rowof(mt), p.M - 1)];
    u32x4 CC[4][2][2][2], SS[4][2][2][2];             // [batch of 2 row tiles][m][gq][hx]
    auto ld = [&](const int b2) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const bf16_t* cp = p.rope_cos + (long long)ps[b2 * 2 + m] * 128 + g * GW;
        const bf16_t* sp = p.rope_sin + (long long)ps[b2 * 2 + m] * 128 + g * GW;
#pragma unroll
        for (int gq = 0; gq < 2; ++gq)
#pragma unroll
          for (int hx = 0; hx < 2; ++hx) {
            CC[b2][m][gq][hx] = *(const u32x4*)(cp + gq * 64 + hx * H2);
            SS[b2][m][gq][hx] = *(const u32x4*)(sp + gq * 64 + hx * H2);
          }
      }
    };
    auto cmp = [&](const int b2) {                    // results replace CC
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int mt = b2 * 2 + m;
        u32x4 x1[2], x2[2];                           // first half (features < 64) / second half, [hx]
        own(mt, 0, x1[0], x1[1]);
        own(mt, 1, x2[0], x2[1]);
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) {
          u32x4 o1, o2;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float a0 = bflo(x1[hx][k]), a1 = bfhi(x1[hx][k]), b0 = bflo(x2[hx][k]), b1 = bfhi(x2[hx][k]);
            const uint32_t c1 = CC[b2][m][0][hx][k], s1 = SS[b2][m][0][hx][k], c2 = CC[b2][m][1][hx][k], s2 = SS[b2][m][1][hx][k];
            // first half: x1*cos + (-x2)*sin ; second half: x2*cos + x1*sin  (rope_kernel's roundings)
            o1[k] = pack2bf(bfround(a0 * bflo(c1)) + bfround(-b0 * bflo(s1)), bfround(a1 * bfhi(c1)) + bfround(-b1 * bfhi(s1)));
            o2[k] = pack2bf(bfround(b0 * bflo(c2)) + bfround(a0 * bflo(s2)), bfround(b1 * bfhi(c2)) + bfround(a1 * bfhi(s2)));
          }
          CC[b2][m][0][hx] = o1;
          CC[b2][m][1][hx] = o2;
        }
      }
    };
    auto pin = [&](const int b2) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
        asm volatile("" : "+v"(CC[b2][m][0][0]), "+v"(CC[b2][m][0][1]), "+v"(CC[b2][m][1][0]), "+v"(CC[b2][m][1][1]) : : "memory");
    };
    auto st = [&](const int b2) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int row = rowof(b2 * 2 + m);
        if (row >= Mv) continue;
        bf16_t* op = Cb5 + (long long)row * p.ldc + cb;
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) { st_c(op + gq * 64, CC[b2][m][gq][0]); st_c(op + gq * 64 + H2, CC[b2][m][gq][1]); }
      }
    };
    ld(0); cmp(0); pin(0); ld(1); st(0); cmp(1); pin(1); ld(2); st(1); cmp(2); pin(2); ld(3); st(2); cmp(3); st(3);
    return;
  }
  auto epi_bf16 = [&](const int gq, const int mt) {          // the common case: bf16 C = act(acc + bias), vector stores
    constexpr bool SPLIT = (MODE == 7) && G4_SPLIT_COLS;       // v[0..7] | v[8..15]: columns g*8.. and 32 + g*8.. (see g4_bcol)
    const int cb = col0 + wc * 128 + gq * 64 + (SPLIT ? g * 8 : g * 16);
    const int row = row0 + wr * 128 + mt * 16 + li;
    if (row >= Mv || cb >= p.N) return;
    float v[16];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) v[nt * 4 + q] = acc[mt][gq * 4 + nt][q] + biav[gq][nt * 4 + q];
    if (p.act) {
#pragma unroll
      for (int x = 0; x < 16; ++x) v[x] = act_apply(bfround(v[x]), p.act);
    }
    bf16_t* cp = (bf16_t*)Cb + (long long)row * p.ldc + cb;
    constexpr int H2 = SPLIT ? 32 : 8;                          // element distance between the lane's two 8-column runs
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c0 = cb + h * H2;
      if ((c0 + 8 <= p.N) && p.vec_ok) {
        *(u32x4*)(cp + h * H2) = (u32x4){pack2bf(v[8 * h], v[8 * h + 1]), pack2bf(v[8 * h + 2], v[8 * h + 3]),
                                         pack2bf(v[8 * h + 4], v[8 * h + 5]), pack2bf(v[8 * h + 6], v[8 * h + 7])};
      } else {
#pragma unroll
        for (int x = 0; x < 8; ++x) if (c0 + x < p.N) cp[h * H2 + x] = f2bf(v[8 * h + x]);
      }
    }
    // persistent form: 64 registers of the next tile's fragments stay live across the epilogue; left free, hipcc converts all 16
    // pieces before the first store (128 packed registers) and spills the fragments around it
    if constexpr (PERSIST) __builtin_amdgcn_sched_barrier(0);
  };
  auto epi_f32_acc = [&](const int gq, const int mt) {       // weight gradients: fp32 C +=, vector path
    const int cb = col0 + wc * 128 + gq * 64 + g * 16;
    const int row = row0 + wr * 128 + mt * 16 + li;
    if (row >= Mv || cb >= p.N) return;
    float* cp = (float*)Cb + (long long)row * p.ldc + cb;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      if (cb + 4 * x + 4 <= p.N) {
        f32x4 o = acc[mt][gq * 4 + x];
        o += *(f32x4*)(cp + 4 * x);
        *(f32x4*)(cp + 4 * x) = o;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) if (cb + 4 * x + q < p.N) cp[4 * x + q] += acc[mt][gq * 4 + x][q];
      }
    }
  };
  auto epi_generic = [&](const int gq, const int mt) {       // the rest (f32 store, bias / act with f32, bf16 accumulate)
    const int cb = col0 + wc * 128 + gq * 64 + g * 16;
    const int row = row0 + wr * 128 + mt * 16 + li;
    if (row >= Mv || cb >= p.N) return;
    float v[16];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) v[nt * 4 + q] = acc[mt][gq * 4 + nt][q] + biav[gq][nt * 4 + q];
    if (p.act) {
#pragma unroll
      for (int x = 0; x < 16; ++x) v[x] = act_apply(bfround(v[x]), p.act);
    }
    if (p.out_f32) {
      float* cp = (float*)Cb + (long long)row * p.ldc + cb;
      if ((cb + 16 <= p.N) && p.vec_ok) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          f32x4 o = {v[4 * x], v[4 * x + 1], v[4 * x + 2], v[4 * x + 3]};
          if (p.accumulate) o += *(f32x4*)(cp + 4 * x);
          *(f32x4*)(cp + 4 * x) = o;
        }
      } else {
#pragma unroll
        for (int x = 0; x < 16; ++x) if (cb + x < p.N) cp[x] = p.accumulate ? cp[x] + v[x] : v[x];
      }
    } else {
      bf16_t* cp = (bf16_t*)Cb + (long long)row * p.ldc + cb;
#pragma unroll
      for (int x = 0; x < 16; ++x) if (cb + x < p.N) cp[x] = f2bf(p.accumulate ? bf2f(cp[x]) + v[x] : v[x]);
    }
  };
  // MODE 7 / 4 / 6: ONE epilogue variant per instantiation (the launcher guarantees its preconditions).  With all of them behind run-time
  // branches in MODE 0 the 256 accumulators per lane leave the register allocator no room: 109 spilled VGPRs and 1271 instead of
  // ~1450 TF on the plain GEMM.  MODE 0 keeps every option (split-K, f32 store, bf16 accumulate, ...).
  if constexpr (MODE == 7) G4_FOR_ALL_TILES(epi_bf16);                 // bf16 C = act(acc + bias)
  else if constexpr (MODE == 4) {
    // fused SwiGLU backward, memory order as in gemm_256_kernel<4>: the [gate | up] loads of 4 of the 16 pieces at a time, each batch's loads issued
    // AHEAD of the previous batch's stores (8 at a time spilled beside the 256 accumulators) (piece = (gq, mt): 16 rows x 64 columns; a lane holds 16 columns of it)
    const int Mz4 = min((Mv + 7) & ~7, p.M);
    auto rowof = [&](int mt) { return row0 + wr * 128 + mt * 16 + li; };
    auto colof = [&](int gq) { return col0 + wc * 128 + gq * 64 + g * 16; };
    u32x4 G[4][4][2], U[4][4][2];                    // [batch][piece in batch: gq * 2 + (mt & 1)][hx]; batch = mt >> 1
    auto ld = [&](const int b) {
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) {
        const int gq = pc >> 1, mt = b * 2 + (pc & 1);
        const bf16_t* gp = (const bf16_t*)p.C2 + (long long)bz * p.sC2 + (long long)min(rowof(mt), p.M - 1) * p.ldc2 + min(colof(gq), p.N - 16);
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) { G[b][pc][hx] = *(const u32x4*)(gp + hx * 8); U[b][pc][hx] = *(const u32x4*)(gp + p.N + hx * 8); }
      }
    };
    auto cmp = [&](const int b) {
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) {
        const int gq = pc >> 1, mt = b * 2 + (pc & 1);
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) {
          float d8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) d8[e] = acc[mt][gq * 4 + hx * 2 + (e >> 2)][e & 3];
          u32x4 og, ou;
          swiglu_bwd8(d8, G[b][pc][hx], U[b][pc][hx], og, ou);
          G[b][pc][hx] = og; U[b][pc][hx] = ou;
        }
      }
    };
    auto pin = [&](const int b) {
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) asm volatile("" : "+v"(G[b][pc][0]), "+v"(G[b][pc][1]), "+v"(U[b][pc][0]), "+v"(U[b][pc][1]) : : "memory");
    };
    auto st = [&](const int b) {
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) {
        const int gq = pc >> 1, mt = b * 2 + (pc & 1), row = rowof(mt), cb = colof(gq);
        if (cb >= p.N || row >= Mv) continue;
        bf16_t* op = (bf16_t*)p.C + (long long)bz * p.sC + (long long)row * p.ldc + cb;
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) { *(u32x4*)(op + hx * 8) = G[b][pc][hx]; *(u32x4*)(op + p.N + hx * 8) = U[b][pc][hx]; }
      }
    };
#ifndef G4_SB_DEPTH
#define G4_SB_DEPTH 2                // [gate | up] batches in flight (round 6 experiment, bit-identical): 1 / 2 / 3 measure THE SAME (profiles/
#endif                               //   r06_epilogue_ab.md: 0.898 / 0.90 / 0.897 ms dense) — this epilogue is not a chain of memory round trips.
                                     //   Its ISA is 3454 VALU instructions per lane and tile, 512 of them quarter-rate (v_exp, v_rcp): ~11 us of
                                     //   issue time beside a 49 us K loop, plus the [gate | up] read and [dgate | dup] write of 512 KiB per tile.
#if G4_SB_DEPTH == 3
    ld(0); ld(1); ld(2); cmp(0); pin(0); st(0); ld(3); cmp(1); pin(1); st(1); cmp(2); pin(2); st(2); cmp(3); st(3);
#elif G4_SB_DEPTH == 2
    ld(0); ld(1); cmp(0); pin(0); st(0); ld(2); cmp(1); pin(1); st(1); ld(3); cmp(2); pin(2); st(2); cmp(3); st(3);
#else
    ld(0); cmp(0); pin(0); ld(1); st(0); cmp(1); pin(1); ld(2); st(1); cmp(2); pin(2); ld(3); st(2); cmp(3); st(3);
#endif
    if (Mz4 != Mv) {                                 // rows Mv .. roundup8(Mv)-1 are zeroed (a k_valid wgrad reads whole 8-row chunks)
#pragma unroll
      for (int pc = 0; pc < 16; ++pc) {
        const int gq = pc >> 3, mt = pc & 7, row = rowof(mt), cb = colof(gq);
        if (cb >= p.N || row < Mv || row >= Mz4) continue;
        bf16_t* op = (bf16_t*)p.C + (long long)bz * p.sC + (long long)row * p.ldc + cb;
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) { *(u32x4*)(op + hx * 8) = (u32x4){0u, 0u, 0u, 0u}; *(u32x4*)(op + p.N + hx * 8) = (u32x4){0u, 0u, 0u, 0u}; }
      }
    }
  } else if constexpr (MODE == 8) {
    // bf16 C = bf16(res + bf16(acc)): the decoder layer's residual add (qwen2/modeling_qwen2.py:757-775: hidden = residual +
    // o_proj(...) / + mlp(...)) in the epilogue of the o / down projection — the roundings of GEMM -> bf16, then the add of
    // rmsnorm_fwd_kernel's residual path (rowops.hip), so the result is that pair's, bit for bit, and the norm that follows reads
    // ONE tensor instead of two and writes one instead of two.  res = p.C2 (row stride ldc2), N % 8 == 0, 16-byte aligned rows
    // (host-checked).  Memory order as in MODE 6, in batches of one row tile (2 pieces: four at a time spilled beside the 256 accumulators): each
    // batch's residual loads are issued ahead of the previous batch's stores.  Columns: the split layout of MODE 7 (runs g*8 .. +7 and 32 + g*8 .. +7 of every 64-column group).
    constexpr int GW = G4_SPLIT_COLS ? 8 : 16, H2 = G4_SPLIT_COLS ? 32 : 8;
    auto rowof = [&](int mt) { return row0 + wr * 128 + mt * 16 + li; };
    auto colof = [&](int gq) { return col0 + wc * 128 + gq * 64 + g * GW; };
    const bf16_t* Rb = (const bf16_t*)p.C2 + (long long)bz * p.sC2;
    u32x4 RR[8][2][2];                               // eight batches of 2 pieces (one row tile, both 64-column groups): 16 registers each
    auto ld = [&](const int b) {
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {
        const int gq = pc, mt = b;
        const bf16_t* rp = Rb + (long long)min(rowof(mt), p.M - 1) * p.ldc2;
#pragma unroll
        for (int h = 0; h < 2; ++h) RR[b][pc][h] = *(const u32x4*)(rp + min(colof(gq) + h * H2, p.N - 8));
      }
    };
    auto cmp = [&](const int b) {
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {
        const int gq = pc, mt = b;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          u32x4 o;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int e = 2 * k;
            const uint32_t d = pack2bf(acc[mt][gq * 4 + h * 2 + (e >> 2)][e & 3], acc[mt][gq * 4 + h * 2 + ((e + 1) >> 2)][(e + 1) & 3]);
            const uint32_t r = RR[b][pc][h][k];
            o[k] = pack2bf(bflo(d) + bflo(r), bfhi(d) + bfhi(r));
          }
          RR[b][pc][h] = o;
        }
      }
    };
    auto pin = [&](const int b) {
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) asm volatile("" : "+v"(RR[b][pc][0]), "+v"(RR[b][pc][1]) : : "memory");
    };
    auto st = [&](const int b) {
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {
        const int gq = pc, mt = b, row = rowof(mt);
        if (row >= Mv) continue;
        bf16_t* cp = (bf16_t*)Cb + (long long)row * p.ldc + colof(gq);
#pragma unroll
        for (int h = 0; h < 2; ++h) if (colof(gq) + h * H2 + 8 <= p.N) st_c(cp + h * H2, RR[b][pc][h]);
      }
    };
#ifndef G4_RES_DEPTH
#define G4_RES_DEPTH 4               // residual batches in flight (round 6; 1 = the round-4 order: one 16 KiB batch at a time).  Bit-identical; measured
#endif                               //   +1.5 ... 2 % at [32768 x 2048 x 2048] and [32768 x 4096 x 4096], level at K >= 5504 (profiles/r06_epilogue_ab.md)
#if G4_RES_DEPTH > 1
#pragma unroll
    for (int b = 0; b < G4_RES_DEPTH; ++b) ld(b);
#pragma unroll
    for (int b = 0; b < 8; ++b) { cmp(b); pin(b); st(b); if (b + G4_RES_DEPTH < 8) ld(b + G4_RES_DEPTH); }
#else
    ld(0); cmp(0); pin(0);
#pragma unroll
    for (int b = 1; b < 8; ++b) { ld(b); st(b - 1); cmp(b); pin(b); }
    st(7);
#endif
  } else if constexpr (MODE == 6) {
    // fp32 C += acc (no bias / act / split; C and ldc 16-byte aligned, N % 16 == 0 not required: whole 4-column groups), four
    // batches of 4 pieces as above
    auto rowof = [&](int mt) { return row0 + wr * 128 + mt * 16 + li; };
    auto colof = [&](int gq) { return col0 + wc * 128 + gq * 64 + g * 16; };
    const bool vec = (p.N & 3) == 0;                 // (N % 4 != 0: the generic piece-by-piece variant)
    if (!vec) { G4_FOR_ALL_TILES(epi_f32_acc); }
    else {
      f32x4 R[4][4][4];
      auto ld = [&](const int b) {
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
          const int gq = pc >> 1, mt = b * 2 + (pc & 1);
          const float* cp = (const float*)Cb + (long long)min(rowof(mt), p.M - 1) * p.ldc;
#pragma unroll
          for (int x = 0; x < 4; ++x) R[b][pc][x] = *(const f32x4*)(cp + min(colof(gq) + 4 * x, p.N - 4));
        }
      };
      auto add = [&](const int b) {
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
          const int gq = pc >> 1, mt = b * 2 + (pc & 1);
#pragma unroll
          for (int x = 0; x < 4; ++x) R[b][pc][x] += acc[mt][gq * 4 + x];
        }
      };
      auto pin = [&](const int b) {
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) asm volatile("" : "+v"(R[b][pc][0]), "+v"(R[b][pc][1]), "+v"(R[b][pc][2]), "+v"(R[b][pc][3]) : : "memory");
      };
      auto st = [&](const int b) {
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
          const int gq = pc >> 1, mt = b * 2 + (pc & 1), row = rowof(mt), cb = colof(gq);
          if (row >= Mv) continue;
          float* cp = (float*)Cb + (long long)row * p.ldc + cb;
#pragma unroll
          for (int x = 0; x < 4; ++x) if (cb + 4 * x + 4 <= p.N) *(f32x4*)(cp + 4 * x) = R[b][pc][x];
        }
      };
      ld(0); add(0); pin(0); ld(1); st(0); add(1); pin(1); ld(2); st(1); add(2); pin(2); ld(3); st(2); add(3); st(3);
    }
  }
  else {
  if (p.act == 3) G4_FOR_ALL_TILES(epi_swiglu_bwd);
  else if (p.splitk > 1) G4_FOR_ALL_TILES(epi_partial);
  else if (!p.out_f32 && !p.accumulate) G4_FOR_ALL_TILES(epi_bf16);
  else if (p.out_f32 && p.accumulate && !p.act && !p.bias && p.vec_ok) G4_FOR_ALL_TILES(epi_f32_acc);
  else G4_FOR_ALL_TILES(epi_generic);
  }
#undef G4_FOR_ALL_TILES
  if (MODE == 0 && p.splitk > 1) {
    // deterministic split-K reduction (see gemm_256_kernel): the last split to arrive adds all partials in split order
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = (int*)smem;
    if (tid == 0)
      *flag = (__hip_atomic_fetch_add(p.counters + id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p.splitk - 1);
    __syncthreads();
    if (!*flag) return;
    float* Cf = (float*)p.C;
    for (int mg = 0; mg < 16; ++mg) {               // (mt, gq) pairs
      const int mt = mg >> 1, gq = mg & 1;
      const int row = row0 + wr * 128 + mt * 16 + li;
      const int cb = col0 + wc * 128 + gq * 64 + g * 16;
      const bool live = (row < Mv) && (cb < p.N);   // N % 16 == 0 on this path
      f32x4 o[4];
#pragma unroll
      for (int x = 0; x < 4; ++x)
        o[x] = live ? *(f32x4*)(Cf + (long long)row * p.ldc + cb + 4 * x) : (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int s0 = 0; s0 < p.splitk; s0 += 4) {
        f32x4 part[4][4];
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
          const float* wp = p.ws + (((long long)min(s0 + ds, p.splitk - 1) * tpb + id) * 64 + mg * 4) * 1024 + tid * 4;
#pragma unroll
          for (int x = 0; x < 4; ++x)
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(part[ds][x]) : "v"(wp + x * 1024) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int ds = 0; ds < 4; ++ds)
          if (s0 + ds < p.splitk) {
#pragma unroll
            for (int x = 0; x < 4; ++x) {
              asm volatile("" : "+v"(part[ds][x]));
              o[x] += part[ds][x];
            }
          }
      }
      if (live) {
#pragma unroll
        for (int x = 0; x < 4; ++x) *(f32x4*)(Cf + (long long)row * p.ldc + cb + 4 * x) = o[x];
      }
    }
    if (tid == 0) __hip_atomic_store(p.counters + id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  };
  epilogue();
  if constexpr (!PERSIST) return;
  if (!more) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }       // (the dead look-ahead pieces of the last tile)
  T = Nx;
  first = false;
  }                                                        // tile loop
}

// =============================================================================================
// gemm4t_kernel (round 4): the weight-gradient form on the 4-wave tile, WITHOUT transposed copies — fp32 C[b] (M x N) += A[b]^T B[b]
// with A = dY [K x M] and B = X [K x N] reduction-major, as autograd holds them (K = tokens).  The step used to transpose both
// operands (transpose_bf16: 2.8 % of GPU time, a quarter of the weight-gradient GEMMs' own time) and run the NT kernel.
// Tile, wave layout, accumulators and epilogue batching are gemm4_kernel's; the K loop is its own asm statement
// (gemm4t_loop_asm.h, written by tools/gen_gemm4t_loop.py — LDS image, fragment registers and the walking descriptors are
// described there).  A reduction-major operand's K edge is a ROW boundary: the descriptors' byte counts cut it exactly, so any K
// and a per-batch k_valid (MoE routed rows) need no rounding, no zero-filled padding and no per-lane mask.
// C layout: natural column order — a lane owns 4 contiguous columns per 16-column MFMA tile (columns nt*16 + 4g .. +3 of row li),
// i.e. one 16-byte fp32 access, four lanes = 64 contiguous bytes per row.
// SPLIT: deterministic split-K as in gemm_256_kernel (partials to the workspace, the last split to arrive adds them in split order).
// Host-checked: M % 8 == 0, N % 8 == 0, lda / ldb % 8 == 0, 16-byte aligned operands, every operand window below 2 GiB.
// =============================================================================================
#include "gemm4t_loop_asm.h"
#ifndef G4T_OB
#define G4T_OB 0                    // 1: the one-barrier schedule of round 5 (tools/gen_gemm4t_loop.py --variant ob): reads at 2 per 3 MFMAs, barrier at MFMA 56
#endif
#if G4T_OB
#include "gemm4t_loop_asm_ob.h"
#endif
#define G4T_SUB 8448
#define G4T_PIECE 1056
#define G4T_OPB 33792
#define G4T_STAGE 67584
// KIND 0: C += A^T B; 1: deterministic split-K; 2: attention dQ = scale * dS K from the spilled dS^T (see GemmP::at_*): one workgroup per
// (sample, head, 256-query block), reduction over the keys the block attends to, bf16 store with the rotary embedding's gradient map.
template <int KIND>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm4t_kernel(GemmP p) {
  constexpr bool SPLIT = KIND == 1, ATT = KIND == 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x 67584
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  const int tpb = p.tiles_m * p.tiles_n;
  int id = ATT ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
  int bz = id / tpb, split = 0;
  if (SPLIT) { split = bz; bz = 0; id -= split * tpb; }
  int r = id - bz * tpb;
  if (!SPLIT && p.k_valid && p.batch > 1 && p.batch <= GEMM_MAX_GROUPS && !(tpb & 7)) {
    // a live reduction length per batch (MoE experts): every XCD takes 1/8 of EVERY expert's tiles, longest first (see gemm_256_kernel)
    const int tpb8 = tpb >> 3, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int rnk = k / tpb8;
    int sel = rnk;
    for (int e = 0; e < p.batch; ++e) {
      const int ke = p.k_valid[e];
      int rank = 0;
      for (int f = 0; f < p.batch; ++f) {
        const int kf = p.k_valid[f];
        rank += (kf > ke || (kf == ke && f < e)) ? 1 : 0;
      }
      if (rank == rnk) sel = e;
    }
    bz = sel;
    r = xcd * tpb8 + (k - rnk * tpb8);
    id = bz * tpb + r;
  }
  const int GROUP_M = G256_GROUP_M;
  const int grp = r / (GROUP_M * p.tiles_n);
  const int first_m = grp * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int rr = r - grp * GROUP_M * p.tiles_n;
  const int tm = first_m + rr % gsz, tn = rr / gsz;
  int Kv = p.k_valid ? min(p.k_valid[bz], p.K) : p.K;
  int kbeg = 0;
  if (SPLIT) {
    kbeg = split * p.kchunk;
    Kv = max(0, min(Kv - kbeg, p.kchunk));      // an empty split still arrives at the semaphore with a zero tile
  }
  int row0 = tm * 256, col0 = tn * 256;
  int colsA = min(256, p.M - row0), colsB = min(256, p.N - col0);
  const bf16_t* Ab = p.A + (long long)bz * p.sA + (long long)kbeg * p.lda + row0;
  const bf16_t* Bb = p.B + (long long)bz * p.sB + (long long)kbeg * p.ldb + col0;
  int at_b = 0, at_h = 0, at_q0 = 0;
  if constexpr (ATT) {
    // workgroup id = (sample, query block, head) with the head fastest: consecutive ids go to the 8 XCDs round-robin, so with a multiple
    // of 8 heads every block of a head meets its K rows in ONE XCD's L2; the longest reductions (last query blocks) are dispatched first
    at_h = id % p.at_nh;
    const int r2 = id / p.at_nh, qb = p.at_nqb - 1 - r2 % p.at_nqb;
    at_b = r2 / p.at_nqb;
    at_q0 = qb * 256;
    Kv = p.at_causal ? min(p.at_S, at_q0 + 256) : p.at_S;
    if (p.at_seqlens) Kv = min(Kv, (min(p.at_seqlens[at_b], p.at_S) + 255) & ~255);     // key blocks past the sample's length were never written
    row0 = 0; col0 = 0; colsA = 256; colsB = 128;
    Ab = p.A + ((long long)at_b * p.at_nh + at_h) * p.at_S * p.at_S + (long long)(at_q0 >> 5) * p.at_S * 32;     // the block's first query tile
    Bb = p.B + (long long)at_b * p.at_S * p.ldb + (at_h / p.at_group) * 128;
  }
  const int nkt = (Kv + 63) >> 6;
  if (!SPLIT && !ATT && nkt == 0) return;
  const uint32_t ksA = ATT ? 4096u : (uint32_t)p.lda * 128u, ksB = (uint32_t)p.ldb * 128u;          // bytes per K tile (64 rows)
  // live bytes of the window.  ATT: the 8 query tiles of the block are S * 64 bytes apart, a K tile is two 2 KiB key strips in each of them
  const uint32_t remA = ATT ? (uint32_t)(7 * p.at_S * 64 + Kv * 64)
                            : (Kv > 0 ? (uint32_t)(((long long)(Kv - 1) * p.lda + colsA) * 2) : 0u);
  const uint32_t remB = Kv > 0 ? (uint32_t)(((long long)(Kv - 1) * p.ldb + colsB) * 2) : 0u;

  // staging: wave w fills sub-image w (64 columns) of both operands, 8 pieces of 8 k-rows x 128 bytes; piece j, LDS row lane >> 3
  // holds k-row 8q + u with u = (j & 3) + 4 (row & 1), q = (j >> 2) + 2 (row >> 1)
  uint32_t voA[8], voB[8];
  {
    const int lr = lane >> 3, col = wave * 64 + (lane & 7) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * ((j >> 2) + 2 * (lr >> 1)) + (j & 3) + 4 * (lr & 1);
      if constexpr (ATT) {
        // The reduction order inside a K tile is free when both operands agree: LDS row (piece j, lane row lr) holds key 8 j + lr of
        // the tile for A and B alike, so that a piece of A is 8 CONSECUTIVE keys of the spilled dS^T — in the dK/dV kernel's layout
        // [query tile][key strip][half][key in strip][16 queries] (attn_bwd2.hip) four runs of 256 contiguous bytes — and a piece of
        // B 8 consecutive K rows.
        const int kk = 8 * j + lr;
        voA[j] = (uint32_t)((col >> 5) * p.at_S * 64 + (kk >> 5) * 2048 + ((col >> 4) & 1) * 1024 + (kk & 31) * 32 + (col & 15) * 2);
        voB[j] = (col < colsB) ? (uint32_t)((kk * p.ldb + col) * 2) : GEMM_OOB;
      } else {
        voA[j] = (col < colsA) ? (uint32_t)((k * p.lda + col) * 2) : GEMM_OOB;
        voB[j] = (col < colsB) ? (uint32_t)((k * p.ldb + col) * 2) : GEMM_OOB;
      }
    }
  }
  auto rsrc_at = [&](const bf16_t* base, const uint32_t ks, const uint32_t rem, const int t) {     // the window from K tile t on
    const unsigned long long q = (unsigned long long)base + (unsigned long long)ks * (unsigned)t;
    const unsigned long long qu = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(q >> 32)) << 32) |
                                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)q);
    const unsigned long long off = (unsigned long long)ks * (unsigned)t;
    const uint32_t left = rem > off ? (uint32_t)(rem - off) : 0u;
    return __builtin_amdgcn_make_buffer_rsrc((void*)qu, 0, __builtin_amdgcn_readfirstlane((int)left), 0x00020000);
  };

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (nkt > 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {            // K tiles 0 and 1 -> the two stages (a tile past the end: zero records, zero fill)
      const __amdgpu_buffer_rsrc_t ra = rsrc_at(Ab, ksA, remA, t), rb = rsrc_at(Bb, ksB, remB, t);
      char* dst = smem + t * G4T_STAGE + wave * G4T_SUB;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(dst + j * G4T_PIECE), 16, voA[j], 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(dst + G4T_OPB + j * G4T_PIECE), 16, voB[j], 0, 0, 0);
      }
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
    // this lane's piece of a transposing read: k-row 8 g + (li >> 2) (+4: second half, +32: k-step 1 — immediates), 8 bytes at (li & 3) * 8
    const uint32_t lbase = (uint32_t)(((li >> 2) + 4 * (g & 1)) * G4T_PIECE + (g >> 1) * 256 + (li & 3) * 8);
    uint32_t g4_ra1 = lds0 + (2 * wr) * G4T_SUB + lbase, g4_rb1 = lds0 + G4T_OPB + (2 * wc) * G4T_SUB + lbase;      // stage 0 (K tile 0)
    uint32_t g4_ra0 = g4_ra1 + G4T_STAGE, g4_rb0 = g4_rb1 + G4T_STAGE;                                               // stage 1 (K tile 1)
    const uint32_t g4_sa = g4_ra0 + g4_ra1, g4_sb = g4_rb0 + g4_rb1;
    uint32_t g4_dma = __builtin_amdgcn_readfirstlane(lds0 + wave * G4T_SUB);                                         // K tile 2 -> stage 0
    const uint32_t g4_dsum = 2u * g4_dma + G4T_STAGE;
    uint32_t g4_nk = __builtin_amdgcn_readfirstlane(nkt);
    const uint32_t g4_ksA = __builtin_amdgcn_readfirstlane(ksA), g4_ksB = __builtin_amdgcn_readfirstlane(ksB);
    const unsigned long long pa = (unsigned long long)Ab + 2ull * ksA, pb = (unsigned long long)Bb + 2ull * ksB;
    const uint32_t g4_dA[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pa), (uint32_t)__builtin_amdgcn_readfirstlane((int)((pa >> 32) & 0xffffu)),
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)(remA > 2ull * ksA ? remA - 2u * ksA : 0u)), 0x00020000u};
    const uint32_t g4_dB[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pb), (uint32_t)__builtin_amdgcn_readfirstlane((int)((pb >> 32) & 0xffffu)),
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)(remB > 2ull * ksB ? remB - 2u * ksB : 0u)), 0x00020000u};
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");     // K tile 0 landed (tile 1's 16 pieces may still fly)
    asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
#if G4T_OB
    G4T_ASM_LOOP_OB();
#else
    G4T_ASM_LOOP();
#endif
    // the asm MFMAs are invisible to the hazard recognizer: let the last accumulator writes retire before reading them
    asm volatile("s_waitcnt vmcnt(0)\n s_nop 15\n s_nop 15" ::: "memory");
  }

  auto rowof = [&](int mt) { return row0 + wr * 128 + mt * 16 + li; };
  auto colof = [&](int gq, int x) { return col0 + wc * 128 + gq * 64 + x * 16 + g * 4; };
  if constexpr (ATT) {
    // dQ rows at_q0 + rowof(mt) of head at_h: the 128 columns belong to the wc = 0 waves (the B image's upper half is out of range:
    // zeros).  A lane holds features d = 16 x + 4 g .. + 3 (acc[mt][x]) and d + 64 (acc[mt][4 + x]) of its row: both halves of every
    // rotate_half pair, so the rotary embedding's gradient map is in-lane — the arithmetic of attn_bwd2.hip's store_pair_rope / rope_kernel's
    // backward (rowops.hip): g = bf16(acc * scale); dx1 = g1 cos1 + g2 sin2, dx2 = g2 cos2 - g1 sin1, every product rounded to bf16.
    if (wc != 0) return;
    bf16_t* Cb = (bf16_t*)p.C;
    const float mul = p.at_scale;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const long long tok = (long long)at_b * p.at_S + at_q0 + rowof(mt);
      bf16_t* dst = Cb + tok * p.ldc + at_h * 128 + g * 4;
      if (p.rope_pos) {
        const long long ro = (long long)p.rope_pos[tok] * 128 + g * 4;
        const bf16_t* cp = p.rope_cos + ro; const bf16_t* sp = p.rope_sin + ro;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const u32x2 c1 = *(const u32x2*)(cp + x * 16), c2 = *(const u32x2*)(cp + 64 + x * 16);
          const u32x2 s1 = *(const u32x2*)(sp + x * 16), s2 = *(const u32x2*)(sp + 64 + x * 16);
          const f32x4 v1 = acc[mt][x], v2 = acc[mt][4 + x];
          u32x2 o1, o2;
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const float a0 = bfround(v1[2 * k] * mul), a1 = bfround(v1[2 * k + 1] * mul);
            const float b0 = bfround(v2[2 * k] * mul), b1 = bfround(v2[2 * k + 1] * mul);
            o1[k] = pack2bf(bfround(a0 * bflo(c1[k])) + bfround(b0 * bflo(s2[k])), bfround(a1 * bfhi(c1[k])) + bfround(b1 * bfhi(s2[k])));
            o2[k] = pack2bf(bfround(b0 * bflo(c2[k])) - bfround(a0 * bflo(s1[k])), bfround(b1 * bfhi(c2[k])) - bfround(a1 * bfhi(s1[k])));
          }
          *(u32x2*)(dst + x * 16) = o1;
          *(u32x2*)(dst + 64 + x * 16) = o2;
        }
      } else {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const f32x4 v1 = acc[mt][x], v2 = acc[mt][4 + x];
          *(u32x2*)(dst + x * 16) = (u32x2){pack2bf(v1[0] * mul, v1[1] * mul), pack2bf(v1[2] * mul, v1[3] * mul)};
          *(u32x2*)(dst + 64 + x * 16) = (u32x2){pack2bf(v2[0] * mul, v2[1] * mul), pack2bf(v2[2] * mul, v2[3] * mul)};
        }
      }
    }
  } else if constexpr (!SPLIT) {
    // fp32 C += acc: four batches of 4 pieces (16 rows x 64 columns each), every batch's loads issued ahead of the previous one's stores
    float* Cb = (float*)p.C + (long long)bz * p.sC;
    f32x4 R[4][4][4];
    auto ld = [&](const int b) {
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) {
        const int gq = pc >> 1, mt = b * 2 + (pc & 1);
        const float* cp = Cb + (long long)min(rowof(mt), p.M - 1) * p.ldc;
#pragma unroll
        for (int x = 0; x < 4; ++x) R[b][pc][x] = *(const f32x4*)(cp + min(colof(gq, x), p.N - 4));
      }
    };
    auto add = [&](const int b) {
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) {
        const int gq = pc >> 1, mt = b * 2 + (pc & 1);
#pragma unroll
        for (int x = 0; x < 4; ++x) R[b][pc][x] += acc[mt][gq * 4 + x];
      }
    };
    auto pin = [&](const int b) {
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) asm volatile("" : "+v"(R[b][pc][0]), "+v"(R[b][pc][1]), "+v"(R[b][pc][2]), "+v"(R[b][pc][3]) : : "memory");
    };
    auto st = [&](const int b) {
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) {
        const int gq = pc >> 1, mt = b * 2 + (pc & 1), row = rowof(mt);
        if (row >= p.M) continue;
        float* cp = Cb + (long long)row * p.ldc;
#pragma unroll
        for (int x = 0; x < 4; ++x) if (colof(gq, x) + 4 <= p.N) *(f32x4*)(cp + colof(gq, x)) = R[b][pc][x];
      }
    };
    ld(0); add(0); pin(0); ld(1); st(0); add(1); pin(1); ld(2); st(1); add(2); pin(2); ld(3); st(2); add(3); st(3);
  } else {
    // split-K: partial tile -> workspace, lane-linear, agent-scope stores; the last split to arrive adds all partials in split order
#pragma unroll
    for (int mg = 0; mg < 16; ++mg) {
      const int mt = mg >> 1, gq = mg & 1;
      float* wp = p.ws + (((long long)split * tpb + id) * 64 + (mt * 2 + gq) * 4) * 1024 + tid * 4;
#pragma unroll
      for (int x = 0; x < 4; ++x)
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(wp + x * 1024), "v"(acc[mt][gq * 4 + x]) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = (int*)smem;
    if (tid == 0)
      *flag = (__hip_atomic_fetch_add(p.counters + id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p.splitk - 1);
    __syncthreads();
    if (!*flag) return;
    float* Cf = (float*)p.C;
    for (int mg = 0; mg < 16; ++mg) {               // (mt, gq) pairs
      const int mt = mg >> 1, gq = mg & 1;
      const int row = rowof(mt);
      f32x4 o[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const bool live = (row < p.M) && (colof(gq, x) + 4 <= p.N);
        o[x] = live ? *(f32x4*)(Cf + (long long)row * p.ldc + colof(gq, x)) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      for (int s0 = 0; s0 < p.splitk; s0 += 4) {
        f32x4 part[4][4];
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
          const float* wp = p.ws + (((long long)min(s0 + ds, p.splitk - 1) * tpb + id) * 64 + mg * 4) * 1024 + tid * 4;
#pragma unroll
          for (int x = 0; x < 4; ++x)
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(part[ds][x]) : "v"(wp + x * 1024) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int ds = 0; ds < 4; ++ds)
          if (s0 + ds < p.splitk) {
#pragma unroll
            for (int x = 0; x < 4; ++x) {
              asm volatile("" : "+v"(part[ds][x]));
              o[x] += part[ds][x];
            }
          }
      }
#pragma unroll
      for (int x = 0; x < 4; ++x)
        if ((row < p.M) && (colof(gq, x) + 4 <= p.N)) *(f32x4*)(Cf + (long long)row * p.ldc + colof(gq, x)) = o[x];
    }
    if (tid == 0) __hip_atomic_store(p.counters + id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 transpose  out[C, ld_out] = in[R, C]^T  (batched).  Each thread transposes an 8x8 block in
// registers: 8 x 16-byte loads (row-contiguous), 8 x 16-byte stores; lanes are laid out 8 x 8 so
// both sides move whole 128-byte lines.  Columns R..ld_out-1 of `out` are written as zeros so the
// result can be used directly as a K-contiguous GEMM operand with K = roundup(R, 8).
// Requires C % 8 == 0, ld_in % 8 == 0, ld_out % 8 == 0, ld_out >= roundup(R, 8).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void transpose8x8(const u32x4 (&in)[8], u32x4 (&out)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t a = in[2 * w][j >> 1], b = in[2 * w + 1][j >> 1];
      o[w] = (j & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
    }
    out[j] = o;
  }
}

__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                            int R, int C, int ld_in, int ld_out,
                                                            long long s_in, long long s_out, int tiles_c,
                                                            const int* __restrict__ r_valid) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bz = blockIdx.y;
  const int tile = blockIdx.x * 4 + wave;               // one 64x64 tile per wave
  const int tr = tile / tiles_c, tc = tile - tr * tiles_c;
  const int r0 = tr * 64 + (lane >> 3) * 8, c0 = tc * 64 + (lane & 7) * 8;
  if (c0 >= C || r0 >= ld_out) return;
  if (r_valid) {       // grouped (MoE capacity slab) use: rows past the live count are never read by the k_valid GEMM —
    R = min(R, r_valid[bz]);                            // the columns up to the next multiple of 64 are zero-filled, so that a
    if (r0 >= ((R + 63) & ~63)) return;                 // caller may round k_valid up to 64 (whole K tiles: the asm K loop)
  }
  const bf16_t* ip = in + (long long)bz * s_in;
  bf16_t* op = out + (long long)bz * s_out;
  u32x4 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (r0 + i < R) a[i] = *(const u32x4*)(ip + (long long)(r0 + i) * ld_in + c0);
    else a[i] = (u32x4){0u, 0u, 0u, 0u};
  }
  transpose8x8(a, b);
#pragma unroll
  for (int j = 0; j < 8; ++j) *(u32x4*)(op + (long long)(c0 + j) * ld_out + r0) = b[j];
}

// Which 256x256 kernel runs a launch.  The 8-wave kernel is the default everywhere: 1390-1410 TF at the teacher QKV shape
// against the 4-wave kernel's ~1270, and the fused SwiGLU backward is 7-13 % faster as its own 8-wave instantiation
// (MODE 4) than on the 4-wave kernel.  That epilogue must NOT sit inside gemm_256_kernel<0>: there it cost the plain GEMM
// 22 % (1390 -> 1075 TF, found by timing library builds of three commits in one process, tools/gemm_ab.py).
// LMOD_GEMM_WAVES=4 runs the 4-wave kernel everywhere (A/B runs).
// Process-wide state of the launchers (all of it, see include/lmod_hip.h "State"): this routing switch, read ONCE (C++11 static
// initialisation: thread-safe); the per-kernel-instance "LDS attribute set" flags (idempotent: a race only repeats the call); the CU
// count per device id; and the A/B environment switches (GemmRouting below), read once unless LMOD_GEMM_ENV_DYNAMIC=1.
static int gemm_waves() {
  static const int w = [] {
    const char* e = getenv("LMOD_GEMM_WAVES");
    const int v = e ? atoi(e) : 0;
    return (v == 8 || v == 4 || v == 44) ? v : 0;       // 44: default routing + the 4-wave MODE 4 / 6 instantiations (A/B)
  }();
  return w;
}
template <typename KT>
static void allow_lds(KT kern, int bytes, bool& done) {      // once per kernel instance, not once per launch
  if (!__atomic_load_n(&done, __ATOMIC_ACQUIRE)) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    __atomic_store_n(&done, true, __ATOMIC_RELEASE);
  }
}
// one launch of the 8-wave kernel: the K64 instantiation when every reduction length of the launch is a multiple of 64
static inline bool k64_ok(const GemmP& p) { return (p.K & 63) == 0 && !p.k_valid; }      // (split-K chunks are multiples of 64)
template <int MODE>
static void launch_256x(const GemmP& p, long long nwg, hipStream_t stream) {
  static bool a0 = false, a1 = false;
  if (k64_ok(p)) {
    allow_lds(gemm_256_kernel<MODE, true>, 8 * G256_SLOT, a1);
    hipLaunchKernelGGL((gemm_256_kernel<MODE, true>), dim3((unsigned)nwg), dim3(512), 8 * G256_SLOT, stream, p);
  } else {
    allow_lds(gemm_256_kernel<MODE, false>, 8 * G256_SLOT, a0);
    hipLaunchKernelGGL((gemm_256_kernel<MODE, false>), dim3((unsigned)nwg), dim3(512), 8 * G256_SLOT, stream, p);
  }
}
// Persistent form of the 4-wave kernel (one workgroup per CU walks the tiles, the next tile's operands in flight under the epilogue):
// plain launches with more tiles than CUs.  LMOD_GEMM_PERSIST=0 / 1 overrides the build's default (A/B runs).
#ifndef G4_PERSIST_DEFAULT
#define G4_PERSIST_DEFAULT 1
#endif
// Compute units of the CURRENT device (the persistent grid is one workgroup per CU), cached per device id: a process that drives
// several devices, or a partitioned one, sizes each device's walk from that device's own count.  (A CU-masked stream still gets
// the full count: the tile loop keeps the results correct, the walk is merely longer per resident workgroup.)
static int gemm_cus() {
  static int cus[64];                                  // 0 = not asked yet; written once per device id (idempotent)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  int n = __atomic_load_n(&cus[dev], __ATOMIC_RELAXED);
  if (!n) {
    hipDeviceProp_t pr;
    n = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    __atomic_store_n(&cus[dev], n, __ATOMIC_RELAXED);
  }
  return n;
}
// The routing switches of this file (A/B arms of measured decisions; defaults = the measured winners) are read from the environment
// ONCE, at the first launch (C++11 static initialisation: thread-safe) — a product process pays no getenv per launch.  A process that
// wants to run both arms of a switch (tests/test_kernels_gpu.py, tools/bench_r5_routing.py, tools/bench_wgrad.py) sets
// LMOD_GEMM_ENV_DYNAMIC=1 before its first launch: the switches are then re-read on every launch.
struct GemmRouting {
  int persist;          // LMOD_GEMM_PERSIST: plain launches with more tiles than CUs on the persistent walk (default: the build's G4_PERSIST_DEFAULT)
  int min_rounds;       // LMOD_GEMM_PERSIST_ROUNDS: from this many rounds of the CUs up (4)
  int grouped;          // LMOD_GEMM_PERSIST_GROUPED: 0 none, 1 fused SwiGLU forward (default), 2 plain grouped launches too
  int sb4;              // LMOD_GEMM_SB4: dense fused SwiGLU backward on the persistent 4-wave kernel (1)
  int sb4g;             // LMOD_GEMM_SB4G: GROUPED (MoE) fused SwiGLU backward on the persistent grouped walk of the 4-wave kernel (round 6; 1)
  int kv4;              // LMOD_GEMM_KV4: k_valid batches (MoE expert weight gradients) on the 4-wave kernel (1)
  int tn4;              // LMOD_GEMM_TN4: fp32-accumulate TN launches on gemm4t_kernel (1)
};
static GemmRouting read_gemm_routing() {
  auto geti = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
  GemmRouting r;
  r.persist = geti("LMOD_GEMM_PERSIST", G4_PERSIST_DEFAULT) != 0;
  r.min_rounds = geti("LMOD_GEMM_PERSIST_ROUNDS", 4);
  r.grouped = geti("LMOD_GEMM_PERSIST_GROUPED", 1);
  r.sb4 = geti("LMOD_GEMM_SB4", 1) != 0;
  r.sb4g = geti("LMOD_GEMM_SB4G", 1) != 0;
  r.kv4 = geti("LMOD_GEMM_KV4", 1) != 0;
  r.tn4 = geti("LMOD_GEMM_TN4", 1) != 0;
  return r;
}
static GemmRouting gemm_routing() {
  static const bool dynamic = [] { const char* e = getenv("LMOD_GEMM_ENV_DYNAMIC"); return e && e[0] == '1'; }();
  static const GemmRouting once = read_gemm_routing();
  return dynamic ? read_gemm_routing() : once;
}
static bool gemm_persist(const GemmP& p, long long nwg) {
  // measured (profiles/r04_gemm_loop.md): +1.5 % at 24 rounds of the CUs (teacher QKV), +4.8 % at 12 rounds with K 2048.  The first
  // threshold was 10 rounds (4 - 8 rounds measured level on the build of that day); on the final build (split-column stores) the
  // persistent form is ahead from 4 rounds up — +14 % at [32768 x 2048 x 2048] (4 rounds, 32 K tiles per output tile: the
  // prologue / epilogue share is largest there), +3.9 % at 8 rounds with K 4096, +1 % elsewhere — and in the step 10 -> 4 is
  // +0.8 ... 1.0 % on two boxes (24.18 -> 24.42, 25.08 -> 25.30 samples/s); 3 and 2 are level with 4.
  const GemmRouting r = gemm_routing();
  return r.persist && G4_ASM && !p.m_valid && !p.k_valid && p.splitk <= 1 && nwg > gemm_cus() && nwg >= (long long)r.min_rounds * gemm_cus() &&
         (p.K & 63) == 0 && p.K >= 256;
}
// The persistent form of a GROUPED launch (round 5; MoE capacity slabs: m_valid live rows per batch): `nwg` counts the slabs' tiles,
// dead ones included (capacity factor 1.5: about a third) — the live count is only known on the device, where the kernel computes
// it.  Measured (profiles/r05_grouped_persistent.jsonl, config-2 MoE shapes, same box, bit-identical): the grouped fused SwiGLU
// forward +3.1 ... 4.4 % (1229 -> 1268 TF; 128-column tiles, 32 K tiles per tile: the largest epilogue share), the plain grouped
// launches level (down projection +4.5 / -1.0 %, gate/up dgrad -0.6 %: 256-column tiles, K 5504 / 11008).  So the default (1) routes
// the fused SwiGLU forward only; LMOD_GEMM_PERSIST_GROUPED=2 adds the plain grouped launches, 0 keeps one tile per workgroup.
static bool gemm_persist_grouped(const GemmP& p, long long nwg, const int mode) {
  const GemmRouting r = gemm_routing();
  const int on = r.persist && (mode == 1 ? r.grouped >= 1 : mode == 4 ? r.sb4g : r.grouped >= 2);
  return on && G4_ASM && p.m_valid && !p.k_valid && p.splitk <= 1 && p.batch >= 1 && p.batch <= GEMM_MAX_GROUPS &&
         nwg >= (long long)r.min_rounds * gemm_cus() && (p.K & 63) == 0 && p.K >= 256;
}
template <int MODE>
static void launch_4(const GemmP& p0, long long nwg, hipStream_t stream) {
  static bool a = false, ap = false, apg = false;
  if constexpr (MODE == 1 || MODE == 7 || MODE == 4) {
    if (gemm_persist_grouped(p0, nwg, MODE)) {
      allow_lds(gemm4_kernel<MODE, true, true>, 2 * G4_STAGE, apg);
      hipLaunchKernelGGL((gemm4_kernel<MODE, true, true>), dim3((unsigned)gemm_cus()), dim3(256), 2 * G4_STAGE, stream, p0);
      return;
    }
  }
  if (gemm_persist(p0, nwg)) {
    GemmP p = p0;
    p.ptotal = (int)nwg;
    allow_lds(gemm4_kernel<MODE, true>, 2 * G4_STAGE, ap);
    hipLaunchKernelGGL((gemm4_kernel<MODE, true>), dim3((unsigned)gemm_cus()), dim3(256), 2 * G4_STAGE, stream, p);
  } else {
    allow_lds(gemm4_kernel<MODE, false>, 2 * G4_STAGE, a);
    hipLaunchKernelGGL((gemm4_kernel<MODE, false>), dim3((unsigned)nwg), dim3(256), 2 * G4_STAGE, stream, p0);
  }
}
static bool gemm_sb4() { return gemm_routing().sb4; }      // (A/B: 0 keeps the dense fused SwiGLU backward on the 8-wave kernel)
static bool gemm_kv4() { return gemm_routing().kv4; }      // (A/B: 0 keeps k_valid batches on the 8-wave kernel)
template <int MODE>
static void launch_256(const GemmP& p, long long nwg, hipStream_t stream) {
  static bool a4 = false, a44 = false, a46 = false, a47 = false;
  const int w = gemm_waves();
  // the 4-wave kernel (128x128 per wave, one wave per SIMD, the vendor kernel's shape and loop structure) is ~2 % ahead of the 8-wave
  // one at K >= 4096 and level or slightly behind at K 2048, depending on N (profiles/r03_vendor_ab.md; in the step: -0.5 % / -0.7 % of
  // the plain / fused-SwiGLU kernel time): the plain bf16 store and the fused SwiGLU forward take it by default; split-K, k_valid
  // batches (their XCD balancing lives in the 8-wave kernel) and the rare option mixes stay on 8 waves
  if ((w == 0 || w == 44) && MODE == 1) {
    launch_4<1>(p, nwg, stream);
  } else if (w == 4) {
    allow_lds(gemm4_kernel<MODE>, 2 * G4_STAGE, a4);
    hipLaunchKernelGGL(gemm4_kernel<MODE>, dim3((unsigned)nwg), dim3(256), 2 * G4_STAGE, stream, p);
  } else if (MODE == 0 && p.splitk <= 1 && !p.k_valid && p.act == 3 &&
             (w == 44 || (w == 0 && gemm_sb4() && gemm_persist(p, nwg)) || (w == 0 && gemm_persist_grouped(p, nwg, 4)))) {
    // (round 6) GROUPED launches (MoE experts: m_valid live rows per slab) take the persistent grouped walk of the same instantiation:
    // live tiles only, the next tile's operands in flight under this tile's [gate | up] loads and [dgate | dup] stores (LMOD_GEMM_SB4G=0:
    // the 8-wave instantiation, one tile per workgroup)
    // fused SwiGLU backward: launches that take the PERSISTENT 4-wave form (dense layers: no m_valid, >= 4 rounds of the CUs) run
    // 2.8 - 3.8 % faster there, bit-identical (round 5 routing; tools/probe/swiglu_bwd_variants.py; LMOD_GEMM_SB4=0 is the A/B arm);
    // grouped (MoE) launches keep the 8-wave instantiation with its two-batch epilogue (one tile per workgroup: 1124 vs 1177 us)
    (void)a44;
    launch_4<4>(p, nwg, stream);
  } else if ((w == 0 || w == 44) && MODE == 0 && p.splitk <= 1 && !p.k_valid && !p.out_f32 && !p.accumulate && p.act != 3) {
    launch_4<7>(p, nwg, stream);
  } else if (w == 44 && MODE == 0 && p.splitk <= 1 && !p.k_valid && p.out_f32 && p.accumulate && !p.act && !p.bias && p.vec_ok) {   // (not routed: the 8-wave MODE 6 has the two-batch read-modify-write)
    allow_lds(gemm4_kernel<6>, 2 * G4_STAGE, a46);
    hipLaunchKernelGGL(gemm4_kernel<6>, dim3((unsigned)nwg), dim3(256), 2 * G4_STAGE, stream, p);
  } else if (MODE == 0 && p.k_valid && !p.m_valid && p.out_f32 && p.accumulate && !p.act && !p.bias && p.vec_ok && p.splitk <= 1 &&
             (p.K & 63) == 0 && (p.N & 3) == 0 && gemm_kv4()) {
    // MoE expert weight gradients (a live reduction length per expert): the 4-wave kernel, whose K loop is the asm statement for
    // every expert whose k_valid is a multiple of 64 (the MoE backward rounds it up; the transposes zero-fill to that boundary)
    allow_lds(gemm4_kernel<6>, 2 * G4_STAGE, a46);
    hipLaunchKernelGGL(gemm4_kernel<6>, dim3((unsigned)nwg), dim3(256), 2 * G4_STAGE, stream, p);
  } else if (MODE == 0 && p.act == 3) {       // the SwiGLU-backward epilogue is its own 8-wave instantiation
    launch_256x<4>(p, nwg, stream);
  } else if (MODE == 0 && p.out_f32 && p.accumulate && p.splitk <= 1 && p.vec_ok) {   // fp32 read-modify-write: see MODE 6
    launch_256x<6>(p, nwg, stream);
  } else {
    launch_256x<MODE>(p, nwg, stream);
  }
}

extern "C" {

int lmod_gemm_bf16_nt(const void* A, const void* B, void* C, const void* bias,
                      int M, int N, int K, int lda, int ldb, int ldc,
                      int batch, long long strideA, long long strideB, long long strideC,
                      const int* m_valid, const int* k_valid,
                      int act, int out_f32, int accumulate, hipStream_t stream) {
  if (M < 0 || N < 0 || K < 0 || batch < 0) return LMOD_EINVAL;
  if (M == 0 || N == 0 || batch == 0) return LMOD_OK;          // an empty problem dereferences nothing
  if (!A || !B || !C) return LMOD_EINVAL;
  if ((K & 7) || (lda & 7) || (ldb & 7) || lda < K || ldb < K) return LMOD_EINVAL;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return LMOD_EINVAL;
  if (act < 0 || act > 2 || ldc < N) return LMOD_EINVAL;
  if ((long long)127 * lda * 2 + (long long)K * 2 >= 0x7fffffffLL) return LMOD_EUNSUPPORTED;
  if ((long long)127 * ldb * 2 + (long long)K * 2 >= 0x7fffffffLL) return LMOD_EUNSUPPORTED;
  GemmP p;
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.bias = (const bf16_t*)bias;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.batch = batch; p.sA = strideA; p.sB = strideB; p.sC = strideC;
  p.m_valid = m_valid; p.k_valid = k_valid;
  p.act = act; p.out_f32 = out_f32; p.accumulate = accumulate;
  p.C2 = nullptr; p.ldc2 = 0; p.sC2 = 0; p.splitk = 1; p.kchunk = 0; p.ws = nullptr; p.counters = nullptr;
  const int esz = out_f32 ? 4 : 2;
  p.vec_ok = (((uintptr_t)C & 15) == 0) && ((((long long)ldc * esz) & 15) == 0) &&
             (((strideC * esz) & 15) == 0);
  static int force_tile = -1;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_128, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)gemm_256_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * G256_SLOT);
    const char* e = getenv("LMOD_GEMM_TILE");
    force_tile = e ? atoi(e) : 0;
    attr_set = true;
  }
  // 256-tiles pay when the grid still fills the 256 CUs and the padded tile area is not wasteful
  const long long t256 = (long long)((M + 255) / 256) * ((N + 255) / 256) * batch;
  bool big = (M >= 512 && N >= 256 && (t256 >= 160 || (t256 >= 96 && K >= 8192))) &&
             ((long long)((M + 255) / 256 * 256) * ((N + 255) / 256 * 256) <= (long long)M * N * 115 / 100 + 65536);
  if (force_tile == 128) big = false;
  if (force_tile == 256) big = true;
  if (big) {
    if ((long long)255 * lda * 2 + (long long)K * 2 >= 0x7fffffffLL || (long long)255 * ldb * 2 + (long long)K * 2 >= 0x7fffffffLL)
      big = false;
  }
  const int T = big ? 256 : 128;
  p.tiles_m = (M + T - 1) / T; p.tiles_n = (N + T - 1) / T;
  const long long nwg = (long long)p.tiles_m * p.tiles_n * batch;
  if (nwg > 0x7fffffffLL) return LMOD_EUNSUPPORTED;
  if (big) launch_256<0>(p, nwg, stream);
  else hipLaunchKernelGGL(gemm_nt_128, dim3((unsigned)nwg), dim3(256), 65536, stream, p);
  return lmod_launch_status();
}

// C[M, N] = bf16(res + bf16(A W^T)): a projection (no bias) whose output is added to the residual stream (the decoder layer's
// hidden = residual + o_proj(attn) / + down_proj(mlp), qwen2/modeling_qwen2.py:757-775) with the add in the GEMM epilogue — what
// lmod_gemm_bf16_nt followed by the residual path of lmod_rmsnorm_fwd computes, bit for bit, one pass over [M, N] less on each side.
// Only on the 4-wave 256-tile kernel: shapes that lmod_gemm_bf16_nt would send elsewhere return LMOD_EUNSUPPORTED (the caller keeps
// the two-step form; kernels.gemm_res_fusable mirrors the test).  N % 8 == 0, C / res rows 16-byte aligned, C != res.
static bool gemm_res_shape_ok(int M, int N, int K, int lda, int ldb) {
  const long long t256 = (long long)((M + 255) / 256) * ((N + 255) / 256);
  const bool big = (M >= 512 && N >= 256 && (t256 >= 160 || (t256 >= 96 && K >= 8192))) &&
                   ((long long)((M + 255) / 256 * 256) * ((N + 255) / 256 * 256) <= (long long)M * N * 115 / 100 + 65536);
  return big && (long long)255 * lda * 2 + (long long)K * 2 < 0x7fffffffLL && (long long)255 * ldb * 2 + (long long)K * 2 < 0x7fffffffLL &&
         gemm_waves() == 0 && G4_ASM;
}
int lmod_gemm_bf16_nt_res(const void* A, const void* W, void* C, const void* bias, const void* res, int M, int N, int K, int lda,
                          int ldw, int ldc, int ldr, hipStream_t stream) {
  if (M < 0 || N < 0 || K < 0) return LMOD_EINVAL;
  if (M == 0 || N == 0) return LMOD_OK;
  if (!A || !W || !C || !res || C == res || bias) return LMOD_EINVAL;        // (no bias on this path: the decoder's o / down projections have none)
  if ((K & 7) || (lda & 7) || (ldw & 7) || lda < K || ldw < K || ldc < N || ldr < N || (ldc & 7) || (ldr & 7) || (N & 7)) return LMOD_EINVAL;
  if (((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)C & 15) || ((uintptr_t)res & 15)) return LMOD_EINVAL;
  if (!gemm_res_shape_ok(M, N, K, lda, ldw)) return LMOD_EUNSUPPORTED;
  GemmP p;
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)W; p.C = C; p.bias = (const bf16_t*)bias;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldw; p.ldc = ldc;
  p.batch = 1; p.sA = 0; p.sB = 0; p.sC = 0; p.m_valid = nullptr; p.k_valid = nullptr;
  p.act = 0; p.out_f32 = 0; p.accumulate = 0; p.vec_ok = 1;
  p.C2 = const_cast<void*>(res); p.ldc2 = ldr; p.sC2 = 0; p.splitk = 1; p.kchunk = 0; p.ws = nullptr; p.counters = nullptr;
  p.tiles_m = (M + 255) / 256; p.tiles_n = (N + 255) / 256;
  launch_4<8>(p, (long long)p.tiles_m * p.tiles_n, stream);
  return lmod_launch_status();
}

// C[M, N] = rope(A W^T + bias): the decoder's fused q/k/v projection with the rotary embedding applied in the GEMM epilogue
// (qwen2/modeling_qwen2.py:262-264 + apply_rotary_pos_emb :146-171).  Heads are 128 wide; columns [0, rope_cols) are the q and
// k heads (rotated with cos/sin [max_pos, 128] bf16 rows pos[row]), the rest (v) is stored as computed.  Bit-identical to
// lmod_gemm_bf16_nt followed by lmod_rope.  rope_cols % 256 == 0, N % 16 == 0, C 16-byte aligned with ldc % 8 == 0.
int lmod_gemm_qkv_rope_bf16(const void* A, const void* W, void* C, const void* bias, int M, int N, int K, int lda, int ldw,
                            int ldc, const void* cos_t, const void* sin_t, const int* pos, int rope_cols, hipStream_t stream) {
  if (M < 0 || N < 0 || K < 0) return LMOD_EINVAL;
  if (M == 0 || N == 0) return LMOD_OK;
  if (!A || !W || !C || !cos_t || !sin_t || !pos) return LMOD_EINVAL;
  if ((K & 7) || (lda & 7) || (ldw & 7) || lda < K || ldw < K || ldc < N || (ldc & 7) || (N & 15)) return LMOD_EINVAL;
  if (((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)C & 15)) return LMOD_EINVAL;
  if (rope_cols < 0 || rope_cols > N || (rope_cols & 255)) return LMOD_EINVAL;
  if ((long long)255 * lda * 2 + (long long)K * 2 >= 0x7fffffffLL || (long long)255 * ldw * 2 + (long long)K * 2 >= 0x7fffffffLL)
    return LMOD_EUNSUPPORTED;
  GemmP p;
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)W; p.C = C; p.bias = (const bf16_t*)bias;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldw; p.ldc = ldc;
  p.batch = 1; p.sA = 0; p.sB = 0; p.sC = 0; p.m_valid = nullptr; p.k_valid = nullptr;
  p.act = 0; p.out_f32 = 0; p.accumulate = 0; p.vec_ok = 1;
  p.C2 = nullptr; p.ldc2 = 0; p.sC2 = 0; p.splitk = 1; p.kchunk = 0; p.ws = nullptr; p.counters = nullptr;
  p.rope_cos = (const bf16_t*)cos_t; p.rope_sin = (const bf16_t*)sin_t; p.rope_pos = pos; p.rope_cols = rope_cols;
  p.tiles_m = (M + 255) / 256; p.tiles_n = (N + 255) / 256;
  const long long nwg = (long long)p.tiles_m * p.tiles_n;
  if (nwg > 0x7fffffffLL) return LMOD_EUNSUPPORTED;
  // 4-wave kernel by default: a wave column is one 128-feature head, the rotate_half partner sits in the same lane (no LDS exchange)
  static bool a45 = false;
  const int w = gemm_waves();
  if ((w == 0 || w == 4 || w == 44) && (N & 127) == 0) {   // +2 % (student shape) ... +3 % (teacher shape) over the 8-wave kernel's LDS-exchange epilogue, bit-identical
    (void)a45;
    launch_4<5>(p, nwg, stream);                    // persistent from 10 rounds of the CUs up
  } else {
    launch_256x<5>(p, nwg, stream);
  }
  return lmod_launch_status();
}

// act_out[b] (M x N) = silu(A Wg^T) * (A Wu^T) with W = [Wg; Wu] the [2N, K] gate-over-up weight (row stride ldw);
// gu_out (optional, M x 2N, ld_gu) receives the bf16 pre-activations [gate | up] for the backward.
// Grouped use as lmod_gemm_bf16_nt (m_valid: live rows per batch; rows up to the next multiple of 8 are zeroed in act_out).
int lmod_gemm_swiglu_bf16(const void* A, const void* W, void* act_out, void* gu_out, int M, int N, int K, int lda,
                          int ldw, int ld_act, int ld_gu, int batch, long long strideA, long long strideW,
                          long long stride_act, long long stride_gu, const int* m_valid, hipStream_t stream) {
  if (!A || !W || !act_out || M < 0 || N < 0 || K < 0 || batch < 0) return LMOD_EINVAL;
  if (M == 0 || N == 0 || batch == 0) return LMOD_OK;
  if ((K & 7) || (N & 7) || (lda & 7) || (ldw & 7) || lda < K || ldw < K || ld_act < N || (ld_act & 7)) return LMOD_EINVAL;
  if (((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)act_out & 15) || (stride_act & 7)) return LMOD_EINVAL;
  if (gu_out && (ld_gu < 2 * N || (ld_gu & 7) || ((uintptr_t)gu_out & 15) || (stride_gu & 7))) return LMOD_EINVAL;
  if ((long long)255 * lda * 2 + (long long)K * 2 >= 0x7fffffffLL) return LMOD_EUNSUPPORTED;
  if (((long long)N + 127) * ldw * 2 + (long long)K * 2 >= 0x7fffffffLL) return LMOD_EUNSUPPORTED;
  GemmP p;
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)W; p.C = act_out; p.bias = nullptr;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldw; p.ldc = ld_act;
  p.batch = batch; p.sA = strideA; p.sB = strideW; p.sC = stride_act;
  p.m_valid = m_valid; p.k_valid = nullptr;
  p.act = 0; p.out_f32 = 0; p.accumulate = 0; p.vec_ok = 1;
  p.C2 = gu_out; p.ldc2 = ld_gu; p.sC2 = stride_gu; p.splitk = 1; p.kchunk = 0; p.ws = nullptr; p.counters = nullptr;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_256_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * G256_SLOT);
    attr_set = true;
  }
  p.tiles_m = (M + 255) / 256; p.tiles_n = (N + 127) / 128;
  const long long nwg = (long long)p.tiles_m * p.tiles_n * batch;
  if (nwg > 0x7fffffffLL) return LMOD_EUNSUPPORTED;
  launch_256<1>(p, nwg, stream);
  return lmod_launch_status();
}

// dgu[b] (M x 2N) = SwiGLU'(gu[b]) applied to dact = A[b] (M x K) * Bt[b]^T (N x K): the down-projection dgrad with the
// SwiGLU backward in its epilogue.  gu = [gate | up] pre-activations saved by lmod_gemm_swiglu_bf16, dgu = [dgate | dup]
// (may alias gu).  Grouped use / m_valid as lmod_gemm_bf16_nt; rows m_valid..roundup8(m_valid)-1 of dgu are zeroed.
int lmod_gemm_swiglu_bwd_bf16(const void* A, const void* Bt, const void* gu, void* dgu, int M, int N, int K, int lda,
                              int ldb, int ld_gu, int ld_dgu, int batch, long long strideA, long long strideB,
                              long long stride_gu, long long stride_dgu, const int* m_valid, hipStream_t stream) {
  if (M < 0 || N < 0 || K < 0 || batch < 0) return LMOD_EINVAL;
  if (M == 0 || N == 0 || batch == 0) return LMOD_OK;          // an empty problem dereferences nothing
  if (!A || !Bt || !gu || !dgu) return LMOD_EINVAL;
  if ((K & 7) || (N & 15) || (lda & 7) || (ldb & 7) || lda < K || ldb < K || ld_gu < 2 * N || ld_dgu < 2 * N ||
      (ld_gu & 7) || (ld_dgu & 7)) return LMOD_EINVAL;
  if (((uintptr_t)A & 15) || ((uintptr_t)Bt & 15) || ((uintptr_t)gu & 15) || ((uintptr_t)dgu & 15) || (stride_gu & 7) ||
      (stride_dgu & 7)) return LMOD_EINVAL;
  if ((long long)255 * lda * 2 + (long long)K * 2 >= 0x7fffffffLL || (long long)255 * ldb * 2 + (long long)K * 2 >= 0x7fffffffLL)
    return LMOD_EUNSUPPORTED;
  GemmP p;
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)Bt; p.C = dgu; p.bias = nullptr;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ld_dgu;
  p.batch = batch; p.sA = strideA; p.sB = strideB; p.sC = stride_dgu;
  p.m_valid = m_valid; p.k_valid = nullptr;
  p.act = 3; p.out_f32 = 0; p.accumulate = 0; p.vec_ok = 1;
  p.C2 = (void*)gu; p.ldc2 = ld_gu; p.sC2 = stride_gu; p.splitk = 1; p.kchunk = 0; p.ws = nullptr; p.counters = nullptr;
  p.tiles_m = (M + 255) / 256; p.tiles_n = (N + 255) / 256;
  const long long nwg = (long long)p.tiles_m * p.tiles_n * batch;
  if (nwg > 0x7fffffffLL) return LMOD_EUNSUPPORTED;
  launch_256<0>(p, nwg, stream);
  return lmod_launch_status();
}

// Weight-gradient accumulate C (fp32, M x N) += At (M x K) * X for long K (tokens); X is either Bt [N x K] (K-contiguous,
// b_kmajor 0) or the activation as autograd holds it, [K x N] (b_kmajor 1: no transposed copy of X).  Few output tiles:
// deterministic split-K.  `workspace` (16-byte aligned, zeroed ONCE by the caller, used by one stream at a time) holds
// 16 KiB of tile semaphores followed by up to 8 partial images (256 KiB per 256x256 tile); its size bounds the split.  With a NULL /
// small workspace, or when splitting does not pay, this is lmod_gemm_bf16_nt(out_f32, accumulate).
#define WGRAD_MAX_TILES 4096
static int wgrad_pick_split(int M, int N, int K, int max_s) {
  if (M < 256 || N < 256 || K < 4096 || (N & 15)) return 1;
  static int forced = -1;              // LMOD_WGRAD_SPLIT=s: force the split (tuning of the cost model below)
  if (forced < 0) { const char* e = getenv("LMOD_WGRAD_SPLIT"); forced = e ? atoi(e) : 0; }
  if (forced > 0) return forced < max_s ? forced : max_s;
  const long long t256 = (long long)((M + 255) / 256) * ((N + 255) / 256);
  if (t256 >= 512 || t256 > WGRAD_MAX_TILES) return 1;
  int best_s = 1;
  double best = 1e30;
  for (int s = 1; s <= max_s; ++s) {
    const int kc = ((K + s - 1) / s + 63) / 64 * 64;
    if (kc < 1024) break;
    const double rounds = (double)((t256 * s + 255) / 256);
    const double tg = rounds * 131072.0 * kc / 5300.0;                     // ns: one 256x256xkc tile at ~5.3 TF/CU
    const double ta = (s > 1 ? 2.0 * s + 2.0 : 0.0) * (double)M * N * 4.0 / 2000.0;   // ns: partial write+read, C RMW (measured ~2 TB/s)
    if (tg + ta < best * 0.97) { best = tg + ta; best_s = s; }
  }
  return best_s;
}

// b_kmajor 2: BOTH operands reduction-major (At is dY [K x M], row stride lda; B is X [K x N]) on gemm4t_kernel — no transposed
// copy of either.  An operand window must stay below 2 GiB (32-bit buffer offsets, out-of-range marker 0x80000000): longer
// reductions run as several launches over K chunks (the result accumulates either way).
static bool gemm_tn4() { return gemm_routing().tn4; }      // (A/B: 0 = the 8-wave TN kernel)
static void launch_4t(const GemmP& p, long long nwg, hipStream_t stream) {
  static bool a0 = false, a1 = false;
  if (p.splitk > 1) {
    allow_lds(gemm4t_kernel<1>, 2 * G4T_STAGE, a1);
    hipLaunchKernelGGL(gemm4t_kernel<1>, dim3((unsigned)nwg), dim3(256), 2 * G4T_STAGE, stream, p);
  } else {
    allow_lds(gemm4t_kernel<0>, 2 * G4T_STAGE, a0);
    hipLaunchKernelGGL(gemm4t_kernel<0>, dim3((unsigned)nwg), dim3(256), 2 * G4T_STAGE, stream, p);
  }
}
// Attention backward, dQ from the spilled dS^T (attn_bwd2.hip's dK/dV kernel wrote it): dQ[b, t, h, :] = scale * sum_o dS[t, o] K[b, o, h_kv, :]
// as ONE launch of gemm4t_kernel<2> — reduction-major operands exactly as stored (A = dS^T [key][query], B = K [key][feature]), one
// workgroup per (sample, head, 256-query block), causal blocks reduce over the keys up to their own end.  N = 128 fills half of the
// 256-column tile: the launch streams 2 S^2 B nh bytes of dS (x 1/2 causal) and is about as HBM- as MFMA-bound, so the idle half
// costs little; against the dQ kernel it replaces (which recomputes S, dP and the exponentials) it removes 2 of the backward's 7 matmuls.
// Declared in attn_common.h (AttnP is not visible here: plain arguments); internal C++ linkage, not part of the C-ABI.
}  // extern "C"
void lmod_launch_attn_dq_gemm_raw(const void* ds_ws, const void* K, void* dQ, const int* seqlens, int B, int S, int nh, int group, int ldk,
                                  int lddq, float scale, int causal, const void* rope_cos, const void* rope_sin, const int* rope_pos,
                                  hipStream_t stream) {
  static bool a2 = false;
  GemmP p = {};
  p.A = (const bf16_t*)ds_ws; p.B = (const bf16_t*)K; p.C = dQ;
  p.M = 256; p.N = 128; p.K = S; p.lda = S; p.ldb = ldk; p.ldc = lddq;
  p.batch = B * nh * (S / 256); p.tiles_m = 1; p.tiles_n = 1; p.splitk = 1;
  p.rope_cos = (const bf16_t*)rope_cos; p.rope_sin = (const bf16_t*)rope_sin; p.rope_pos = rope_pos;
  p.at_nh = nh; p.at_group = group; p.at_S = S; p.at_nqb = S / 256; p.at_causal = causal; p.at_scale = scale; p.at_seqlens = seqlens;
  allow_lds(gemm4t_kernel<2>, 2 * G4T_STAGE, a2);
  hipLaunchKernelGGL(gemm4t_kernel<2>, dim3((unsigned)p.batch), dim3(256), 2 * G4T_STAGE, stream, p);
}
extern "C" {
static inline int tn4_max_rows(int lda, int ldb) {      // reduction rows per launch that keep both operand windows below 2 GiB
  const long long ld = lda > ldb ? lda : ldb;
  const long long rows = (0x7fffffffLL - 1024) / (ld * 2);
  return (int)(rows > 0x40000000LL ? 0x40000000LL : rows) & ~63;
}
static int wgrad_tn4(const void* A, const void* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                     void* workspace, long long workspace_bytes, hipStream_t stream) {
  if ((M & 7) || (N & 7) || (lda & 7) || (ldb & 7) || lda < M || ldb < N || ldc < N || (ldc & 3)) return LMOD_EINVAL;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)C & 15)) return LMOD_EINVAL;
  if (K == 0) return LMOD_OK;
  const int kmax = tn4_max_rows(lda, ldb);
  if (kmax < 64) return LMOD_EUNSUPPORTED;
  long long cap = 0;
  if (workspace && !((uintptr_t)workspace & 15))
    cap = (workspace_bytes - (long long)WGRAD_MAX_TILES * 4) / ((long long)((M + 255) / 256) * ((N + 255) / 256) * 262144);
  for (int k0 = 0; k0 < K; k0 += kmax) {
    const int kc = (K - k0 < kmax) ? K - k0 : kmax;
    GemmP p;
    p.A = (const bf16_t*)A + (long long)k0 * lda; p.B = (const bf16_t*)B + (long long)k0 * ldb; p.C = C; p.bias = nullptr;
    p.M = M; p.N = N; p.K = kc; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.batch = 1; p.sA = 0; p.sB = 0; p.sC = 0;
    p.m_valid = nullptr; p.k_valid = nullptr;
    p.act = 0; p.out_f32 = 1; p.accumulate = 1; p.vec_ok = 1;
    p.C2 = nullptr; p.ldc2 = 0; p.sC2 = 0;
    const int s = wgrad_pick_split(M, N, kc, (int)(cap < 1 ? 1 : (cap > 8 ? 8 : cap)));
    p.splitk = s; p.kchunk = ((kc + s - 1) / s + 63) / 64 * 64;
    p.counters = (int*)workspace; p.ws = workspace ? (float*)((char*)workspace + (long long)WGRAD_MAX_TILES * 4) : nullptr;
    p.tiles_m = (M + 255) / 256; p.tiles_n = (N + 255) / 256;
    launch_4t(p, (long long)p.tiles_m * p.tiles_n * s, stream);
    const int st = lmod_launch_status();
    if (st != LMOD_OK) return st;
  }
  return LMOD_OK;
}

int lmod_gemm_wgrad_bf16_nt(const void* At, const void* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                            int b_kmajor, void* workspace, long long workspace_bytes, hipStream_t stream) {
  if (M < 0 || N < 0 || K < 0) return LMOD_EINVAL;
  if (M == 0 || N == 0) return LMOD_OK;          // an empty problem dereferences nothing
  if (!At || !B || !C) return LMOD_EINVAL;
  if (b_kmajor == 2) return wgrad_tn4(At, B, C, M, N, K, lda, ldb, ldc, workspace, workspace_bytes, stream);
  // the workspace bounds the split: WGRAD_MAX_TILES semaphores (16 KiB) + s partial images of 256 KiB per tile
  long long cap = 0;
  if (workspace && !((uintptr_t)workspace & 15))
    cap = (workspace_bytes - (long long)WGRAD_MAX_TILES * 4) / ((long long)((M + 255) / 256) * ((N + 255) / 256) * 262144);
  int s = wgrad_pick_split(M, N, K, (int)(cap < 1 ? 1 : (cap > 8 ? 8 : cap)));
  if (s > 1 && (((uintptr_t)C & 15) || (ldc & 3))) s = 1;
  if (!b_kmajor && (s == 1 || (long long)255 * ldb * 2 + (long long)K * 2 >= 0x7fffffffLL ||
                    (long long)255 * lda * 2 + (long long)K * 2 >= 0x7fffffffLL))
    return lmod_gemm_bf16_nt(At, B, C, nullptr, M, N, K, lda, ldb, ldc, 1, 0, 0, 0, nullptr, nullptr, 0, 1, 1, stream);
  if ((K & 7) || (lda & 7) || (ldb & 7) || lda < K || ldc < N || ((uintptr_t)At & 15) || ((uintptr_t)B & 15)) return LMOD_EINVAL;
  if (b_kmajor) {        // B = X as stored: [K, N], N contiguous
    if ((N & 7) || ldb < N || ((uintptr_t)C & 15) || (ldc & 3)) return LMOD_EINVAL;
    if ((long long)64 * ldb * 2 >= 0x7fffffffLL || (long long)255 * lda * 2 + (long long)K * 2 >= 0x7fffffffLL) return LMOD_EUNSUPPORTED;
  } else if (ldb < K) return LMOD_EINVAL;
  GemmP p;
  p.A = (const bf16_t*)At; p.B = (const bf16_t*)B; p.C = C; p.bias = nullptr;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.batch = 1; p.sA = 0; p.sB = 0; p.sC = 0;
  p.m_valid = nullptr; p.k_valid = nullptr;
  p.act = 0; p.out_f32 = 1; p.accumulate = 1; p.vec_ok = 1;
  p.C2 = nullptr; p.ldc2 = 0; p.sC2 = 0;
  p.splitk = s; p.kchunk = ((K + s - 1) / s + 63) / 64 * 64;
  p.counters = (int*)workspace; p.ws = (float*)((char*)workspace + (long long)WGRAD_MAX_TILES * 4);
  p.tiles_m = (M + 255) / 256; p.tiles_n = (N + 255) / 256;
  const long long nwg = (long long)p.tiles_m * p.tiles_n * s;
  if (b_kmajor) {
    (void)hipFuncSetAttribute((const void*)gemm_256_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * G256_SLOT);
    hipLaunchKernelGGL(gemm_256_kernel<3>, dim3((unsigned)nwg), dim3(512), 8 * G256_SLOT, stream, p);
  } else {
    launch_256<0>(p, nwg, stream);
  }
  return lmod_launch_status();
}

// C[b] (M x N) (+)= A[b]^T B[b] with A [K x M] (row stride lda) and B [K x N] (row stride ldb): the weight-gradient
// form dW = dY^T X on the tensors as autograd holds them (tokens major).  k_valid: live reduction rows per batch
// (exact, no 8-row granularity).  Split-K: pass batch = S with strideA = Kc*lda, strideB = Kc*ldb and a [S, M, N]
// workspace as C.  Requires M % 8 == 0, N % 8 == 0, lda/ldb % 8 == 0, 16-byte aligned A/B/C, ldc % 4 == 0.
int lmod_gemm_bf16_tn(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int batch,
                      long long strideA, long long strideB, long long strideC, const int* k_valid, int out_f32,
                      int accumulate, hipStream_t stream) {
  if (M < 0 || N < 0 || K < 0 || batch < 0) return LMOD_EINVAL;
  if (M == 0 || N == 0 || batch == 0) return LMOD_OK;          // an empty problem dereferences nothing
  if (!A || !B || !C) return LMOD_EINVAL;
  if ((M & 7) || (N & 7) || (lda & 7) || (ldb & 7) || lda < M || ldb < N || ldc < N || (ldc & 3)) return LMOD_EINVAL;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)C & 15) || (strideA & 7) || (strideB & 7) || (strideC & 3))
    return LMOD_EINVAL;
  if ((long long)64 * lda * 2 >= 0x7fffffffLL || (long long)64 * ldb * 2 >= 0x7fffffffLL) return LMOD_EUNSUPPORTED;
  GemmP p;
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.bias = nullptr;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.batch = batch; p.sA = strideA; p.sB = strideB; p.sC = strideC;
  p.m_valid = nullptr; p.k_valid = k_valid;
  p.act = 0; p.out_f32 = out_f32; p.accumulate = accumulate; p.vec_ok = 1;
  p.C2 = nullptr; p.ldc2 = 0; p.sC2 = 0; p.splitk = 1; p.kchunk = 0; p.ws = nullptr; p.counters = nullptr;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_256_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * G256_SLOT);
    attr_set = true;
  }
  p.tiles_m = (M + 255) / 256; p.tiles_n = (N + 255) / 256;
  const long long nwg = (long long)p.tiles_m * p.tiles_n * batch;
  if (nwg > 0x7fffffffLL) return LMOD_EUNSUPPORTED;
  if (out_f32 && accumulate && gemm_tn4() && K <= tn4_max_rows(lda, ldb)) {      // the fp32 accumulate of the step: the 4-wave asm loop
    launch_4t(p, nwg, stream);
    return lmod_launch_status();
  }
  hipLaunchKernelGGL(gemm_256_kernel<2>, dim3((unsigned)nwg), dim3(512), 8 * G256_SLOT, stream, p);
  return lmod_launch_status();
}

int lmod_transpose_bf16(const void* in, void* out, int R, int C, int ld_in, int ld_out,
                        int batch, long long stride_in, long long stride_out, const int* r_valid, hipStream_t stream) {
  if (!in || !out || R < 0 || C < 0) return LMOD_EINVAL;
  if (R == 0 || C == 0 || batch == 0) return LMOD_OK;
  if ((C & 7) || (ld_in & 7) || (ld_out & 7) || ld_in < C || ld_out < ((R + 7) & ~7)) return LMOD_EINVAL;
  if (((uintptr_t)in & 15) || ((uintptr_t)out & 15) || (stride_in & 7) || (stride_out & 7)) return LMOD_EINVAL;
  const int tiles_r = (ld_out + 63) / 64, tiles_c = (C + 63) / 64;
  const int tiles = tiles_r * tiles_c;
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((tiles + 3) / 4, batch), dim3(256), 0, stream,
                     (const bf16_t*)in, (bf16_t*)out, R, C, ld_in, ld_out, stride_in, stride_out, tiles_c, r_valid);
  return lmod_launch_status();
}

}  // extern "C"
