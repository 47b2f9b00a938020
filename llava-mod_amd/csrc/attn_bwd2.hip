// Flash attention backward for head dim 128 (and 64, see the end of this comment) on gfx950 (the autograd of qwen2/modeling_qwen2.py:700-708 / :290-309 that
// the reference gets from torch SDPA / flash-attn): dQ in one kernel, dK + dV in another, from Q, K, V, dO, the forward's
// log-sum-exp rows and delta = rowsum(dO * O).
//
// One workgroup = 4 waves = ONE wave per SIMD with the full 512-register file: 256 accumulator registers owned by inline
// asm (attn_acc256.h) + 256 VGPRs.  A wave OWNS 64 rows (dK/dV kernel: 64 keys; dQ kernel: 64 queries) and streams the
// other side in 32-row tiles through LDS; every operand fragment read from LDS feeds TWO MFMAs (the wave's two 32-row
// owner strips), which halves the LDS traffic per flop of a 32-row-per-wave tiling (LDS pipe ~35 % busy at the MFMA rate).
//
//   owner side   X (K | Q): fragments in registers for the whole block (B operands, 64 VGPRs)
//                Y (V | dO): 256-row image resident in LDS (272-byte rows: conflict-free ds_read_b128 with immediates only)
//   streamed     Xs (Q | K), Ys (dO | V): 32-row tiles, double buffered, 320-byte rows with the 16-byte chunk index XORed by
//                (row >> 3) & 3.  That one image serves BOTH read patterns without bank conflicts: ds_read_b128 operand rows
//                (S, dP) and ds_read_b64_tr_b16 transposed fragments (dK/dV/dQ), see `rowa` / `tra`.
//
// MFMA 32x32x16 (operand index = lane&31, 8 reduction slots at (lane>>5)*8; D: column = lane&31, row = (r&3)+8(r>>2)+4(lane>>5)):
//   S  [t x o] = Xs X^T      dP [t x o] = Ys Y^T          (t: streamed row, o: owner row; a lane owns ONE owner row)
//   P = exp2(S c - lse),  dS = P (dP - delta);  packed to bf16 in place they ARE the B operands of
//   accY [d x o] += Ys^T P   (dV^T;  dK/dV kernel only)   accX [d x o] += Xs^T dS   (dK^T | dQ^T)
//   — the reduction-slot order of an MFMA is free when both operands agree, so no LDS round trip and no cross-lane move.
// Schedule per 32-row tile (dK/dV: 64 MFMAs; dQ: 48), all VALU work placed in the shadow of MFMAs whose inputs are ready:
//   dK/dV:  S (16)  |  dP (16) || exp2, pack P  |  dV += (16) || dS, pack  |  dK += (16) || LDS writes of the next tile
//   dQ:     dP (16) |  S strip 0 (8) || dP -= delta  |  S strip 1 (8) || strip 0: exp2, dS, pack
//                   |  dQ strip 0 += (8) || strip 1: exp2, dS, pack  |  dQ strip 1 += (8)
// One workgroup barrier per tile.  Built by hipcc_agpr.sh with "amdgpu-agpr-alloc"="256".
//
// Head dim 64 (round 5; Qwen2-0.5B student, CLIP tower): the same kernels with HD = 64 — images, strips and phases unchanged, 4 reduction
// steps in S / dP, 2 feature strips per accumulator group, 32 (dK/dV) and 24 (dQ) MFMAs per tile, two VALU pieces behind every MFMA, and
// dP - delta for free: -delta is the dP accumulator's initial value (B2_DFOLD).  The hd-128 instantiations compile to the code they had.
// Grouped-query launches with too few dK/dV workgroups for the chip are head-split (AttnP::nsplit, lmod_attn_bwd_split in attn.hip): a
// workgroup takes a PART of its KV head's query heads and stores fp32 partial sums; attn_dkv_reduce_kernel (attn.hip) adds the parts.
// Work ids go through xcd_work_id (attn_common.h) when the grid's head extent is no multiple of the 8 XCDs.
//
// dS spill (round 6; template flag SPILL of the hd-128 dK/dV kernel): the two kernels compute S, dP and the exponentials twice — 7 matmul
// units for 5 algorithmic ones.  With a workspace the dK/dV kernel also STORES the packed bf16 dS^T it feeds to its dK MFMAs (tile-major
// 1 KiB blocks, see `ds_store_half`), the dQ kernel is not launched, and dQ = scale * dS K runs as one batched TN GEMM
// (gemm.hip: gemm4t_kernel<2>).  dK / dV are bit-identical to the two-kernel form.
#include "attn_common.h"
#include "attn_acc256.h"
#include <type_traits>

#define B2_PITCH 320
#define B2_TILE (32 * B2_PITCH)            // one streamed tensor's tile image
#define B2_LD (2 * B2_TILE)                // lse2[32], delta[32] of the tile (dK/dV kernel)
#define B2_BUF (2 * B2_TILE + 256)
#define B2_YPITCH 272
#define B2_YIMG (256 * B2_YPITCH)
#define B2_LDS (B2_YIMG + 2 * B2_BUF)
#define B2_PIN(x) asm volatile("" : "+v"(x))
#define B2_SB() __builtin_amdgcn_sched_barrier(0)
#ifndef B2_ABL
#define B2_ABL 0                           // timing ablations, a bit mask (WRONG RESULTS): 1 no barrier, 2 no global->LDS staging in
#endif                                     //   the loop, 4 no exp2 / dS VALU work, 8 no operand reads from LDS
#ifndef B2_DFOLD
#define B2_DFOLD 1                         // hd 64: dP - delta without VALU work: -delta is the dP accumulator's INITIAL value (the MFMA's C
#endif                                     //   operand).  dK/dV kernel: read from the tile's LDS image straight into the accumulator registers
                                           //   during the S phase; dQ kernel: two constant blocks of -delta[own row]
#ifndef B2_SPREAD
#define B2_SPREAD 0                        // (measured -0.5 ... -2 %, off; bit 1: dK/dV kernel, bit 2: dQ kernel) hd 64: the softmax / dS pieces of a tile spread over more MFMA gaps (dK/dV kernel: 12 exp pieces in
#endif                                     //   the dP phase, 4 in the first half of dV, dS pieces in dV's second half and dK's first; dQ kernel:
                                           //   one strip per phase — S0 | S1 || exp 0 | dP0 || exp 1 | dP1 || dS 0 | dQ0 || dS 1 | dQ1)
#ifndef B2_DEPTH
#define B2_DEPTH 2                         // operand fragments are read from LDS this many steps ahead
#endif

template <bool FIRST>
__device__ __forceinline__ void sd_mfma(f32x16& acc, const bf16x8 a, const bf16x8 b) {
  if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b) : B2_CLOB_ALL);
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : B2_CLOB_ALL);
}
__device__ __forceinline__ void sd_mfma_c(f32x16& acc, const bf16x8 a, const bf16x8 b, const f32x16& c) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "v"(b), "v"(c) : B2_CLOB_ALL);
}
__device__ __forceinline__ bf16x8 tr2(const char* p0, const char* p1) {      // reduction slots 0-3 | 4-7
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
  return (bf16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
template <int N>
__device__ __forceinline__ void store_strip(bf16_t* dst, const float mul) {   // 16 contiguous features of this lane's row
  float v[16];
  acc_read16<N>(v);
  u32x4 w0, w1;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    w0[k] = pack2bf(v[2 * k] * mul, v[2 * k + 1] * mul);
    w1[k] = pack2bf(v[8 + 2 * k] * mul, v[8 + 2 * k + 1] * mul);
  }
  *(u32x4*)(dst) = w0;
  *(u32x4*)(dst + 8) = w1;
}
template <int N>
__device__ __forceinline__ void store_strip_f32(float* dst) {                  // the same 16 features, fp32, unscaled
  float v[16];
  acc_read16<N>(v);
#pragma unroll
  for (int k = 0; k < 4; ++k) *(f32x4*)(dst + 4 * k) = (f32x4){v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
}
// RoPE backward on the strip pair (N1: features f .. f+15 < 64, N2: the same + 64) of this lane's row, in registers: the lane holds both
// halves of every rotate_half pair.  g = the bf16-rounded gradients the plain epilogue would store; then rope_kernel's backward
// arithmetic (rowops.hip): dx1 = g1*cos1 + g2*sin2, dx2 = g2*cos2 - g1*sin1, each product rounded to bf16 — bit-identical to
// store + lmod_rope(backward).  cp / sp: cos / sin rows of the token's position, at feature f.
template <int N1, int N2>
__device__ __forceinline__ void store_pair_rope(bf16_t* dst, const float mul, const bf16_t* cp, const bf16_t* sp) {
  float v1[16], v2[16];
  acc_read16<N1>(v1);
  acc_read16<N2>(v2);
#pragma unroll
  for (int hx = 0; hx < 2; ++hx) {
    const u32x4 c1 = *(const u32x4*)(cp + hx * 8), c2 = *(const u32x4*)(cp + 64 + hx * 8);
    const u32x4 s1 = *(const u32x4*)(sp + hx * 8), s2 = *(const u32x4*)(sp + 64 + hx * 8);
    u32x4 o1, o2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float a0 = bfround(v1[hx * 8 + 2 * k] * mul), a1 = bfround(v1[hx * 8 + 2 * k + 1] * mul);
      const float b0 = bfround(v2[hx * 8 + 2 * k] * mul), b1 = bfround(v2[hx * 8 + 2 * k + 1] * mul);
      o1[k] = pack2bf(bfround(a0 * bflo(c1[k])) + bfround(b0 * bflo(s2[k])), bfround(a1 * bfhi(c1[k])) + bfround(b1 * bfhi(s2[k])));
      o2[k] = pack2bf(bfround(b0 * bflo(c2[k])) - bfround(a0 * bflo(s1[k])), bfround(b1 * bfhi(c2[k])) - bfround(a1 * bfhi(s1[k])));
    }
    *(u32x4*)(dst + hx * 8) = o1;
    *(u32x4*)(dst + 64 + hx * 8) = o2;
  }
}
template <int I> using IC = std::integral_constant<int, I>;
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(IC<I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
// MFMA result -> first VALU read.  hipcc does not see inside the asm MFMAs, so (a) it inserts no wait states and (b) it is free
// to hoist the VALU consumers above them.  The pad is an asm statement that "modifies" the accumulators: every consumer is
// data-dependent on it and it is itself ordered after the MFMAs (all asm volatile).  It is placed after the MFMAs of the step
// FOLLOWING the producer's last one, so >= 1 full MFMA issue slot (32 cycles) has passed; the s_nop covers the rest.
__device__ __forceinline__ void hazard_pad(f32x16& a, f32x16& b) {
  asm volatile("s_nop 3" : "+v"(a), "+v"(b)::B2_CLOB_ALL);
}
__device__ __forceinline__ void hazard_pad(f32x16& a) { asm volatile("s_nop 3" : "+v"(a)::B2_CLOB_ALL); }

template <bool DKV, bool CAUSAL, int HD, bool SPILL = false>
__device__ __forceinline__ void bwd2_block(const AttnP& p, char* smem, int ob, int ho, int b, int sp) {
  static_assert(!SPILL || (DKV && HD == 128), "the dS spill belongs to the hd-128 dK/dV kernel");
  // Head dim 64 keeps the tile images (pitches, swizzle, owner strips) and the phase order and drops what belongs to the absent upper
  // feature half: KS reduction steps in S / dP, DT 32-feature strips per accumulator group, CH 16-byte chunks per staged row.  The
  // softmax arithmetic per tile is the same, so VP of its two-element pieces go behind every MFMA instead of one.
  constexpr bool DF = B2_DFOLD && HD == 64;      // hd 128 keeps its subtractions: no registers to spare (the fold costs 22 more scalar spills there)
  constexpr int KS = HD / 16, DT = HD / 32, NST = 2 * DT, CH = HD / 8, CHS = HD == 128 ? 4 : 3, VP = 128 / HD, NLD = HD / 64;
  static_assert(HD == 128 || HD == 64, "head dim 64 or 128");
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = p.cu ? (p.cu[b + 1] - p.cu[b]) : p.S;
  const int len = p.seqlens ? min(p.seqlens[b], S) : S;
  const int o0 = ob * 256, ow0 = o0 + wave * 64;
  if (o0 >= S) return;
  const long long tok0 = p.cu ? (long long)p.cu[b] : (long long)b * S;
  const float c = p.scale * 1.4426950408889634f;
  // dK/dV kernel: the query heads h0 .. h0 + gsz - 1 of this KV head's group (all of them unless the launch is head-split)
  int h0 = 0, gsz = DKV ? p.group : 1;
  if constexpr (DKV && !SPILL) {
    if (p.nsplit > 1) {
      const int per = (p.group + p.nsplit - 1) / p.nsplit;
      h0 = sp * per;
      gsz = max(0, min(p.group, h0 + per) - h0);
    }
  }

  const bf16_t* Xo = DKV ? p.K + tok0 * p.ldk + ho * HD : p.Q + tok0 * p.ldq + ho * HD;
  const bf16_t* Yo = DKV ? p.V + tok0 * p.ldv + ho * HD : p.dO + tok0 * p.lddo + ho * HD;
  const int ldxo = DKV ? p.ldk : p.ldq, ldyo = DKV ? p.ldv : p.lddo;
  const bf16_t* Xs = DKV ? p.Q + tok0 * p.ldq + (ho * p.group + h0) * HD : p.K + tok0 * p.ldk + (ho / p.group) * HD;
  const bf16_t* Ys = DKV ? p.dO + tok0 * p.lddo + (ho * p.group + h0) * HD : p.V + tok0 * p.ldv + (ho / p.group) * HD;
  const int ldxs = DKV ? p.ldq : p.ldk, ldys = DKV ? p.lddo : p.ldv;

  // ---- owner fragments X (B operands of S): row ow0 + 32*os + l31, features ks*16 + hi*8 .. +7
  bf16x8 xf[2][KS];
  float olse[2] = {0.f, 0.f}, odl[2] = {0.f, 0.f};             // dQ kernel: the lane's own rows' lse (log2 units) and delta
#pragma unroll
  for (int os = 0; os < 2; ++os) {
    const int row = ow0 + os * 32 + l31;
    const bf16_t* xp = Xo + (long long)min(row, S - 1) * ldxo + hi * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      xf[os][ks] = *(const bf16x8*)(xp + ks * 16);
      if (row >= S) xf[os][ks] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
    if constexpr (!DKV) {
      const long long si = ((long long)b * p.nh + ho) * p.S + min(row, S - 1);
      olse[os] = p.LSE[si] * 1.4426950408889634f;
      odl[os] = p.Delta[si];
    }
  }
  // ---- owner image Y: rows o0 .. o0+255 (zeros past the end of the sequence) -> LDS, 272-byte rows
  {
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)Yo, 0, (int)(((long long)(S - 1) * ldyo + HD) * 2), 0x00020000);
    u32x4 v[CH];                                                 // all loads in flight (the main loop's registers are not live yet)
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int id = tid + 256 * i, row = id >> CHS, ch = id & (CH - 1);
      v[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, (uint32_t)((o0 + row) * ldyo + ch * 8) * 2u, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int id = tid + 256 * i, row = id >> CHS, ch = id & (CH - 1);
      *(u32x4*)(smem + row * B2_YPITCH + ch * 16) = v[i];
    }
  }
  acc_zero_all();

  // ---- streamed tiles
  const int nt = DKV ? (S + 31) >> 5 : ((CAUSAL ? min(o0 + 256, len) : len) + 31) >> 5;
  const int first = (DKV && CAUSAL) ? (o0 >> 5) : 0;
  const int per_head = nt - first;
  int total = per_head * gsz;
  if (DKV && o0 >= len) total = 0;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)Xs, 0, (int)(((long long)(S - 1) * ldxs + gsz * HD) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rys = __builtin_amdgcn_make_buffer_rsrc((void*)Ys, 0, (int)(((long long)(S - 1) * ldys + gsz * HD) * 2), 0x00020000);
  const int srow = tid >> CHS, sch = tid & (CH - 1);                    // staging: chunk sch of rows srow (hd 128: and srow + 16)
  const uint32_t xg0 = (uint32_t)(srow * ldxs + sch * 8) * 2u, xg1 = xg0 + (uint32_t)(16 * ldxs) * 2u;
  const uint32_t yg0 = (uint32_t)(srow * ldys + sch * 8) * 2u, yg1 = yg0 + (uint32_t)(16 * ldys) * 2u;
  const int wa0 = B2_YIMG + srow * B2_PITCH + ((sch ^ ((srow >> 3) & 3)) << 4);
  const int wa1 = B2_YIMG + (srow + 16) * B2_PITCH + ((sch ^ (((srow + 16) >> 3) & 3)) << 4);
  u32x4 xr0, xr1, yr0, yr1;
  uint32_t lsr = 0, dlr = 0;                                    // dK/dV kernel: one lse / delta value of the staged tile
  const long long ld0 = ((long long)b * p.nh + (DKV ? ho * p.group + h0 : ho)) * p.S;
  const __amdgpu_buffer_rsrc_t rlse = __builtin_amdgcn_make_buffer_rsrc((void*)(p.LSE + ld0), 0, gsz * p.S * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdel = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Delta + ld0), 0, gsz * p.S * 4, 0x00020000);
  int ghh = 0, gj = first;                                      // (head, tile) of the next gload: consecutive calls
  auto gload = [&](const int) {
    const int hh = ghh, j = gj;
    if (++gj == nt) { gj = first; ++ghh; }
    const uint32_t ax = (uint32_t)(j * 32 * ldxs + hh * HD) * 2u, ay = (uint32_t)(j * 32 * ldys + hh * HD) * 2u;
    xr0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xg0 + ax, 0, 0));
    if constexpr (NLD == 2) xr1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xg1 + ax, 0, 0));
    yr0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rys, yg0 + ay, 0, 0));
    if constexpr (NLD == 2) yr1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rys, yg1 + ay, 0, 0));
    if constexpr (DKV) {        // lse / delta of the tile's 32 rows: lanes 0-31 of every wave fetch them (no branch, no arithmetic
      const int t = j * 32 + l31;      // on the loaded value here: either would make hipcc wait for the loads at once)
      const uint32_t o = (hi == 0 && t < S) ? (uint32_t)((hh * p.S + t) * 4) : 0xffffffffu;
      lsr = __builtin_amdgcn_raw_buffer_load_b32(rlse, o, 0, 0);
      dlr = __builtin_amdgcn_raw_buffer_load_b32(rdel, o, 0, 0);
    }
  };
  auto lwrite = [&](const int buf) {
    char* bb = smem + buf * B2_BUF;
    *(u32x4*)(bb + wa0) = xr0;
    if constexpr (NLD == 2) *(u32x4*)(bb + wa1) = xr1;
    *(u32x4*)(bb + B2_TILE + wa0) = yr0;
    if constexpr (NLD == 2) *(u32x4*)(bb + B2_TILE + wa1) = yr1;
    if constexpr (DKV) {
      if (tid < 32) {
        *(float*)(bb + B2_YIMG + B2_LD + tid * 4) = __uint_as_float(lsr) * 1.4426950408889634f;
        *(float*)(bb + B2_YIMG + B2_LD + 128 + tid * 4) = DF ? -__uint_as_float(dlr) : __uint_as_float(dlr);
      }
    }
  };

  // ---- operand read addresses (bytes from smem; immediates carry feature group, row group and tensor)
  int rowa[2], tra[4];
  {
    const int hrow = (l31 >> 3) & 3;
#pragma unroll
    for (int par = 0; par < 2; ++par) rowa[par] = B2_YIMG + l31 * B2_PITCH + (((2 * par + hi) ^ hrow) << 4);
    const int i16 = lane & 15, gi = (lane >> 4) & 1, r4 = i16 >> 2, cc = i16 & 3;
#pragma unroll
    for (int hc = 0; hc < 4; ++hc) tra[hc] = B2_YIMG + (4 * hi + r4) * B2_PITCH + (((2 * (cc & 1) + gi) ^ hc) << 4) + 8 * (cc >> 1);
  }
  int lda = B2_YIMG + B2_LD + hi * 16;
  const int ya = (wave * 64 + l31) * B2_YPITCH + hi * 16;

  auto rd_xs = [&](const int ks) { if (B2_ABL & 8) return xf[0][ks % KS]; return *(const bf16x8*)(smem + rowa[ks & 1] + (ks >> 1) * 64); };
  auto rd_ys = [&](const int ks) { if (B2_ABL & 8) return xf[1][ks % KS]; return *(const bf16x8*)(smem + rowa[ks & 1] + B2_TILE + (ks >> 1) * 64); };
  auto rd_y = [&](const int os, const int ks) { if (B2_ABL & 8) return xf[os][ks % KS]; return *(const bf16x8*)(smem + ya + os * 32 * B2_YPITCH + ks * 32); };
  auto rd_tr = [&](const int tensor, const int tk, const int dt) {
    if (B2_ABL & 8) return xf[tensor][(tk * DT + dt) % KS];
    return tr2(smem + tra[(2 * tk) & 3] + tensor * B2_TILE + (16 * tk) * B2_PITCH + dt * 64,
               smem + tra[(2 * tk + 1) & 3] + tensor * B2_TILE + (16 * tk + 8) * B2_PITCH + dt * 64);
  };

  f32x16 s[2], dp[2];
  constexpr bool DQC = !DKV && DF;
  f32x16 ndl[DQC ? 2 : 1];             // dQ kernel, hd 64: -delta[own row] in every element, the C operand of the tile's first dP MFMAs
  if constexpr (DQC) {
#pragma unroll
    for (int os = 0; os < 2; ++os)
#pragma unroll
      for (int r = 0; r < 16; ++r) ndl[os][r] = -odl[os];
  }
  uint32_t pP[2][8], pS[2][8];         // bf16-packed P and dS: [owner strip][4 * reduction step of 16 streamed rows + dword]
  f32x4 l4c[2], d4c[2];                // dK/dV kernel: lse / delta of the 4 streamed rows of row group G4, in slot G4 & 1;
                                       // read from LDS one row group AHEAD of their use (a read placed at its use costs
                                       // an LDS round trip per group with nothing else for this one wave to issue)
  auto ld_l4 = [&](const int g4) { if constexpr (DKV) l4c[g4 & 1] = *(const f32x4*)(smem + lda + g4 * 32); };
  auto ld_d4 = [&](const int g4) { if constexpr (DKV) d4c[g4 & 1] = *(const f32x4*)(smem + lda + 128 + g4 * 32); };

  // VALU work comes in PIECES of two elements so that one piece fits behind one MFMA.
  // exp piece N (0..15): rows 4*G4 + 2*H, +1 of strip OS, with G4 = N >> 2, OS = (N >> 1) & 1, H = N & 1 (dK/dV kernel order:
  // both strips of a row group share the lse / delta values read from LDS); p = exp2(s c - lse), masked entries -> 0.
  // The bf16 packing of a piece is issued with the NEXT piece (v_exp -> consumer needs a wait state: hipcc pads with s_nop
  // unless an independent instruction sits between them).
  auto exp_piece = [&](auto masked_t, auto os_t, auto g4_t, auto h_t, const int thr) {
    constexpr bool MASKED = decltype(masked_t)::value;
    constexpr int OS = decltype(os_t)::value, G4 = decltype(g4_t)::value, H = decltype(h_t)::value;
    if (B2_ABL & 4) return;
    if constexpr (DKV) { if (OS == 0 && H == 0 && G4 < 3) ld_l4(G4 + 1); }
#pragma unroll
    for (int e = 2 * H; e < 2 * H + 2; ++e) {
      const int r = 4 * G4 + e, base = (r & 3) + 8 * (r >> 2);
      float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[OS][r], c, DKV ? -l4c[G4 & 1][e] : -olse[OS]));
      if constexpr (MASKED) { if (DKV ? (base < thr) : (base > thr)) pv = 0.f; }
      s[OS][r] = pv;
    }
    if constexpr (!DKV) B2_PIN(s[OS][4 * G4 + 2 * H]);
  };
  auto pack_piece = [&](auto os_t, auto g4_t, auto h_t) {        // dK/dV kernel: P of the piece -> bf16 pair
    constexpr int OS = decltype(os_t)::value, G4 = decltype(g4_t)::value, H = decltype(h_t)::value;
    if (B2_ABL & 4) { pP[OS][2 * G4 + H] = 0x3c003c00u; return; }
    pP[OS][2 * G4 + H] = pack2bf(s[OS][4 * G4 + 2 * H], s[OS][4 * G4 + 2 * H + 1]);
    B2_PIN(pP[OS][2 * G4 + H]);
  };
  // dS piece: dS = p (dp - delta) of the same two rows, packed.  SUB: delta still to be subtracted (dK/dV kernel)
  auto ds_piece = [&](auto sub_t, auto os_t, auto g4_t, auto h_t) {
    constexpr bool SUB = decltype(sub_t)::value;
    constexpr int OS = decltype(os_t)::value, G4 = decltype(g4_t)::value, H = decltype(h_t)::value;
    if (B2_ABL & 4) { pS[OS][2 * G4 + H] = 0x3c003c00u; return; }
    if constexpr (SUB && DKV && !DF) { if (OS == 0 && H == 0 && G4 < 3) ld_d4(G4 + 1); }
    const int r = 4 * G4 + 2 * H;
    constexpr bool SUBD = SUB && !DF;
    const float v0 = s[OS][r] * (SUBD ? dp[OS][r] - d4c[G4 & 1][2 * H] : dp[OS][r]);
    const float v1 = s[OS][r + 1] * (SUBD ? dp[OS][r + 1] - d4c[G4 & 1][2 * H + 1] : dp[OS][r + 1]);
    pS[OS][2 * G4 + H] = pack2bf(v0, v1);
    B2_PIN(pS[OS][2 * G4 + H]);
  };
  auto opnd = [](const uint32_t (&w)[8], const int tk) {
    return __builtin_bit_cast(bf16x8, (u32x4){w[4 * tk], w[4 * tk + 1], w[4 * tk + 2], w[4 * tk + 3]});
  };
  using T = BoolTag<true>;
  using F = BoolTag<false>;

  // ---- dS spill (dK/dV kernel, SPILL): the tile's dS^T — the bf16 words the dK MFMAs consume — goes to the workspace in the layout
  //   ws[b * nh + head][query tile j (S / 32)][key strip (S / 32)][half (2)][key in strip (32)][16 queries]        (1 KiB per half block)
  // chosen so that ONE store instruction of a wave writes ONE contiguous KiB (8 whole 128-byte lines): a lane holds, per owner strip,
  // queries {0-3, 8-11, 16-19, 24-27} + 4 hi of the tile (r -> (r & 3) + 8 (r >> 2) + 4 hi); v_permlane32_swap exchanges the 4-query runs
  // of the two lane halves, after which lane (key, hi) owns the 16-byte chunk hi of its key's 32-byte row in half block 0 (queries 0-15)
  // and in half block 1 (queries 16-31).  (A [key][query] row-major image — 32 rows x 32 bytes at a 4 KiB stride per instruction — doubled
  // this kernel's time: partial lines in 32 DRAM pages per store.)  pS is dead after the dK MFMAs that read it: the exchange is in place.
  // gemm4t_kernel<2> (gemm.hip) reads this layout through its per-lane staging offsets.
  constexpr bool spill = SPILL;
  char* const ds_blk = spill ? (char*)(p.ds_ws + ((long long)b * p.nh + ho * p.group + h0) * S * S + (long long)ow0 * 32) : nullptr;
  // (hipcc may set up an asm operand with a v_mov right in front of the statement and does not know the VALU-write -> v_permlane-read
  // hazard inside it: the wait states are part of the statement.  Without them the first exchange of a strip read a stale register.)
  auto ds_swap = [&](uint32_t& a, uint32_t& c) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(c)); };
  // the lane offset is recomputed from the thread id where it is used: a value hoisted to kernel entry would cost a register across the
  // whole tile loop, and this kernel has none to spare
  auto ds_lane = [&]() {
    int t = tid;
    asm volatile("" : "+v"(t));
    return (uint32_t)((t & 31) * 32 + ((t >> 5) & 1) * 16);
  };
  auto ds_store_half = [&](char* base, const uint32_t lo, const int os, const int half) {   // half 0: queries 0-15 of the tile, 1: 16-31
    uint32_t* w = &pS[os][4 * half];
    ds_swap(w[0], w[2]);
    ds_swap(w[1], w[3]);
    __builtin_nontemporal_store((u32x4){w[0], w[1], w[2], w[3]}, (u32x4*)(base + lo + (os * 2048 + half * 1024)));
  };
  auto ds_store_zero = [&](char* base) {                                            // a tile this wave skips (wholly above the diagonal)
    const uint32_t lo = ds_lane();
#pragma unroll
    for (int q = 0; q < 4; ++q) __builtin_nontemporal_store((u32x4){0u, 0u, 0u, 0u}, (u32x4*)(base + lo + q * 1024));
  };

  // ================================================================================ dK/dV tile body
  // VALU slots: slot N (0..15) = exp piece N + the bf16 packing of piece N - 1 | dS piece N.  VP slots go behind MFMA number M of a phase.
  auto exp_slot = [&](auto masked_t, auto n_t, const int lo0, const int lo1) {
    constexpr int N = decltype(n_t)::value;
    exp_piece(masked_t, IC<((N >> 1) & 1)>{}, IC<(N >> 2)>{}, IC<(N & 1)>{}, ((N >> 1) & 1) ? lo1 : lo0);
    if constexpr (N > 0) pack_piece(IC<(((N - 1) >> 1) & 1)>{}, IC<((N - 1) >> 2)>{}, IC<((N - 1) & 1)>{});
  };
  constexpr bool SP = (B2_SPREAD & 1) && HD == 64, SPQ = (B2_SPREAD & 2) && HD == 64;
  auto exp_range = [&](auto masked_t, auto lo_t, auto hi_t, const int lo0, const int lo1) {       // exp slots LO .. HI - 1
    constexpr int LO = decltype(lo_t)::value, HI = decltype(hi_t)::value;
    if constexpr (HI > LO) static_for<HI - LO>([&](auto v_t) { exp_slot(masked_t, IC<(LO + decltype(v_t)::value)>{}, lo0, lo1); });
  };
  auto ds_range = [&](auto lo_t, auto hi_t) {                                                       // dS pieces LO .. HI - 1
    constexpr int LO = decltype(lo_t)::value, HI = decltype(hi_t)::value;
    if constexpr (HI > LO) static_for<HI - LO>([&](auto v_t) {
      constexpr int N = LO + decltype(v_t)::value;
      ds_piece(T{}, IC<((N >> 1) & 1)>{}, IC<(N >> 2)>{}, IC<(N & 1)>{});
    });
  };
  // which pieces go behind MFMA number M of a phase.  Plain: VP exp pieces behind every dP MFMA, VP dS pieces behind every dV MFMA.
  // Spread (hd 64; a tile phase is 8 MFMAs, two reduction halves tk of 4): exp 0-11 behind the dP MFMAs (2, 1, 2, 1, ...), exp 12-15 behind
  // dV's tk 0 MFMAs (they feed tk 1), dS 0-7 behind dV's tk 1 MFMAs, dS 8-15 (feed dK's tk 1) behind dK's tk 0 MFMAs.
  auto dp_gap = [&](auto masked_t, auto m_t, const int lo0, const int lo1) {
    constexpr int M = decltype(m_t)::value;
    if constexpr (SP) exp_range(masked_t, IC<((3 * M + 1) / 2)>{}, IC<((3 * M + 4) / 2)>{}, lo0, lo1);
    else exp_range(masked_t, IC<(VP * M)>{}, IC<(VP * M + VP)>{}, lo0, lo1);
  };
  auto dv_gap = [&](auto masked_t, auto m_t, const int lo0, const int lo1) {
    constexpr int M = decltype(m_t)::value;
    if constexpr (SP) {
      if constexpr (M < 4) exp_range(masked_t, IC<(12 + M)>{}, IC<(13 + M)>{}, lo0, lo1);
      else { if constexpr (M == 4) hazard_pad(dp[0], dp[1]); ds_range(IC<(2 * (M - 4))>{}, IC<(2 * (M - 4) + 2)>{}); }
    } else {
      if constexpr (M == 0) hazard_pad(dp[0], dp[1]);
      ds_range(IC<(VP * M)>{}, IC<(VP * M + VP)>{});
    }
  };
  auto dk_gap = [&](auto m_t) {
    constexpr int M = decltype(m_t)::value;
    if constexpr (SP && M < 4) ds_range(IC<(8 + 2 * M)>{}, IC<(10 + 2 * M)>{});
  };
  auto body_dkv = [&](auto masked_t, const int lo0, const int lo1, char* const ds_base) {
    // ---- S = Xs X^T
    {
      bf16x8 af[B2_DEPTH + 1];
#pragma unroll
      for (int st = 0; st < B2_DEPTH; ++st) af[st] = rd_xs(st);
      static_for<KS>([&](auto ks_t) {
        constexpr int ks = decltype(ks_t)::value;
        if (ks + B2_DEPTH < KS) af[(ks + B2_DEPTH) % (B2_DEPTH + 1)] = rd_xs(ks + B2_DEPTH);
        const bf16x8 a = af[ks % (B2_DEPTH + 1)];
        sd_mfma<ks == 0>(s[0], a, xf[0][ks]);
        sd_mfma<ks == 0>(s[1], a, xf[1][ks]);
        if (ks == KS - 3) ld_l4(0);
        if constexpr (DF) {                // dP starts at -delta[streamed row]: 8 reads (4 row groups x 2 strips) spread over the S phase
          static_for<8 / KS>([&](auto j_t) {
            constexpr int i = ks * (8 / KS) + decltype(j_t)::value, os = i & 1, g4 = i >> 1;
            const f32x4 v = *(const f32x4*)(smem + lda + 128 + g4 * 32);
#pragma unroll
            for (int e = 0; e < 4; ++e) dp[os][4 * g4 + e] = v[e];
          });
        }
        B2_SB();
      });
    }
    // ---- dP = Ys Y^T  ||  P = exp2(S c - lse), packed: VP pieces behind every MFMA
    {
      bf16x8 af[B2_DEPTH + 1], b0[B2_DEPTH + 1], b1[B2_DEPTH + 1];
#pragma unroll
      for (int st = 0; st < B2_DEPTH; ++st) { af[st] = rd_ys(st); b0[st] = rd_y(0, st); b1[st] = rd_y(1, st); }
      static_for<KS>([&](auto ks_t) {
        constexpr int ks = decltype(ks_t)::value, nx = ks + B2_DEPTH, sl = nx % (B2_DEPTH + 1), cu = ks % (B2_DEPTH + 1);
        if (nx < KS) { af[sl] = rd_ys(nx); b0[sl] = rd_y(0, nx); b1[sl] = rd_y(1, nx); }
        sd_mfma<(ks == 0 && !DF)>(dp[0], af[cu], b0[cu]);
        if (ks == 0) hazard_pad(s[0], s[1]);
        dp_gap(masked_t, IC<(2 * ks)>{}, lo0, lo1);
        B2_SB();
        sd_mfma<(ks == 0 && !DF)>(dp[1], af[cu], b1[cu]);
        dp_gap(masked_t, IC<(2 * ks + 1)>{}, lo0, lo1);
        if (ks == KS - 2 && !DF) ld_d4(0);
        B2_SB();
      });
    }
    // ---- dV^T += Ys^T P  ||  dS = P (dP - delta), packed
    {
      bf16x8 af[B2_DEPTH + 1];
#pragma unroll
      for (int st = 0; st < B2_DEPTH; ++st) af[st] = rd_tr(1, st / DT, st % DT);
      static_for<NST>([&](auto st_t) {
        constexpr int st = decltype(st_t)::value, tk = st / DT, dt = st % DT, nx = st + B2_DEPTH;
        if (nx < NST) af[nx % (B2_DEPTH + 1)] = rd_tr(1, nx / DT, nx % DT);
        const bf16x8 a = af[st % (B2_DEPTH + 1)];
        if (st == (SP ? DT : 0)) pack_piece(IC<1>{}, IC<3>{}, IC<1>{});    // the last exp piece's pair (feeds tk = 1 only)
        acc_mfma<8 + dt>(a, opnd(pP[0], tk));
        dv_gap(masked_t, IC<(2 * st)>{}, lo0, lo1);
        B2_SB();
        acc_mfma<12 + dt>(a, opnd(pP[1], tk));
        dv_gap(masked_t, IC<(2 * st + 1)>{}, lo0, lo1);
        B2_SB();
      });
    }
    // ---- dK^T += Xs^T dS
    {
      uint32_t ds_lo = 0;
      if constexpr (spill) ds_lo = ds_lane();              // once per tile, live through this phase only
      bf16x8 af[B2_DEPTH + 1];
#pragma unroll
      for (int st = 0; st < B2_DEPTH; ++st) af[st] = rd_tr(0, st / DT, st % DT);
      static_for<NST>([&](auto st_t) {
        constexpr int st = decltype(st_t)::value, tk = st / DT, dt = st % DT, nx = st + B2_DEPTH;
        if (nx < NST) af[nx % (B2_DEPTH + 1)] = rd_tr(0, nx / DT, nx % DT);
        const bf16x8 a = af[st % (B2_DEPTH + 1)];
        acc_mfma<dt>(a, opnd(pS[0], tk));
        dk_gap(IC<(2 * st)>{});
        if constexpr (SP) B2_SB();
        acc_mfma<4 + dt>(a, opnd(pS[1], tk));
        dk_gap(IC<(2 * st + 1)>{});
        if constexpr (DKV && HD == 128) {
          // dS spill: the tk = 0 words (queries 0-15) have fed their last MFMA once the tk = 1 steps begin: one strip behind each of
          // the first two of them; the tk = 1 words follow the phase
          if constexpr (spill && tk == 1 && dt < 2) ds_store_half(ds_base, ds_lo, dt, 0);
        }
        B2_SB();
      });
      if constexpr (spill) { ds_store_half(ds_base, ds_lo, 0, 1); ds_store_half(ds_base, ds_lo, 1, 1); }
    }
  };

  // ================================================================================ dQ tile body
  auto body_dq = [&](auto masked_t, const int up0, const int up1) {
    auto exp_grp = [&](auto m_t, auto os_t, auto g4_t, const int thr) {
      exp_piece(m_t, os_t, g4_t, IC<0>{}, thr);
      exp_piece(m_t, os_t, g4_t, IC<1>{}, thr);
    };
    auto ds_grp = [&](auto sub_t, auto os_t, auto g4_t) {
      ds_piece(sub_t, os_t, g4_t, IC<0>{});
      ds_piece(sub_t, os_t, g4_t, IC<1>{});
    };
    // the softmax / dS work of one owner strip as 7 slots of row groups; VP slots go behind MFMA number M of a phase
    auto grp_slot = [&](auto m_t, auto os_t, auto n_t, const int thr) {
      constexpr int N = decltype(n_t)::value;
      if constexpr (N == 0) exp_grp(m_t, os_t, IC<0>{}, thr);
      if constexpr (N == 1) { ds_grp(F{}, os_t, IC<0>{}); exp_grp(m_t, os_t, IC<1>{}, thr); }
      if constexpr (N == 2) ds_grp(F{}, os_t, IC<1>{});
      if constexpr (N == 3) exp_grp(m_t, os_t, IC<2>{}, thr);
      if constexpr (N == 4) ds_grp(F{}, os_t, IC<2>{});
      if constexpr (N == 5) exp_grp(m_t, os_t, IC<3>{}, thr);
      if constexpr (N == 6) ds_grp(F{}, os_t, IC<3>{});
    };
    auto grp_slots = [&](auto m_t, auto os_t, auto mf_t, const int thr) {
      constexpr int M = decltype(mf_t)::value;
      static_for<VP>([&](auto v_t) { grp_slot(m_t, os_t, IC<(VP * M + decltype(v_t)::value)>{}, thr); });
    };
    if constexpr (SPQ && !DKV) {
      // hd 64, one owner strip per phase: every phase's MFMAs carry the other strip's softmax / dS work of ONE row group each
      //   S0 | S1 || exp 0 | dP0 || exp 1 | dP1 || dS 0 | dQ0 || dS 1 | dQ1      (24 MFMAs; the K and V row fragments are read twice)
      static_for<2>([&](auto os_t) {                                       // ---- S strip OS  ||  (OS = 1) exp of strip 0
        constexpr int OS = decltype(os_t)::value;
        bf16x8 af[B2_DEPTH + 1];
#pragma unroll
        for (int st = 0; st < B2_DEPTH; ++st) af[st] = rd_xs(st);
        static_for<KS>([&](auto ks_t) {
          constexpr int ks = decltype(ks_t)::value;
          if (ks + B2_DEPTH < KS) af[(ks + B2_DEPTH) % (B2_DEPTH + 1)] = rd_xs(ks + B2_DEPTH);
          sd_mfma<ks == 0>(s[OS], af[ks % (B2_DEPTH + 1)], xf[OS][ks]);
          if constexpr (OS == 1) {
            if (ks == 0) hazard_pad(s[0]);
            exp_grp(masked_t, IC<0>{}, ks_t, up0);
          }
          B2_SB();
        });
      });
      static_for<2>([&](auto os_t) {                                       // ---- dP strip OS  ||  exp of strip 1 | dS of strip 0
        constexpr int OS = decltype(os_t)::value;
        bf16x8 af[B2_DEPTH + 1], bb[B2_DEPTH + 1];
#pragma unroll
        for (int st = 0; st < B2_DEPTH; ++st) { af[st] = rd_ys(st); bb[st] = rd_y(OS, st); }
        static_for<KS>([&](auto ks_t) {
          constexpr int ks = decltype(ks_t)::value, nx = ks + B2_DEPTH, sl = nx % (B2_DEPTH + 1), cu = ks % (B2_DEPTH + 1);
          if (nx < KS) { af[sl] = rd_ys(nx); bb[sl] = rd_y(OS, nx); }
          if constexpr (DQC && ks == 0) sd_mfma_c(dp[OS], af[cu], bb[cu], ndl[OS]);
          else sd_mfma<ks == 0>(dp[OS], af[cu], bb[cu]);
          if constexpr (OS == 0) {
            if (ks == 0) hazard_pad(s[1]);
            exp_grp(masked_t, IC<1>{}, ks_t, up1);
          } else {
            if (ks == 0) hazard_pad(dp[0]);
            if constexpr (!DQC) {
#pragma unroll
              for (int e = 0; e < 4; ++e) dp[0][4 * ks + e] -= odl[0];
            }
            ds_grp(F{}, IC<0>{}, ks_t);
          }
          B2_SB();
        });
      });
      static_for<2>([&](auto os_t) {                                       // ---- dQ^T strip OS += K^T dS  ||  (OS = 0) dS of strip 1
        constexpr int OS = decltype(os_t)::value;
        bf16x8 af[B2_DEPTH + 1];
#pragma unroll
        for (int st = 0; st < B2_DEPTH; ++st) af[st] = rd_tr(0, st / DT, st % DT);
        static_for<NST>([&](auto st_t) {
          constexpr int st = decltype(st_t)::value, tk = st / DT, dt = st % DT, nx = st + B2_DEPTH;
          if (nx < NST) af[nx % (B2_DEPTH + 1)] = rd_tr(0, nx / DT, nx % DT);
          acc_mfma<4 * OS + dt>(af[st % (B2_DEPTH + 1)], opnd(pS[OS], tk));
          if constexpr (OS == 0) {
            if (st == 0) hazard_pad(dp[1]);
            if constexpr (!DQC) {
#pragma unroll
              for (int e = 0; e < 4; ++e) dp[1][4 * st + e] -= odl[1];
            }
            ds_grp(F{}, IC<1>{}, st_t);
          }
          B2_SB();
        });
      });
      return;
    }
    // ---- dP = Ys Y^T   (Ys = V rows, Y = dO image)
    {
      bf16x8 af[B2_DEPTH + 1], b0[B2_DEPTH + 1], b1[B2_DEPTH + 1];
#pragma unroll
      for (int st = 0; st < B2_DEPTH; ++st) { af[st] = rd_ys(st); b0[st] = rd_y(0, st); b1[st] = rd_y(1, st); }
      static_for<KS>([&](auto ks_t) {
        constexpr int ks = decltype(ks_t)::value, nx = ks + B2_DEPTH, sl = nx % (B2_DEPTH + 1), cu = ks % (B2_DEPTH + 1);
        if (nx < KS) { af[sl] = rd_ys(nx); b0[sl] = rd_y(0, nx); b1[sl] = rd_y(1, nx); }
        if constexpr (DQC && ks == 0) { sd_mfma_c(dp[0], af[cu], b0[cu], ndl[0]); sd_mfma_c(dp[1], af[cu], b1[cu], ndl[1]); }
        else { sd_mfma<ks == 0>(dp[0], af[cu], b0[cu]); sd_mfma<ks == 0>(dp[1], af[cu], b1[cu]); }
        B2_SB();
      });
    }
    // ---- S strip 0  ||  dP -= delta ; S strip 1  ||  strip 0: exp2, dS, pack    (K row fragments are read twice)
    {
      bf16x8 af[B2_DEPTH + 1];
#pragma unroll
      for (int st = 0; st < B2_DEPTH; ++st) af[st] = rd_xs(st);
      static_for<KS>([&](auto ks_t) {
        constexpr int ks = decltype(ks_t)::value, PER = 32 / KS;      // dP values per MFMA: 4 (hd 128) | 8 (hd 64)
        if (ks + B2_DEPTH < KS) af[(ks + B2_DEPTH) % (B2_DEPTH + 1)] = rd_xs(ks + B2_DEPTH);
        const bf16x8 a = af[ks % (B2_DEPTH + 1)];
        sd_mfma<ks == 0>(s[0], a, xf[0][ks]);
        if constexpr (!DQC) {
          if (ks == 0) hazard_pad(dp[0], dp[1]);
#pragma unroll
          for (int e = 0; e < PER; ++e) {
            const int id = ks * PER + e, os = id >> 4, r = id & 15;
            dp[os][r] -= odl[os];
          }
          B2_PIN(dp[(ks * PER) >> 4]);
        }
        B2_SB();
      });
    }
    {
      bf16x8 af[B2_DEPTH + 1];
#pragma unroll
      for (int st = 0; st < B2_DEPTH; ++st) af[st] = rd_xs(st);
      static_for<KS>([&](auto ks_t) {
        constexpr int ks = decltype(ks_t)::value;
        if (ks + B2_DEPTH < KS) af[(ks + B2_DEPTH) % (B2_DEPTH + 1)] = rd_xs(ks + B2_DEPTH);
        const bf16x8 a = af[ks % (B2_DEPTH + 1)];
        sd_mfma<ks == 0>(s[1], a, xf[1][ks]);
        if (ks == 0) hazard_pad(s[0]);
        grp_slots(masked_t, IC<0>{}, ks_t, up0);
        B2_SB();
      });
    }
    // ---- dQ^T strip 0 += K^T dS  ||  strip 1: exp2, dS, pack ; then strip 1
    {
      bf16x8 af[B2_DEPTH + 1];
#pragma unroll
      for (int st = 0; st < B2_DEPTH; ++st) af[st] = rd_tr(0, st / DT, st % DT);
      static_for<NST>([&](auto st_t) {
        constexpr int st = decltype(st_t)::value, tk = st / DT, dt = st % DT, nx = st + B2_DEPTH;
        if (nx < NST) af[nx % (B2_DEPTH + 1)] = rd_tr(0, nx / DT, nx % DT);
        const bf16x8 a = af[st % (B2_DEPTH + 1)];
        acc_mfma<dt>(a, opnd(pS[0], tk));
        if (st == 0) hazard_pad(s[1]);
        grp_slots(masked_t, IC<1>{}, st_t, up1);
        B2_SB();
      });
    }
    {
      bf16x8 af[B2_DEPTH + 1];
#pragma unroll
      for (int st = 0; st < B2_DEPTH; ++st) af[st] = rd_tr(0, st / DT, st % DT);
      static_for<NST>([&](auto st_t) {
        constexpr int st = decltype(st_t)::value, tk = st / DT, dt = st % DT, nx = st + B2_DEPTH;
        if (nx < NST) af[nx % (B2_DEPTH + 1)] = rd_tr(0, nx / DT, nx % DT);
        const bf16x8 a = af[st % (B2_DEPTH + 1)];
        acc_mfma<4 + dt>(a, opnd(pS[1], tk));
        B2_SB();
      });
    }
  };

  // ---- prologue: tile 0 in LDS, tile 1 in flight towards the staging registers
  if (total > 0) { gload(0); lwrite(0); }
  if (total > 1 && !(B2_ABL & 2)) gload(1);
  __syncthreads();

  // Staging (one register set): the registers hold tile it+1, loaded during iteration it-1.  They are written to LDS right
  // AFTER the barrier that ended iteration it-1 (its readers of that buffer are done), the loads of tile it+2 are re-issued
  // into the same registers at once, and both overlap the MFMAs of tile it; nothing but the barrier follows the last MFMA.
  int hh = 0, j = first;                                        // (head of the group, tile) of iteration `it`
  for (int it = 0; it < total; ++it) {
    const int t0 = j * 32;
    if (!(B2_ABL & 2)) {
      if (it + 1 < total) lwrite((it + 1) & 1);
      if (it + 2 < total) gload(it + 2);
    }
    bool skip, masked;
    int th0, th1;
    if constexpr (DKV) {
      // valid(streamed query row, owner key): key < len and (non-causal or row >= key); rows past S carry zero dO / delta
      skip = CAUSAL && t0 + 31 < ow0;
      masked = (CAUSAL && t0 < ow0 + 63) || ow0 + 64 > len;
      const int k0 = ow0 + l31, k1 = k0 + 32;
      th0 = (k0 >= len) ? 1000 : (CAUSAL ? k0 - t0 - 4 * hi : -1000);      // row index inside the tile must be >= th
      th1 = (k1 >= len) ? 1000 : (CAUSAL ? k1 - t0 - 4 * hi : -1000);
    } else {
      // valid(streamed key row, owner query): key < len and (non-causal or key <= query)
      skip = (CAUSAL && t0 > ow0 + 63) || ow0 >= S;
      masked = (CAUSAL && t0 + 31 > ow0) || t0 + 32 > len;
      const int q0 = ow0 + l31, q1 = q0 + 32;
      th0 = (CAUSAL ? min(len - 1, q0) : len - 1) - t0 - 4 * hi;             // row index inside the tile must be <= th
      th1 = (CAUSAL ? min(len - 1, q1) : len - 1) - t0 - 4 * hi;
    }
    char* const ds_base = spill ? ds_blk + ((long long)hh * S * S + (long long)j * S * 32) * 2 : nullptr;      // wave-uniform: (head, query tile)
    if (!skip) {
      if constexpr (DKV) { if (masked) body_dkv(BoolTag<true>{}, th0, th1, ds_base); else body_dkv(BoolTag<false>{}, th0, th1, ds_base); }
      else { if (masked) body_dq(BoolTag<true>{}, th0, th1); else body_dq(BoolTag<false>{}, th0, th1); }
    } else if (spill) {
      ds_store_zero(ds_base);           // the dQ GEMM reads the whole diagonal block: what this wave skips is zero there
    }
    // the next tile lives in the other buffer
    const int flip = (it & 1) ? -B2_BUF : B2_BUF;
    rowa[0] += flip; rowa[1] += flip;
#pragma unroll
    for (int hc = 0; hc < 4; ++hc) tra[hc] += flip;
    lda += flip;
    if (++j == nt) { j = first; ++hh; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(B2_ABL & 1)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  // ---- epilogue: lane (owner row l31 of strip os, half hi) holds features dt*32 + hi*16 + r of strip dt
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
  for (int os = 0; os < 2; ++os) {
    const int row = ow0 + os * 32 + l31;
    if (row >= S) continue;
    if constexpr (DKV && !SPILL) {
      if (p.split_ws) {              // head-split launch: this part's fp32 sums, unscaled; attn_dkv_reduce_kernel finishes
        const long long width = (long long)(p.nh / p.group) * HD, plane = p.split_rows * width;
        float* wk = p.split_ws + (long long)(2 * sp) * plane + (tok0 + row) * width + ho * HD + hi * 16;
        float* wv = wk + plane;
        if (os == 0) {
          static_for<DT>([&](auto dt_t) { constexpr int dt = decltype(dt_t)::value; store_strip_f32<dt>(wk + dt * 32); store_strip_f32<8 + dt>(wv + dt * 32); });
        } else {
          static_for<DT>([&](auto dt_t) { constexpr int dt = decltype(dt_t)::value; store_strip_f32<4 + dt>(wk + dt * 32); store_strip_f32<12 + dt>(wv + dt * 32); });
        }
        continue;
      }
    }
    if constexpr (HD == 64) {       // strips dt = 0, 1 of every accumulator group; no fused RoPE at this head dim (attn.hip refuses it)
      bf16_t* dx = (DKV ? p.dK + (tok0 + row) * p.lddk : p.dQ + (tok0 + row) * p.lddq) + ho * 64 + hi * 16;
      if (os == 0) { store_strip<0>(dx, p.scale); store_strip<1>(dx + 32, p.scale); }
      else { store_strip<4>(dx, p.scale); store_strip<5>(dx + 32, p.scale); }
      if constexpr (DKV) {
        bf16_t* dv = p.dV + (tok0 + row) * p.lddv + ho * 64 + hi * 16;
        if (os == 0) { store_strip<8>(dv, 1.f); store_strip<9>(dv + 32, 1.f); }
        else { store_strip<12>(dv, 1.f); store_strip<13>(dv + 32, 1.f); }
      }
    } else if constexpr (DKV) {
      bf16_t* dk = p.dK + (tok0 + row) * p.lddk + ho * 128 + hi * 16;
      bf16_t* dv = p.dV + (tok0 + row) * p.lddv + ho * 128 + hi * 16;
      const bf16_t* cp = nullptr; const bf16_t* sp = nullptr;
      if (p.rope_pos) {
        const long long ro = (long long)p.rope_pos[tok0 + row] * 128 + hi * 16;
        cp = p.rope_cos + ro; sp = p.rope_sin + ro;
      }
      if (os == 0) {
        if (cp) { store_pair_rope<0, 2>(dk, p.scale, cp, sp); store_pair_rope<1, 3>(dk + 32, p.scale, cp + 32, sp + 32); }
        else { store_strip<0>(dk, p.scale); store_strip<1>(dk + 32, p.scale); store_strip<2>(dk + 64, p.scale); store_strip<3>(dk + 96, p.scale); }
        store_strip<8>(dv, 1.f); store_strip<9>(dv + 32, 1.f); store_strip<10>(dv + 64, 1.f); store_strip<11>(dv + 96, 1.f);
      } else {
        if (cp) { store_pair_rope<4, 6>(dk, p.scale, cp, sp); store_pair_rope<5, 7>(dk + 32, p.scale, cp + 32, sp + 32); }
        else { store_strip<4>(dk, p.scale); store_strip<5>(dk + 32, p.scale); store_strip<6>(dk + 64, p.scale); store_strip<7>(dk + 96, p.scale); }
        store_strip<12>(dv, 1.f); store_strip<13>(dv + 32, 1.f); store_strip<14>(dv + 64, 1.f); store_strip<15>(dv + 96, 1.f);
      }
    } else {
      bf16_t* dq = p.dQ + (tok0 + row) * p.lddq + ho * 128 + hi * 16;
      if (p.rope_pos) {
        const long long ro = (long long)p.rope_pos[tok0 + row] * 128 + hi * 16;
        const bf16_t* cp = p.rope_cos + ro; const bf16_t* sp = p.rope_sin + ro;
        if (os == 0) { store_pair_rope<0, 2>(dq, p.scale, cp, sp); store_pair_rope<1, 3>(dq + 32, p.scale, cp + 32, sp + 32); }
        else { store_pair_rope<4, 6>(dq, p.scale, cp, sp); store_pair_rope<5, 7>(dq + 32, p.scale, cp + 32, sp + 32); }
      } else if (os == 0) { store_strip<0>(dq, p.scale); store_strip<1>(dq + 32, p.scale); store_strip<2>(dq + 64, p.scale); store_strip<3>(dq + 96, p.scale); }
      else { store_strip<4>(dq, p.scale); store_strip<5>(dq + 32, p.scale); store_strip<6>(dq + 64, p.scale); store_strip<7>(dq + 96, p.scale); }
    }
  }
}

// Causal work per owner block is linear in its index (dQ: grows, dK/dV: shrinks): every workgroup takes a pair of blocks
// from opposite ends, so all workgroups carry the same number of tiles.
// Grid (heads, owner blocks, batch), head fastest: see attn_fwd2.hip — the workgroups of one head share an XCD's L2.
template <bool DKV, bool CAUSAL, int HD, bool SPILL = false>
__global__ __launch_bounds__(256, 1) void attn_bwd2_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int bx, by, bz;
  xcd_work_id(p.xcd_remap, bx, by, bz);
  int ho = bx, sp = 0;
  if constexpr (DKV && !SPILL) { if (p.nsplit > 1) { ho = bx / p.nsplit; sp = bx - ho * p.nsplit; } }
  if constexpr (CAUSAL) {
    const int nb = (p.S + 255) / 256, x = by;
    const int npass = (2 * x + 1 < nb) ? 2 : 1;
#pragma nounroll
    for (int pass = 0; pass < npass; ++pass) {
      const int big = DKV ? x : nb - 1 - x, small = DKV ? nb - 1 - x : x;
      bwd2_block<DKV, true, HD, SPILL>(p, smem, pass ? small : big, ho, bz, sp);
      __syncthreads();
    }
  } else {
    bwd2_block<DKV, false, HD, SPILL>(p, smem, by, ho, bz, sp);
  }
}

template <bool DKV, bool CAUSAL, int HD, bool SPILL = false>
static void launch_bwd2(const AttnP& p, const dim3 grid, hipStream_t stream) {
  static bool attr = false;          // per instantiation
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_bwd2_kernel<DKV, CAUSAL, HD, SPILL>, hipFuncAttributeMaxDynamicSharedMemorySize, B2_LDS);
    attr = true;
  }
  hipLaunchKernelGGL((attn_bwd2_kernel<DKV, CAUSAL, HD, SPILL>), grid, dim3(256), B2_LDS, stream, p);
}

void lmod_launch_attn_bwd2(const AttnP& p, int causal, hipStream_t stream, int hd) {
  const int nb = (p.S + 255) / 256, nkv = p.nh / p.group, gx = causal ? (nb + 1) / 2 : nb;
  const dim3 gq(p.nh, gx, p.B), gk(nkv * (p.nsplit > 1 ? p.nsplit : 1), gx, p.B);
  if (p.ds_ws && hd == 128) {          // dS spill: dQ comes from lmod_launch_attn_dq_gemm
    if (causal) launch_bwd2<true, true, 128, true>(p, gk, stream);
    else launch_bwd2<true, false, 128, true>(p, gk, stream);
    return;
  }
  if (hd == 64) {
    if (causal) { launch_bwd2<false, true, 64>(p, gq, stream); launch_bwd2<true, true, 64>(p, gk, stream); }
    else { launch_bwd2<false, false, 64>(p, gq, stream); launch_bwd2<true, false, 64>(p, gk, stream); }
  } else {
    if (causal) { launch_bwd2<false, true, 128>(p, gq, stream); launch_bwd2<true, true, 128>(p, gk, stream); }
    else { launch_bwd2<false, false, 128>(p, gq, stream); launch_bwd2<true, false, 128>(p, gk, stream); }
  }
}
