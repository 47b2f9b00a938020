// Shared declarations of the attention kernels (attn.hip: entry points and dispatch; attn_fwd2.hip / attn_bwd2.hip: the kernels).
#pragma once
#include "common.h"

#ifndef NWAVE
#define NWAVE 8
#endif
#define NTHR (NWAVE * 64)
#ifndef ATT_SCHED
#define ATT_SCHED 1
#endif
#ifndef ATT_PRIO
#define ATT_PRIO 1
#endif

struct AttnP {
  const bf16_t* Q; const bf16_t* K; const bf16_t* V; bf16_t* O; float* LSE;
  const bf16_t* dO; const float* Delta; bf16_t* dQ; bf16_t* dK; bf16_t* dV;
  const int* seqlens;
  const int* cu;                 // varlen: token offsets [B+1] of the PACKED samples (S = longest sample), or NULL (b*S)
  int B, S, nh, group;           // group = nh / nkv
  int ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  float scale;
  // lmod_attn_bwd_rope: the rotary embedding's gradient map applied to dQ / dK in the backward kernels' epilogues (hd 128)
  const bf16_t* rope_cos; const bf16_t* rope_sin; const int* rope_pos;
  // lmod_attn_bwd_split (attn_bwd2.hip): the dK/dV kernel's query-head group cut into nsplit parts, one workgroup each, that store
  // fp32 partial sums into split_ws [nsplit][dK | dV][split_rows][nkv * hd]; a reduction kernel adds them in split order.
  float* split_ws; int nsplit; long long split_rows;
  // dS spill (round 6; attn_bwd2.hip dK/dV kernel + gemm.hip lmod_launch_attn_dq_gemm): the dK/dV kernel stores dS^T — the bf16 values its
  // dK MFMAs consume — as [B * nh][S keys][S queries]; dQ = scale * dS K is then ONE batched TN GEMM and the dQ kernel (which
  // recomputes S, dP and the exponentials: 3 of the backward's 7 matmuls) is not launched.  NULL: the two-kernel form.
  bf16_t* ds_ws;
  int xcd_remap;                 // see xcd_work_id
};

// XCD-aware work mapping for grids whose x extent (heads) is not a multiple of the 8 XCDs.  The dispatcher deals consecutive workgroup ids
// round-robin to the XCDs; with a multiple of 8 heads as the fastest index every workgroup of a head lands on one XCD and the blocks of a
// (batch, head) share that XCD's L2 (attn_fwd2.hip).  With 14 heads (Qwen2-0.5B) they scatter over all eight.  Here hardware id
// -> (id % 8) * chunk + id / 8 hands every XCD a CONTIGUOUS range of the logical order (head fastest, then block, then batch), i.e.
// whole batch items: their K / V (shared by all query heads of a KV group) are fetched into ONE private L2.
__device__ __forceinline__ void xcd_work_id(const int on, int& bx, int& by, int& bz) {
  bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
  const int nx = gridDim.x, ny = gridDim.y;
  if (!on || !(nx & 7)) return;
  const int total = nx * ny * (int)gridDim.z, lin = bx + nx * (by + ny * bz);
  const int xcd = lin & 7, idx = lin >> 3, base = total >> 3, rem = total & 7;
  const int l = xcd * base + min(xcd, rem) + idx, t = l / nx;
  bx = l - t * nx; bz = t / ny; by = t - bz * ny;
}

typedef __attribute__((ext_vector_type(4))) short s16x4;
template <bool V> struct BoolTag { static constexpr bool value = V; };


typedef __attribute__((ext_vector_type(16))) float f32x16;

// hd-128 forward, 32x32x16-MFMA software-pipelined kernel (attn_fwd2.hip)
void lmod_launch_attn_fwd2(const AttnP& p, int causal, hipStream_t stream, int hd = 128);

// LAB arms (`make LAB=1`, -DLMOD_LAB=1): hd-128 forward, one wave per SIMD with 64 query rows per wave (attn_fwd3.hip, round 5), and
// the round-1 generic 16x16x32 kernels for hd 64 / 128 (attn_lab.hip)
void lmod_launch_attn_fwd3(const AttnP& p, int causal, hipStream_t stream);
void lmod_launch_attn_fwd_generic(const AttnP& p, int causal, hipStream_t stream, int hd);
void lmod_launch_attn_bwd_generic(const AttnP& p, int causal, hipStream_t stream, int hd);

// backward (dQ and dK/dV kernels), one wave per SIMD with 256 asm-owned accumulators (attn_bwd2.hip); hd 128 or 64.
// With p.ds_ws set (hd 128, dense layout, S % 256 == 0, unsplit) only the dK/dV kernel runs and spills dS^T; the caller follows with
// lmod_launch_attn_dq_gemm (gemm.hip: gemm4t_kernel's K loop, bf16 + scale + RoPE-backward epilogue).
void lmod_launch_attn_bwd2(const AttnP& p, int causal, hipStream_t stream, int hd = 128);
void lmod_launch_attn_dq_gemm_raw(const void* ds_ws, const void* K, void* dQ, const int* seqlens, int B, int S, int nh, int group, int ldk,
                                  int lddq, float scale, int causal, const void* rope_cos, const void* rope_sin, const int* rope_pos,
                                  hipStream_t stream);                  // gemm.hip
inline void lmod_launch_attn_dq_gemm(const AttnP& p, int causal, hipStream_t stream) {
  lmod_launch_attn_dq_gemm_raw(p.ds_ws, p.K, p.dQ, p.seqlens, p.B, p.S, p.nh, p.group, p.ldk, p.lddq, p.scale, causal, p.rope_cos, p.rope_sin,
                               p.rope_pos, stream);
}
