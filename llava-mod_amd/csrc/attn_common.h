// Shared declarations of the attention kernels (attn.hip: hd 64 + backward; attn_fwd2.hip: hd 128 forward).
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

// hd-128 forward, one wave per SIMD with 64 query rows per wave (attn_fwd3.hip, round 5)
void lmod_launch_attn_fwd3(const AttnP& p, int causal, hipStream_t stream);

// backward (dQ and dK/dV kernels), one wave per SIMD with 256 asm-owned accumulators (attn_bwd2.hip); hd 128 or 64
void lmod_launch_attn_bwd2(const AttnP& p, int causal, hipStream_t stream, int hd = 128);
