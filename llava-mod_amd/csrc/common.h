// Shared device helpers for the LLaVA-MoD distillation-step kernels (gfx950 / CDNA4 only).
// wave = 64 lanes everywhere; bf16 is carried as raw uint16_t bit patterns.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LMOD_OK 0
#define LMOD_EINVAL (-1)
#define LMOD_ELAUNCH (-2)
#define LMOD_EUNSUPPORTED (-3)

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;    // one 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

static inline int lmod_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? LMOD_OK : LMOD_ELAUNCH;
}

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// fp32 -> bf16, round-to-nearest-even: gfx950 has v_cvt_pk_bf16_f32; clang emits it for __bf16
// conversions (a hand-rolled integer rounding costs ~8 VALU ops per element and made the attention
// softmax VALU-issue-bound).
typedef __bf16 lmod_bf2 __attribute__((ext_vector_type(2)));
typedef float lmod_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  lmod_f2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, lmod_bf2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float bfround(float f) { return __uint_as_float(pack2bf(f, 0.f) << 16); }
__device__ __forceinline__ float bflo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// sigmoid / SiLU with ONE v_rcp_f32 (<= 1 ulp) instead of the IEEE division sequence (v_div_scale x2, v_rcp, 3 fma, v_div_fmas,
// v_div_fixup = 10 VALU per element): the SwiGLU GEMM epilogues hold 64-128 elements per lane and were VALU-bound on it.  The
// standalone row kernels use the same helpers, so fused and unfused paths stay bit-identical to each other.
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float fast_silu(float x) { return x * fast_sigmoid(x); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum over NW waves; `red` is NW floats of LDS. Every thread gets the result.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) t += red[i];
  return t;
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) t = fmaxf(t, red[i]);
  return t;
}

// XCD-aware, bijective remap of a linear workgroup id: the dispatcher puts block b on XCD b%8,
// so give every XCD one contiguous chunk of the logical tile space (shared operand panels then
// hit the same private L2).  Speed only — never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}
