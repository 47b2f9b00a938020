// Sustained MFMA throughput from registers only (no LDS / memory in the loop): 16x16x32 vs 32x32x16 bf16, random operands.
// Tells whether the chip's power limit favours one shape (operand-register reads per flop differ 2x).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void spin(const bf16x8* __restrict__ in, float* out, int iters) {
  const int tid = blockIdx.x * 512 + threadIdx.x;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(tid * 8 + i) & 65535]; b[i] = in[(tid * 8 + 4 + i) & 65535]; }
  if (SHAPE == 16) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    out[tid] = s;
  } else {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r)          // same flops per iteration as the 16x16x32 arm: 8 x 32x32x16
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i + 2 * r], b[j + 2 * r], acc[i * 2 + j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    out[tid] = s;
  }
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 3;      // 2 x ~11 ms launches per rep
  bf16x8* in; float* out;
  hipMalloc(&in, 65536 * 16); hipMalloc(&out, 2048 * 512 * 4);
  uint16_t* h = (uint16_t*)malloc(65536 * 16);
  uint32_t x = 12345;
  for (int i = 0; i < 65536 * 8; ++i) { x = x * 1664525u + 1013904223u; float f = ((int)(x >> 8) % 2001 - 1000) / 1000.f; uint32_t u; memcpy(&u, &f, 4); h[i] = u >> 16; }
  hipMemcpy(in, h, 65536 * 16, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, blocks = 512;
  for (int rep = 0; rep < reps; ++rep)
    for (int shape : {16, 32}) {
      if (shape == 16) hipLaunchKernelGGL(spin<16>, dim3(blocks), dim3(512), 0, 0, in, out, 100);
      else hipLaunchKernelGGL(spin<32>, dim3(blocks), dim3(512), 0, 0, in, out, 100);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      if (shape == 16) hipLaunchKernelGGL(spin<16>, dim3(blocks), dim3(512), 0, 0, in, out, iters);
      else hipLaunchKernelGGL(spin<32>, dim3(blocks), dim3(512), 0, 0, in, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double fl = (double)blocks * 8 * iters * 16 * 16384.0;     // per wave per iteration: 16 x (2*16*16*32)
      printf("shape %dx%d: %.2f ms  %.0f TFLOP/s\n", shape, shape, ms, fl / ms / 1e9);
    }
  return 0;
}
