// L2 -> CU operand-stream ceiling under the 256x256x64 GEMM's own tile walk (gemm_256_kernel<0>, teacher QKV shape):
// every workgroup streams the A panel (256 rows x K) and the B panel (256 rows x K) of its tile, 64 KiB per K step, with
//   mode 0: buffer_load_dwordx4 ... lds (LDS-DMA, what the GEMM uses), `depth` K-steps' worth kept in flight per wave
//   mode 1: plain buffer_load_dwordx4 into VGPRs (consumed by a dummy OR), same addresses, same depth
// No barriers, no ds_reads, no MFMA: the pure arrival rate of operand bytes at a CU while all 256 CUs do the same.
// Prints GB/s chip-wide, B/clk/CU at the measured kernel time, and the MFMA-equivalent TF (8.39 MFLOP per 64 KiB).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/dma_rate tools/probe/dma_rate.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}
struct P { const uint16_t* A; const uint16_t* B; int M, N, K, tiles_m, tiles_n; uint32_t* sink; int waves; };

template <int MODE, int DEPTH, int SWZ = 1>
__global__ __launch_bounds__(512, 2) void stream_kernel(P p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int GROUP_M = 4;
  const int grp = id / (GROUP_M * p.tiles_n), first_m = grp * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int rr = id - grp * GROUP_M * p.tiles_n;
  const int tm = first_m + rr % gsz, tn = rr / gsz;
  const uint16_t* Ab = p.A + (long long)tm * 256 * p.K;
  const uint16_t* Bb = p.B + (long long)tn * 256 * p.K;
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, 256 * p.K * 2, 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, 256 * p.K * 2, 0x00020000);
  // piece j (0..3) of operand X for this wave: rows wave*32 + j*8 + (lane>>3), 16-byte chunk (lane&7)^(lane>>3)
  // SWZ 1: the GEMM's XOR-swizzled source (16-byte chunks permuted inside each 128-byte row); 0: linear rows (round 4: does the
  // permutation itself cost arrival rate?); 2: chunk pairs permuted only (32-byte granules stay contiguous)
  const int cchunk = SWZ == 1 ? ((lane & 7) ^ (lane >> 3)) : SWZ == 2 ? ((lane & 7) ^ ((lane >> 3) & 6)) : (lane & 7);
  uint32_t vo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) vo[j] = (uint32_t)(((wave * 32 + j * 8 + (lane >> 3)) * p.K + cchunk * 8) * 2);
  const int nkt = p.K / 64;
  u32x4 accv = {0u, 0u, 0u, 0u};
  if (wave >= p.waves) return;
  for (int t = 0; t < nkt; ++t) {
    const int k0 = t * 64;
    if constexpr (MODE == 0) {
      char* dst = smem + (t % 2) * 65536 + wave * 8192;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(dst + j * 1024), 16, vo[j], k0 * 2, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(dst + 4096 + j * 1024), 16, vo[j], k0 * 2, 0, 0);
      }
      if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    } else {
      u32x4 v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[2 * j] = __builtin_amdgcn_raw_buffer_load_b128(rsA, vo[j], k0 * 2, 0);
        v[2 * j + 1] = __builtin_amdgcn_raw_buffer_load_b128(rsB, vo[j], k0 * 2, 0);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) accv |= v[j];      // the compiler places its own counted waits (loads of t+1 hoist over these)
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (MODE == 1 && (accv[0] | accv[1] | accv[2] | accv[3]) == 0x12345u) p.sink[tid] = accv[0];
}

template <int MODE, int DEPTH, int SWZ = 1>
static void run(const P& p, int nwg, const char* tag, double clk_ghz) {
  hipFuncSetAttribute((const void*)stream_kernel<MODE, DEPTH, SWZ>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_kernel<MODE, DEPTH, SWZ>), dim3(nwg), dim3(512), 131072, 0, p);
  hipEventRecord(e0);
  const int n = 40;
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL((stream_kernel<MODE, DEPTH, SWZ>), dim3(nwg), dim3(512), 131072, 0, p);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= n;
  const double bytes = (double)nwg * (p.K / 64) * 65536.0 * p.waves / 8.0;
  const double flops = (double)nwg * (p.K / 64) * 2.0 * 256 * 256 * 64 * p.waves / 8.0;
  printf("%-34s waves %d  %.3f ms  %.2f TB/s  %.1f B/clk/CU @%.2f GHz  = %.0f TF-equivalent\n", tag, p.waves, ms, bytes / ms / 1e9,
         bytes / (ms * 1e-3) / 256 / (clk_ghz * 1e9), clk_ghz, flops / ms / 1e9);
}

int main(int argc, char** argv) {
  const int M = 32768, N = 12288, K = 4096;
  uint16_t *A, *B; uint32_t* sink;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&sink, 4096);
  hipMemset(A, 1, (size_t)M * K * 2); hipMemset(B, 1, (size_t)N * K * 2);
  P p{A, B, M, N, K, M / 256, N / 256, sink, 8};
  const int nwg = p.tiles_m * p.tiles_n;
  const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
  for (int w : {8, 4}) {
    p.waves = w;
    run<0, 1>(p, nwg, "LDS-DMA, 1 K-step in flight", ghz);
    run<0, 2>(p, nwg, "LDS-DMA, 2 K-steps in flight", ghz);
    run<0, 3>(p, nwg, "LDS-DMA, 3 K-steps in flight", ghz);
    run<0, 4>(p, nwg, "LDS-DMA, 4 K-steps in flight", ghz);
    run<1, 0>(p, nwg, "plain loads to VGPRs", ghz);
    run<0, 2, 0>(p, nwg, "LDS-DMA, 2 K-steps, LINEAR source", ghz);
    run<0, 3, 0>(p, nwg, "LDS-DMA, 3 K-steps, LINEAR source", ghz);
    run<0, 3, 2>(p, nwg, "LDS-DMA, 3 K-steps, 32-B granules", ghz);
    run<1, 0, 0>(p, nwg, "plain loads, LINEAR source", ghz);
  }
  return 0;
}
