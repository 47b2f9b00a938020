// Probe ds_read_b64_tr_b16 semantics: LDS holds u16 value == its element index; every lane passes its own
// byte address; print what each lane gets.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
__global__ void probe(const int* addr_in, uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  uint32_t a = (uint32_t)(uintptr_t)lds + addr_in[threadIdx.x];
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
  out[threadIdx.x * 2] = r[0]; out[threadIdx.x * 2 + 1] = r[1];
}
int main() {
  int h_addr[64]; uint32_t h_out[128];
  int *d_addr; uint32_t* d_out;
  hipMalloc(&d_addr, 256); hipMalloc(&d_out, 512);
  for (int mode = 0; mode < 3; ++mode) {
    for (int l = 0; l < 64; ++l) {
      if (mode == 0) h_addr[l] = 0;                                   // all lanes same address
      if (mode == 1) h_addr[l] = l * 8;                               // lane-linear 8 bytes
      if (mode == 2) h_addr[l] = ((l & 15) >> 2) * 256 + (l & 3) * 8 + (l >> 4) * 32;   // 4 rows of 128 elements, row = (l&15)>>2
    }
    hipMemcpy(d_addr, h_addr, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, 512, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d addr_el %4d -> %4u %4u %4u %4u\n", l, h_addr[l] / 2, h_out[2*l] & 0xffff, h_out[2*l] >> 16, h_out[2*l+1] & 0xffff, h_out[2*l+1] >> 16);
      if (l == 19 && mode != 2) { l = 47; }
    }
  }
  return 0;
}
