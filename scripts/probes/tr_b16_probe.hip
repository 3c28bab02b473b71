#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(uint16_t* out, int stride_bytes) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int lane = threadIdx.x;
  // lane supplies the address of its own 8-byte chunk: chunk index = lane (row-major [16 rows? ...])
  uint32_t addr = (uint32_t)(uintptr_t)(&lds[0]) + lane * stride_bytes;
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)v[j];
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int stride : {8, 32, 64}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("stride %d bytes (lane l points at element %d*l):\n", stride, stride / 2);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  return 0;
}
