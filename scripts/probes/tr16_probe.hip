// What does ds_read_b64_tr_b16 return?  LDS holds element index i at position i (16-bit); lane l passes address a(l).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* __restrict__ addr_bytes, short* __restrict__ out) {
  __shared__ short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  typedef __attribute__((address_space(3))) s16x4* lp;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)((char*)lds + addr_bytes[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
  int h[64]; short o[256];
  int* d; short* od;
  hipMalloc(&d, sizeof(h)); hipMalloc(&od, sizeof(o));
  for (int variant = 0; variant < 2; ++variant) {
    // variant 0: lane l -> byte 8*l (contiguous).  variant 1: 16-lane group g reads a [4 rows][16 cols] block of a
    // row-major matrix with 256-byte rows: lane i of the group -> row i/4, cols 4*(i%4); group g at cols 16*(g&1), rows 4*(g>>1)
    for (int l = 0; l < 64; ++l) {
      if (variant == 0) h[l] = 8 * l;
      else { int g = l >> 4, i = l & 15; h[l] = ((4 * (g >> 1) + (i >> 2)) * 128 + 16 * (g & 1) + 4 * (i & 3)) * 2; }
    }
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, od);
    hipMemcpy(o, od, sizeof(o), hipMemcpyDeviceToHost);
    printf("variant %d\n", variant);
    for (int l = 0; l < 64; ++l) printf("lane %2d addr(elem) %4d -> %4d %4d %4d %4d\n", l, h[l] / 2, o[4 * l], o[4 * l + 1], o[4 * l + 2], o[4 * l + 3]);
  }
  return 0;
}
