// Probe: LDS atomic throughput on gfx950 — ds_add_f32 vs ds_add_u32 vs ds_add_u64 vs plain RMW.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ void __launch_bounds__(1024) k(const int* __restrict__ idx, int n, float* out) {
  extern __shared__ float slab[];
  for (int i = threadIdx.x; i < 32768; i += 1024) slab[i] = 0.f;
  __syncthreads();
  for (int it = 0; it < 16; ++it)
    for (int i = threadIdx.x; i < n; i += 1024) {
      const int a = idx[i] & 16383;
      if (MODE == 0) atomicAdd(&slab[a], 1.5f);
      else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(slab) + a, 3u);
      else if (MODE == 2) atomicAdd(reinterpret_cast<unsigned long long*>(slab) + a, 3ull);
      else slab[a] += 1.5f;
    }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = slab[5];
}
int main() {
  const int n = 65536;
  int* h = (int*)malloc(n * 4);
  for (int i = 0; i < n; ++i) h[i] = (i * 2654435761u) >> 7;
  int* d; float* o;
  hipMalloc(&d, n * 4); hipMalloc(&o, 4096);
  hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[4] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "plain rmw"};
  for (int m = 0; m < 4; ++m) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (m == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 131072, 0, d, n, o);
      if (m == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 131072, 0, d, n, o);
      if (m == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 131072, 0, d, n, o);
      if (m == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(1024), 131072, 0, d, n, o);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%-12s %8.1f us  -> %.2f lane-ops/clk/CU (2.4GHz)\n", names[m], ms * 1e3, 16.0 * n / (ms * 1e-3 * 2.4e9));
    }
  }
  return 0;
}
