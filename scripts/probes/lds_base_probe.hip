// What does HW_REG_LDS_ALLOC say in the first / second block of a CU?  (conv_fast.h conv_stagger relies on LDS_BASE != 0
// identifying the second co-resident block.)  512 blocks of 256 threads with 68 KB of LDS each: two per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void __launch_bounds__(256) k(unsigned* out, unsigned long long* t) {
  __shared__ float s[17408];
  s[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_getreg((31 << 11) | 6);    // LDS_ALLOC, all 32 bits
    out[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
    out[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
    out[blockIdx.x * 4 + 3] = (unsigned)s[17];
    t[blockIdx.x] = __builtin_readcyclecounter();
  }
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(32);     // keep the block resident (~4 ms)
}
int main() {
  const int nb = 512;
  unsigned* d; unsigned long long* t;
  hipMalloc(&d, nb * 16); hipMalloc(&t, nb * 8);
  hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, t);
  std::vector<unsigned> h(nb * 4);
  hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost);
  std::map<unsigned, int> lo, hi;
  for (int b = 0; b < nb; ++b) (b < nb / 2 ? lo : hi)[h[b * 4] & 0xfffff]++;
  printf("LDS_ALLOC (low 20 bits) of blocks 0..255:\n");
  for (auto& e : lo) printf("  %05x x%d\n", e.first, e.second);
  printf("LDS_ALLOC (low 20 bits) of blocks 256..511:\n");
  for (auto& e : hi) printf("  %05x x%d\n", e.first, e.second);
  // CU identity from HW_ID (cu_id [11:8], sh_id [12], se_id [15:13]) + XCC_ID: do blocks b and b+256 share a CU?
  std::map<unsigned long long, std::vector<int>> cu;
  for (int b = 0; b < nb; ++b) {
    const unsigned hw = h[b * 4 + 1], xcc = h[b * 4 + 2] & 0xf;
    cu[((unsigned long long)xcc << 32) | (hw & 0xff00)].push_back(b);
  }
  int both = 0, same_half = 0;
  for (auto& e : cu) {
    if (e.second.size() == 2) { ++both; if ((e.second[0] < 256) == (e.second[1] < 256)) ++same_half; }
  }
  printf("distinct (xcc, se, sh, cu): %zu; with exactly 2 blocks: %d; of those with both blocks in the same grid half: %d\n",
         cu.size(), both, same_half);
  for (int b = 0; b < 4; ++b) printf("block %d: lds_alloc %08x hw_id %08x xcc %x\n", b, h[b * 4], h[b * 4 + 1], h[b * 4 + 2]);
  for (int b = 256; b < 260; ++b) printf("block %d: lds_alloc %08x hw_id %08x xcc %x\n", b, h[b * 4], h[b * 4 + 1], h[b * 4 + 2]);
  return 0;
}
