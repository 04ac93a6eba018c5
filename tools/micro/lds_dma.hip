// Microbenchmark / feasibility probe: global -> LDS loads without VGPRs (global_load_lds_dword, gfx950) through the compiler
// builtin; checks the data and how the compiler waits for it (development aid; next round's entry ring of k_rho_sp).
//   hipcc -O3 --offload-arch=gfx950 -o lds_dma tools/micro/lds_dma.hip && ./lds_dma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ __launch_bounds__(256) void k(const unsigned* __restrict__ src, unsigned* __restrict__ dst, int trips) {
  __shared__ unsigned ring[4 * 8 * 64];   // 4 waves x 8 trips x 64 entries
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned* mine = ring + wv * 8 * 64;
  const unsigned* g = src + ((size_t)blockIdx.x * 4 + wv) * trips * 64;
  unsigned acc = 0;
  for (int t0 = 0; t0 < trips; t0 += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j)   // each lane's 4 bytes land at (LDS base) + lane * 4
      __builtin_amdgcn_global_load_lds(g + (size_t)(t0 + j) * 64 + lane, mine + j * 64, 4, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += mine[j * 64 + ((lane + j) & 63)];
    __builtin_amdgcn_wave_barrier();
  }
  dst[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
  const int blocks = 1024, trips = 64; const size_t n = (size_t)blocks * 4 * trips * 64;
  std::vector<unsigned> h(n); for (size_t i = 0; i < n; ++i) h[i] = (unsigned)(i * 2654435761u);
  unsigned *s, *d; hipMalloc(&s, n * 4); hipMalloc(&d, blocks * 256 * 4); hipMemcpy(s, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, s, d, trips); hipDeviceSynchronize();
  std::vector<unsigned> o(blocks * 256); hipMemcpy(o.data(), d, o.size() * 4, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (int b = 0; b < blocks; ++b) for (int t = 0; t < 256; ++t) { int wv = t >> 6, lane = t & 63; unsigned acc = 0;
    for (int tr = 0; tr < trips; ++tr) { int j = tr & 7; acc += h[((size_t)b * 4 + wv) * trips * 64 + (size_t)tr * 64 + ((lane + j) & 63)]; }
    bad += acc != o[b * 256 + t]; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, s, d, trips);
  hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("global_load_lds_dword: %zu mismatches of %d; %.1f GB/s through LDS\n", bad, blocks * 256, 20.0 * n * 4 / ms / 1e6);
  return bad != 0;
}
