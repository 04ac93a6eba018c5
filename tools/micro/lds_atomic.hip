// Microbenchmark: LDS atomic throughput on gfx950 by type, active lanes and address pattern (development aid).
//   hipcc -O3 --offload-arch=gfx950 -o lds_atomic tools/micro/lds_atomic.hip && ./lds_atomic
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void kern(double* out, int iters, int active, int pattern, double addend) {
  __shared__ double sh[4096];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += 256) sh[i] = 0.0;
  __syncthreads();
  // address pattern: 0 = lane-contiguous (conflict-free), 1 = pairs share an address, 2 = 4 lanes share, 3 = pseudo-random,
  //                  4 = all lanes one address, 5 = stride 16 B (every other bank pair)
  int idx;
  switch (pattern) {
    case 0: idx = lane; break;
    case 1: idx = lane >> 1; break;
    case 2: idx = lane >> 2; break;
    case 3: idx = (lane * 2654435761u >> 20) & 1023; break;
    case 4: idx = 0; break;
    default: idx = lane * 2; break;
  }
  idx += (tid >> 6) * 1024;   // each wave its own region
  const bool on = lane < active;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (on) {
      if (MODE == 0) atomicAdd(&sh[idx], addend);
      else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned long long*>(&sh[idx]), 1ull);
      else if (MODE == 2) atomicAdd(reinterpret_cast<float*>(&sh[idx]), 1.0f);
      else if (MODE == 3) atomicAdd(reinterpret_cast<unsigned*>(&sh[idx]), 1u);
      else if (MODE == 4) sh[idx] = (double)it;                 // plain 64-bit store
      else if (MODE == 5) { double v = sh[idx]; sh[idx] = v + 1.0; }   // read + write (non-atomic RMW)
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) out[blockIdx.x] = (double)(t1 - t0) / iters;
  if (tid == 1) out[gridDim.x + blockIdx.x] = sh[idx];
}

int main() {
  double* d; hipMalloc(&d, 8 * 4096);
  const char* mn[] = {"ds_add_f64", "ds_add_u64", "ds_add_f32", "ds_add_u32", "ds_write_b64", "read+write b64"};
  const char* pn[] = {"contig", "pairs", "quads", "random", "one addr", "stride16B"};
  const int iters = 2000;
  for (double addend : {1.0, 0.0, 1e-310, 1e-200})
  for (int wgs : {1, 2}) {
    if (addend != 1.0 && wgs != 2) continue;
    printf("addend %g\n", addend);   // workgroups per CU (4 waves each)
    printf("== %d workgroup(s) of 4 waves per CU: cycles per wave-instruction, as seen by one wave (x waves per CU for the LDS cost)\n", wgs);
    for (int mode = 0; mode < (addend == 1.0 ? 6 : 1); ++mode)
      for (int pat = 0; pat < 6; ++pat)
        for (int act : {64, 32, 16, 8}) {
          if (pat != 0 && act != 64) continue;
          void (*k)(double*, int, int, int, double) = mode == 0 ? kern<0> : mode == 1 ? kern<1> : mode == 2 ? kern<2> : mode == 3 ? kern<3> : mode == 4 ? kern<4> : kern<5>;
          hipLaunchKernelGGL(k, dim3(256 * wgs), dim3(256), 0, 0, d, iters, act, pat, addend);
          hipDeviceSynchronize();
          std::vector<double> h(256 * wgs);
          hipMemcpy(h.data(), d, 8 * 256 * wgs, hipMemcpyDeviceToHost);
          double s = 0; for (double v : h) s += v;
          s /= h.size();
          printf("%-16s %-10s active %2d : %7.1f cycles/instr/wave -> %6.1f LDS cycles per instr (CU view)\n", mn[mode], pn[pat], act, s, s / (4.0 * wgs));
        }
  }
  return 0;
}
