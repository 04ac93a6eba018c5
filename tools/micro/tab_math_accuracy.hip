// Accuracy of the rho pass' table-driven exp / log (exp_tab, log_tab of vimure_hip.hip, compiled from that file) against
// the host's libm in long double.  Build: hipcc -O3 --offload-arch=gfx950 -DVMR_DEV -Iinclude tools/micro/tab_math_accuracy.hip
#include "../../vimure_amd/csrc/vimure_hip.hip"
#include <stdio.h>
__global__ void k_tab(const double* x, double* ye, double* yl, int n) {
  __shared__ double xt[64], lt[256];
  sp_math_tables(xt, lt, threadIdx.x, blockDim.x);
  __syncthreads();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { ye[i] = exp_tab(x[i], xt); yl[i] = log_tab(fabs(x[i]) + 1e-300, lt); }
}
int main() {
  const int n = 1 << 21; std::vector<double> hx(n), he(n), hl(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0);
    hx[i] = (u - 0.5) * 1399.0; if (i < 200000) hx[i] = (u - 0.5) * 2.0; if (i >= 200000 && i < 400000) hx[i] = ldexp(0.5 + u, (int)(s % 90) - 45); }
  double *dx, *de, *dl; hipMalloc(&dx, n * 8); hipMalloc(&de, n * 8); hipMalloc(&dl, n * 8); hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_tab, dim3(n / 256), dim3(256), 0, 0, dx, de, dl, n);
  hipMemcpy(he.data(), de, n * 8, hipMemcpyDeviceToHost); hipMemcpy(hl.data(), dl, n * 8, hipMemcpyDeviceToHost);
  double we = 0, wl_abs = 0, wl_rel = 0;
  for (int i = 0; i < n; ++i) {
    if (fabs(hx[i]) < 700.0) { long double r = expl((long double)hx[i]); double e = (double)fabsl(((long double)he[i] - r) / r); if (e > we) we = e; }
    long double a = fabsl((long double)hx[i]) + 1e-300L, rl = logl(a); double ea = (double)fabsl((long double)hl[i] - rl);
    if (ea > wl_abs) wl_abs = ea; if (fabsl(rl) > 0.1L) { double er = (double)(ea / fabsl(rl)); if (er > wl_rel) wl_rel = er; }
  }
  printf("exp_tab: worst relative error %.3e over |x| < 700;  log_tab: worst absolute error %.3e, worst relative %.3e where |log| > 0.1  (%d arguments)\n", we, wl_abs, wl_rel, n);
  return 0;
}
