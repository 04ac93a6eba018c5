// How accurate is v_rcp_f64 (+ n Newton steps) on gfx950?  Prints max relative error vs IEEE 1/x.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k(const double* x, double* e0, double* e1, double* e2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d = x[i], ex = 1.0 / d;
  double r = __builtin_amdgcn_rcp(d);
  e0[i] = fabs(r - ex) / ex;
  r = fma(fma(-d, r, 1.0), r, r);
  e1[i] = fabs(r - ex) / ex;
  r = fma(fma(-d, r, 1.0), r, r);
  e2[i] = fabs(r - ex) / ex;
}
int main() {
  const int n = 1 << 22;
  double *x, *e0, *e1, *e2;
  hipMallocManaged(&x, n * 8); hipMallocManaged(&e0, n * 8); hipMallocManaged(&e1, n * 8); hipMallocManaged(&e2, n * 8);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); x[i] = exp((u - 0.5) * 80.0) * (1.0 + u); }
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, e0, e1, e2, n);
  hipDeviceSynchronize();
  double m0 = 0, m1 = 0, m2 = 0;
  for (int i = 0; i < n; ++i) { m0 = fmax(m0, e0[i]); m1 = fmax(m1, e1[i]); m2 = fmax(m2, e2[i]); }
  printf("max rel err: rcp %.3e  +1 NR %.3e  +2 NR %.3e\n", m0, m1, m2);
  return 0;
}
