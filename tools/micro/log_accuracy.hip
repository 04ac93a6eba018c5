#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
__device__ __forceinline__ double log_pos(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  double m = __builtin_amdgcn_frexp_mant(x);
  int k = __builtin_amdgcn_frexp_exp(x);
  const bool lo = m < 0.70710678118654752440;
  m = lo ? m + m : m;
  k = lo ? k - 1 : k;
  const double f = m - 1.0, dk = (double)k;
  const double s = f / (2.0 + f);
  const double z = s * s, w = z * z;
  const double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
  const double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
  const double R = t2 + t1, hfsq = 0.5 * f * f;
  return dk * ln2_hi - ((hfsq - fma(s, hfsq + R, dk * ln2_lo)) - f);
}
__global__ void k(const double* x, double* y, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = log_pos(x[i]); }
int main() {
  const int n = 1 << 20; std::vector<double> hx(n), hy(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); int e = (int)((s >> 3) % 120) - 60;
    hx[i] = ldexp(0.5 + u, e); if (i < 1000) hx[i] = 1.0 + (u - 0.5) * 1e-6; if (i >= 1000 && i < 2000) hx[i] = 1e-12 * (1 + u); }
  double *dx, *dy; hipMalloc(&dx, n * 8); hipMalloc(&dy, n * 8); hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dy, n); hipMemcpy(hy.data(), dy, n * 8, hipMemcpyDeviceToHost);
  double worst = 0, worst_abs = 0; for (int i = 0; i < n; ++i) { double r = log(hx[i]); double e = fabs(hy[i] - r); double rel = e / fmax(fabs(r), 1e-300); if (fabs(r) > 1e-9 && rel > worst) worst = rel; if (e > worst_abs) worst_abs = e; }
  printf("log_pos: worst relative error %.3e (|log| > 1e-9), worst absolute %.3e over %d arguments\n", worst, worst_abs, n); return 0; }
