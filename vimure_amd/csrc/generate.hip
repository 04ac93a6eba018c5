// generate.hip -- the synthetic generators on the device: the ground truth Y of a stochastic block model and the observed reports X
// given Y (reference synthetic.py:548-571, 639-667 and `_build_X`, synthetic.py:159-231).  The reference draws X in Python loops of
// L * M * N^2 trips (32 s at N = 500, M = 50; BASELINE config 3 is 3.2e9 trips); here one thread draws a (pair, reporter):
//     for every unordered pair {i, j}, reporter m, layer l:   a = lambda[l,i,j] theta[l,m],  b = lambda[l,j,i] theta[l,m]
//     a fair coin picks which direction is drawn first:        first  ~ Poisson((own + eta mirror) / (1 - eta^2))
//                                                              second ~ Poisson(own + eta first)              (synthetic.py:213-231)
// in float64 rates (small rates keep their tail), from a counter-based stream: Philox4x32-10 keyed by the seed, counter = (layer, pair,
// reporter, draw) -- a draw depends on its coordinates only, so the result does not depend on the launch shape.  Output: the uint8
// [L][N][N][M] tensor vmr_create takes (counts clamped to 255), written once, coalesced along the reporters.
// Not the reference's RandomState stream (a GPU cannot follow MT19937 draw by draw): the host classes keep that exact mode
// (vimure_amd/synthetic.py, pinned bit for bit by tests/golden/K_generators.npz); this one is held to it by moments.
#include "vmr_internal.h"

namespace {

struct Rng {   // a stream of uniforms for one (layer, pair, reporter): Philox calls as needed
  unsigned k0, k1, c0, c1, c2, n, have;
  unsigned w[4];
  __device__ Rng(unsigned long long seed, unsigned l, unsigned long long pair, unsigned m)
      : k0((unsigned)seed), k1((unsigned)(seed >> 32)), c0((unsigned)pair), c1((unsigned)(pair >> 32)), c2(m ^ (l << 20)), n(0), have(0) {}
  __device__ double uniform() {   // (0, 1): 53 bits, never 0
    if (have < 2) {
      unsigned c[4] = {c0, c1, c2, n++};
      philox4x32_10(c, k0, k1);
      w[0] = c[0]; w[1] = c[1]; w[2] = c[2]; w[3] = c[3];
      have = 4;
    }
    const unsigned a = w[have - 1], b = w[have - 2];
    have -= 2;
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6) + 0.5) * (1.0 / 9007199254740992.0);
  }
};

// Poisson(rate): inversion by sequential search below 30 (one uniform, about `rate` steps), Hoermann's transformed rejection (PTRS,
// 1993) above -- the algorithm NumPy's legacy generator uses for rate >= 10 -- both exact.
__device__ unsigned poisson_draw(double rate, Rng& g) {
  if (!(rate > 0.0)) return 0u;
  if (rate < 30.0) {
    const double u = g.uniform();
    double p = exp(-rate), cdf = p;
    unsigned k = 0;
    while (u > cdf && k < 1000u) { ++k; p *= rate / (double)k; cdf += p; }
    return k;
  }
  const double slam = sqrt(rate), loglam = log(rate), b = 0.931 + 2.53 * slam, a = -0.059 + 0.02483 * b;
  const double invalpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2.0);
  for (int it = 0; it < 64; ++it) {
    const double U = g.uniform() - 0.5, V = g.uniform(), us = 0.5 - fabs(U);
    const double kf = floor((2.0 * a / us + b) * U + rate + 0.43);
    if (us >= 0.07 && V <= vr) return (unsigned)kf;
    if (kf < 0.0 || (us < 0.013 && V > us)) continue;
    if (log(V) + log(invalpha) - log(a / (us * us) + b) <= -rate + kf * loglam - lgamma(kf + 1.0)) return (unsigned)kf;
  }
  return (unsigned)(rate + 0.5);   // (not reached in practice: acceptance is > 0.9 per trial)
}

// Y[l,i,j] ~ Poisson(w[grp_i][grp_j]) clipped to K - 1, zero diagonal (synthetic.py:548-571, 639-667)
__global__ __launch_bounds__(256) void k_gen_y(uint8_t* __restrict__ Y, const double* __restrict__ w, const int* __restrict__ grp, int L, int N, int C, int K,
                                               unsigned long long seed) {
  const size_t T = (size_t)N * N, n = (size_t)L * T;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (size_t)gridDim.x * 256) {
    const size_t l = q / T, t = q - l * T, i = t / N, j = t - i * N;
    unsigned y = 0;
    if (i != j) {
      Rng g(seed ^ 0x9E3779B97F4A7C15ull, (unsigned)l, (unsigned long long)t, 0xffffu);
      y = poisson_draw(w[(size_t)grp[i] * C + grp[j]], g);
      if (y > (unsigned)(K - 1)) y = (unsigned)(K - 1);
    }
    Y[q] = (uint8_t)y;
  }
}

// lambda of a tie from Y: 0.01 where Y = 0, else Y (or 0.01 + lambda_diff) (synthetic.py:140-157)
__device__ __forceinline__ double lam_of(unsigned y, double lambda_diff) {
  return y == 0u ? 0.01 : (lambda_diff > 0.0 ? 0.01 + lambda_diff : (double)y);
}

// One workgroup per run of pairs, its threads over the reporters: X[l,i,j,:] and X[l,j,i,:] are written as two contiguous rows.
// lamd != null: lambda as doubles [L][N][N] (the classes' lambda_k); else from Y.
__global__ __launch_bounds__(256) void k_gen_x(uint8_t* __restrict__ X, const uint8_t* __restrict__ Y, const double* __restrict__ lamd,
                                               const double* __restrict__ theta, int L, int N, int M, double eta, double lambda_diff,
                                               unsigned long long seed, int self_rep) {
  const unsigned long long T = (unsigned long long)N * N, npair = (unsigned long long)L * T;
  const double inv = 1.0 / (1.0 - eta * eta);
  for (unsigned long long q = blockIdx.x; q < npair; q += gridDim.x) {
    const unsigned long long l = q / T, t = q - l * T, i = t / N, j = t - i * N;
    if (j <= i) continue;   // every unordered pair once
    const size_t tij = (size_t)(l * T + i * N + j), tji = (size_t)(l * T + j * N + i);
    const double la = lamd ? lamd[tij] : lam_of(Y[tij], lambda_diff), lb = lamd ? lamd[tji] : lam_of(Y[tji], lambda_diff);
    uint8_t* xa = X + tij * M;
    uint8_t* xb = X + tji * M;
    for (int m = threadIdx.x; m < M; m += 256) {
      if (self_rep && (unsigned long long)m != i && (unsigned long long)m != j) continue;   // (the tensor was zeroed by the caller)
      const double th = theta[l * M + m], a = la * th, b = lb * th;
      Rng g(seed, (unsigned)l, t, (unsigned)m);
      const bool ij_first = g.uniform() < 0.5;
      unsigned xij, xji;
      if (ij_first) {
        xij = poisson_draw((a + eta * b) * inv, g);
        xji = poisson_draw(b + eta * (double)xij, g);
      } else {
        xji = poisson_draw((b + eta * a) * inv, g);
        xij = poisson_draw(a + eta * (double)xji, g);
      }
      xa[m] = (uint8_t)(xij > 255u ? 255u : xij);
      xb[m] = (uint8_t)(xji > 255u ? 255u : xji);
    }
  }
}

}  // namespace

extern "C" {

// Ground truth of a stochastic block model on the device.  w [C][C] (host): expected ties per ordered pair of groups; grp [N] (host):
// group of every node.  Y_dev: uint8 [L][N][N] in device memory.
int vmr_generate_y(int device, int L, int N, int K, int C, const double* w, const int32_t* grp, uint64_t seed, uint8_t* Y_dev) {
  if (L < 1 || N < 1 || K < 2 || C < 1 || !w || !grp || !Y_dev) return fail(nullptr, VMR_EINVAL, "vmr_generate_y: bad arguments");
  CK(hipSetDevice(device));
  double* wd = nullptr;
  int* gd = nullptr;
  CK(hipMalloc(&wd, (size_t)C * C * 8));
  CK(hipMalloc(&gd, (size_t)N * 4));
  CK(hipMemcpy(wd, w, (size_t)C * C * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(gd, grp, (size_t)N * 4, hipMemcpyHostToDevice));
  const size_t n = (size_t)L * N * N;
  hipLaunchKernelGGL(k_gen_y, dim3((unsigned)std::min<size_t>(65535, (n + 255) / 256)), dim3(256), 0, 0, Y_dev, wd, gd, L, N, C, K, (unsigned long long)seed);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  (void)hipFree(wd); (void)hipFree(gd);
  if (e != hipSuccess) { g_create_err = std::string("vmr_generate_y: ") + hipGetErrorString(e); (void)hipGetLastError(); return VMR_EHIP; }
  return VMR_OK;
}

// The reports X given the ground truth.  Y_dev (uint8 [L][N][N], device) with lambda = 0.01 | Y | 0.01 + lambda_diff (lambda_diff <= 0:
// lambda = Y), or lam_dev (double [L][N][N], device; Y_dev then ignored).  theta [L][M] (host).  self_reporter: only the reports of
// a tie's own two nodes are drawn (M == N); X_dev (uint8 [L][N][N][M], device) must then be zeroed by the caller.
int vmr_generate_x(int device, int L, int N, int M, const uint8_t* Y_dev, const double* lam_dev, const double* theta, double eta,
                   double lambda_diff, uint64_t seed, int self_reporter, uint8_t* X_dev) {
  if (L < 1 || N < 1 || M < 1 || (!Y_dev && !lam_dev) || !theta || !X_dev) return fail(nullptr, VMR_EINVAL, "vmr_generate_x: bad arguments");
  if (!(eta >= 0.0 && eta < 1.0)) return fail(nullptr, VMR_EINVAL, "The mutuality parameter has to be in [0, 1)!");
  if (self_reporter && M != N) return fail(nullptr, VMR_EINVAL, "vmr_generate_x: the self-reporter mask needs M == N");
  CK(hipSetDevice(device));
  double* td = nullptr;
  CK(hipMalloc(&td, (size_t)L * M * 8));
  CK(hipMemcpy(td, theta, (size_t)L * M * 8, hipMemcpyHostToDevice));
  if (!self_reporter) {   // the diagonal ties (i == i) hold no report
    for (int l = 0; l < L; ++l)
      CK(hipMemset2D(X_dev + (size_t)l * N * N * M, (size_t)(N + 1) * M, 0, (size_t)M, (size_t)N));
  }
  const unsigned long long npair = (unsigned long long)L * N * N;
  hipLaunchKernelGGL(k_gen_x, dim3((unsigned)std::min<unsigned long long>(1u << 20, npair)), dim3(256), 0, 0, X_dev, Y_dev, lam_dev, td, L, N, M, eta,
                     lambda_diff, (unsigned long long)seed, self_reporter ? 1 : 0);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  (void)hipFree(td);
  if (e != hipSuccess) { g_create_err = std::string("vmr_generate_x: ") + hipGetErrorString(e); (void)hipGetLastError(); return VMR_EHIP; }
  return VMR_OK;
}

}  // extern "C"
