// libvimure_hip.so -- MI355X (gfx950 / CDNA4) CAVI engine for the VIMuRe model.
//
// Implements the reference's hot path (latentnetworks/vimure, src/python/vimure/model.py:
// _update_CAVI :623-660, _update_cache :662-696, _update_gamma :698-727, _update_phi :729-761,
// _update_rho :763-818, _update_nu :820-830, __ELBO :948-1019) as FP64 streaming passes over a
// dense uint8 report tensor X[L,N,N,Mp] and a bit-packed reporter mask R, behind the C-ABI of
// include/vimure_hip.h.  Design notes, data layout and byte accounting: DESIGN.md.
//
// Kernel families
//   k_gamma_mask   lane <-> reporter m, tie wave-uniform: R-row words arrive by scalar loads and
//                  become the EXEC mask of K v_add_f64 -- A[l,m,k] = sum_ij R[l,i,j,m] rho[l,i,j,k].
//   k_gamma_counts / k_phi / k_rho   "tile-pair" sweeps over X: a workgroup stages the (I,J) tile
//                  of ties and its mirror (J,I) in LDS so that X[l,j,i,m] (the reference's
//                  data_T_vals) is an LDS byte read; S lanes own one tie row, skip zero 16-B
//                  chunks and run the per-report arithmetic only on non-zero counts.
//   k_fin_*        single-workgroup reductions of the per-workgroup partials + the Gamma
//                  expectations (digamma/log/exp) -- no host round trip inside a sweep.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <string>
#include <vector>

#include "vimure_hip.h"

#define VMR_VERSION "vimure_hip 0.1 (gfx950)"
#define TPB 256
#define KMAX 8

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
struct Geo {
  int L, N, M, K, mut;
  int Mp;       // row bytes of X on device (M rounded up to 16)
  int nchunk;   // Mp / 16
  int stride;   // LDS row stride in bytes (odd multiple of 16)
  int W;        // 64-bit words per R row
  int b, lb;    // tile edge (ties), log2
  int nb;       // tiles per side
  int nt;       // tie slots per tile pair = 2 b^2
  int S, lS;    // lanes per tie
  long long P;  // tile pairs per layer
  int Gl;       // workgroups per layer for tile-pair kernels
  int Gm;       // workgroups per layer for the mask kernel
  double eps;
};

struct vmr_ctx {
  Geo g;
  int device;
  hipStream_t stream;
  std::string err;
  // data
  uint8_t* X = nullptr;        // [L][N*N][Mp]
  uint64_t* Rb = nullptr;      // [L][N*N][W]
  uint8_t* cov = nullptr;      // [L][N*N]
  unsigned long long* sumx = nullptr;
  // state
  double *rho = nullptr, *logpr = nullptr;
  double* par = nullptr;       // parameter block, see P_* offsets
  size_t par_doubles = 0;
  // partials
  double *partS1 = nullptr, *partA = nullptr, *partP = nullptr, *partR = nullptr, *Atot = nullptr;
  double* elbo_dev = nullptr;  // [0] elbo
  bool have_priors = false, have_state = false;
  // profiling
  bool prof = false;
  struct Ev { int cls; hipEvent_t a, b; };
  std::vector<Ev> evs;
  double prof_ms[VMR_KERNEL_COUNT];
  int64_t prof_n[VMR_KERNEL_COUNT];
};

static std::string g_create_err;

// parameter block layout (doubles); LM = L*Mp, LK = L*K
struct ParOff {
  size_t a_th, b_th, g_shp, g_rte, E_th, G_th, l_th;   // each L*Mp
  size_t a_la, b_la, p_shp, p_rte, p_rte_pend, E_la, G_la, l_la;  // each L*K
  size_t sc;   // scalars: see SC_*
  size_t total;
};
enum { SC_A_ETA = 0, SC_B_ETA, SC_NU_SHP, SC_NU_RTE, SC_G_NU, SC_G_NU_STALE, SC_E_NU, SC_COUNT = 8 };

__host__ __device__ static inline ParOff par_off(int L, int Mp, int K) {
  ParOff o;
  size_t LM = (size_t)L * Mp, LK = (size_t)L * K, p = 0;
  o.a_th = p; p += LM; o.b_th = p; p += LM; o.g_shp = p; p += LM; o.g_rte = p; p += LM;
  o.E_th = p; p += LM; o.G_th = p; p += LM; o.l_th = p; p += LM;
  o.a_la = p; p += LK; o.b_la = p; p += LK; o.p_shp = p; p += LK; o.p_rte = p; p += LK;
  o.p_rte_pend = p; p += LK; o.E_la = p; p += LK; o.G_la = p; p += LK; o.l_la = p; p += LK;
  o.sc = p; p += SC_COUNT;
  o.total = p;
  return o;
}

#define HIPCHK(h, call)                                                              \
  do {                                                                               \
    hipError_t e_ = (call);                                                          \
    if (e_ != hipSuccess) {                                                          \
      char buf_[512];                                                                \
      snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      if (h) (h)->err = buf_; else g_create_err = buf_;                              \
      return VMR_EHIP;                                                               \
    }                                                                                \
  } while (0)

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double digamma_pos(double x) {
  // psi(x), x > 0: upward recurrence to x >= 10, then the asymptotic series
  // (same construction as cephes/scipy.special.psi, which the reference calls at model.py:676).
  double r = 0.0;
  while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
  double f = 1.0 / (x * x);
  double t = f * (-1.0 / 12.0 + f * (1.0 / 120.0 + f * (-1.0 / 252.0 + f * (1.0 / 240.0 +
             f * (-1.0 / 132.0 + f * (691.0 / 32760.0 + f * (-1.0 / 12.0)))))));
  return r + log(x) - 0.5 / x + t;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum over the S (power of two, group-aligned) lanes that share a tie; every lane gets the sum
__device__ __forceinline__ double group_sum(double v, int S) {
  for (int o = S >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ unsigned group_sum_u(unsigned v, int S) {
  for (int o = S >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum (TPB threads); result valid in thread 0. `red` = >= 4 doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    for (int w = 0; w < TPB / 64; ++w) r += red[w];
  }
  return r;
}

// 16-bit mask of the non-zero bytes of a 16-byte chunk
__device__ __forceinline__ unsigned nz_mask4(unsigned d) {
  unsigned t = (((d & 0x7f7f7f7fu) + 0x7f7f7f7fu) | d) & 0x80808080u;
  return (((t >> 7) * 0x00204081u) >> 21) & 0xfu;
}
__device__ __forceinline__ unsigned nz_mask16(uint4 v) {
  return nz_mask4(v.x) | (nz_mask4(v.y) << 4) | (nz_mask4(v.z) << 8) | (nz_mask4(v.w) << 12);
}

// Visit the non-zero counts of the chunks {s, s+S, ...} of one LDS row: f(m, x).
// Every lane advances through its own non-zeros; zero chunks cost one 16-B LDS read.
template <class F>
__device__ __forceinline__ void scan_row(const unsigned char* row, int s, int S, int nchunk, F&& f) {
  int c = s, cc = 0;
  unsigned nzm = 0;
  for (;;) {
    while (nzm == 0 && c < nchunk) {
      uint4 v = *reinterpret_cast<const uint4*>(row + c * 16);
      nzm = nz_mask16(v);
      cc = c;
      c += S;
    }
    if (nzm == 0) break;
    int i = __builtin_ctz(nzm);
    nzm &= nzm - 1;
    int m = cc * 16 + i;
    f(m, (unsigned)row[m]);
  }
}

// weight of the theta*lambda part of a report (model.py:685-693): z1 / (z1 + z2), 0-safe
__device__ __forceinline__ double w1_of(double z1, double z2) {
  double den = z1 + z2;
  den = (den == 0.0) ? 1.0 : den;
  return z1 / den;
}

struct TileCtx {
  int N, b, lb, bb;
  int I0, J0;
  bool diag;
};

// tie slot tau of the pair (I,J) -> (i,j); false when outside the network
__device__ __forceinline__ bool tie_coords(const TileCtx& t, int tau, int& i, int& j) {
  bool second = tau >= t.bb;
  int u = second ? tau - t.bb : tau;
  int p = u >> t.lb, q = u & (t.b - 1);
  i = (second ? t.J0 : t.I0) + p;
  j = (second ? t.I0 : t.J0) + q;
  return i < t.N && j < t.N;
}
__device__ __forceinline__ int mirror_slot(const TileCtx& t, int tau) {
  bool second = tau >= t.bb;
  int u = second ? tau - t.bb : tau;
  int m = ((u & (t.b - 1)) << t.lb) | (u >> t.lb);
  return t.diag ? m : (second ? m : t.bb + m);
}

// decode pair index p -> (I,J), I <= J, row-major over the upper triangle of an nb x nb grid
__device__ __forceinline__ void pair_decode(long long p, int nb, int& I, int& J) {
  double d = (2.0 * nb + 1.0);
  long long i = (long long)((d - sqrt(d * d - 8.0 * (double)p)) * 0.5);
  if (i < 0) i = 0;
  if (i > nb - 1) i = nb - 1;
  auto off = [&](long long r) { return r * nb - r * (r - 1) / 2; };
  while (i > 0 && off(i) > p) --i;
  while (i < nb - 1 && off(i + 1) <= p) ++i;
  I = (int)i;
  J = (int)(i + (p - off(i)));
}

// cooperative load of the tile pair's X rows into LDS (zero-fill outside the network)
__device__ __forceinline__ void load_tile(const uint8_t* __restrict__ Xl, unsigned char* xt, const Geo& g,
                                          const TileCtx& t, int nt_cur) {
  const int total = nt_cur * g.nchunk;
  int q = threadIdx.x;
  int tau = q / g.nchunk, c = q - tau * g.nchunk;
  const int dt = TPB / g.nchunk, dc = TPB - dt * g.nchunk;
  for (; q < total; q += TPB) {
    int i, j;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (tie_coords(t, tau, i, j))
      v = *reinterpret_cast<const uint4*>(Xl + ((size_t)i * g.N + j) * g.Mp + c * 16);
    *reinterpret_cast<uint4*>(xt + tau * g.stride + c * 16) = v;
    tau += dt; c += dc;
    if (c >= g.nchunk) { c -= g.nchunk; ++tau; }
  }
}

// ------------------------------------------------------------------------------------------
// set-up kernels
// ------------------------------------------------------------------------------------------
__global__ void k_pack_x(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t rows, int M, int Mp) {
  // one 16-byte output chunk per thread
  size_t nchunk = Mp / 16, total = rows * nchunk;
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < total; q += (size_t)gridDim.x * blockDim.x) {
    size_t r = q / nchunk; int c = (int)(q - r * nchunk);
    unsigned char b[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { int m = c * 16 + u; b[u] = (m < M) ? src[r * M + m] : 0; }
    *reinterpret_cast<uint4*>(dst + r * Mp + c * 16) = *reinterpret_cast<uint4*>(b);
  }
}

__global__ void k_pack_r(const uint8_t* __restrict__ src, uint64_t* __restrict__ dst, size_t rows, int M, int W) {
  size_t total = rows * W;
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < total; q += (size_t)gridDim.x * blockDim.x) {
    size_t r = q / W; int w = (int)(q - r * W);
    uint64_t bits = 0;
    for (int u = 0; u < 64; ++u) {
      int m = w * 64 + u;
      bool on = (m < M) && (src == nullptr || src[r * M + m] != 0);
      bits |= (uint64_t)on << u;
    }
    dst[q] = bits;
  }
}

// coverage flag per tie + sum(X)
__global__ void k_stats(const uint8_t* __restrict__ X, const uint64_t* __restrict__ Rb, uint8_t* __restrict__ cov,
                        unsigned long long* sumx, size_t rows, int Mp, int W) {
  unsigned long long local = 0;
  for (size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x; r < rows; r += (size_t)gridDim.x * blockDim.x) {
    bool anyx = false, anyr = false;
    const uint4* p = reinterpret_cast<const uint4*>(X + r * Mp);
    for (int c = 0; c < Mp / 16; ++c) {
      uint4 v = p[c];
      if (v.x | v.y | v.z | v.w) {
        anyx = true;
        unsigned d[4] = {v.x, v.y, v.z, v.w};
        for (int u = 0; u < 4; ++u)
          local += (d[u] & 0xff) + ((d[u] >> 8) & 0xff) + ((d[u] >> 16) & 0xff) + (d[u] >> 24);
      }
    }
    for (int w = 0; w < W; ++w) anyr |= Rb[r * W + w] != 0;
    cov[r] = (anyx && anyr) ? 1 : 0;
  }
  if (local) atomicAdd(sumx, local);
}

__global__ void k_init_rho(const double* __restrict__ pr, double* __restrict__ rho, double* __restrict__ logpr,
                           size_t n, double eps) {
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
    double v = pr[q];
    rho[q] = v;
    logpr[q] = log(v + eps);
  }
}

// derived expectations of all Gamma factors from shp/rte (used after vmr_set_state)
__global__ void k_derive_all(double* par, Geo g) {
  ParOff o = par_off(g.L, g.Mp, g.K);
  int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
  for (int q = tid; q < g.L * g.Mp; q += nth) {
    int m = q % g.Mp;
    if (m < g.M) {
      double s = par[o.g_shp + q], r = par[o.g_rte + q];
      double l = digamma_pos(s) - log(r);
      par[o.E_th + q] = s / r; par[o.l_th + q] = l; par[o.G_th + q] = exp(l);
    } else {
      par[o.E_th + q] = 0.0; par[o.l_th + q] = 0.0; par[o.G_th + q] = 0.0;
    }
  }
  for (int q = tid; q < g.L * g.K; q += nth) {
    double s = par[o.p_shp + q], r = par[o.p_rte + q];
    double l = digamma_pos(s) - log(r);
    par[o.E_la + q] = s / r; par[o.l_la + q] = l; par[o.G_la + q] = exp(l);
  }
  if (tid == 0) {
    double* sc = par + o.sc;
    double gn = g.mut ? exp(digamma_pos(sc[SC_NU_SHP]) - log(sc[SC_NU_RTE])) : 0.0;  // model.py:596-600
    sc[SC_G_NU] = gn; sc[SC_G_NU_STALE] = gn;
    sc[SC_E_NU] = sc[SC_NU_SHP] / sc[SC_NU_RTE];
  }
}

// ------------------------------------------------------------------------------------------
// gamma, mask half:  A[l,m,k] = sum_{i,j} R[l,i,j,m] rho[l,i,j,k]
// (gives gamma_rte = beta + sum_k E[lambda_k] A  -- model.py:704-718 -- and, with the new
//  E[theta], phi_rte = beta + sum_m E[theta_m] A -- model.py:742-749 -- from ONE pass over R)
// ------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(TPB) void k_gamma_mask(const uint64_t* __restrict__ Rb, const double* __restrict__ rho,
                                                    double* __restrict__ partA, Geo g) {
  __shared__ double sacc[256 * K];
  const int l = blockIdx.x / g.Gm, gb = blockIdx.x - l * g.Gm;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const long long T = (long long)g.N * g.N;
  const long long nw = (long long)g.Gm * (TPB / 64), gw = (long long)gb * (TPB / 64) + wave;
  const long long t0 = gw * T / nw, t1 = (gw + 1) * T / nw;
  const uint64_t* Rl = Rb + (size_t)l * T * g.W;
  const double* rl = rho + (size_t)l * T * K;
  const int Wp = g.W * 64;
  for (int cg = 0; cg < g.W; cg += 4) {
    double acc[4][K];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int k = 0; k < K; ++k) acc[c][k] = 0.0;
    const int nc = min(4, g.W - cg);
#pragma unroll 2
    for (long long t = t0; t < t1; ++t) {
      const uint64_t* rw = Rl + t * g.W + cg;
      const double* rp = rl + t * K;
      double r[K];
#pragma unroll
      for (int k = 0; k < K; ++k) r[k] = rp[k];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c < nc) {
          uint64_t w = rw[c];
          if (__builtin_amdgcn_inverse_ballot_w64(w)) {
#pragma unroll
            for (int k = 0; k < K; ++k) acc[c][k] += r[k];
          }
        }
      }
    }
    // combine the 4 waves and emit this block's partial for reporters [64 cg, 64 cg + 256)
    for (int q = threadIdx.x; q < 256 * K; q += TPB) sacc[q] = 0.0;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int k = 0; k < K; ++k)
        if (c < nc) atomicAdd(&sacc[(c * 64 + lane) * K + k], acc[c][k]);
    __syncthreads();
    double* out = partA + ((size_t)blockIdx.x * Wp + cg * 64) * K;
    for (int q = threadIdx.x; q < nc * 64 * K; q += TPB) out[q] = sacc[q];
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// gamma, counts half (model.py:698-703, 832-859):
//   S1[l,m] = sum_{ij: x>0} x * sum_k rho_k w1_k        (gamma_shp - alpha)
//   mutuality off also P[l,k] = sum x rho_k               (phi_shp - alpha, model.py:861-887)
// ------------------------------------------------------------------------------------------
struct CountArgs {
  const uint8_t* X; const double* rho; const double* par;
  double* partS1; double* partP;
};

template <int K, bool MUT>
__global__ __launch_bounds__(TPB, 4) void k_gamma_counts(CountArgs a, Geo g) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char* xt = smem;
  double* S1 = reinterpret_cast<double*>(smem + (size_t)g.nt * g.stride);
  double* Gth = S1 + g.Mp;
  double* red = Gth + g.Mp;
  const ParOff o = par_off(g.L, g.Mp, g.K);
  const int l = blockIdx.x / g.Gl, gb = blockIdx.x - l * g.Gl;
  const long long p0 = (long long)gb * g.P / g.Gl, p1 = (long long)(gb + 1) * g.P / g.Gl;
  for (int m = threadIdx.x; m < g.Mp; m += TPB) { S1[m] = 0.0; Gth[m] = a.par[o.G_th + (size_t)l * g.Mp + m]; }
  double Gla[K];
#pragma unroll
  for (int k = 0; k < K; ++k) Gla[k] = a.par[o.G_la + l * K + k];
  const double gnu = a.par[o.sc + SC_G_NU];
  double Pk[K];
#pragma unroll
  for (int k = 0; k < K; ++k) Pk[k] = 0.0;

  TileCtx t; t.N = g.N; t.b = g.b; t.lb = g.lb; t.bb = g.b * g.b;
  int I, J;
  pair_decode(p0, g.nb, I, J);
  const int tau = threadIdx.x >> g.lS, s = threadIdx.x & (g.S - 1);
  const uint8_t* Xl = a.X + (size_t)l * g.N * g.N * g.Mp;
  const double* rl = a.rho + (size_t)l * g.N * g.N * K;
  __syncthreads();
  for (long long p = p0; p < p1; ++p) {
    t.I0 = I * g.b; t.J0 = J * g.b; t.diag = (I == J);
    const int nt_cur = t.diag ? t.bb : 2 * t.bb;
    load_tile(Xl, xt, g, t, nt_cur);
    int i, j;
    const bool act = tau < nt_cur && tie_coords(t, tau, i, j);
    double r[K];
    if (act) {
#pragma unroll
      for (int k = 0; k < K; ++k) r[k] = rl[((size_t)i * g.N + j) * K + k];
    }
    __syncthreads();
    if (act) {
      const unsigned char* row = xt + tau * g.stride;
      const unsigned char* mrow = xt + mirror_slot(t, tau) * g.stride;
      scan_row(row, s, g.S, g.nchunk, [&](int m, unsigned x) {
        double dx = (double)x, sum = 0.0;
        if (MUT) {
          unsigned y = mrow[m];
          if (y == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) sum += r[k] * ((Gth[m] * Gla[k] != 0.0) ? 1.0 : 0.0);
          } else {
            double z2 = gnu * (double)y;
#pragma unroll
            for (int k = 0; k < K; ++k) sum += r[k] * w1_of(Gth[m] * Gla[k], z2);
          }
        } else {
#pragma unroll
          for (int k = 0; k < K; ++k) { sum += r[k]; Pk[k] += dx * r[k]; }
        }
        atomicAdd(&S1[m], dx * sum);
      });
    }
    __syncthreads();
    if (++J == g.nb) { ++I; J = I; }
  }
  for (int m = threadIdx.x; m < g.Mp; m += TPB) a.partS1[(size_t)blockIdx.x * g.Mp + m] = S1[m];
  if (!MUT) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      double v = block_sum(Pk[k], red);
      if (threadIdx.x == 0) a.partP[(size_t)blockIdx.x * K + k] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------
// phi, counts (mutuality on; model.py:731-733, 861-887): P[l,k] = sum x rho_k w1_k with the
// NEW E[log theta] -- the cache refresh of model.py:647 sits between the two updates.
// ------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(TPB, 4) void k_phi(CountArgs a, Geo g) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char* xt = smem;
  double* Gth = reinterpret_cast<double*>(smem + (size_t)g.nt * g.stride);
  double* red = Gth + g.Mp;
  const ParOff o = par_off(g.L, g.Mp, g.K);
  const int l = blockIdx.x / g.Gl, gb = blockIdx.x - l * g.Gl;
  const long long p0 = (long long)gb * g.P / g.Gl, p1 = (long long)(gb + 1) * g.P / g.Gl;
  for (int m = threadIdx.x; m < g.Mp; m += TPB) Gth[m] = a.par[o.G_th + (size_t)l * g.Mp + m];
  double Gla[K], Pk[K];
#pragma unroll
  for (int k = 0; k < K; ++k) { Gla[k] = a.par[o.G_la + l * K + k]; Pk[k] = 0.0; }
  const double gnu = a.par[o.sc + SC_G_NU];
  TileCtx t; t.N = g.N; t.b = g.b; t.lb = g.lb; t.bb = g.b * g.b;
  int I, J;
  pair_decode(p0, g.nb, I, J);
  const int tau = threadIdx.x >> g.lS, s = threadIdx.x & (g.S - 1);
  const uint8_t* Xl = a.X + (size_t)l * g.N * g.N * g.Mp;
  const double* rl = a.rho + (size_t)l * g.N * g.N * K;
  __syncthreads();
  for (long long p = p0; p < p1; ++p) {
    t.I0 = I * g.b; t.J0 = J * g.b; t.diag = (I == J);
    const int nt_cur = t.diag ? t.bb : 2 * t.bb;
    load_tile(Xl, xt, g, t, nt_cur);
    int i, j;
    const bool act = tau < nt_cur && tie_coords(t, tau, i, j);
    double r[K];
    if (act) {
#pragma unroll
      for (int k = 0; k < K; ++k) r[k] = rl[((size_t)i * g.N + j) * K + k];
    }
    __syncthreads();
    if (act) {
      const unsigned char* row = xt + tau * g.stride;
      const unsigned char* mrow = xt + mirror_slot(t, tau) * g.stride;
      scan_row(row, s, g.S, g.nchunk, [&](int m, unsigned x) {
        double dx = (double)x;
        unsigned y = mrow[m];
        if (y == 0) {
#pragma unroll
          for (int k = 0; k < K; ++k) Pk[k] += dx * r[k] * ((Gth[m] * Gla[k] != 0.0) ? 1.0 : 0.0);
        } else {
          double z2 = gnu * (double)y;
#pragma unroll
          for (int k = 0; k < K; ++k) Pk[k] += dx * r[k] * w1_of(Gth[m] * Gla[k], z2);
        }
      });
    }
    __syncthreads();
    if (++J == g.nb) { ++I; J = I; }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double v = block_sum(Pk[k], red);
    if (threadIdx.x == 0) a.partP[(size_t)blockIdx.x * K + k] = v;
  }
}

// ------------------------------------------------------------------------------------------
// rho (+ nu partial, + ELBO data terms)   model.py:763-830, 889-923, 948-995, 1013
// partR per workgroup: [0] nu partial, [1] ELBO linear+entropy terms, [2] ELBO log terms,
//                      [3] sum_t (sum_k rho_k) Q_t  (multiplied by -E[nu] in k_fin_rho)
// ------------------------------------------------------------------------------------------
struct RhoArgs {
  const uint8_t* X; const uint64_t* Rb; double* rho; const double* logpr; const double* par;
  double* partR;
};

template <int K, bool MUT, bool UPDATE, bool ELBO>
__global__ __launch_bounds__(TPB, 2) void k_rho(RhoArgs a, Geo g) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char* xt = smem;
  size_t off = (size_t)g.nt * g.stride;
  uint64_t* rw = reinterpret_cast<uint64_t*>(smem + off); off += (size_t)g.nt * g.W * 8;
  double* lut = reinterpret_cast<double*>(smem + off); off += (size_t)g.W * 16 * 16 * 8;
  double* lth = reinterpret_cast<double*>(smem + off); off += (size_t)g.Mp * 8;
  double* Gth = reinterpret_cast<double*>(smem + off); off += (size_t)g.Mp * 8;
  double* red = reinterpret_cast<double*>(smem + off); off += 8 * 8;
  unsigned* qs = reinterpret_cast<unsigned*>(smem + off);
  const ParOff o = par_off(g.L, g.Mp, g.K);
  const int l = blockIdx.x / g.Gl, gb = blockIdx.x - l * g.Gl;
  const long long p0 = (long long)gb * g.P / g.Gl, p1 = (long long)(gb + 1) * g.P / g.Gl;
  const double* Eth = a.par + o.E_th + (size_t)l * g.Mp;
  for (int m = threadIdx.x; m < g.Mp; m += TPB) {
    lth[m] = a.par[o.l_th + (size_t)l * g.Mp + m];
    Gth[m] = a.par[o.G_th + (size_t)l * g.Mp + m];
  }
  // nibble LUT: lut[n][e] = sum of E[theta_m] over the set bits e of reporters 4n..4n+3
  for (int q = threadIdx.x; q < g.W * 16 * 16; q += TPB) {
    int n = q >> 4, e = q & 15;
    double v = 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int m = n * 4 + u;
      if ((e >> u) & 1) v += (m < g.Mp) ? Eth[m] : 0.0;
    }
    lut[q] = v;
  }
  double Ela[K], lla[K], Gla[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    Ela[k] = a.par[o.E_la + l * K + k]; lla[k] = a.par[o.l_la + l * K + k]; Gla[k] = a.par[o.G_la + l * K + k];
  }
  // UPDATE: the weights use the current G_nu; stand-alone ELBO: the stale one (model.py:970)
  const double gnu = a.par[o.sc + (UPDATE ? SC_G_NU : SC_G_NU_STALE)];
  const double eps = g.eps;
  double nu_acc = 0.0, e_lin = 0.0, e_log = 0.0, e_q = 0.0;

  TileCtx t; t.N = g.N; t.b = g.b; t.lb = g.lb; t.bb = g.b * g.b;
  int I, J;
  pair_decode(p0, g.nb, I, J);
  const int tau = threadIdx.x >> g.lS, s = threadIdx.x & (g.S - 1);
  const size_t T = (size_t)g.N * g.N;
  const uint8_t* Xl = a.X + (size_t)l * T * g.Mp;
  const uint64_t* Rl = a.Rb + (size_t)l * T * g.W;
  double* rl = a.rho + (size_t)l * T * K;
  const double* lpl = a.logpr + (size_t)l * T * K;
  const int nnib = g.W * 16;
  __syncthreads();
  for (long long p = p0; p < p1; ++p) {
    t.I0 = I * g.b; t.J0 = J * g.b; t.diag = (I == J);
    const int nt_cur = t.diag ? t.bb : 2 * t.bb;
    load_tile(Xl, xt, g, t, nt_cur);
    for (int q = threadIdx.x; q < nt_cur * g.W; q += TPB) {
      int tq = q / g.W, w = q - tq * g.W, i2, j2;
      rw[q] = tie_coords(t, tq, i2, j2) ? Rl[((size_t)i2 * g.N + j2) * g.W + w] : 0ull;
    }
    int i, j;
    const bool act = tau < nt_cur && tie_coords(t, tau, i, j);
    const size_t tg = act ? ((size_t)i * g.N + j) : 0;
    double lp[K], r[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { lp[k] = 0.0; r[k] = 0.0; }
    if (act) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        lp[k] = lpl[tg * K + k];
        if (!UPDATE) r[k] = rl[tg * K + k];
      }
    }
    __syncthreads();
    const int mtau = act ? mirror_slot(t, tau) : 0;
    const unsigned char* row = xt + tau * g.stride;
    const unsigned char* mrow = xt + mtau * g.stride;
    double Tt = 0.0;
    if (act) {
      // T = sum_m R E[theta_m] (model.py:766-792) by nibble look-up
      const unsigned char* rb = reinterpret_cast<const unsigned char*>(rw + (size_t)tau * g.W);
      for (int n = s; n < nnib; n += g.S) {
        unsigned nib = (rb[n >> 1] >> ((n & 1) * 4)) & 15u;
        Tt += lut[n * 16 + nib];
      }
    }
    Tt = group_sum(Tt, g.S);
    if (UPDATE) {
      double U[K];
#pragma unroll
      for (int k = 0; k < K; ++k) U[k] = 0.0;
      if (act) {
        scan_row(row, s, g.S, g.nchunk, [&](int m, unsigned x) {
          double dx = (double)x;
          unsigned y = MUT ? (unsigned)mrow[m] : 0u;
          if (y == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) U[k] += (lth[m] + lla[k]) * (dx * ((!MUT || Gth[m] * Gla[k] != 0.0) ? 1.0 : 0.0));
          } else {
            double z2 = gnu * (double)y;
#pragma unroll
            for (int k = 0; k < K; ++k) U[k] += (lth[m] + lla[k]) * (dx * w1_of(Gth[m] * Gla[k], z2));
          }
        });
      }
      double sum = 0.0;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        U[k] = group_sum(U[k], g.S);
        r[k] = exp((lp[k] + U[k]) - Tt * Ela[k]);   // no max-subtraction, as model.py:807
        sum += r[k];
      }
      if (sum > 0.0) {
#pragma unroll
        for (int k = 0; k < K; ++k) r[k] /= sum;
      }
      if (act && s == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) rl[tg * K + k] = r[k];
      }
    }
    if ((UPDATE && MUT) || ELBO) {
      unsigned qloc = 0;
      if (act) {
        double er[K];
        if (ELBO) {
#pragma unroll
          for (int k = 0; k < K; ++k) er[k] = exp(r[k]);   // exp(rho), model.py:971
        }
        const uint64_t* rwt = rw + (size_t)tau * g.W;
        const uint64_t* rwm = rw + (size_t)mtau * g.W;
        scan_row(row, s, g.S, g.nchunk, [&](int m, unsigned x) {
          double dx = (double)x;
          unsigned y = MUT ? (unsigned)mrow[m] : 0u;
          if (UPDATE && MUT && y != 0) {
            double z2 = gnu * (double)y;
#pragma unroll
            for (int k = 0; k < K; ++k) {
              double z1 = Gth[m] * Gla[k];
              nu_acc += dx * (z2 / (z1 + z2)) * r[k];   // x w2_k rho_k, model.py:694-696, 822-825
            }
          }
          if (ELBO) {
            bool in_r = (rwt[m >> 6] >> (m & 63)) & 1ull;
            double inner = 0.0;
            if (in_r) {
              double z2 = gnu * (double)y;
#pragma unroll
              for (int k = 0; k < K; ++k) inner += er[k] * (Gth[m] * Gla[k] + z2);
            }
            e_log += dx * log(inner + eps);
            if (MUT && ((rwm[m >> 6] >> (m & 63)) & 1ull)) qloc += x;   // R[mirror] * X^T[mirror]
          }
        });
      }
      if (ELBO) {
        qloc = group_sum_u(qloc, g.S);
        if (act && s == 0) qs[mtau] = qloc;
        __syncthreads();
        if (act && s == 0) {
          double sr = 0.0, se = 0.0, ent = 0.0;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            sr += r[k]; se += r[k] * Ela[k];
            ent += r[k] * lp[k] - r[k] * log(r[k] + eps);   // model.py:1306-1313
          }
          e_lin += ent - se * Tt;
          e_q += sr * (double)qs[tau];
        }
      }
    }
    __syncthreads();
    if (++J == g.nb) { ++I; J = I; }
  }
  double v0 = block_sum(nu_acc, red);
  double v1 = block_sum(e_lin, red);
  double v2 = block_sum(e_log, red);
  double v3 = block_sum(e_q, red);
  if (threadIdx.x == 0) {
    double* out = a.partR + (size_t)blockIdx.x * 4;
    out[0] = v0; out[1] = v1; out[2] = v2; out[3] = v3;
  }
}

// ------------------------------------------------------------------------------------------
// finalize kernels (one workgroup per layer / one workgroup)
// ------------------------------------------------------------------------------------------
// gamma_shp/rte (model.py:700-718), then phi_rte from the same A with the new E[theta]
// (model.py:742-749); mutuality off: phi_shp too, and commit phi.
__global__ __launch_bounds__(TPB) void k_fin_gamma(double* par, const double* __restrict__ partS1,
                                                   const double* __restrict__ partA, const double* __restrict__ partP,
                                                   Geo g) {
  __shared__ double red[8];
  __shared__ double ela_old[KMAX];
  const ParOff o = par_off(g.L, g.Mp, g.K);
  const int l = blockIdx.x, K = g.K, Wp = g.W * 64;
  if (threadIdx.x < K) ela_old[threadIdx.x] = par[o.p_shp + l * K + threadIdx.x] / par[o.p_rte + l * K + threadIdx.x];
  __syncthreads();
  double pr[KMAX];
  for (int k = 0; k < KMAX; ++k) pr[k] = 0.0;
  for (int m = threadIdx.x; m < g.M; m += TPB) {
    double s1 = 0.0;
    for (int gb = 0; gb < g.Gl; ++gb) s1 += partS1[((size_t)l * g.Gl + gb) * g.Mp + m];
    double A[KMAX], rte = 0.0;
    for (int k = 0; k < K; ++k) {
      double ak = 0.0;
      for (int gb = 0; gb < g.Gm; ++gb) ak += partA[(((size_t)l * g.Gm + gb) * Wp + m) * K + k];
      A[k] = ak;
      rte += ela_old[k] * ak;
    }
    size_t q = (size_t)l * g.Mp + m;
    double shp = par[o.a_th + q] + s1;
    rte = par[o.b_th + q] + rte;
    par[o.g_shp + q] = shp; par[o.g_rte + q] = rte;
    double e = shp / rte, lg = digamma_pos(shp) - log(rte);
    par[o.E_th + q] = e; par[o.l_th + q] = lg; par[o.G_th + q] = exp(lg);
    for (int k = 0; k < K; ++k) pr[k] += e * A[k];
  }
  for (int k = 0; k < K; ++k) {
    double v = block_sum(pr[k], red);
    if (threadIdx.x == 0) {
      double rte = par[o.b_la + l * K + k] + v;
      if (g.mut) {
        par[o.p_rte_pend + l * K + k] = rte;
      } else {
        double ps = 0.0;
        for (int gb = 0; gb < g.Gl; ++gb) ps += partP[((size_t)l * g.Gl + gb) * K + k];
        double shp = par[o.a_la + l * K + k] + ps;
        par[o.p_shp + l * K + k] = shp; par[o.p_rte + l * K + k] = rte;
        double lg = digamma_pos(shp) - log(rte);
        par[o.E_la + l * K + k] = shp / rte; par[o.l_la + l * K + k] = lg; par[o.G_la + l * K + k] = exp(lg);
      }
    }
  }
}

// phi commit, mutuality on (model.py:731-749)
__global__ void k_fin_phi(double* par, const double* __restrict__ partP, Geo g) {
  const ParOff o = par_off(g.L, g.Mp, g.K);
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= g.L * g.K) return;
  int l = q / g.K, k = q - l * g.K;
  double ps = 0.0;
  for (int gb = 0; gb < g.Gl; ++gb) ps += partP[((size_t)l * g.Gl + gb) * g.K + k];
  double shp = par[o.a_la + q] + ps, rte = par[o.p_rte_pend + q];
  par[o.p_shp + q] = shp; par[o.p_rte + q] = rte;
  double lg = digamma_pos(shp) - log(rte);
  par[o.E_la + q] = shp / rte; par[o.l_la + q] = lg; par[o.G_la + q] = exp(lg);
}

__device__ __forceinline__ double gamma_elbo_term(double pa, double pb, double qa, double qb) {
  // model.py:1300-1303
  return lgamma(qa) - pa * log(qb) + (pa - qa) * digamma_pos(qa) + qa * (1.0 - pb / qb);
}

// nu commit (model.py:822-825) and/or ELBO assembly (model.py:997-1013)
__global__ __launch_bounds__(TPB) void k_fin_rho(double* par, const double* __restrict__ partR, double* elbo_out,
                                                 int nblocks, int do_nu, int do_elbo, Geo g) {
  __shared__ double red[8];
  const ParOff o = par_off(g.L, g.Mp, g.K);
  double* sc = par + o.sc;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  for (int q = threadIdx.x; q < nblocks; q += TPB) {
    a0 += partR[(size_t)q * 4 + 0]; a1 += partR[(size_t)q * 4 + 1];
    a2 += partR[(size_t)q * 4 + 2]; a3 += partR[(size_t)q * 4 + 3];
  }
  a0 = block_sum(a0, red); a1 = block_sum(a1, red); a2 = block_sum(a2, red); a3 = block_sum(a3, red);
  double gt = 0.0;
  if (do_elbo) {
    for (int q = threadIdx.x; q < g.L * g.Mp; q += TPB)
      if (q % g.Mp < g.M) gt += gamma_elbo_term(par[o.a_th + q], par[o.b_th + q], par[o.g_shp + q], par[o.g_rte + q]);
    for (int q = threadIdx.x; q < g.L * g.K; q += TPB)
      gt += gamma_elbo_term(par[o.a_la + q], par[o.b_la + q], par[o.p_shp + q], par[o.p_rte + q]);
    gt = block_sum(gt, red);
  }
  if (threadIdx.x == 0) {
    if (do_nu && g.mut) {
      sc[SC_G_NU_STALE] = sc[SC_G_NU];           // what the last cache refresh held (model.py:684)
      sc[SC_NU_SHP] = sc[SC_A_ETA] + a0;
      sc[SC_G_NU] = exp(digamma_pos(sc[SC_NU_SHP]) - log(sc[SC_NU_RTE]));
      sc[SC_E_NU] = sc[SC_NU_SHP] / sc[SC_NU_RTE];
    }
    if (do_elbo) {
      double e = a1 + a2 - sc[SC_E_NU] * a3 + gt;
      e += gamma_elbo_term(sc[SC_A_ETA], sc[SC_B_ETA], sc[SC_NU_SHP], sc[SC_NU_RTE]);
      elbo_out[0] = e;
    }
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static int fail(vmr_handle h, int code, const char* msg) {
  if (h) h->err = msg; else g_create_err = msg;
  return code;
}

static size_t shmem_counts(const Geo& g) { return (size_t)g.nt * g.stride + (size_t)g.Mp * 16 + 64; }
static size_t shmem_phi(const Geo& g) { return (size_t)g.nt * g.stride + (size_t)g.Mp * 8 + 64; }
static size_t shmem_rho(const Geo& g) {
  return (size_t)g.nt * g.stride + (size_t)g.nt * g.W * 8 + (size_t)g.W * 16 * 16 * 8 + (size_t)g.Mp * 16 + 64 +
         (size_t)g.nt * 4 + 16;
}

struct Prof {
  vmr_ctx* h; int cls; hipEvent_t a = nullptr, b = nullptr;
  Prof(vmr_ctx* h_, int c) : h(h_), cls(c) {
    if (h->prof) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, h->stream); }
  }
  ~Prof() {
    if (h->prof) { (void)hipEventRecord(b, h->stream); h->evs.push_back({cls, a, b}); }
  }
};

#define DISPATCH_K(K_, ...)                         \
  switch (K_) {                                     \
    case 2: { constexpr int KK = 2; __VA_ARGS__; } break; \
    case 3: { constexpr int KK = 3; __VA_ARGS__; } break; \
    case 4: { constexpr int KK = 4; __VA_ARGS__; } break; \
    case 5: { constexpr int KK = 5; __VA_ARGS__; } break; \
    case 6: { constexpr int KK = 6; __VA_ARGS__; } break; \
    case 7: { constexpr int KK = 7; __VA_ARGS__; } break; \
    case 8: { constexpr int KK = 8; __VA_ARGS__; } break; \
    default: break;                                 \
  }

template <class Kern>
static hipError_t set_smem(Kern k, size_t bytes) {
  if (bytes > 48 * 1024) return hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return hipSuccess;
}

static int launch_gamma(vmr_ctx* h) {
  const Geo& g = h->g;
  {
    Prof p(h, VMR_KERNEL_GAMMA_MASK);
    DISPATCH_K(g.K, hipLaunchKernelGGL(k_gamma_mask<KK>, dim3(g.L * g.Gm), dim3(TPB), 0, h->stream, h->Rb, h->rho, h->partA, g));
  }
  {
    Prof p(h, VMR_KERNEL_GAMMA_COUNTS);
    CountArgs a{h->X, h->rho, h->par, h->partS1, h->partP};
    size_t sm = shmem_counts(g);
    if (g.mut) {
      DISPATCH_K(g.K, HIPCHK(h, set_smem(k_gamma_counts<KK, true>, sm));
                 hipLaunchKernelGGL((k_gamma_counts<KK, true>), dim3(g.L * g.Gl), dim3(TPB), sm, h->stream, a, g));
    } else {
      DISPATCH_K(g.K, HIPCHK(h, set_smem(k_gamma_counts<KK, false>, sm));
                 hipLaunchKernelGGL((k_gamma_counts<KK, false>), dim3(g.L * g.Gl), dim3(TPB), sm, h->stream, a, g));
    }
  }
  {
    Prof p(h, VMR_KERNEL_FINALIZE);
    hipLaunchKernelGGL(k_fin_gamma, dim3(g.L), dim3(TPB), 0, h->stream, h->par, h->partS1, h->partA, h->partP, g);
  }
  HIPCHK(h, hipGetLastError());
  return VMR_OK;
}

static int launch_phi(vmr_ctx* h) {
  const Geo& g = h->g;
  if (!g.mut) return VMR_OK;   // committed by k_fin_gamma
  {
    Prof p(h, VMR_KERNEL_PHI);
    CountArgs a{h->X, h->rho, h->par, h->partS1, h->partP};
    size_t sm = shmem_phi(g);
    DISPATCH_K(g.K, HIPCHK(h, set_smem(k_phi<KK>, sm));
               hipLaunchKernelGGL(k_phi<KK>, dim3(g.L * g.Gl), dim3(TPB), sm, h->stream, a, g));
  }
  {
    Prof p(h, VMR_KERNEL_FINALIZE);
    int n = g.L * g.K;
    hipLaunchKernelGGL(k_fin_phi, dim3((n + 63) / 64), dim3(64), 0, h->stream, h->par, h->partP, g);
  }
  HIPCHK(h, hipGetLastError());
  return VMR_OK;
}

// mode: 0 = rho update (+nu), 1 = rho update + fused ELBO, 2 = ELBO only
static int launch_rho(vmr_ctx* h, int mode, bool commit_nu) {
  const Geo& g = h->g;
  RhoArgs a{h->X, h->Rb, h->rho, h->logpr, h->par, h->partR};
  size_t sm = shmem_rho(g);
  dim3 grid(g.L * g.Gl), blk(TPB);
  {
    Prof p(h, mode == 2 ? VMR_KERNEL_ELBO : VMR_KERNEL_RHO);
#define LRHO(MUT_, UPD_, ELB_)                                                            \
  DISPATCH_K(g.K, HIPCHK(h, set_smem(k_rho<KK, MUT_, UPD_, ELB_>, sm));                    \
             hipLaunchKernelGGL((k_rho<KK, MUT_, UPD_, ELB_>), grid, blk, sm, h->stream, a, g))
    if (g.mut) {
      if (mode == 0) { LRHO(true, true, false); } else if (mode == 1) { LRHO(true, true, true); } else { LRHO(true, false, true); }
    } else {
      if (mode == 0) { LRHO(false, true, false); } else if (mode == 1) { LRHO(false, true, true); } else { LRHO(false, false, true); }
    }
#undef LRHO
  }
  {
    Prof p(h, VMR_KERNEL_FINALIZE);
    hipLaunchKernelGGL(k_fin_rho, dim3(1), dim3(TPB), 0, h->stream, h->par, h->partR, h->elbo_dev, g.L * g.Gl,
                       (mode != 2 && commit_nu) ? 1 : 0, mode != 0 ? 1 : 0, g);
  }
  HIPCHK(h, hipGetLastError());
  return VMR_OK;
}

static int choose_geo(Geo& g, int ncu, std::string& err) {
  g.Mp = (g.M + 15) / 16 * 16;
  g.nchunk = g.Mp / 16;
  g.stride = (g.nchunk % 2 == 1) ? g.Mp : g.Mp + 16;
  g.W = (g.M + 63) / 64;
  const size_t budget = 48 * 1024;
  int b = 8;
  while (b > 1 && (size_t)2 * b * b * g.stride > budget) b >>= 1;
  if ((size_t)2 * b * b * g.stride > 96 * 1024) { err = "M too large for the LDS tile (M <= ~49000 supported)"; return VMR_EINVAL; }
  g.b = b; g.lb = (b == 8) ? 3 : (b == 4) ? 2 : (b == 2) ? 1 : 0;
  g.nb = (g.N + b - 1) / b;
  g.nt = 2 * b * b;
  g.S = TPB / g.nt; if (g.S > 64) g.S = 64;
  g.lS = 0; while ((1 << g.lS) < g.S) ++g.lS;
  g.P = (long long)g.nb * (g.nb + 1) / 2;
  int target = ncu * 4;
  long long gl = target / g.L; if (gl < 1) gl = 1; if (gl > g.P) gl = g.P;
  g.Gl = (int)gl;
  long long T = (long long)g.N * g.N;
  long long gm = target / g.L; if (gm < 1) gm = 1;
  long long maxgm = (T + 255) / 256; if (gm > maxgm) gm = maxgm; if (gm < 1) gm = 1;
  g.Gm = (int)gm;
  return VMR_OK;
}

extern "C" {

const char* vmr_version(void) { return VMR_VERSION; }

const char* vmr_last_error(vmr_handle h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int vmr_create(vmr_handle* out, int device, int L, int N, int M, int K, int mutuality, const uint8_t* X,
               const uint8_t* R, int data_on_device, double eps) {
  if (!out) return fail(nullptr, VMR_EINVAL, "out is NULL");
  *out = nullptr;
  if (L < 1 || N < 1 || M < 1) return fail(nullptr, VMR_EINVAL, "L, N, M must be positive");
  if (K < 2 || K > KMAX) return fail(nullptr, VMR_EINVAL, "K must be in [2, 8]");
  if (!X) return fail(nullptr, VMR_EINVAL, "X is NULL");
  int ndev = 0;
  HIPCHK((vmr_ctx*)nullptr, hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(nullptr, VMR_EINVAL, "no such HIP device");
  HIPCHK((vmr_ctx*)nullptr, hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK((vmr_ctx*)nullptr, hipGetDeviceProperties(&prop, device));
  vmr_ctx* h = new vmr_ctx();
  h->device = device;
  Geo& g = h->g;
  g.L = L; g.N = N; g.M = M; g.K = K; g.mut = mutuality ? 1 : 0; g.eps = eps;
  std::string err;
  if (choose_geo(g, prop.multiProcessorCount, err) != VMR_OK) { delete h; return fail(nullptr, VMR_EINVAL, err.c_str()); }
  memset(h->prof_ms, 0, sizeof h->prof_ms); memset(h->prof_n, 0, sizeof h->prof_n);
#define CCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { g_create_err = std::string(#call) + ": " + hipGetErrorString(e_); vmr_destroy(h); return VMR_EHIP; } } while (0)
  CCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  const size_t rows = (size_t)L * N * N;
  const size_t raw = rows * M;
  CCHK(hipMalloc(&h->X, rows * g.Mp));
  CCHK(hipMalloc(&h->Rb, rows * g.W * 8));
  CCHK(hipMalloc(&h->cov, rows));
  CCHK(hipMalloc(&h->sumx, 8));
  CCHK(hipMemsetAsync(h->sumx, 0, 8, h->stream));
  {
    uint8_t* tmp = nullptr;
    const uint8_t* src = X;
    if (!data_on_device) { CCHK(hipMalloc(&tmp, raw)); CCHK(hipMemcpyAsync(tmp, X, raw, hipMemcpyHostToDevice, h->stream)); src = tmp; }
    hipLaunchKernelGGL(k_pack_x, dim3(4096), dim3(256), 0, h->stream, src, h->X, rows, M, g.Mp);
    CCHK(hipStreamSynchronize(h->stream));
    if (tmp) CCHK(hipFree(tmp));
    tmp = nullptr; src = R;
    if (R && !data_on_device) { CCHK(hipMalloc(&tmp, raw)); CCHK(hipMemcpyAsync(tmp, R, raw, hipMemcpyHostToDevice, h->stream)); src = tmp; }
    hipLaunchKernelGGL(k_pack_r, dim3(4096), dim3(256), 0, h->stream, src, h->Rb, rows, M, g.W);
    hipLaunchKernelGGL(k_stats, dim3(2048), dim3(256), 0, h->stream, h->X, h->Rb, h->cov, h->sumx, rows, g.Mp, g.W);
    CCHK(hipStreamSynchronize(h->stream));
    if (tmp) CCHK(hipFree(tmp));
  }
  CCHK(hipMalloc(&h->rho, rows * K * 8));
  CCHK(hipMalloc(&h->logpr, rows * K * 8));
  ParOff o = par_off(L, g.Mp, K);
  h->par_doubles = o.total;
  CCHK(hipMalloc(&h->par, o.total * 8));
  CCHK(hipMemsetAsync(h->par, 0, o.total * 8, h->stream));
  CCHK(hipMalloc(&h->partS1, (size_t)L * g.Gl * g.Mp * 8));
  CCHK(hipMalloc(&h->partA, (size_t)L * g.Gm * g.W * 64 * K * 8));
  CCHK(hipMalloc(&h->partP, (size_t)L * g.Gl * K * 8));
  CCHK(hipMalloc(&h->partR, (size_t)L * g.Gl * 4 * 8));
  CCHK(hipMalloc(&h->elbo_dev, 8 * 8));
  CCHK(hipStreamSynchronize(h->stream));
#undef CCHK
  *out = h;
  return VMR_OK;
}

void vmr_destroy(vmr_handle h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (auto& e : h->evs) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  void* ptrs[] = {h->X, h->Rb, h->cov, h->sumx, h->rho, h->logpr, h->par, h->partS1, h->partA, h->partP, h->partR, h->elbo_dev};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int vmr_data_stats(vmr_handle h, double* sum_x, uint8_t* coverage) {
  if (!h) return VMR_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  if (sum_x) {
    unsigned long long v = 0;
    HIPCHK(h, hipMemcpy(&v, h->sumx, 8, hipMemcpyDeviceToHost));
    *sum_x = (double)v;
  }
  if (coverage) HIPCHK(h, hipMemcpy(coverage, h->cov, (size_t)h->g.L * h->g.N * h->g.N, hipMemcpyDeviceToHost));
  return VMR_OK;
}

// host [L,M] -> device [L,Mp]
static int upload_lm(vmr_ctx* h, size_t off, const double* src, double pad) {
  const Geo& g = h->g;
  std::vector<double> buf((size_t)g.L * g.Mp, pad);
  for (int l = 0; l < g.L; ++l) memcpy(&buf[(size_t)l * g.Mp], src + (size_t)l * g.M, (size_t)g.M * 8);
  HIPCHK(h, hipMemcpyAsync(h->par + off, buf.data(), buf.size() * 8, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return VMR_OK;
}

int vmr_set_priors(vmr_handle h, const double* alpha_theta, const double* beta_theta, const double* alpha_lambda,
                   const double* beta_lambda, double alpha_eta, double beta_eta) {
  if (!h || !alpha_theta || !beta_theta || !alpha_lambda || !beta_lambda) return fail(h, VMR_EINVAL, "NULL prior array");
  HIPCHK(h, hipSetDevice(h->device));
  const Geo& g = h->g;
  ParOff o = par_off(g.L, g.Mp, g.K);
  int rc;
  if ((rc = upload_lm(h, o.a_th, alpha_theta, 1.0))) return rc;
  if ((rc = upload_lm(h, o.b_th, beta_theta, 1.0))) return rc;
  HIPCHK(h, hipMemcpy(h->par + o.a_la, alpha_lambda, (size_t)g.L * g.K * 8, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->par + o.b_la, beta_lambda, (size_t)g.L * g.K * 8, hipMemcpyHostToDevice));
  double ab[2] = {alpha_eta, beta_eta};
  HIPCHK(h, hipMemcpy(h->par + o.sc + SC_A_ETA, ab, 16, hipMemcpyHostToDevice));
  h->have_priors = true;
  return VMR_OK;
}

int vmr_set_state(vmr_handle h, const double* gamma_shp, const double* gamma_rte, const double* phi_shp,
                  const double* phi_rte, double nu_shp, double nu_rte, const double* pr_rho, int pr_rho_on_device) {
  if (!h || !gamma_shp || !gamma_rte || !phi_shp || !phi_rte || !pr_rho) return fail(h, VMR_EINVAL, "NULL state array");
  if (!h->have_priors) return fail(h, VMR_ESTATE, "vmr_set_priors must be called before vmr_set_state");
  HIPCHK(h, hipSetDevice(h->device));
  const Geo& g = h->g;
  ParOff o = par_off(g.L, g.Mp, g.K);
  int rc;
  if ((rc = upload_lm(h, o.g_shp, gamma_shp, 1.0))) return rc;
  if ((rc = upload_lm(h, o.g_rte, gamma_rte, 1.0))) return rc;
  HIPCHK(h, hipMemcpy(h->par + o.p_shp, phi_shp, (size_t)g.L * g.K * 8, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->par + o.p_rte, phi_rte, (size_t)g.L * g.K * 8, hipMemcpyHostToDevice));
  double nu[2] = {nu_shp, nu_rte};
  HIPCHK(h, hipMemcpy(h->par + o.sc + SC_NU_SHP, nu, 16, hipMemcpyHostToDevice));
  const size_t n = (size_t)g.L * g.N * g.N * g.K;
  const double* src = pr_rho;
  if (!pr_rho_on_device) {
    // stage through logpr (overwritten by k_init_rho element-wise after being read)
    HIPCHK(h, hipMemcpyAsync(h->logpr, pr_rho, n * 8, hipMemcpyHostToDevice, h->stream));
    src = h->logpr;
  }
  hipLaunchKernelGGL(k_init_rho, dim3(4096), dim3(256), 0, h->stream, src, h->rho, h->logpr, n, g.eps);
  hipLaunchKernelGGL(k_derive_all, dim3(8), dim3(256), 0, h->stream, h->par, g);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->have_state = true;
  return VMR_OK;
}

static int read_elbo(vmr_ctx* h, double* out) {
  HIPCHK(h, hipMemcpyAsync(out, h->elbo_dev, 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (isnan(*out)) return fail(h, VMR_ENAN, "ELBO is NaN!!!!");
  return VMR_OK;
}

int vmr_step(vmr_handle h, int n_iters, double* elbo_out) {
  if (!h) return VMR_EINVAL;
  if (!h->have_state) return fail(h, VMR_ESTATE, "vmr_set_state must be called before vmr_step");
  if (n_iters < 0) return fail(h, VMR_EINVAL, "n_iters < 0");
  HIPCHK(h, hipSetDevice(h->device));
  int rc;
  for (int it = 0; it < n_iters; ++it) {
    if ((rc = launch_gamma(h))) return rc;
    if ((rc = launch_phi(h))) return rc;
    bool last = (it == n_iters - 1) && elbo_out;
    if ((rc = launch_rho(h, last ? 1 : 0, true))) return rc;
  }
  if (elbo_out) {
    if (n_iters == 0) return vmr_elbo(h, elbo_out);
    return read_elbo(h, elbo_out);
  }
  return VMR_OK;
}

int vmr_elbo(vmr_handle h, double* out) {
  if (!h || !out) return VMR_EINVAL;
  if (!h->have_state) return fail(h, VMR_ESTATE, "vmr_set_state must be called before vmr_elbo");
  HIPCHK(h, hipSetDevice(h->device));
  int rc;
  if ((rc = launch_rho(h, 2, false))) return rc;
  return read_elbo(h, out);
}

int vmr_sub_step(vmr_handle h, int which) {
  if (!h) return VMR_EINVAL;
  if (!h->have_state) return fail(h, VMR_ESTATE, "vmr_set_state must be called before vmr_sub_step");
  HIPCHK(h, hipSetDevice(h->device));
  switch (which) {
    case VMR_STEP_GAMMA: return launch_gamma(h);
    case VMR_STEP_PHI: return launch_phi(h);
    // rho and nu come out of one pass over X; the nu value is committed by the NU sub-step
    case VMR_STEP_RHO: return launch_rho(h, 0, false);
    case VMR_STEP_NU: {
      if (!h->g.mut) return VMR_OK;
      Prof p(h, VMR_KERNEL_FINALIZE);
      hipLaunchKernelGGL(k_fin_rho, dim3(1), dim3(TPB), 0, h->stream, h->par, h->partR, h->elbo_dev,
                         h->g.L * h->g.Gl, 1, 0, h->g);
      HIPCHK(h, hipGetLastError());
      return VMR_OK;
    }
    default: return fail(h, VMR_EINVAL, "unknown sub-step");
  }
}

static int download_lm(vmr_ctx* h, size_t off, double* dst) {
  const Geo& g = h->g;
  std::vector<double> buf((size_t)g.L * g.Mp);
  HIPCHK(h, hipMemcpy(buf.data(), h->par + off, buf.size() * 8, hipMemcpyDeviceToHost));
  for (int l = 0; l < g.L; ++l) memcpy(dst + (size_t)l * g.M, &buf[(size_t)l * g.Mp], (size_t)g.M * 8);
  return VMR_OK;
}

int vmr_get_state(vmr_handle h, double* gamma_shp, double* gamma_rte, double* phi_shp, double* phi_rte,
                  double* nu_shp, double* nu_rte, double* rho) {
  if (!h) return VMR_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const Geo& g = h->g;
  ParOff o = par_off(g.L, g.Mp, g.K);
  int rc;
  if (gamma_shp && (rc = download_lm(h, o.g_shp, gamma_shp))) return rc;
  if (gamma_rte && (rc = download_lm(h, o.g_rte, gamma_rte))) return rc;
  if (phi_shp) HIPCHK(h, hipMemcpy(phi_shp, h->par + o.p_shp, (size_t)g.L * g.K * 8, hipMemcpyDeviceToHost));
  if (phi_rte) HIPCHK(h, hipMemcpy(phi_rte, h->par + o.p_rte, (size_t)g.L * g.K * 8, hipMemcpyDeviceToHost));
  if (nu_shp) HIPCHK(h, hipMemcpy(nu_shp, h->par + o.sc + SC_NU_SHP, 8, hipMemcpyDeviceToHost));
  if (nu_rte) HIPCHK(h, hipMemcpy(nu_rte, h->par + o.sc + SC_NU_RTE, 8, hipMemcpyDeviceToHost));
  if (rho) HIPCHK(h, hipMemcpy(rho, h->rho, (size_t)g.L * g.N * g.N * g.K * 8, hipMemcpyDeviceToHost));
  return VMR_OK;
}

int vmr_get_geometric(vmr_handle h, double* g_theta, double* g_lambda, double* g_nu, double* g_nu_cache) {
  if (!h) return VMR_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const Geo& g = h->g;
  ParOff o = par_off(g.L, g.Mp, g.K);
  int rc;
  if (g_theta && (rc = download_lm(h, o.G_th, g_theta))) return rc;
  if (g_lambda) HIPCHK(h, hipMemcpy(g_lambda, h->par + o.G_la, (size_t)g.L * g.K * 8, hipMemcpyDeviceToHost));
  if (g_nu) HIPCHK(h, hipMemcpy(g_nu, h->par + o.sc + SC_G_NU, 8, hipMemcpyDeviceToHost));
  if (g_nu_cache) HIPCHK(h, hipMemcpy(g_nu_cache, h->par + o.sc + SC_G_NU_STALE, 8, hipMemcpyDeviceToHost));
  return VMR_OK;
}

int vmr_sync(vmr_handle h) {
  if (!h) return VMR_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return VMR_OK;
}

static int prof_collect(vmr_ctx* h) {
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (auto& e : h->evs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { h->prof_ms[e.cls] += ms; h->prof_n[e.cls] += 1; }
    (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
  }
  h->evs.clear();
  return VMR_OK;
}

int vmr_profile(vmr_handle h, int enable) {
  if (!h) return VMR_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  int rc = prof_collect(h);
  if (rc) return rc;
  h->prof = enable != 0;
  if (enable) { memset(h->prof_ms, 0, sizeof h->prof_ms); memset(h->prof_n, 0, sizeof h->prof_n); }
  return VMR_OK;
}

int vmr_profile_read(vmr_handle h, int kernel_class, double* total_ms, int64_t* launches) {
  if (!h || kernel_class < 0 || kernel_class >= VMR_KERNEL_COUNT) return VMR_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  int rc = prof_collect(h);
  if (rc) return rc;
  if (total_ms) *total_ms = h->prof_ms[kernel_class];
  if (launches) *launches = h->prof_n[kernel_class];
  return VMR_OK;
}

int vmr_kernel_bytes(vmr_handle h, int kernel_class, double* bytes) {
  if (!h || !bytes) return VMR_EINVAL;
  const Geo& g = h->g;
  const double V = (double)g.L * g.N * g.N * g.M;   // canonical: X 1 B/elt, R 1 bit/elt
  const double SX = V, SR = V / 8.0, Srho = 8.0 * g.L * (double)g.N * g.N * g.K;
  switch (kernel_class) {
    case VMR_KERNEL_GAMMA_MASK: *bytes = SR + Srho; break;
    case VMR_KERNEL_GAMMA_COUNTS: *bytes = SX + Srho; break;
    case VMR_KERNEL_PHI: *bytes = SX + Srho; break;
    case VMR_KERNEL_RHO: *bytes = SX + SR + 2.0 * Srho; break;
    case VMR_KERNEL_ELBO: *bytes = SX + SR + 2.0 * Srho; break;
    default: *bytes = 0.0; break;
  }
  return VMR_OK;
}

}  // extern "C"
