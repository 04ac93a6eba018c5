// vmr_internal.h -- shared by the translation units of libvimure_hip.so (not part of the C-ABI: that is include/vimure_hip.h).
// Context and geometry of a handle, the parameter block layout, error macros, and the small device helpers every kernel file uses.
#ifndef VMR_INTERNAL_H
#define VMR_INTERNAL_H
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <utility>
#include <algorithm>
#include <chrono>
#include <type_traits>

#include "vimure_hip.h"

#define VMR_VERSION "vimure_hip 0.1 (gfx950)"
#define TPB 256
#ifndef VMR_LB_COUNTS
#define VMR_LB_COUNTS 4   // resident workgroups per CU the gamma/phi sweeps are compiled for
#endif
#ifndef VMR_LB_RHO
#define VMR_LB_RHO 3
#endif
#ifndef VMR_NR_STEPS
#define VMR_NR_STEPS 1   // v_rcp_f64 is good to 4.6e-8; one step gives 2e-15, two are exact (tools/rcp_accuracy.hip)
#endif
#define KMAX 8        // categories the specialised kernels are compiled for (one sweep object per K, per-K register budgets)
#define KGEN_MAX 256  // ... and the general kernels (sweep_gen.hip: a category per lane) take: K = max(X) + 1 is the reference's default (model.py:179-197)

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
  int Y;        // mirror-count levels of the statistics H: max count + 1 (1 when mutuality is off)
  int hc;       // how many of them are accumulated in LDS (dense tiles: 0..HC_MAX); the rest goes to global atomics
  int yt;       // report lists: levels of the factor table F the rho pass keeps in LDS; the rest is read from global
  int fuse_full; // rows of R that are all ones are summed by the rho pass itself (A gets sum_t rho_k for them)
  int ml;        // report lists with mask lists: the rho / statistics pass sums rho over the listed reporters too (A[Mp][K] in LDS)
  int two_pass; // wide reporter dimension: the LDS levels do not fit beside the rho pass' tables, so H is rebuilt by
                // k_hist after the rho pass (two passes over X per sweep instead of one)
  int pf;       // 16-B chunks of a tile pair each thread stages (prefetch depth)
  int heavy;    // a lane's share of a row with more non-zeros than this is processed by the whole wave
  int dbg;      // timing experiments only (env VMR_DEBUG; results are wrong): dense path 1 = skip per-report math, 2 = skip
                // the scan; report lists 8 = no H flush, 16 = no walk 1, 32 = no walk 2, 64 = no exp in the tie update
  double eps;
  int det;      // VMR_DETERMINISTIC=1: bit-reproducible sweeps (one wave per workgroup, fixed step shares, integer cross-workgroup sums)
  int det_sh;   // ... whose fixed point for count-weighted sums (H, the nu share) is 2^-det_sh: sum x of the dataset < 2^(62 - det_sh)
  int det_shr;  // ... and 2^-det_shr for the ELBO partials (bounded by 64 (sum x + ties K))
  int gen;      // the general kernels (sweep_gen.hip): K > KMAX, or entries too wide for the packed format.  H is then ONE copy
                // [L][Y][Mp][K] holding every category (no constant C, no deficits), nu and the mask sums go through k_fin_rho / the pass
  int farl;     // report lists, one pass per sweep although not every level of F / H fits in LDS: the reports of the levels beyond
                // (y >= hc = yt; a few per cent) are ALSO kept as a compact list (vmr_ctx::far_pos / far_ent); the pass takes their
                // factors by formula and leaves their statistics to k_far_hist, which follows it
  int wide;     // entries in two words (vmr_ctx::EX): counts beyond 2047 or (max count + 1) * Mp beyond 2^20 table rows
};
// fixed point of the deterministic mode: count-weighted sums 2^-g.det_sh; ELBO partials 2^-g.det_shr; sums of rho over ties (< 2^31 ties) 2^-30
#define DET_SH_A 30
__device__ __forceinline__ unsigned long long det_fx(double v, int sh) { return (unsigned long long)__double2ll_rn(ldexp(v, sh)); }
__device__ __forceinline__ double det_back(unsigned long long u, int sh) { return ldexp((double)(long long)u, -sh); }

#define NSLOT 8   // accumulation slots per layer for cross-workgroup sums (global f64 atomics)

struct vmr_ctx {
  Geo g;
  int device;
  hipStream_t stream;
  hipStream_t stream2 = nullptr;     // the mask half of the gamma update runs beside the counts half
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  std::string err;
  // data
  uint8_t* X = nullptr;        // [L][N*N][Mp]
  uint64_t* Rb = nullptr;      // [L][N*N][W]
  uint8_t* cov = nullptr;      // [L][N*N]
  uint8_t* rcls = nullptr;     // [L][N*N] class of the mask row: 0 empty, 1 all ones, 2 partial
  // report lists (sparse format, see k_rho_sp); X is freed once they exist
  int sparse = 0;
  unsigned* E = nullptr;       // one entry per non-zero count, layer after layer
  unsigned* EX = nullptr;      // wide entries (Geo::wide): E holds the table row y * Mp + m in 32 bits, EX the count << 1 | R[l,i,j,m], same slots
  double* gen_s1 = nullptr;    // [L][Mp] + [L][K] scratch of the general finalize (sums over H by reporter, by category) + [L] step counters of the pass
  unsigned* rs = nullptr;      // [L][N*N/64+1] first entry of every 64-tie step, relative to ebase[l]
  double* Cg = nullptr;        // [L][Y][Mp] sum of the counts x per (mirror count, reporter): what H_0 is rebuilt from
  int sp_tpb = 256;            // threads per workgroup of the report-list passes that update rho or reduce the ELBO
  int st_tpb = 256;            // ... of the statistics-only pass (a lighter variant: bigger workgroups)
  unsigned* Qt = nullptr;      // [L][N*N] sum_m R[t,m] X[mirror(t),m]
  unsigned long long* ebase = nullptr;   // device [L]
  unsigned long long nnz = 0;  // non-zero counts in X
  unsigned long long n_slots = 0;   // entry slots of the report lists: nnz + the padding of the full rounds
  int all_full = 0;            // every mask row is all ones
  // report lists (sweep_sl.h): rho, logpr and the per-tie arrays below are stored BY SORTED POSITION; perm translates at the
  // boundary (vmr_set_state, vmr_get_state, vmr_readout, vmr_sample)
  unsigned* perm = nullptr;    // [L][NS*64] position -> tie (0xffffffff past the last tie)
  unsigned* sy = nullptr;      // [L][NS] highest mirror-count level of a step's reports
  unsigned* far_pos = nullptr; // Geo::farl: sorted position (layer-local) and entry word of every report of a level beyond the LDS ones,
  unsigned* far_ent = nullptr; //            layer l at [far_off[l], far_off[l + 1])
  unsigned long long* far_base = nullptr;   // [L + 1] device copy of far_off
  std::vector<unsigned long long> far_off;
  double* h0s = nullptr;       // level-0 rounds without LDS adds (SlArgs::h0s): [L][NSLOT][K] sums over ties of rho_k times the tie's counts in
                               // such rounds, then [L] sum_m C[l][0][m]
  unsigned long long stat_slots = 0;   // ... the entry slots its statistics pass still reads (the rounds before a step's first all-level-0 one)
  unsigned* x0p = nullptr;     // ... and [L][NS * 64], by position: a tie's summed counts at mirror count 0 (SlArgs::x0p)
  bool elbo_split = false;     // Geo::farl: the fused rho + ELBO variant does not keep g.hc levels in LDS (its logarithm table): an ELBO sweep is
                               // the plain update pass followed by the ELBO-only pass (same numbers: the stale G_nu, model.py:970)
  uint8_t* cls_p = nullptr;    // [L][T] rcls by position (null when every row is all ones)
  unsigned* Qt_p = nullptr;    // [L][T] Qt by position
  double* nat = nullptr;       // [L][T][K] scratch in tie order for the boundary copies of rho / pr_rho (allocated on first use)
  // mask lists (partial rows with few reporters), see k_mask_lists
  unsigned* rq = nullptr;              // [L][N*N+1]
  unsigned short* Rm = nullptr;
  unsigned long long* rbase = nullptr; // device [L]
  unsigned long long n_rm = 0;         // listed reporters in all
  unsigned rm_maxrow = 0;              // longest list
  unsigned* rm2 = nullptr;             // [L][NS * 64] by sorted position, when no list is longer than 2 (SlArgs::rm2)
  unsigned long long* sumx = nullptr;
  // state
  double *rho = nullptr, *logpr = nullptr;
  double* par = nullptr;       // parameter block, see P_* offsets
  // Steady-state sweeps as hipGraphs (vmr_step and the fit loop: up to 9 plain sweeps per launch; env VMR_GRAPH=1).  Off by default:
  // measured at BASELINE config 3 (round 4, with the unwritten rho inside the graphs) 5797-5959 iterations/s replayed against
  // 5882-5906 queued launch by launch -- the two launches of a sweep are not what it waits for; on the Karnataka-shaped batch (48
  // fits, 8 host threads) 22.3 fits/s against 22.9 -- and a capture is invalidated when another host thread creates or destroys a
  // handle meanwhile (hipMalloc / hipFree during capture; step_n then falls back to the eager loop for good).
  // A graph is the sweeps' launches with their ARGUMENTS: keyed by the count, by whether its last sweep leaves rho unwritten and by
  // the handle's bookkeeping state at capture (sig0; sig1 = the state it leaves), dropped whenever vmr_set_state may change an argument.
  struct GraphEntry { int n; bool lazy_last; unsigned sig0, sig1; hipGraphExec_t ex; };
  std::vector<GraphEntry> graphs;
  bool use_graphs = false;
  unsigned long long* det_buf = nullptr;   // deterministic mode: integer shadows of H (one copy), the mask sums, the ELBO partials, the nu share
  double* fr_slots = nullptr;              // ... and k_fin_rho's per-workgroup partial sums [L * FR_G][2]
  double *rho_snap = nullptr, *par_snap = nullptr;   // vmr_snapshot: the best realisation so far (model.py:925-942)
  bool have_snap = false;
  bool lp0 = false;            // every tie of a step without reports carries the one-hot prior: the sweeps do not read those steps' log prior (SlArgs::lp0)
  bool rho_stale = false;      // the last sweep used its new rho without writing it (sweep_body's STORE = false): ensure_rho re-writes it before anyone reads
  bool restored = false;       // vmr_restore brought back rho and the parameters of another realisation, not its log prior: no sweeps until vmr_set_state
  size_t par_doubles = 0;
  // partials
  double *slotA = nullptr, *slotR = nullptr;   // NSLOT accumulation slots
  double* Hg = nullptr;        // sufficient statistics H[L][Y][Mp][K]
  bool h_valid = false;        // H matches the current rho
  double* slotF = nullptr;     // [L][NSLOT][K]: sum of the new rho over ties whose mask row is all ones (rho pass)
  bool f_valid = false;        // slotF matches the current rho
  bool a_valid = false;        // slotA holds the mask-list sums of the current rho (summed by the last rho / statistics pass)
  bool a_zero = true;          // slotA is known to be all zero
  bool h_reduced = false;      // the NH copies of H are folded into copy 0 (what the finalize kernels read)
  bool h_zero = false;         // k_fin_gamma consumed H and slotF: both are all zero, ready for the rho pass
  bool fin_attr = false, ml_attr = false;
  unsigned long long* npartial = nullptr;   // rows of R that are neither empty nor all ones
  unsigned long long n_partial = 0;
  unsigned* xmax = nullptr;
  double* elbo_dev = nullptr;  // [0] elbo
  double* fin_g = nullptr;     // [L][2 KMAX + 2] scratch of k_fin_gamma
  double* nu_acc = nullptr;    // [2 + L] the nu update inside the sweep (SlArgs::nu_acc)
  double* lutg = nullptr;      // wide masks (W > 4): the nibble LUT of E[theta] lives in global memory [L][W*256]
  bool have_priors = false, have_state = false;
  bool serial = false;
  int ncu = 256;
  std::vector<std::pair<const void*, int>> occ;   // kernel -> resident workgroups per CU   // a rho sub-step left an unconsumed nu partial in slotR
  // profiling
  bool prof = false;
  unsigned prof_mask = ~0u;    // kernel classes whose launches are bracketed by events (vmr_profile)
  struct Ev { int cls; hipEvent_t a, b; };
  std::vector<Ev> evs;
  double prof_ms[VMR_KERNEL_COUNT];
  int64_t prof_n[VMR_KERNEL_COUNT];
};

extern thread_local std::string g_create_err;   // create-time errors (vmr_last_error(NULL)); defined in vimure_hip.hip

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
      (void)hipGetLastError(); /* do not leave the error sticky for the next call */ \
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


// weight of the theta*lambda part of a report (model.py:685-693): z1 / (z1 + z2), 0-safe
__device__ __forceinline__ double w1_of(double z1, double z2) {
  double den = z1 + z2;
  den = (den == 0.0) ? 1.0 : den;
  return z1 / den;
}


// 1/d: v_rcp_f64 (4.6e-8) + VMR_NR_STEPS Newton steps (one: 2e-15, two: exact); the IEEE divide costs ~2x
__device__ __forceinline__ double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
#pragma unroll
  for (int i = 0; i < VMR_NR_STEPS; ++i) r = fma(fma(-d, r, 1.0), r, r);
  return r;
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int lane) {
  unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, lane);
  unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  return __longlong_as_double((long long)readlane64((uint64_t)__double_as_longlong(v), lane));
}


#ifndef NH
#define NH 8
#endif
// NH copies of H in global memory (workgroup gb adds into copy gb % NH): spreads the atomics
#ifndef HC_MAX
#define HC_MAX 3
#endif
// HC_MAX: mirror-count levels cached in LDS when they fit (93 % of the reports at BASELINE config 3); Geo.hc

// Per-report factor of the rho update, a function of (reporter m, mirror count y, category k) only:
//   F[l][y][m][k] = (E[log theta_lm] + E[log lambda_lk]) * w1_k(m, y),   w1 = z1 / (z1 + z2), den == 0 -> 1   (model.py:685-693, 911-921)
// so a report contributes x * F to its tie's U_k: one table read and K multiplies instead of a reciprocal per report.
// Built once per sweep for all levels (a few KB..MB, L2-resident); the rho pass keeps the populous low levels in LDS.
__device__ __forceinline__ double f_entry(int mut, double lth, double gth, double lla, double gla, double gnu, int y) {
  return (lth + lla) * (mut ? w1_of(gth * gla, gnu * (double)y) : 1.0);   // mutuality off: data_z1 = x (model.py:680)
}

// Orders a wave's LDS traffic across lanes.  The LDS executes one wave's operations in issue order (a ds_read issued
// after another lane's ds_add to the same address sees it), so all that is needed is that the COMPILER keeps them in
// program order.  A __builtin_amdgcn_fence here -- even at wavefront scope -- also emits s_waitcnt vmcnt(0), which
// drains the next step's prefetched global loads at every call and serialises memory latency with the walks.
__device__ __forceinline__ void wave_sync() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

// block-wide sum for any block size (<= 1024 threads); result valid in thread 0.  `red` = 16 doubles of LDS.
__device__ __forceinline__ double block_sum_n(double v, double* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    for (unsigned w = 0; w < (blockDim.x >> 6); ++w) r += red[w];
  }
  return r;
}

// log(x) for positive normal x (what the ELBO terms feed it: x >= eps), < 1 ulp: the classic reduction x = 2^k m,
// m in [sqrt(1/2), sqrt(2)), f = m - 1, s = f / (2 + f), log(1 + f) = f - f^2/2 + s (f^2/2 + R(s^2)) with the degree-14 even
// minimax R of FreeBSD msun's e_log.c (coefficients Lg1..Lg7 are that algorithm's published constants).  About 40 VALU
// instructions against ~100 of the library's log(), which also serves zero, subnormal, negative and infinite arguments;
// NaN propagates.  One evaluation per report on ELBO sweeps.
__device__ __forceinline__ double log_pos(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  double m = __builtin_amdgcn_frexp_mant(x);   // [0.5, 1)
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

// Table-driven exp / log for the rho pass (tables in LDS, filled by sp_math_tables): less than half the instructions of the
// polynomial-only versions, and a shorter dependent chain per tie.
//   exp_tab(x), |x| < 700:  n = rint(x 64/ln2), r = x - n ln2/64 (|r| <= 0.0055), exp(x) = 2^(n>>6) T[n&63] (1 + r + .. + r^5/120);
//                            truncation r^6/720 < 4e-17, measured against the library over [-700, 700]: <= 1 ulp
//   log_tab(x), x > 0 normal: x = 2^k m, m in [0.5, 1), c = midpoint of m's 1/256-wide cell, r = m/c - 1 (|r| <= 2^-8, 1/c tabulated),
//                            log x = k ln2 + log c + (r - r^2/2 + .. - r^6/6); truncation 2^-56/7: ABSOLUTE error ~2e-16 (what sums of
//                            ELBO terms need; near x = 1 the relative error is large, unlike log_pos)
#ifndef SP_TABLE_MATH
#define SP_TABLE_MATH 1
#endif
#define SP_MATH_DOUBLES (64 + 256)
__device__ __forceinline__ void sp_math_tables(double* xt /*64*/, double* lt /*128 x (1/c, log c)*/, int tid, int nthr, bool with_log = true) {
  for (int j = tid; j < 64; j += nthr) xt[j] = exp2((double)j * (1.0 / 64.0));
  if (!with_log) return;
  for (int i = tid; i < 128; i += nthr) {
    const double c = 0.5 + ((double)i + 0.5) * (1.0 / 256.0);
    lt[2 * i] = 1.0 / c;
    lt[2 * i + 1] = log(c);
  }
}
__device__ __forceinline__ double exp_tab(double x, const double* xt) {
  const double nf = __builtin_rint(x * 92.33248261689366);            // 64 / ln 2
  double r = fma(-nf, 0x1.62e42fee00000p-7, x);                        // ln2/64, upper 32 bits: nf * hi is exact
  r = fma(-nf, 2.9815858269852933e-12, r);
  const int n = (int)nf;
  const double t = xt[n & 63];
  const double q = r * fma(r, fma(r, fma(r, fma(r, 1.0 / 120.0, 1.0 / 24.0), 1.0 / 6.0), 0.5), 1.0);
  return __builtin_amdgcn_ldexp(fma(t, q, t), n >> 6);
}
__device__ __forceinline__ double log_tab(double x, const double* lt) {
  const double m = __builtin_amdgcn_frexp_mant(x);   // [0.5, 1)
  const int k = __builtin_amdgcn_frexp_exp(x);
  const int i = (__double2hiint(m) >> 13) & 127;
  const double2 cl = *reinterpret_cast<const double2*>(lt + 2 * i);
  const double r = fma(m, cl.x, -1.0);
  const double p = r * fma(r, fma(r, fma(r, fma(r, fma(r, -1.0 / 6.0, 0.2), -0.25), 1.0 / 3.0), -0.5), 1.0);
  return fma((double)k, 0.6931471805599453, cl.y + p);
}

// K consecutive doubles of a [.][K] array: 16-byte accesses when K is even (the arrays are 256-byte aligned)
template <int K>
__device__ __forceinline__ void load_k(const double* __restrict__ p, double (&v)[K]) {
  if (K % 2 == 0) {
#pragma unroll
    for (int k = 0; k < K; k += 2) {
      const double2 t = *reinterpret_cast<const double2*>(p + k);
      v[k] = t.x; v[k + 1 < K ? k + 1 : k] = t.y;
    }
  } else {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = p[k];
  }
}
template <int K>
__device__ __forceinline__ void store_k(double* __restrict__ p, const double (&v)[K]) {
  if (K % 2 == 0) {
#pragma unroll
    for (int k = 0; k < K; k += 2) *reinterpret_cast<double2*>(p + k) = make_double2(v[k], v[k + 1 < K ? k + 1 : k]);
  } else {
#pragma unroll
    for (int k = 0; k < K; ++k) p[k] = v[k];
  }
}

// Philox4x32-10 (Salmon et al. 2011): the counter-based generator of the device sampler and the device generator
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const unsigned n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
static inline int fail(vmr_handle h, int code, const char* msg) {
  if (h) h->err = msg; else g_create_err = msg;
  return code;
}

struct Prof {
  vmr_ctx* h; int cls; hipStream_t st; hipEvent_t a = nullptr, b = nullptr;
  Prof(vmr_ctx* h_, int c, hipStream_t st_ = nullptr) : h(h_), cls(c), st(st_ ? st_ : h_->stream) {
    if (h->prof && ((h->prof_mask >> cls) & 1u)) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, st); }
  }
  ~Prof() {
    if (a) { (void)hipEventRecord(b, st); h->evs.push_back({cls, a, b}); }
  }
};

// Opt in to > 48 KB of dynamic LDS and size the (persistent) grid to what is resident at once:
// workgroups per layer = resident workgroups per CU x CUs / L, never more than tile pairs.
template <class Kern>
static int grid_per_layer(vmr_ctx* h, Kern k, size_t smem, int* gl, long long cap = 0, int tpb = TPB) {
  const void* fn = reinterpret_cast<const void*>(k);
  int per_cu = 0;
  for (auto& e : h->occ) if (e.first == fn) per_cu = e.second;
  if (!per_cu) {
    if (smem > 48 * 1024) HIPCHK(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, tpb, smem));
    if (per_cu < 1) per_cu = 1;
    h->occ.push_back({fn, per_cu});
  }
  long long gl_ = (long long)per_cu * h->ncu / h->g.L;
  if (gl_ < 1) gl_ = 1;
  if (cap <= 0) cap = h->g.P;
  if (gl_ > cap) gl_ = cap;
  *gl = (int)gl_;
  return VMR_OK;
}

// in-place exclusive scan of n u32 items on the handle's stream (bsum: scratch of ceil(n / 2048) items); vimure_hip.hip
int scan_u32(vmr_ctx* h, unsigned* a, unsigned* bsum, size_t n);

#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { g_create_err = std::string(#call) + ": " + hipGetErrorString(e_); (void)hipGetLastError(); return VMR_EHIP; } } while (0)

#endif  // VMR_INTERNAL_H