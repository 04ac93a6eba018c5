// libvimure_hip.so -- MI355X (gfx950 / CDNA4) CAVI engine for the VIMuRe model.
//
// Implements the reference's hot path (latentnetworks/vimure, src/python/vimure/model.py:
// _update_CAVI :623-660, _update_cache :662-696, _update_gamma :698-727, _update_phi :729-761,
// _update_rho :763-818, _update_nu :820-830, __ELBO :948-1019) as FP64 streaming passes over a
// dense uint8 report tensor X[L,N,N,Mp] and a bit-packed reporter mask R, behind the C-ABI of
// include/vimure_hip.h.  Design notes, data layout and byte accounting: DESIGN.md.
//
// One sweep touches X once:  k_gamma_mask (partial mask rows) -> k_fin_gamma -> k_rho -> k_fin_rho.
//   k_rho          "tile-pair" sweep over X: a workgroup stages the (I,J) tile of ties and its mirror (J,I) in
//                  LDS so that X[l,j,i,m] (the reference's data_T_vals) is an LDS byte read; the non-zero counts
//                  of the pair are compacted into wave-local queues and walked twice: per-tie sums for the rho
//                  update, then the sufficient statistics H[l,y,m,k] of the new rho (and the ELBO data terms).
//   k_hist         the same sweep building H from the current rho (start of a realisation).
//   k_gamma_mask   A[l,m,k] = sum_ij R rho: mask words become the EXEC mask of K v_add_f64 (lane <-> reporter).
//   k_fin_*        one workgroup per layer: gamma, phi, nu from H and A, the Gamma expectations
//                  (digamma/log/exp), ELBO assembly -- no host round trip inside a sweep.
#include "vmr_internal.h"
#include "sweep_sl.h"
#include "sweep_gen.h"

thread_local std::string g_create_err;



// Non-zero-byte flags of a 16-byte chunk, 16 bits spread over a dword: bit 8*b + i (+4 when hi) is set
// iff byte b of dword i is non-zero.  Two chunks (hi = 0/1) share one dword, four one 64-bit word.
template <int SH>   // SH = 0 or 4
__device__ __forceinline__ unsigned nz_flags16(uint4 v) {
  const unsigned M = 0x7f7f7f7fu;
  unsigned t0 = ((v.x & M) + M) | v.x, t1 = ((v.y & M) + M) | v.y;
  unsigned t2 = ((v.z & M) + M) | v.z, t3 = ((v.w & M) + M) | v.w;   // bit 7 of each byte = byte != 0
  unsigned f = (t0 >> (7 - SH)) & (0x01010101u << SH);
  f |= (t1 >> (6 - SH)) & (0x02020202u << SH);
  f |= (t2 >> (5 - SH)) & (0x04040404u << SH);
  f |= (t3 >> (4 - SH)) & (0x08080808u << SH);
  return f;
}
// flag position (0..63 in a 64-bit word of four chunks) -> chunk-in-word (0..3) and byte-in-chunk (0..15)
__device__ __forceinline__ void flag_pos(int bit, int& chunk, int& byte) {
  const int lo = bit & 31;                       // position inside the dword of a chunk pair
  chunk = ((bit >> 5) << 1) | ((lo >> 2) & 1);   // dword half, then the +4 flag
  byte = ((lo & 3) << 2) | (lo >> 3);            // dword i = lo & 3, byte b = lo >> 3  ->  4*i + b
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

// Register-staged stream of tile pairs.  The 16-B chunks a thread stages are the same slots of
// every tile pair, so their byte offsets from the two sub-tile bases (I,J) / (J,I) and their LDS
// offsets are computed once; fetch() is one saddr+voffset global_load_dwordx4 per slot and the
// loads stay in flight while the previous tile pair is processed.  Reads never need a bounds
// check: X is allocated with b*N+b rows of slack, and rows outside the network are never used.
template <int PF>
struct TileStream {
  uint4 buf[PF];
  unsigned goff[PF];
  unsigned loff[PF];
  unsigned valid, second;   // bit u: slot exists / belongs to the mirrored sub-tile
  __device__ __forceinline__ void init(const Geo& g) {
    valid = 0; second = 0;
    const int bb = g.b * g.b;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      int q = threadIdx.x + u * TPB;
      int tau = q / g.nchunk, c = q - tau * g.nchunk;
      bool ok = tau < g.nt, sec = tau >= bb;
      int v = sec ? tau - bb : tau;
      int p = v >> g.lb, qq = v & (g.b - 1);
      goff[u] = ok ? (unsigned)((p * g.N + qq) * g.Mp + c * 16) : 0u;
      loff[u] = ok ? (unsigned)(tau * g.stride + c * 16) : 0u;
      valid |= (ok ? 1u : 0u) << u;
      second |= (sec ? 1u : 0u) << u;
    }
  }
  __device__ __forceinline__ void fetch(const uint8_t* __restrict__ Xl, const Geo& g, int I0, int J0) {
    const uint8_t* baseA = Xl + ((size_t)I0 * g.N + J0) * g.Mp;
    const uint8_t* baseB = Xl + ((size_t)J0 * g.N + I0) * g.Mp;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      // (assign every slot unconditionally: a conditionally written array would live in scratch)
      uint4 v = make_uint4(0, 0, 0, 0);
      if ((valid >> u) & 1u) {
        const uint8_t* base = ((second >> u) & 1u) ? baseB : baseA;
        v = *reinterpret_cast<const uint4*>(base + goff[u]);
      }
      buf[u] = v;
    }
  }
  __device__ __forceinline__ void store(unsigned char* xt) {
#pragma unroll
    for (int u = 0; u < PF; ++u)
      if ((valid >> u) & 1u) *reinterpret_cast<uint4*>(xt + loff[u]) = buf[u];
  }
};

// The same staging for the tile pair's mask rows (64-bit words), 3 words per thread at most.
struct MaskStream {
  uint64_t buf[3];
  unsigned goff[3];
  unsigned valid, second;
  __device__ __forceinline__ void init(const Geo& g) {
    valid = 0; second = 0;
    const int bb = g.b * g.b;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      int q = threadIdx.x + u * TPB;
      int tau = q / g.W, w = q - tau * g.W;
      bool ok = tau < g.nt, sec = tau >= bb;
      int v = sec ? tau - bb : tau;
      int p = v >> g.lb, qq = v & (g.b - 1);
      goff[u] = ok ? (unsigned)((p * g.N + qq) * g.W + w) : 0u;
      valid |= (ok ? 1u : 0u) << u;
      second |= (sec ? 1u : 0u) << u;
    }
  }
  __device__ __forceinline__ void fetch(const uint64_t* __restrict__ Rl, const Geo& g, int I0, int J0) {
    const uint64_t* baseA = Rl + ((size_t)I0 * g.N + J0) * g.W;
    const uint64_t* baseB = Rl + ((size_t)J0 * g.N + I0) * g.W;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      uint64_t v = 0ull;
      if ((valid >> u) & 1u) v = (((second >> u) & 1u) ? baseB : baseA)[goff[u]];
      buf[u] = v;
    }
  }
  __device__ __forceinline__ void store(uint64_t* rw) {
#pragma unroll
    for (int u = 0; u < 3; ++u)
      if ((valid >> u) & 1u) rw[threadIdx.x + u * TPB] = buf[u];
  }
};

// Weights of the cache refresh (model.py:685-693) without a divide per report:
//   w1 = z1/(z1 + z2) = 1/(1 + c y),  c[m][k] = G_nu / (G_theta[m] G_lambda[k])  (LDS table, built per launch)
//   w2 = z2/(z1 + z2) = c y w1.
// y = 0 gives exactly 1; a z1 that underflows to 0 gives c = inf and w1 = 0 (the reference's den==0 -> 1 rule).
template <int K>
__device__ __forceinline__ void build_ct(double* ct, const double* Gth, const double (&Gla)[K], double gnu, int Mp) {
  for (int m = threadIdx.x; m < Mp; m += TPB) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      double z1 = Gth[m] * Gla[k];
      ct[(size_t)m * K + k] = (z1 == 0.0) ? (double)INFINITY : gnu / z1;
    }
  }
}
template <int K>
__device__ __forceinline__ void weights(double (&w)[K], double (&cy)[K], const double* ct, int m, unsigned y) {
  const double dy = (double)y;
  const double* p = ct + (size_t)m * K;
  if (K == 2) {   // one reciprocal for both categories: 1/a0 = a1/(a0 a1)
    const double c0 = p[0], c1 = p[1];
    if (c0 < (double)INFINITY && c1 < (double)INFINITY) {
      cy[0] = c0 * dy; cy[1] = c1 * dy;
      const double a0 = 1.0 + cy[0], a1 = 1.0 + cy[1];
      const double r = fast_rcp(a0 * a1);
      w[0] = a1 * r; w[1] = a0 * r;   // y = 0: a0 = a1 = 1 and r = 1 exactly
      return;
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double c = p[k];
    cy[k] = c * dy;
    const double r = fast_rcp(1.0 + cy[k]);
    w[k] = (y == 0) ? ((c < (double)INFINITY) ? 1.0 : 0.0) : r;
  }
}

// the same weights from cb = G_nu / G_theta_m and 1 / G_lambda_k (report lists: one table entry per reporter)
template <int K>
__device__ __forceinline__ void weights_cb(double (&w)[K], double cb, const double (&iGla)[K], unsigned y) {
  const double dy = (double)y;
  double c[K];
#pragma unroll
  for (int k = 0; k < K; ++k) c[k] = cb * iGla[k];
  if (K == 2) {   // one reciprocal for both categories: 1/a0 = a1/(a0 a1)
    if (c[0] < (double)INFINITY && c[1] < (double)INFINITY) {
      const double a0 = 1.0 + c[0] * dy, a1 = 1.0 + c[1] * dy;
      const double r = fast_rcp(a0 * a1);
      w[0] = a1 * r; w[1] = a0 * r;   // y = 0: a0 = a1 = 1 and r = 1 exactly
      return;
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double r = fast_rcp(1.0 + c[k] * dy);
    w[k] = (y == 0) ? ((c[k] < (double)INFINITY) ? 1.0 : 0.0) : r;
  }
}


// Per-tie accumulators a report may feed (reduced over the wave when a heavy share is processed cooperatively)
struct NoAcc {
  __device__ __forceinline__ void zero() {}
  __device__ __forceinline__ void wave_reduce() {}
};

#define QCAP 6   // a lane queues at most this many non-zero dwords; larger shares are "heavy"

// The non-zero counts of the tile rows owned by one wave.  Lane (tau, s) owns the 16-B chunks {s, s+S, ..}
// of row tau.
//   build(): phase 1 (no divergence): one flag per DWORD of the lane's chunks (v_min_u32 + v_lshl_or); then the
//            flags of all 64 lanes are compacted into a wave-local LDS queue of (lane, dword) entries
//            (ballot/mbcnt prefix sums) -- without it a wave runs max-over-lanes(non-zeros) trips at ~30 % lane
//            utilisation.  Heavy shares (a true tie is reported by most reporters: ~100 non-zeros in a row whose
//            neighbours have 2-5) are not queued.
//   walk():  every lane takes queue entries; f(tau_e, m, x, acc) handles one report of tie slot tau_e,
//            commit(tau_e, acc) adds acc to that tie's sums.  Heavy shares: all lanes take one dword each, sums
//            come back through a wave reduction.  walk() may be called several times per build().
// Both must be called by every lane of the wave (act = false for lanes without a tie).
struct TileScan {
  unsigned lo, hi, tot;
  bool heavy, simple;

  __device__ __forceinline__ void build(const unsigned char* xt, const Geo& g, int tau, int s, bool act,
                                        unsigned short* wq) {
    const int S = g.S, nchunk = g.nchunk;
    const unsigned char* row = xt + tau * g.stride;
    lo = 0; hi = 0; tot = 0; heavy = false;
    simple = (nchunk + S - 1) / S > 12;   // huge M (b = 1): plain chunk-by-chunk walk, nothing to build
    if (simple) return;
    // phase 1: bit 4*j + i <-> dword i of chunk s + j*S   (lo: chunks 0..7, hi: chunks 8..11)
    if (act) {
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        if (j * S < nchunk) {   // wave-uniform
          const int c = s + j * S;
          const int cl = c < nchunk ? c : nchunk - 1;
          const uint4 v = *reinterpret_cast<const uint4*>(row + cl * 16);
          unsigned f4 = min(v.x, 1u) | (min(v.y, 1u) << 1) | (min(v.z, 1u) << 2) | (min(v.w, 1u) << 3);
          f4 = c < nchunk ? f4 : 0u;
          if (j < 8) lo |= f4 << (4 * j); else hi |= f4 << (4 * (j - 8));
        }
      }
    }
    const int cnt = __popc(lo) + __popc(hi);   // non-zero dwords of the share
    if (g.dbg & 4) { lo = (cnt == 0x7fffffff) ? 1u : 0u; hi = 0; return; }   // timing experiment: phase 1 only
    const int lane = threadIdx.x & 63;
    heavy = cnt > QCAP;
    const unsigned cl_ = heavy ? 0u : (unsigned)cnt;   // 0..QCAP (< 8)
    unsigned pre = 0;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const uint64_t bal = __ballot((cl_ >> b) & 1u);
      pre += __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u)) << b;
      tot += (unsigned)__popcll(bal) << b;
    }
    if (!heavy) {
      unsigned l2 = lo, h2 = hi, j = pre;
      while (l2 | h2) {
        int p;
        if (l2) { p = __builtin_ctz(l2); l2 &= l2 - 1; } else { p = 32 + __builtin_ctz(h2); h2 &= h2 - 1; }
        wq[j++] = (unsigned short)((lane << 6) | p);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }

  template <class TieAcc, class F, class C>
  __device__ __forceinline__ void walk(const unsigned char* xt, const Geo& g, int tau, int s, bool act,
                                       const unsigned short* wq, F&& f, C&& commit) const {
    const int S = g.S, nchunk = g.nchunk;
    if (simple) {
      if (!act) return;
      const unsigned char* row = xt + tau * g.stride;
      TieAcc a;
      a.zero();
      for (int c = s; c < nchunk; c += S) {
        unsigned nzm = nz_flags16<0>(*reinterpret_cast<const uint4*>(row + c * 16));
        while (nzm) {
          int bit = __builtin_ctz(nzm), ch, by;
          nzm &= nzm - 1;
          flag_pos(bit, ch, by);
          f(tau, c * 16 + by, (unsigned)row[c * 16 + by], a);
        }
      }
      commit(tau, a);
      return;
    }
    const int lane = threadIdx.x & 63;
    const int wave_base = threadIdx.x & ~63;
    for (unsigned e0 = 0; e0 < tot; e0 += 64) {   // wave-uniform trip count
      const unsigned e = e0 + lane;
      if (e < tot) {
        const unsigned ent = wq[e];
        const int ls = ent >> 6, p = ent & 63;
        const int tau_e = (wave_base | ls) >> g.lS, s_e = ls & (S - 1);
        const int off = (s_e + (p >> 2) * S) * 16 + (p & 3) * 4;
        unsigned d = *reinterpret_cast<const unsigned*>(xt + tau_e * g.stride + off);
        TieAcc a;
        a.zero();
        while (d) {
          const int sh = __builtin_ctz(d) & ~7;
          const unsigned x = (d >> sh) & 0xffu;
          d &= ~(0xffu << sh);
          f(tau_e, off + (sh >> 3), x, a);
        }
        commit(tau_e, a);
      }
    }
    // heavy shares: lane p < 48 takes dword p of the share
    uint64_t hm = __ballot(heavy);
    while (hm) {
      const int h = __builtin_ctzll(hm);
      hm &= hm - 1;
      const int tau_h = (wave_base | h) >> g.lS, s_h = h & (S - 1);
      const unsigned lo_h = __builtin_amdgcn_readlane((int)lo, h), hi_h = __builtin_amdgcn_readlane((int)hi, h);
      TieAcc a;
      a.zero();
      const bool mine = lane < 32 ? ((lo_h >> lane) & 1u) : (lane < 48 ? ((hi_h >> (lane - 32)) & 1u) : false);
      if (mine) {
        const int off = (s_h + (lane >> 2) * S) * 16 + (lane & 3) * 4;
        unsigned d = *reinterpret_cast<const unsigned*>(xt + tau_h * g.stride + off);
        while (d) {
          const int sh = __builtin_ctz(d) & ~7;
          const unsigned x = (d >> sh) & 0xffu;
          d &= ~(0xffu << sh);
          f(tau_h, off + (sh >> 3), x, a);
        }
      }
      a.wave_reduce();
      if (lane == 0) commit(tau_h, a);
    }
    __builtin_amdgcn_wave_barrier();
  }
};

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
                        uint8_t* __restrict__ rcls,
                        unsigned long long* sumx, unsigned* xmax, unsigned long long* npartial, size_t rows, int Mp,
                        int W, int M) {
  unsigned long long local = 0, lpart = 0, lempty = 0;
  unsigned lmax = 0;
  for (size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x; r < rows; r += (size_t)gridDim.x * blockDim.x) {
    bool anyx = false, anyr = false;
    const uint4* p = reinterpret_cast<const uint4*>(X + r * Mp);
    for (int c = 0; c < Mp / 16; ++c) {
      uint4 v = p[c];
      if (v.x | v.y | v.z | v.w) {
        anyx = true;
        unsigned d[4] = {v.x, v.y, v.z, v.w};
        for (int u = 0; u < 4; ++u) {
          local += (d[u] & 0xff) + ((d[u] >> 8) & 0xff) + ((d[u] >> 16) & 0xff) + (d[u] >> 24);
          lmax = max(lmax, max(max(d[u] & 0xff, (d[u] >> 8) & 0xff), max((d[u] >> 16) & 0xff, d[u] >> 24)));
        }
      }
    }
    bool full = true;
    for (int w = 0; w < W; ++w) {
      const uint64_t bits = Rb[r * W + w];
      const int rem = M - w * 64;
      const uint64_t fm = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
      anyr |= bits != 0;
      full = full && (bits == fm);
    }
    cov[r] = (anyx && anyr) ? 1 : 0;
    rcls[r] = !anyr ? 0 : (full ? 1 : 2);
    if (anyr && !full) ++lpart;
    if (!anyr) ++lempty;
  }
  if (local) atomicAdd(sumx, local);
  if (lpart) atomicAdd(npartial, lpart);
  if (lempty) atomicAdd(npartial + 1, lempty);   // [1]: rows with no reporter at all
  if (lmax) atomicMax(xmax, lmax);
}

__global__ void k_init_rho(const double* __restrict__ pr, double* __restrict__ rho, double* __restrict__ logpr,
                           size_t n, double eps) {
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
    double v = pr[q];
    rho[q] = v;
    logpr[q] = log(v + eps);
  }
}

// the same for a handle whose rho / log prior live in sorted position order (sweep_sl.h): pr is in tie order
// (rs / not_onehot: does every tie of a step WITHOUT reports carry the one-hot prior (1, 0, .., 0) the reference gives ties nobody
// reported on (model.py:536-556)?  Then the sweeps need not read the log prior of those steps -- SlArgs::lp0)
__global__ void k_init_rho_pos(const double* __restrict__ pr, double* __restrict__ rho, double* __restrict__ logpr,
                               const unsigned* __restrict__ perm, size_t T, size_t NS, int L, int K, double eps,
                               const unsigned* __restrict__ rs, int* __restrict__ not_onehot) {
  const size_t n = (size_t)L * T * K;
  bool bad = false;
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
    const size_t row = q / K, k = q - row * K, l = row / T, pos = row - l * T;
    const double v = pr[(l * T + perm[l * NS * 64 + pos]) * K + k];
    rho[q] = v;
    logpr[q] = log(v + eps);
    const unsigned* rsl = rs + l * (NS + 1);
    const size_t st = pos >> 6;
    if (rsl[st + 1] == rsl[st] && v != (k == 0 ? 1.0 : 0.0)) bad = true;
  }
  if (bad) atomicOr(not_onehot, 1);
}

// nibble LUT of E[theta] for wide masks: lut[l][n][e] = sum of E[theta_m] over the set bits e of reporters 4n..4n+3
__global__ void k_build_lut(const double* __restrict__ par, double* __restrict__ lutg, Geo g) {
  const ParOff o = par_off(g.L, g.Mp, g.K);
  const int l = blockIdx.x;
  const double* Eth = par + o.E_th + (size_t)l * g.Mp;
  for (int q = threadIdx.x; q < g.W * 256; q += blockDim.x) {
    int n = q >> 4, e = q & 15;
    double v = 0.0;
    for (int u = 0; u < 4; ++u) {
      int m = n * 4 + u;
      if (((e >> u) & 1) && m < g.Mp) v += Eth[m];
    }
    lutg[(size_t)l * g.W * 256 + q] = v;
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
// Output: slotA[l][slot][m][k] accumulated with global f64 atomics (zeroed by k_fin_gamma).
// ------------------------------------------------------------------------------------------
// A wave takes 64 consecutive ties per trip, lane <-> tie for the (coalesced, prefetched) loads of
// the R-row words and rho.  Rows whose words are all ones only feed a per-lane sum (one v_add_f64
// per k and 64 ties); every other row is broadcast with v_readlane and added under EXEC = its bits
// (lane <-> reporter).
template <int K, int NC>
__global__ __launch_bounds__(TPB) void k_gamma_mask(const uint64_t* __restrict__ Rb, const double* __restrict__ rho,
                                                    const uint8_t* __restrict__ rcls, double* __restrict__ slotA,
                                                    int skip_full, const unsigned* __restrict__ perm /* rho by sorted position, or null */, Geo g) {
  __shared__ double sacc[NC * 64 * K];
  const int l = blockIdx.x / g.Gm, gb = blockIdx.x - l * g.Gm;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const long long T = (long long)g.N * g.N;
  const long long nw = (long long)g.Gm * (TPB / 64), gw = (long long)gb * (TPB / 64) + wave;
  const long long t0 = gw * T / nw, t1 = (gw + 1) * T / nw;
  const uint64_t* Rl = Rb + (size_t)l * T * g.W;
  const double* rl = rho + (size_t)l * T * K;
  const uint8_t* cl = rcls + (size_t)l * T;
  const unsigned* pl = perm ? perm + (size_t)l * ((T + 63) / 64) * 64 : nullptr;
  const int Wp = g.W * 64;
  for (int cg = 0; cg < g.W; cg += NC) {
    const int nc = min(NC, g.W - cg);   // == NC except in the last group of a wide mask
    double acc[NC][K], accF[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      accF[k] = 0.0;
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c][k] = 0.0;
    }
    uint64_t wq[NC];
    double rq[K];
    unsigned cq = 0;   // the row's class: "all ones" must mean the whole row, not just this group's words
    auto fetch = [&](long long tb) {
      long long t = tb + lane;
      bool ok = t < t1;
      const long long pc = ok ? t : t1 - 1;              // position (= tie when rho is in tie order)
      const long long tc = pl ? (long long)pl[pc] : pc;   // the tie whose mask row this is
#pragma unroll
      for (int c = 0; c < NC; ++c) wq[c] = (ok && c < nc) ? Rl[tc * g.W + cg + c] : 0ull;
#pragma unroll
      for (int k = 0; k < K; ++k) rq[k] = ok ? rl[pc * K + k] : 0.0;
      cq = ok ? (unsigned)cl[tc] : 0u;
    };
    if (t0 < t1) fetch(t0);
    for (long long tb = t0; tb < t1; tb += 64) {
      uint64_t w[NC];
      double r[K];
#pragma unroll
      for (int c = 0; c < NC; ++c) w[c] = wq[c];
#pragma unroll
      for (int k = 0; k < K; ++k) r[k] = rq[k];
      const bool full = cq == 1u;
      if (tb + 64 < t1) fetch(tb + 64);   // in flight during this batch
      bool any = false;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (c < nc) any = any || (w[c] != 0ull);
      }
#pragma unroll
      for (int k = 0; k < K; ++k) accF[k] += (full && !skip_full) ? r[k] : 0.0;   // skip_full: the rho pass summed them
      uint64_t todo = __ballot(any && !full);
      while (todo) {
        int i = __builtin_ctzll(todo);
        todo &= todo - 1;
        double ri[K];
#pragma unroll
        for (int k = 0; k < K; ++k) ri[k] = readlane_f64(r[k], i);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          if (c < nc) {
            uint64_t wi = readlane64(w[c], i);
            if (__builtin_amdgcn_inverse_ballot_w64(wi)) {
#pragma unroll
              for (int k = 0; k < K; ++k) acc[c][k] += ri[k];
            }
          }
        }
      }
    }
    // all-ones rows: every reporter of the group gets the same sum
#pragma unroll
    for (int k = 0; k < K; ++k) {
      double u = wave_sum(accF[k]);
#pragma unroll
      for (int c = 0; c < NC; ++c)
        if (c < nc && (cg + c) * 64 + lane < g.M) acc[c][k] += u;
    }
    // combine the 4 waves in LDS, then one global atomic per (m,k) and workgroup
    for (int q = threadIdx.x; q < NC * 64 * K; q += TPB) sacc[q] = 0.0;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int k = 0; k < K; ++k) atomicAdd(&sacc[(c * 64 + lane) * K + k], acc[c][k]);
    __syncthreads();
    double* out = slotA + (((size_t)l * NSLOT + (gb % NSLOT)) * Wp + cg * 64) * K;
    for (int q = threadIdx.x; q < nc * 64 * K; q += TPB) atomicAdd(&out[q], sacc[q]);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// Sufficient statistics of the reports:  H[l,y,m,k] = sum over ties t of x * rho_k(t), taken over the non-zero
// counts x = X[l,t,m] whose mirrored count X^T[l,t,m] equals y (y = 0 always when mutuality is off).
// Everything the gamma, phi and nu updates need from X is linear in rho with weights that depend on (m, y, k) only:
//   gamma_shp[l,m] = alpha + sum_{y,k} w1_k(m,y) H    (model.py:698-703, 832-859; w1 with the OLD parameters)
//   phi_shp[l,k]   = alpha + sum_{m,y} w1_k(m,y) H    (model.py:731-733, 861-887; w1 with the NEW E[log theta])
//   nu_shp         = alpha + sum_{l,m,y>0,k} w2_k(m,y) H   (model.py:822-825; H of the NEW rho)
// so one pass over X per sweep (the rho pass, which rebuilds H from the new rho) replaces three.  H is tiny
// ([L][Y][Mp][K] doubles, Y = max count + 1); mirror counts 0..HC-1 are accumulated in LDS, the rest with global
// f64 atomics.  k_hist builds H from the current rho (start of a fit, sub-step tests).
// ------------------------------------------------------------------------------------------

struct HistArgs {
  const uint8_t* X; const double* rho; double* Hg;
  int Gl;   // workgroups per layer of this launch
};

// common per-tile bookkeeping of the tile-pair kernels
struct TileIter {
  int I, J, nb, b, lb, bb, N;
  __device__ __forceinline__ void init(const Geo& g, long long p0) {
    nb = g.nb; b = g.b; lb = g.lb; bb = g.b * g.b; N = g.N;
    pair_decode(p0, nb, I, J);
  }
  __device__ __forceinline__ void next() { if (++J == nb) { ++I; J = I; } }
  __device__ __forceinline__ bool diag() const { return I == J; }
  // tie slot tau -> (i,j); false when the slot is unused in this pair or lies outside the network
  __device__ __forceinline__ bool coords(int tau, int& i, int& j) const {
    const bool second = tau >= bb;
    const int u = second ? tau - bb : tau;
    const int p = u >> lb, q = u & (b - 1);
    i = (second ? J : I) * b + p;
    j = (second ? I : J) * b + q;
    return tau < (diag() ? bb : 2 * bb) && i < N && j < N;
  }
  __device__ __forceinline__ int mirror(int tau) const {
    const bool second = tau >= bb;
    const int u = second ? tau - bb : tau;
    const int m = ((u & (b - 1)) << lb) | (u >> lb);
    return diag() ? m : (second ? m : bb + m);
  }
};

// one report into H: LDS cache for small mirror counts, global atomics beyond
template <int K>
__device__ __forceinline__ void hist_add(double* Hc, double* Hl /*this workgroup's copy [Y][Mp][K]*/, int Mp, int m, unsigned y,
                                         double dx, const double* r, unsigned hc) {
  if (y < hc) {
    double* d = Hc + ((size_t)y * Mp + m) * K;
#pragma unroll
    for (int k = 0; k < K; ++k) atomicAdd(&d[k], dx * r[k]);
  } else {
    double* d = Hl + ((size_t)y * Mp + m) * K;
#pragma unroll
    for (int k = 0; k < K; ++k) atomicAdd(&d[k], dx * r[k]);
  }
}
__device__ __forceinline__ void hist_flush(const double* Hc, double* Hl, int n) {
  for (int q = threadIdx.x; q < n; q += TPB) {
    const double v = Hc[q];
    if (v != 0.0) atomicAdd(&Hl[q], v);
  }
}

template <int K, bool MUT, int PF>
__global__ __launch_bounds__(TPB, VMR_LB_COUNTS) void k_hist(HistArgs a, Geo g) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char* xt = smem;
  double* rt = reinterpret_cast<double*>(smem + (size_t)g.nt * g.stride);   // rho of the pair's ties [nt][K]
  double* Hc = rt + (size_t)g.nt * K;                                        // [HC][Mp][K]
  const int nHc = g.hc * g.Mp * K;
  unsigned short* wq = reinterpret_cast<unsigned short*>(Hc + nHc) + (threadIdx.x >> 6) * (64 * QCAP);
  const int l = blockIdx.x / a.Gl, gb = blockIdx.x - l * a.Gl;
  const long long p0 = (long long)gb * g.P / a.Gl, p1 = (long long)(gb + 1) * g.P / a.Gl;
  for (int q = threadIdx.x; q < nHc; q += TPB) Hc[q] = 0.0;
  double* Hl = a.Hg + ((size_t)l * NH + (gb % NH)) * g.Y * g.Mp * K;
  TileIter it;
  it.init(g, p0);
  const int tau = threadIdx.x >> g.lS, s = threadIdx.x & (g.S - 1);
  const uint8_t* Xl = a.X + (size_t)l * g.N * g.N * g.Mp;
  const double* rl = a.rho + (size_t)l * g.N * g.N * K;
  TileStream<PF> ts;
  ts.init(g);
  double rn[K];   // rho of this lane's tie in the NEXT pair (prefetched like the X chunks)
  auto fetch_rho = [&]() {
    int i, j;
    const bool ok = it.coords(tau, i, j);
#pragma unroll
    for (int k = 0; k < K; ++k) rn[k] = ok ? rl[((size_t)i * g.N + j) * K + k] : 0.0;
  };
  if (p0 < p1) { ts.fetch(Xl, g, it.I * g.b, it.J * g.b); fetch_rho(); }
  __syncthreads();
  for (long long p = p0; p < p1; ++p) {
    ts.store(xt);
    int i, j;
    const bool act = it.coords(tau, i, j);
    if (s == 0 && tau < g.nt) {
#pragma unroll
      for (int k = 0; k < K; ++k) rt[tau * K + k] = rn[k];
    }
    const TileIter cur = it;
    __syncthreads();
    it.next();
    if (p + 1 < p1) { ts.fetch(Xl, g, it.I * g.b, it.J * g.b); fetch_rho(); }   // flies while this pair is scanned
    TileScan sc;
    sc.build(xt, g, tau, s, act, wq);
    sc.walk<NoAcc>(xt, g, tau, s, act, wq,
      [&](int te, int m, unsigned x, NoAcc&) {
        const unsigned y = MUT ? (unsigned)xt[cur.mirror(te) * g.stride + m] : 0u;
        hist_add<K>(Hc, Hl, g.Mp, m, y, (double)x, rt + te * K, (unsigned)g.hc);
      },
      [&](int, const NoAcc&) {});
    __syncthreads();
  }
  hist_flush(Hc, Hl, nHc);
}

// ------------------------------------------------------------------------------------------
// rho (+ H of the new rho, + ELBO data terms)   model.py:763-818, 889-923; ELBO :948-995, :1013
// slotR[slot][4]: [1] ELBO linear+entropy terms, [2] ELBO log terms,
//                 [3] sum_t (sum_k rho_k) Q_t  (multiplied by -E[nu] in k_fin_rho)
// Walk 1 over the pair's non-zero counts collects U_k per tie; after the per-tie update the same queue is
// walked again with the NEW rho to rebuild H (and, on ELBO sweeps, the log terms and the mirror sums Q).
// ------------------------------------------------------------------------------------------
struct RhoArgs {
  const uint8_t* X; const uint64_t* Rb; double* rho; const double* logpr; const double* par;
  double* slotR;
  const double* lutg;   // nibble LUT of E[theta], [L][W*256]
  double* Hg;
  double* slotF;
  int Gl;
};

template <int K>
struct SumU {
  double U[K];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int k = 0; k < K; ++k) U[k] = 0.0;
  }
  __device__ __forceinline__ void wave_reduce() {
#pragma unroll
    for (int k = 0; k < K; ++k) U[k] = wave_sum(U[k]);
  }
};
struct SumQ {   // ELBO walk: the mirror tie's masked count
  unsigned q;
  __device__ __forceinline__ void zero() { q = 0u; }
  __device__ __forceinline__ void wave_reduce() {
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) q += __shfl_xor(q, o2, 64);
  }
};

template <int K, bool MUT, bool UPDATE, bool ELBO, int PF>
__global__ __launch_bounds__(TPB, ELBO ? 2 : VMR_LB_RHO) void k_rho(RhoArgs a, Geo g) {   // ELBO variants: 2 resident (LDS), so 256 VGPRs
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char* xt = smem;
  size_t off = (size_t)g.nt * g.stride;
  uint64_t* rw = reinterpret_cast<uint64_t*>(smem + off); off += (size_t)g.nt * g.W * 8;
  double* wsum = reinterpret_cast<double*>(smem + off); off += (size_t)g.W * 8;
  double* lth = reinterpret_cast<double*>(smem + off); off += (size_t)g.Mp * 8;
  double* red = reinterpret_cast<double*>(smem + off); off += 8 * 8;
  double* ct = reinterpret_cast<double*>(smem + off); off += MUT ? (size_t)g.Mp * K * 8 : 0;
  double* ut = reinterpret_cast<double*>(smem + off); off += (size_t)g.nt * K * 8;    // U per tie; exp(rho) in the ELBO walk
  double* rt = reinterpret_cast<double*>(smem + off); off += (size_t)g.nt * K * 8;    // (new) rho per tie
  const bool do_hist = UPDATE && !g.two_pass;
  const int nHc = do_hist ? g.hc * g.Mp * K : 0;
  double* Hc = reinterpret_cast<double*>(smem + off); off += (size_t)nHc * 8;
  double* Gth = reinterpret_cast<double*>(smem + off); off += ELBO ? (size_t)g.Mp * 8 : 0;
  unsigned* qs = reinterpret_cast<unsigned*>(smem + off); off += ELBO ? (size_t)g.nt * 4 : 0;
  unsigned short* wq = reinterpret_cast<unsigned short*>(smem + off) + (threadIdx.x >> 6) * (64 * QCAP);
  const ParOff o = par_off(g.L, g.Mp, g.K);
  const int l = blockIdx.x / a.Gl, gb = blockIdx.x - l * a.Gl;
  const long long p0 = (long long)gb * g.P / a.Gl, p1 = (long long)(gb + 1) * g.P / a.Gl;
  for (int m = threadIdx.x; m < g.Mp; m += TPB) {
    lth[m] = a.par[o.l_th + (size_t)l * g.Mp + m];
    if (ELBO) Gth[m] = a.par[o.G_th + (size_t)l * g.Mp + m];
  }
  for (int q = threadIdx.x; q < nHc; q += TPB) Hc[q] = 0.0;
  // lut[n][e] = sum of E[theta_m] over the set bits e of reporters 4n..4n+3 (global, L1/L2 resident);
  // wsum[w] = sum over the 64 reporters of word w (shortcut for all-ones words)
  const double* lut = a.lutg + (size_t)l * g.W * 256;
  for (int w = threadIdx.x; w < g.W; w += TPB) {
    double v = 0.0;
    for (int n = 0; n < 16; ++n) v += lut[(w * 16 + n) * 16 + 15];
    wsum[w] = v;
  }
  double Ela[K], lla[K], Gla[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    Ela[k] = a.par[o.E_la + l * K + k]; lla[k] = a.par[o.l_la + l * K + k]; Gla[k] = a.par[o.G_la + l * K + k];
  }
  // UPDATE: the weights use the current G_nu; stand-alone ELBO: the stale one (model.py:970)
  const double gnu = a.par[o.sc + (UPDATE ? SC_G_NU : SC_G_NU_STALE)];
  const double eps = g.eps;
  double e_lin = 0.0, e_q = 0.0, e_log = 0.0;
  double accF[K];   // sum of the new rho over this workgroup's ties whose mask row is all ones
#pragma unroll
  for (int k = 0; k < K; ++k) accF[k] = 0.0;
  const int lastrem = g.M - (g.W - 1) * 64;
  const uint64_t lastfull = lastrem >= 64 ? ~0ull : ((1ull << lastrem) - 1ull);
  if (MUT) build_ct<K>(ct, a.par + o.G_th + (size_t)l * g.Mp, Gla, gnu, g.Mp);
  double* Hl = a.Hg + ((size_t)l * NH + (gb % NH)) * g.Y * g.Mp * K;

  TileIter it;
  it.init(g, p0);
  const int tau = threadIdx.x >> g.lS, s = threadIdx.x & (g.S - 1);
  const size_t T = (size_t)g.N * g.N;
  const uint8_t* Xl = a.X + (size_t)l * T * g.Mp;
  const uint64_t* Rl = a.Rb + (size_t)l * T * g.W;
  double* rl = a.rho + (size_t)l * T * K;
  const double* lpl = a.logpr + (size_t)l * T * K;
  TileStream<PF> ts;
  ts.init(g);
  MaskStream ms;
  ms.init(g);
  double lpn[K], rn[K];   // log prior (and rho, ELBO-only) of this lane's tie in the NEXT pair
  auto fetch_tie = [&]() {
    int i, j;
    const bool ok = it.coords(tau, i, j);
    const size_t t_ = ok ? ((size_t)i * g.N + j) : 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      lpn[k] = ok ? lpl[t_ * K + k] : 0.0;
      rn[k] = (!UPDATE && ok) ? rl[t_ * K + k] : 0.0;
    }
  };
  if (p0 < p1) { ts.fetch(Xl, g, it.I * g.b, it.J * g.b); ms.fetch(Rl, g, it.I * g.b, it.J * g.b); fetch_tie(); }
  __syncthreads();
  for (long long p = p0; p < p1; ++p) {
    ts.store(xt);
    ms.store(rw);
    int i, j;
    const bool act = it.coords(tau, i, j);
    const size_t tg = act ? ((size_t)i * g.N + j) : 0;
    const TileIter cur = it;
    double lp[K], r[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { lp[k] = lpn[k]; r[k] = rn[k]; }
    if (UPDATE && s == 0 && tau < g.nt) {
#pragma unroll
      for (int k = 0; k < K; ++k) ut[tau * K + k] = 0.0;
    }
    __syncthreads();
    it.next();
    if (p + 1 < p1) { ts.fetch(Xl, g, it.I * g.b, it.J * g.b); ms.fetch(Rl, g, it.I * g.b, it.J * g.b); fetch_tie(); }
    double Tt = 0.0;
    int rowfull = 1;
    if (act) {
      // T = sum_m R E[theta_m] (model.py:766-792): whole-word shortcut, else nibble look-ups
      const uint64_t* rwt = rw + (size_t)tau * g.W;
      for (int w = s; w < g.W; w += g.S) {
        uint64_t bits = rwt[w];
        rowfull &= (bits == (w == g.W - 1 ? lastfull : ~0ull)) ? 1 : 0;
        if (bits == ~0ull) { Tt += wsum[w]; continue; }
        for (int n = 0; bits != 0; ++n, bits >>= 4) Tt += lut[((w * 16 + n) << 4) + (unsigned)(bits & 15u)];
      }
    }
    Tt = group_sum(Tt, g.S);
    if (UPDATE && g.fuse_full) {
      for (int o2 = g.S >> 1; o2 > 0; o2 >>= 1) rowfull &= __shfl_xor(rowfull, o2, 64);
    }
    TileScan sc;
    sc.build(xt, g, tau, s, act, wq);
    if (UPDATE) {
      sc.walk<SumU<K>>(xt, g, tau, s, act, wq,
        [&](int te, int m, unsigned x, SumU<K>& acc) {
          const double dx = (double)x, lt = lth[m];
          if (MUT) {
            double w[K], cy[K];
            weights<K>(w, cy, ct, m, (unsigned)xt[cur.mirror(te) * g.stride + m]);
#pragma unroll
            for (int k = 0; k < K; ++k) acc.U[k] += (lt + lla[k]) * (dx * w[k]);
          } else {
#pragma unroll
            for (int k = 0; k < K; ++k) acc.U[k] += (lt + lla[k]) * dx;
          }
        },
        [&](int te, const SumU<K>& t) {   // the tie's sums live in LDS; a tie belongs to one wave
#pragma unroll
          for (int k = 0; k < K; ++k) atomicAdd(&ut[te * K + k], t.U[k]);
        });
      double sum = 0.0;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const double u = (tau < g.nt) ? ut[tau * K + k] : 0.0;
        r[k] = exp((lp[k] + u) - Tt * Ela[k]);   // no max-subtraction, as model.py:807
        sum += r[k];
      }
      if (sum > 0.0) {   // model.py:808-811; a true divide: 1/sum overflows when sum is subnormal
#pragma unroll
        for (int k = 0; k < K; ++k) r[k] /= sum;
      }
      if (act && s == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
          rl[tg * K + k] = r[k];
          accF[k] += (g.fuse_full && rowfull) ? r[k] : 0.0;
        }
      }
    }
    if (s == 0 && tau < g.nt) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        rt[tau * K + k] = r[k];
        if (ELBO) ut[tau * K + k] = exp(r[k]);   // exp(rho), model.py:971 (U is consumed)
      }
      if (ELBO) qs[tau] = 0u;
    }
    if (ELBO) __syncthreads(); else __builtin_amdgcn_wave_barrier();   // Q crosses waves, the rest is wave-local
    // walk 2: H of the new rho; ELBO log terms and mirror sums
    if (do_hist || ELBO) sc.walk<SumQ>(xt, g, tau, s, act, wq,
      [&](int te, int m, unsigned x, SumQ& acc) {
        const double dx = (double)x;
        const int mt = cur.mirror(te);
        const unsigned y = MUT ? (unsigned)xt[mt * g.stride + m] : 0u;
        if (do_hist) hist_add<K>(Hc, Hl, g.Mp, m, y, dx, rt + te * K, (unsigned)g.hc);
        if (ELBO) {
          const bool in_r = (rw[te * g.W + (m >> 6)] >> (m & 63)) & 1ull;
          double inner = 0.0;
          if (in_r) {
            const double z2 = gnu * (double)y, gt = Gth[m];
            const double* er = ut + te * K;
#pragma unroll
            for (int k = 0; k < K; ++k) inner += er[k] * (gt * Gla[k] + z2);
          }
          e_log += dx * log(inner + eps);
          if (MUT && ((rw[mt * g.W + (m >> 6)] >> (m & 63)) & 1ull)) acc.q += x;   // R[mirror] X^T[mirror]
        }
      },
      [&](int te, const SumQ& t) {   // Q of the MIRROR tie: sum_m R[mirror,m] X[this,m]
        if (ELBO && t.q) atomicAdd(&qs[cur.mirror(te)], t.q);
      });
    if (ELBO) {
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
    __syncthreads();
  }
  if (do_hist) hist_flush(Hc, Hl, nHc);
  if (UPDATE && g.fuse_full) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      double v = block_sum(accF[k], red);
      if (threadIdx.x == 0) atomicAdd(&a.slotF[((size_t)l * NSLOT + (gb % NSLOT)) * K + k], v);
    }
  }
  if (ELBO) {
    double v1 = block_sum(e_lin, red);
    double v2 = block_sum(e_log, red);
    double v3 = block_sum(e_q, red);
    if (threadIdx.x == 0) {
      double* out = a.slotR + (size_t)(blockIdx.x % NSLOT) * 4;
      atomicAdd(&out[1], v1); atomicAdd(&out[2], v2); atomicAdd(&out[3], v3);
    }
  }
}

// ==========================================================================================
// Report lists (the default data format when X is sparse, which it is: 1.5-5 % of the counts are non-zero)
//
// X is a tensor of COUNTS of which a few per cent are non-zero; every update touches only those, and the only
// thing the mutuality terms need besides a report is the mirrored count X[l,j,i,m].  vmr_create therefore turns
// the dense tensor into one 4-byte entry per non-zero count (layout and limits: sweep_sl.h), sorted by the ties'
// report counts and laid out in steps of 64 ties (sorted_lists.hip); rcls[l][t] = class of the tie's mask row (0 empty,
// 1 all ones, 2 partial; not read when every row is all ones), Qt[l][t] = sum_m R[t,m] X[mirror(t),m] (the ELBO's mirror
// sum, a constant of the data).  A sweep reads 4 B per report + rho/log-prior instead of 1 B per (tie, reporter).  The
// dense tensor is freed once the lists exist.  (Rounds 1-2 kept steps of 64 CONSECUTIVE ties with a scattered rest --
// k_rho_sp, removed in round 3: DESIGN.md section 2.)
// ==========================================================================================
__device__ __forceinline__ unsigned nz_bytes(uint4 v) {
  const unsigned M = 0x7f7f7f7fu;
  unsigned t0 = (((v.x & M) + M) | v.x) & ~M, t1 = (((v.y & M) + M) | v.y) & ~M;
  unsigned t2 = (((v.z & M) + M) | v.z) & ~M, t3 = (((v.w & M) + M) | v.w) & ~M;
  return __popc(t0) + __popc(t1) + __popc(t2) + __popc(t3);
}

// Per tie of one layer (16 lanes per row): rp[t] = its non-zero counts.  The layer total goes to *nnz.
__global__ __launch_bounds__(256) void k_sp_count(const uint8_t* __restrict__ Xl, unsigned* __restrict__ rpl,
                                                  unsigned long long* nnz, Geo g) {
  __shared__ double red[8];
  const int gl = threadIdx.x & 15;
  unsigned long long mine = 0;
  const size_t T = (size_t)g.N * g.N, Tr = (T + 15) / 16 * 16;
  for (size_t t = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 4; t < Tr; t += (size_t)gridDim.x * 16) {
    unsigned c = 0;
    if (t < T) {
      const uint8_t* row = Xl + t * g.Mp;
      for (int ch = gl; ch < g.nchunk; ch += 16) c += nz_bytes(*reinterpret_cast<const uint4*>(row + ch * 16));
    }
    c = group_sum_u(c, 16);
    if (gl == 0 && t < T) { rpl[t] = c; mine += c; }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) rpl[T] = 0u;
  // block total (exact in double: < 2^53)
  double v = block_sum((double)mine, red);
  if (threadIdx.x == 0 && v > 0.0) atomicAdd(nnz, (unsigned long long)v);
}

// in-place exclusive scan of n u32 items in three launches (2048 items per workgroup)
__global__ __launch_bounds__(256) void k_scan_local(unsigned* a, unsigned* bsum, size_t n) {
  __shared__ unsigned sh[256];
  const size_t base = (size_t)blockIdx.x * 2048 + (size_t)threadIdx.x * 8;
  unsigned v[8], s = 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) { unsigned t = (base + u < n) ? a[base + u] : 0u; v[u] = s; s += t; }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o2 = 1; o2 < 256; o2 <<= 1) {
    unsigned t = (threadIdx.x >= (unsigned)o2) ? sh[threadIdx.x - o2] : 0u;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  const unsigned excl = sh[threadIdx.x] - s;
#pragma unroll
  for (int u = 0; u < 8; ++u) if (base + u < n) a[base + u] = v[u] + excl;
  if (threadIdx.x == 255) bsum[blockIdx.x] = sh[255];
}
__global__ __launch_bounds__(256) void k_scan_bsum(unsigned* bsum, int nb) {
  __shared__ unsigned sh[256];
  unsigned carry = 0;
  for (int c0 = 0; c0 < nb; c0 += 256) {
    const int i = c0 + threadIdx.x;
    const unsigned s = (i < nb) ? bsum[i] : 0u;
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o2 = 1; o2 < 256; o2 <<= 1) {
      unsigned t = (threadIdx.x >= (unsigned)o2) ? sh[threadIdx.x - o2] : 0u;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nb) bsum[i] = carry + sh[threadIdx.x] - s;
    carry += sh[255];
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void k_scan_add(unsigned* a, const unsigned* __restrict__ bsum, size_t n) {
  const size_t base = (size_t)blockIdx.x * 2048 + (size_t)threadIdx.x * 8;
  const unsigned add = bsum[blockIdx.x];
#pragma unroll
  for (int u = 0; u < 8; ++u) if (base + u < n) a[base + u] += add;
}
// Write the entries of one layer (16 lanes per tie) and the mirror sums Qt.  rpl is the exclusive scan of the
// per-tie counts: tie t's entries start at rpl[t], reporters ascending.
template <bool MUT>
__global__ __launch_bounds__(256) void k_sp_fill(const uint8_t* __restrict__ Xl, const uint64_t* __restrict__ Rl,
                                                 const unsigned* __restrict__ rpl, unsigned* __restrict__ El, unsigned* __restrict__ El2 /*wide entries: the second words*/,
                                                 unsigned* __restrict__ Qtl, Geo g) {
  const int gl = threadIdx.x & 15;
  const size_t T = (size_t)g.N * g.N, Tr = (T + 15) / 16 * 16;
  for (size_t t = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 4; t < Tr; t += (size_t)gridDim.x * 16) {
    const bool ok = t < T;
    const size_t tc = ok ? t : 0;
    const size_t i = tc / g.N, j = tc - i * g.N, tm = j * g.N + i;
    const uint8_t* row = Xl + tc * g.Mp;
    const uint8_t* mrow = Xl + tm * g.Mp;
    const uint64_t* rr = Rl + tc * g.W;
    const uint64_t* rmr = Rl + tm * g.W;
    size_t pos = rpl[tc];
    unsigned q = 0;
    for (int c0 = 0; c0 < g.nchunk; c0 += 16) {   // uniform over the 16 lanes
      const int ch = c0 + gl;
      uint4 v = make_uint4(0, 0, 0, 0), yv = make_uint4(0, 0, 0, 0);
      if (ok && ch < g.nchunk) {
        v = *reinterpret_cast<const uint4*>(row + ch * 16);
        if (MUT) yv = *reinterpret_cast<const uint4*>(mrow + ch * 16);
      }
      const unsigned xs[4] = {v.x, v.y, v.z, v.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w};
      const unsigned n = nz_bytes(v);
      unsigned incl = n;
#pragma unroll
      for (int o2 = 1; o2 < 16; o2 <<= 1) {
        const unsigned up = __shfl_up(incl, o2, 16);
        if (gl >= o2) incl += up;
      }
      const unsigned tot = __shfl(incl, 15, 16);
      size_t w = pos + incl - n;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        unsigned dw = xs[u];
        while (dw) {
          const int sh = __builtin_ctz(dw) & ~7;
          const unsigned x = (dw >> sh) & 0xffu;
          dw &= ~(0xffu << sh);
          const int m = ch * 16 + u * 4 + (sh >> 3);
          const unsigned inr = (unsigned)((rr[m >> 6] >> (m & 63)) & 1ull);
          unsigned y = 0;
          if (MUT) {
            y = (ys[u] >> sh) & 0xffu;
            if ((rmr[m >> 6] >> (m & 63)) & 1ull) q += x;   // R[mirror,m] X[this,m]
          }
          if (El2) { El[w] = y * (unsigned)g.Mp + (unsigned)m; El2[w] = (x << 1) | inr; ++w; }
          else El[w++] = (y * (unsigned)g.Mp + (unsigned)m) | (inr << 20) | (x << 21);   // (sweep_sl.h)
        }
      }
      pos += tot;
    }
    q = group_sum_u(q, 16);
    if (ok && gl == 0) Qtl[tm] = q;   // every tie is the mirror of exactly one tie
  }
}

// ------------------------------------------------------------------------------------------
// Mask lists: a partial mask row as the list of its reporters (u16), for masks whose partial rows hold few
// reporters (the self-reporter mask of survey data: R[l,i,j,m] = 1 iff m is i or j, two per row).  Reading 4 bytes
// per row instead of W words makes the mask sums A and the per-tie T of such data a few microseconds.
// rq[l][t] (u32, N*N+1 per layer): first reporter of tie t's list, relative to rbase[l]; empty for rows that are
// not partial.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rm_count(const uint64_t* __restrict__ Rl, const uint8_t* __restrict__ cl,
                                                  unsigned* __restrict__ rql, unsigned long long* total, unsigned* maxrow,
                                                  size_t T, int W) {
  __shared__ double red[8];
  unsigned long long mine = 0;
  unsigned mx = 0;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < T; t += (size_t)gridDim.x * 256) {
    unsigned n = 0;
    if (cl[t] == 2) for (int w = 0; w < W; ++w) n += (unsigned)__popcll(Rl[t * W + w]);
    rql[t] = n;
    mine += n;
    mx = max(mx, n);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) rql[T] = 0u;
  double v = block_sum((double)mine, red);
  if (threadIdx.x == 0 && v > 0.0) atomicAdd(total, (unsigned long long)v);
  if (mx) atomicMax(maxrow, mx);
}
__global__ __launch_bounds__(256) void k_rm_fill(const uint64_t* __restrict__ Rl, const uint8_t* __restrict__ cl,
                                                 const unsigned* __restrict__ rql, unsigned short* __restrict__ Rml, size_t T, int W) {
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < T; t += (size_t)gridDim.x * 256) {
    if (cl[t] != 2) continue;
    size_t q = rql[t];
    for (int w = 0; w < W; ++w) {
      uint64_t bits = Rl[t * W + w];
      while (bits) {
        const int b_ = __builtin_ctzll(bits);
        bits &= bits - 1;
        Rml[q++] = (unsigned short)(w * 64 + b_);
      }
    }
  }
}
// A[l,m,k] += rho[l,t,k] over the listed reporters of the partial rows (model.py:704-718, 742-749); all-ones rows
// were summed by the rho pass.  One thread per tie, A of the workgroup in LDS, one global add per value at the end.
template <int K>
__global__ __launch_bounds__(TPB) void k_mask_lists(const unsigned* __restrict__ rq, const unsigned short* __restrict__ Rm,
                                                    const unsigned long long* __restrict__ rbase, const uint8_t* __restrict__ rcls,
                                                    const double* __restrict__ rho, double* __restrict__ slotA, int Gl,
                                                    const unsigned* __restrict__ perm /* rho by sorted position, or null */, Geo g) {
  extern __shared__ double As[];   // [Mp][K]
  const int l = blockIdx.x / Gl, gb = blockIdx.x - l * Gl;
  const size_t T = (size_t)g.N * g.N;
  const unsigned* rql = rq + (size_t)l * (T + 1);
  const unsigned short* Rml = Rm + rbase[l];
  const uint8_t* cl = rcls + (size_t)l * T;
  const double* rl = rho + (size_t)l * T * K;
  for (int q = threadIdx.x; q < g.Mp * K; q += TPB) As[q] = 0.0;
  __syncthreads();
  const size_t t0 = (size_t)gb * T / Gl, t1 = (size_t)(gb + 1) * T / Gl;
  const unsigned* pl = perm ? perm + (size_t)l * ((T + 63) / 64) * 64 : nullptr;
  for (size_t p = t0 + threadIdx.x; p < t1; p += TPB) {
    const size_t t = pl ? (size_t)pl[p] : p;
    if (cl[t] != 2) continue;
    const unsigned q0 = rql[t], q1 = rql[t + 1];
    double r[K];
#pragma unroll
    for (int k = 0; k < K; ++k) r[k] = rl[p * K + k];
    for (unsigned q = q0; q < q1; ++q) {
      const int m = Rml[q];
#pragma unroll
      for (int k = 0; k < K; ++k) atomicAdd(&As[m * K + k], r[k]);
    }
  }
  __syncthreads();
  double* out = slotA + ((size_t)l * NSLOT + (gb % NSLOT)) * (size_t)g.W * 64 * K;
  for (int q = threadIdx.x; q < g.M * K; q += TPB) {
    const double v = As[q];
    if (v != 0.0) atomicAdd(&out[q], v);
  }
}

// ------------------------------------------------------------------------------------------
// H is accumulated in NH copies (workgroup gb adds into copy gb % NH; adding the rare high levels to one copy
// only was tried: the rho pass ran 1.8x longer on the contended addresses).  Before anything reads it the copies
// are folded into copy 0 by many workgroups (k_fin_rho does it on its way, k_h_reduce otherwise): one workgroup
// per layer reading 8 x 50 KB was most of the finalize kernels' time.
__device__ __forceinline__ double h_fold(double* Hl0, size_t copy_stride, size_t idx) {
  double v[NH];
#pragma unroll
  for (int c = 0; c < NH; ++c) v[c] = Hl0[c * copy_stride + idx];   // all copies in flight at once
  double t = 0.0;
#pragma unroll
  for (int c = 0; c < NH; ++c) {
    t += v[c];
    if (c > 0) Hl0[c * copy_stride + idx] = 0.0;
  }
  Hl0[idx] = t;
  return t;
}
// Fold the K values of one (y, m) item.  Report lists (Cl != null) accumulate categories 1..K-1 and, in slot 0, the
// deficit sum x (1 - sum_k rho_k) of irregular ties; with Cl[y][m] = sum x over the item's reports (a constant of the
// data, from the count-mode launch of vmr_create): H_0 = C - deficit - sum_{k>0} H_k.
__device__ __forceinline__ void h_fold_item(double* Hl0, size_t copy_stride, size_t item, int K, const double* Cl, double* out) {
  double rest = 0.0;
  for (int k = K - 1; k >= 0; --k) {
    double v = h_fold(Hl0, copy_stride, item * K + k);
    if (k > 0) rest += v;
    else if (Cl) { v = Cl[item] - v - rest; Hl0[item * K] = v; }
    out[k] = v;
  }
}
__global__ __launch_bounds__(TPB) void k_h_reduce(double* Hg, const double* Cg, Geo g) {
  const size_t items = (size_t)g.Y * g.Mp, hcs = items * g.K, n = (size_t)g.L * items;
  double tmp[KMAX];
  for (size_t q = (size_t)blockIdx.x * TPB + threadIdx.x; q < n; q += (size_t)gridDim.x * TPB) {
    const size_t l = q / items, it = q - l * items;
    h_fold_item(Hg + l * NH * hcs, hcs, it, g.K, Cg ? Cg + l * items : nullptr, tmp);
  }
}
// count-mode result -> the constants C[l][y][m], leaving H zeroed
__global__ __launch_bounds__(TPB) void k_take_counts(double* Hg, double* Cg, Geo g) {
  const size_t items = (size_t)g.Y * g.Mp, hcs = items * g.K, n = (size_t)g.L * items;
  for (size_t q = (size_t)blockIdx.x * TPB + threadIdx.x; q < n; q += (size_t)gridDim.x * TPB) {
    const size_t l = q / items, it = q - l * items;
    double* Hl0 = Hg + l * NH * hcs;
    Cg[q] = h_fold(Hl0, hcs, it * g.K + 1);   // count mode put sum x into category 1
    Hl0[it * g.K + 1] = 0.0;
  }
}
template <bool ZERO>
__device__ __forceinline__ double h_sum(double* Hl0, size_t, size_t idx) {   // folded H: copy 0
  const double v = Hl0[idx];
  if (ZERO) Hl0[idx] = 0.0;
  return v;
}
__device__ __forceinline__ double h_at(const double* Hl0, size_t, size_t idx) { return Hl0[idx]; }

#ifndef FIN_TPB
#define FIN_TPB 512   // (1024: 128 registers, 16-33 spilled, +2 us per launch; 256: +5 us)
#endif
#define FG_G 16   // workgroups per layer of k_fin_gamma: each owns a range of reporters
__device__ __forceinline__ double block_sum_fin(double v, double* red /*16*/) {   // result valid in thread 0
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    for (int w = 0; w < FIN_TPB / 64; ++w) r += red[w];
  }
  return r;
}

// finalize kernels
// ------------------------------------------------------------------------------------------
// gamma_shp from H with the current (old) weights (model.py:698-703), gamma_rte from A (model.py:704-718), then phi_rte from
// the same A with the new E[theta] (model.py:742-749) and phi_shp from H with the new E[log theta] (model.py:731-733, 861-887;
// mutuality off: from the level-0 sums), the factor table F of the rho pass, the LUT of E[theta].
// FG_G workgroups per layer, each owning a range of REPORTERS: everything of gamma is local to a reporter, so a workgroup
// reads the (y, m) items of its own reporters (kept in registers for the second, phi, use when they are few), finishes their
// gamma and adds its share of the 2 K sums that lambda needs to `fin` with device-scope atomics; the workgroup that draws the
// layer's last ticket finishes lambda and builds F and the LUT from the theta values the others published (release before
// the ticket, acquire after it).  One workgroup per layer (rounds 1-2) kept 4 of 256 CUs busy for 25 us per sweep.
// consume = 1 (fused sweep): H and slotF are read for the last time here and left zeroed for the rho pass.
// (bx: the workgroup's index in its handle's grid -- the launch's, or the handle's share of a lockstep launch, k_fin_gamma_b)
template <bool det /*Geo::det: sums whose order would vary are integer or in slots (its own instantiation: the code costs the plain one 3 us)*/>
__device__ __forceinline__ void fin_gamma_body(double* par, double* Hg, const double* __restrict__ Cg, double* slotA, double* slotF,
                                               double* lutg, double* Fg, double* fin /*[L][2 KMAX + 2]*/, double* nu_acc,
                                               int nh /*copies of H to fold: NH, or 1 when folded already*/, int do_phi, int consume, const Geo& g,
                                               const int bx, const int fg = FG_G /*workgroups per layer of this launch (<= FG_G)*/) {
  extern __shared__ double dyn[];   // s1[mper]: sum_{y,k} w1 H per reporter; gthn[mper]: the new G_theta
  __shared__ double red[16];
  __shared__ double ela_old[KMAX], gla_old[KMAX], fk[KMAX];
  __shared__ double lla_n[KMAX], gla_n[KMAX];   // the new E[log lambda], G_lambda (for the factor table F)
  __shared__ int last;
  const ParOff o = par_off(g.L, g.Mp, g.K);
  const int l = bx / fg, gq = bx - l * fg, K = g.K, Wp = g.W * 64, tid = threadIdx.x;
  const int mper = (g.M + fg - 1) / fg, m0 = gq * mper, m1 = min(g.M, m0 + mper), nm = max(0, m1 - m0);
  double* s1 = dyn;
  double* gthn = dyn + mper;
  double* finl = fin + (size_t)l * (2 * KMAX + 2);
  if (tid < K) {
    double fv[NSLOT], f = 0.0;   // all-ones mask rows, summed by the rho pass (zero when the mask kernel handled them)
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) fv[sl] = slotF[((size_t)l * NSLOT + sl) * K + tid];   // loads first, all in flight
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) f += fv[sl];
    fk[tid] = f;
    ela_old[tid] = par[o.p_shp + l * K + tid] / par[o.p_rte + l * K + tid];
    gla_old[tid] = par[o.G_la + l * K + tid];
  }
  for (int m = tid; m < mper; m += FIN_TPB) { s1[m] = 0.0; gthn[m] = 0.0; }
  __syncthreads();
  const double gnu = par[o.sc + SC_G_NU];
  const size_t hcs = (size_t)g.Y * g.Mp * K;
  double* Hl = Hg + (size_t)l * NH * hcs;
  const int items = g.Y * nm;                 // (y, m) items of this workgroup's reporters
  const bool phi2 = do_phi && g.mut;          // a second use of H follows (phi_shp with the new E[log theta])
  const bool cached = items <= FIN_TPB;       // one item per thread: its K values stay in registers for that second use
  double hv[KMAX], p0[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) { hv[k] = 0.0; p0[k] = 0.0; }
  // One (y, m) item: its K values summed over the nh copies the workgroups of the pass added into (all loads in flight at
  // once).  Report lists (Cl != null) accumulate categories 1..K-1 and, in slot 0, the deficit of irregular ties:
  // H_0 = C - deficit - sum_{k>0} H_k with C[y][m] = sum x a constant of the data.  zero: leave the copies zeroed.
  const double* Cl = Cg ? Cg + (size_t)l * g.Y * g.Mp : nullptr;
  auto item = [&](int y, int m, double (&h_)[KMAX], bool zero) {
    const size_t base = ((size_t)y * g.Mp + m) * K;
    double rest = 0.0;
#pragma unroll
    for (int k = KMAX - 1; k >= 0; --k) {
      if (k < K) {
        double v[NH], t = 0.0;
#pragma unroll
        for (int c = 0; c < NH; ++c) v[c] = c < nh ? Hl[(size_t)c * hcs + base + k] : 0.0;   // loads first, all in flight
#pragma unroll
        for (int c = 0; c < NH; ++c) {
          t += v[c];
          if (zero && c < nh) Hl[(size_t)c * hcs + base + k] = 0.0;
        }
        if (k > 0) rest += t;
        else if (Cl && nh > 1) t = Cl[(size_t)y * g.Mp + m] - t - rest;   // (a folded H holds H_0 itself)
        h_[k] = t;
      }
    }
  };
  for (int it = tid; it < items; it += FIN_TPB) {
    const int y = it / nm, m = m0 + (it - y * nm);
    const double gth = par[o.G_th + (size_t)l * g.Mp + m];   // still the old value
    double acc = 0.0;
    item(y, m, hv, consume && (!phi2 || cached));
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      if (k < K) {
        acc += (g.mut ? w1_of(gth * gla_old[k], gnu * (double)y) : 1.0) * hv[k];
        if (y == 0) p0[k] += hv[k];
      }
    }
    if (acc != 0.0) {
      if (det) atomicAdd(reinterpret_cast<unsigned long long*>(&s1[m - m0]), det_fx(acc, g.det_sh));   // (threads of several waves add here)
      else atomicAdd(&s1[m - m0], acc);
    }
  }
  __syncthreads();
  if (det) {
    for (int m = tid; m < mper; m += FIN_TPB) s1[m] = det_back(*reinterpret_cast<unsigned long long*>(&s1[m]), g.det_sh);
    __syncthreads();
  }
  double pr[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) pr[k] = 0.0;
  for (int mi = tid; mi < nm; mi += FIN_TPB) {
    const int m = m0 + mi;
    const size_t q = (size_t)l * g.Mp + m;
    double A[KMAX], rte = 0.0;
    const double pa_th = par[o.a_th + q], pb_th = par[o.b_th + q];
    for (int k = 0; k < K; ++k) {
      double av[NSLOT], ak = 0.0;
      if (slotA) {   // (null: the mask has no partial rows, nothing ever lands in the slots -- a memory round trip less)
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) av[sl] = slotA[(((size_t)l * NSLOT + sl) * Wp + m) * K + k];   // loads first
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {
          ak += av[sl];
          slotA[(((size_t)l * NSLOT + sl) * Wp + m) * K + k] = 0.0;   // consume: the slots are zero again for the next sweep
        }
      }
      ak += fk[k];
      A[k] = ak;
      rte += ela_old[k] * ak;
    }
    double shp = pa_th + s1[mi];
    rte = pb_th + rte;
    par[o.g_shp + q] = shp; par[o.g_rte + q] = rte;
    double e = shp / rte, lg = digamma_pos(shp) - log(rte);
    const double gn = exp(lg);
    par[o.E_th + q] = e; par[o.l_th + q] = lg; par[o.G_th + q] = gn;
    gthn[mi] = gn;
    for (int k = 0; k < K; ++k) pr[k] += e * A[k];
  }
  __syncthreads();
  // phi_shp's share of this workgroup: H with the NEW E[log theta] of its reporters (model.py:731-733, 861-887)
  double ps[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) ps[k] = 0.0;
  if (phi2) {
    if (cached) {
      if (tid < items) {
        const int y = tid / nm, mi = tid - y * nm;
        const double gth = gthn[mi];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) if (k < K) ps[k] = w1_of(gth * gla_old[k], gnu * (double)y) * hv[k];
      }
    } else {
      for (int it = tid; it < items; it += FIN_TPB) {
        const int y = it / nm, mi = it - y * nm;
        const double gth = gthn[mi];
        double h2[KMAX];
        item(y, m0 + mi, h2, consume != 0);
        for (int k = 0; k < K; ++k) ps[k] += w1_of(gth * gla_old[k], gnu * (double)y) * h2[k];
      }
    }
  } else if (!g.mut) {
#pragma unroll
    for (int k = 0; k < KMAX; ++k) ps[k] = p0[k];   // mutuality off: phi_shp = alpha + sum x rho_k
  }
  // this workgroup's shares of the 2 K sums, then its ticket
  for (int k = 0; k < K; ++k) {
    const double v = block_sum_fin(pr[k], red);
    const double w = block_sum_fin(ps[k], red);
    if (tid == 0) {
      if (det) { double* sl_ = fin + (size_t)g.L * (2 * KMAX + 2) + ((size_t)l * FG_G + gq) * 2 * KMAX; sl_[k] = v; sl_[KMAX + k] = w; }   // its own slot
      else { atomicAdd(&finl[k], v); atomicAdd(&finl[KMAX + k], w); }
    }
  }
  // publish this workgroup's theta values (every storing wave drained, then one agent-scope release) before the ticket
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    if (fg == 1) last = 1;   // (one workgroup has the whole layer: nothing to publish, no ticket, no fences)
    else {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const double t = atomicAdd(&finl[2 * KMAX], 1.0);
      last = (t == (double)(fg - 1));
      if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (!last) return;
  // ---- the layer's last workgroup: lambda, the factor table F, the LUT -------------------------------------------------
  if (tid < K) {
    const int k = tid, q = l * K + k;
    double sr, ss;
    if (det) {   // the workgroups' slots in order (published like the theta values: release before the ticket, acquire after it)
      sr = 0.0; ss = 0.0;
      const double* sl_ = fin + (size_t)g.L * (2 * KMAX + 2) + (size_t)l * FG_G * 2 * KMAX;
      for (int q = 0; q < FG_G; ++q) { sr += __builtin_nontemporal_load(&sl_[q * 2 * KMAX + k]); ss += __builtin_nontemporal_load(&sl_[q * 2 * KMAX + KMAX + k]); }
    } else { sr = atomicAdd(&finl[k], 0.0); ss = atomicAdd(&finl[KMAX + k], 0.0); }   // (device-scope reads)
    const double rte = par[o.b_la + q] + sr;
    if (g.mut && !do_phi) {
      par[o.p_rte_pend + q] = rte;   // the PHI sub-step commits (k_fin_phi)
    } else {
      const double shp = par[o.a_la + q] + ss;
      par[o.p_shp + q] = shp; par[o.p_rte + q] = rte;
      if (g.mut) par[o.p_rte_pend + q] = rte;
      const double lg = digamma_pos(shp) - log(rte);
      par[o.E_la + q] = shp / rte; par[o.l_la + q] = lg; par[o.G_la + q] = exp(lg);
      lla_n[k] = lg; gla_n[k] = exp(lg);
    }
  }
  __syncthreads();
  if (tid == 0) { for (int k = 0; k < 2 * KMAX + 1; ++k) finl[k] = 0.0; }   // scratch ready for the next sweep
  if (consume && tid < K) {
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) slotF[((size_t)l * NSLOT + sl) * K + tid] = 0.0;   // (every workgroup of the layer has read it)
  }
  // (fused sweep, report lists) the factor table F of the rho pass from the new theta, lambda and the current nu
  if (do_phi && Fg) {
    const int all = g.Y * g.Mp;
    double* Fl = Fg + (size_t)l * all * K;
    const double* lthn = par + o.l_th + (size_t)l * g.Mp;
    const double* gthg = par + o.G_th + (size_t)l * g.Mp;
    for (int it = tid; it < all; it += FIN_TPB) {
      const int y = it / g.Mp, m = it - y * g.Mp;
      const double lt = (m < g.M) ? lthn[m] : 0.0, gt = (m < g.M) ? gthg[m] : 0.0;
      for (int k = 0; k < K; ++k)
        Fl[(size_t)it * K + k] = (m < g.M) ? f_entry(g.mut, lt, gt, lla_n[k], gla_n[k], gnu, y) : 0.0;
    }
  }
  const double* Eth = par + o.E_th + (size_t)l * g.Mp;
  for (int q = tid; q < g.W * 256; q += FIN_TPB) {
    int n = q >> 4, e = q & 15;
    double v = 0.0;
    for (int u = 0; u < 4; ++u) {
      int m = n * 4 + u;
      if (((e >> u) & 1) && m < g.Mp) v += Eth[m];
    }
    lutg[(size_t)l * g.W * 256 + q] = v;
  }
  // (fused sweep, sorted lists) the constant part of the nu update the rho pass finishes itself (SlArgs::nu_acc):
  // sum_{y>0,m} w2_0(m, y) C[l][y][m] with the new theta, lambda and the current nu
  if (nu_acc && do_phi && g.mut && Cl) {
    const double* gthg = par + o.G_th + (size_t)l * g.Mp;
    double c0 = 0.0;
    for (int it = tid; it < g.Y * g.Mp; it += FIN_TPB) {
      const int y = it / g.Mp, m = it - y * g.Mp;
      if (y > 0 && m < g.M) {
        const double z2 = gnu * (double)y, den = gthg[m] * gla_n[0] + z2;
        if (den != 0.0) c0 += (z2 / den) * Cl[it];
      }
    }
    c0 = block_sum_fin(c0, red);
    if (tid == 0) nu_acc[2 + l] = c0;
  }
}
__global__ __launch_bounds__(FIN_TPB) void k_fin_gamma(double* par, double* Hg, const double* __restrict__ Cg, double* slotA, double* slotF,
                                                       double* lutg, double* Fg, double* fin, double* nu_acc, int nh, int do_phi, int consume, Geo g) {
  fin_gamma_body<false>(par, Hg, Cg, slotA, slotF, lutg, Fg, fin, nu_acc, nh, do_phi, consume, g, (int)blockIdx.x);
}
__global__ __launch_bounds__(FIN_TPB) void k_fin_gamma_det(double* par, double* Hg, const double* __restrict__ Cg, double* slotA, double* slotF,
                                                           double* lutg, double* Fg, double* fin, double* nu_acc, int nh, int do_phi, int consume, Geo g) {
  fin_gamma_body<true>(par, Hg, Cg, slotA, slotF, lutg, Fg, fin, nu_acc, nh, do_phi, consume, g, (int)blockIdx.x);
}
// the finalize kernels of many small handles in one launch (vmr_fit_loop_batch): workgroup -> unit through blk_unit
struct FinUnit {
  double *par, *Hg; const double* Cg; double *slotA, *slotF, *lutg, *fin_g, *nu_acc, *slotR, *elbo;   // (elbo: 8 doubles, as vmr_ctx::elbo_dev)
  Geo g; int fg_blk0, fr_blk0, fr_nblk;
};
// (fg workgroups per layer: 16 of them -- the latency of one handle's finalize -- times many units would queue up on the CUs)
__global__ __launch_bounds__(FIN_TPB) void k_fin_gamma_b(const FinUnit* __restrict__ units, const int* __restrict__ blk_unit, int fg) {
  const FinUnit& u = units[blk_unit[blockIdx.x]];
  fin_gamma_body<false>(u.par, u.Hg, u.Cg, u.slotA, u.slotF, u.lutg, nullptr, u.fin_g, u.nu_acc, NH, 1, 1, u.g, (int)blockIdx.x - u.fg_blk0, fg);
}

// phi commit, mutuality on: phi_shp from H with the NEW E[log theta] (model.py:731-733, 861-887; the cache
// refresh of :647 sits between the two updates), phi_rte as computed by k_fin_gamma (model.py:742-749)
__global__ __launch_bounds__(TPB) void k_fin_phi(double* par, const double* __restrict__ Hg, Geo g) {
  __shared__ double red[8];
  __shared__ double gla_old[KMAX];
  const ParOff o = par_off(g.L, g.Mp, g.K);
  const int l = blockIdx.x, K = g.K;
  if ((int)threadIdx.x < K) gla_old[threadIdx.x] = par[o.G_la + l * K + threadIdx.x];
  __syncthreads();
  const double gnu = par[o.sc + SC_G_NU];
  const size_t hcs = (size_t)g.Y * g.Mp * K;
  const double* Hl = Hg + (size_t)l * NH * hcs;
  double ps[KMAX];
  for (int k = 0; k < KMAX; ++k) ps[k] = 0.0;
  for (int m = threadIdx.x; m < g.M; m += TPB) {
    const double gth = par[o.G_th + (size_t)l * g.Mp + m];   // new
    for (int y = 0; y < g.Y; ++y)
      for (int k = 0; k < K; ++k) ps[k] += w1_of(gth * gla_old[k], gnu * (double)y) * h_at(Hl, hcs, ((size_t)y * g.Mp + m) * K + k);
  }
  for (int k = 0; k < K; ++k) {
    double v = block_sum(ps[k], red);
    if (threadIdx.x == 0) {
      const int q = l * K + k;
      double shp = par[o.a_la + q] + v, rte = par[o.p_rte_pend + q];
      par[o.p_shp + q] = shp; par[o.p_rte + q] = rte;
      double lg = digamma_pos(shp) - log(rte);
      par[o.E_la + q] = shp / rte; par[o.l_la + q] = lg; par[o.G_la + q] = exp(lg);
    }
  }
}

__device__ __forceinline__ double gamma_elbo_term(double pa, double pb, double qa, double qb) {
  // model.py:1300-1303
  return lgamma(qa) - pa * log(qb) + (pa - qa) * digamma_pos(qa) + qa * (1.0 - pb / qb);
}

// nu from H of the new rho (model.py:694-696, 822-825: sum x w2_k rho_k) and/or ELBO assembly (model.py:997-1013).
// One workgroup per layer adds its share to fin[0..1] with device-scope atomics; the workgroup that draws the
// last ticket (fin[2]) finishes the scalars and clears the scratch.
#define FR_G 16   // workgroups per layer of k_fin_rho
__device__ __forceinline__ void fin_rho_body(double* par, double* Hg, const double* Cg, double* slotR, double* elbo_out,
                                             double* fin, int do_nu, int do_elbo, int fold, int skip_nu /* the pass finished nu itself */, const Geo& g,
                                             const int bx, const int gx, double* slots = nullptr /*deterministic mode: [gx][2] partial sums, summed in order*/) {
  __shared__ double red[8];
  __shared__ int last;
  const ParOff o = par_off(g.L, g.Mp, g.K);
  double* sc = par + o.sc;
  const int l = bx / FR_G, gs = bx - l * FR_G;
  double a0 = 0.0, gt = 0.0;
  if (!skip_nu && (g.mut || fold)) {   // threads take (y, m) items of H; fold: the NH copies are summed into copy 0 on the way
    const double gnu = sc[SC_G_NU];
    const size_t hcs = (size_t)g.Y * g.Mp * g.K;
    double* Hl = Hg + (size_t)l * (g.gen ? 1 : NH) * hcs;   // (the general kernels keep one copy of H, every category in it)
    const long long items = (long long)g.Y * g.Mp, i0 = (long long)gs * items / FR_G, i1 = (long long)(gs + 1) * items / FR_G;
    for (long long it = i0 + (int)threadIdx.x; it < i1; it += TPB) {
      const int y = (int)(it / g.Mp), m = (int)(it - (long long)y * g.Mp);
      if (m >= g.M || (y == 0 && !fold)) continue;
      const double gth = par[o.G_th + (size_t)l * g.Mp + m], z2 = gnu * (double)y;
      if (fold) {   // (specialised kernels only: K <= KMAX)
        double hk[KMAX];
        h_fold_item(Hl, hcs, (size_t)it, g.K, Cg ? Cg + (size_t)l * items : nullptr, hk);
        for (int k = 0; k < g.K; ++k) {
          const double z1 = gth * par[o.G_la + l * g.K + k];
          if (g.mut && y > 0) a0 += (z2 / (z1 + z2)) * hk[k];
        }
      } else if (g.mut && y > 0) {
        for (int k = 0; k < g.K; ++k) {
          const double hv = Hl[(size_t)it * g.K + k];
          if (hv != 0.0) a0 += (z2 / (gth * par[o.G_la + l * g.K + k] + z2)) * hv;
        }
      }
    }
  }
  if (do_elbo) {
    for (int m = gs * TPB + threadIdx.x; m < g.M; m += FR_G * TPB) {
      const size_t q = (size_t)l * g.Mp + m;
      gt += gamma_elbo_term(par[o.a_th + q], par[o.b_th + q], par[o.g_shp + q], par[o.g_rte + q]);
    }
    if (gs == 0) {
      for (int k = threadIdx.x; k < g.K; k += TPB) {
        const int q = l * g.K + k;
        gt += gamma_elbo_term(par[o.a_la + q], par[o.b_la + q], par[o.p_shp + q], par[o.p_rte + q]);
      }
    }
  }
  a0 = block_sum(a0, red);
  gt = block_sum(gt, red);
  if (threadIdx.x == 0) {
    if (slots) { slots[2 * bx] = a0; slots[2 * bx + 1] = gt; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
    else { atomicAdd(&fin[0], a0); atomicAdd(&fin[1], gt); }
    // the two adds are performed at the memory side before the ticket is drawn: they stay counted until they are (a full
    // __threadfence() -- L2 write-back and invalidate, ~3.5 us -- orders plain stores, of which the ticket publishes none)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const double t = atomicAdd(&fin[2], 1.0);
    last = (t == (double)(gx - 1));
  }
  __syncthreads();
  if (!last) return;
  double a1 = 0, a2 = 0, a3 = 0;
  if (threadIdx.x < NSLOT) {
    double* ps = slotR + (size_t)threadIdx.x * 4;
    a1 = ps[1]; a2 = ps[2]; a3 = ps[3];
    ps[0] = ps[1] = ps[2] = ps[3] = 0.0;   // consume
  }
  a1 = block_sum(a1, red); a2 = block_sum(a2, red); a3 = block_sum(a3, red);
  if (threadIdx.x == 0) {
    if (slots) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      a0 = 0.0; gt = 0.0;
      for (int q = 0; q < gx; ++q) { a0 += __builtin_nontemporal_load(&slots[2 * q]); gt += __builtin_nontemporal_load(&slots[2 * q + 1]); }
    } else {
      a0 = atomicAdd(&fin[0], 0.0);   // device-scope reads of the other workgroups' sums
      gt = atomicAdd(&fin[1], 0.0);
    }
    fin[0] = 0.0; fin[1] = 0.0; fin[2] = 0.0;
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
    // raw pieces for fits whose layers are spread over several handles (vmr_sweep_local)
    if (!skip_nu) elbo_out[1] = a0;   // sum x w2 rho over the local layers (nu_shp - alpha_eta)
    elbo_out[2] = a1 + a2 + gt;   // local ELBO terms that do not involve nu
    elbo_out[3] = a3;             // local sum_t (sum_k rho) Q_t, enters the ELBO as -E[nu] * (.)
  }
}
__global__ __launch_bounds__(TPB) void k_fin_rho(double* par, double* Hg, const double* Cg, double* slotR, double* elbo_out,
                                                 double* fin, int do_nu, int do_elbo, int fold, int skip_nu, Geo g, double* slots) {
  fin_rho_body(par, Hg, Cg, slotR, elbo_out, fin, do_nu, do_elbo, fold, skip_nu, g, (int)blockIdx.x, (int)gridDim.x, slots);
}
// (lockstep launch: the ELBO of a sweep whose pass finished nu itself)
__global__ __launch_bounds__(TPB) void k_fin_rho_b(const FinUnit* __restrict__ units, const int* __restrict__ blk_unit) {
  const FinUnit& u = units[blk_unit[blockIdx.x]];
  fin_rho_body(u.par, u.Hg, u.Cg, u.slotR, u.elbo, u.elbo + 4, 0, 1, 0, 1, u.g, (int)blockIdx.x - u.fr_blk0, u.fr_nblk);
}

// commit a nu_shp that was summed over several handles (layer-sharded fits)
__global__ void k_commit_nu(double* par, double nu_partial_total, const double* total_dev, Geo g) {
  const ParOff o = par_off(g.L, g.Mp, g.K);
  double* sc = par + o.sc;
  if (total_dev) nu_partial_total = total_dev[0];   // (summed over the owners on the device: no host hop)
  if (threadIdx.x == 0 && blockIdx.x == 0 && g.mut) {
    sc[SC_G_NU_STALE] = sc[SC_G_NU];
    sc[SC_NU_SHP] = sc[SC_A_ETA] + nu_partial_total;
    sc[SC_G_NU] = exp(digamma_pos(sc[SC_NU_SHP]) - log(sc[SC_NU_RTE]));
    sc[SC_E_NU] = sc[SC_NU_SHP] / sc[SC_NU_RTE];
  }
}

// posterior read-out per tie (model.py:1099-1188, utils.py:200-217): argmax_k rho, sum_k k rho_k, rho_1 >= threshold
__global__ __launch_bounds__(256) void k_readout(const double* __restrict__ rho, void* __restrict__ out, size_t ties, int K,
                                                 int method, double threshold, const unsigned* __restrict__ perm, size_t T, size_t NS) {
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < ties; q += (size_t)gridDim.x * blockDim.x) {
    const double* r = rho + q * K;
    size_t t = q;
    if (perm) { const size_t l = q / T, pos = q - l * T; t = l * T + perm[l * NS * 64 + pos]; }   // rho by sorted position, answers by tie
    if (method == VMR_READ_RHO_MAX) {
      int best = 0;
      double bv = r[0];
      for (int k = 1; k < K; ++k) if (r[k] > bv) { bv = r[k]; best = k; }   // first maximum, as np.argmax
      reinterpret_cast<uint8_t*>(out)[t] = (uint8_t)best;
    } else if (method == VMR_READ_RHO_MEAN) {
      double m = 0.0;
      for (int k = 1; k < K; ++k) m += (double)k * r[k];
      reinterpret_cast<double*>(out)[t] = m;
    } else {
      reinterpret_cast<uint8_t*>(out)[t] = r[1] >= threshold ? 1 : 0;
    }
  }
}


// ------------------------------------------------------------------------------------------
// Posterior samples of Y on the device: `sample_inferred_model` (model.py:1062-1096) draws, per tie, n_trials categorical
// trials from rho and keeps the most frequent category (first maximum): Generator.multinomial(n, rho).argmax(-1).  Here the
// uniforms come from Philox4x32-10 with key = seed and counter = (tie index in [L,N,N] order, trial pair), so a draw depends on
// (seed, tie, trial) only -- reproducible, and the same for every data layout -- not on NumPy's PCG64 stream, which a GPU cannot
// follow (the host class keeps that exact mode; this one is checked by distribution).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sample(const double* __restrict__ rho, uint8_t* __restrict__ out, size_t ties, int K, int n_trials,
                                                unsigned long long seed, const unsigned* __restrict__ perm, size_t T, size_t NS) {
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < ties; q += (size_t)gridDim.x * blockDim.x) {
    const double* r = rho + q * K;
    size_t t = q;
    if (perm) { const size_t l = q / T, pos = q - l * T; t = l * T + perm[l * NS * 64 + pos]; }
    unsigned cnt[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) cnt[k] = 0u;
    for (int n = 0; n < n_trials; n += 2) {
      unsigned c[4] = {(unsigned)t, (unsigned)((unsigned long long)t >> 32), (unsigned)(n >> 1), 0u};
      philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (n + i < n_trials) {
          const double u = ((double)(c[2 * i] >> 5) * 67108864.0 + (double)(c[2 * i + 1] >> 6)) * (1.0 / 9007199254740992.0);
          int sel = 0;
          double acc = r[0];
          for (int k = 1; k < K; ++k) { if (u >= acc) sel = k; acc += r[k]; }   // first k with u < cumulative sum; the last one catches the rest
#pragma unroll
          for (int k = 0; k < KMAX; ++k) cnt[k] += (sel == k) ? 1u : 0u;
        }
      }
    }
    int best = 0;
#pragma unroll
    for (int k = 1; k < KMAX; ++k) if (k < K && cnt[k] > cnt[best]) best = k;
    out[t] = (uint8_t)best;
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------

static size_t shmem_ct(const Geo& g) { return g.mut ? (size_t)g.Mp * g.K * 8 : 0; }
static size_t shmem_q() { return (size_t)(TPB / 64) * 64 * QCAP * 2; }
static size_t shmem_hc(const Geo& g) { return (size_t)g.hc * g.Mp * g.K * 8; }
static size_t shmem_hist(const Geo& g) {
  return (size_t)g.nt * g.stride + (size_t)g.nt * g.K * 8 + shmem_hc(g) + shmem_q() + 16;
}
static size_t shmem_rho(const Geo& g, bool update, bool elbo) {
  size_t n = (size_t)g.nt * g.stride + (size_t)g.nt * g.W * 8 + (size_t)g.W * 8 + (size_t)g.Mp * 8 + 64 + shmem_ct(g) +
             2 * (size_t)g.nt * g.K * 8 + shmem_q();
  if (update && !g.two_pass) n += shmem_hc(g);
  if (elbo) n += (size_t)g.Mp * 8 + (size_t)g.nt * 4;
  return n + 16;
}

#define SP_LDS_MAX (160 * 1024)

// Launch shape of one sweep over the sorted lists: the handle's block size and table levels, shrunk until the workgroup fits in LDS
static SlShape sl_shape(const vmr_ctx* h, bool update, bool elbo, bool hist) {
  const Geo& g = h->g;
  SlShape s{std::max(64, std::min((update || elbo) ? h->sp_tpb : h->st_tpb, sl_tpb_max(g.K, elbo, h->all_full != 0, update)) & ~63), update ? g.yt : 0, hist ? g.hc : 0, 0};
  auto bytes = [&]() { return sl_smem(g, s.yt, s.hc, update, elbo, hist); };
  while (bytes() > SP_LDS_MAX && (s.yt > 0 || s.hc > 0)) { if (s.yt >= s.hc && s.yt > 0) --s.yt; else --s.hc; }
  s.smem = bytes();
  return s;
}
static SlArgs sl_args(const vmr_ctx* h, const SlShape& sh, int do_hist, int sum_a = 0) {
  return SlArgs{h->E, h->rs, h->ebase, h->perm, h->sy, h->cls_p, h->Qt_p, h->Rb, h->rq, h->Rm, h->rbase, h->rm2, h->rho, h->logpr, h->par, h->slotR,
                h->lutg, h->Hg, h->slotF, h->slotA, 1, do_hist, sh.yt, sh.hc, sum_a, nullptr, nullptr, 0, 0, nullptr, h->lp0 ? 1 : 0, h->g.farl, (do_hist == 1 && h->g.two_pass && sh.hc >= 1) ? h->h0s : nullptr, (do_hist == 1 && h->g.two_pass && sh.hc >= 1 && h->h0s) ? h->x0p : nullptr,
                (h->h0s && h->g.two_pass && !getenv("VMR_NO_LV0R")) ? 1 : 0, h->E + h->n_slots, 0};   // (level 0 must be among the LDS levels: its deficits go there)
}
static int sl_launch(vmr_ctx* h, int mode, const SlShape& sh, SlArgs& a) {
  sl_launch_fn fn = vmr_sl_launcher(h->g.K);
  if (!fn) return fail(h, VMR_EINVAL, "this build of libvimure_hip.so holds no sweep kernel for this K");
  return fn(h, mode, sh, a);
}

#ifdef VMR_DEV   // development build: K = 2 only (fast compile, ISA inspection)
#define DISPATCH_K(K_, ...)                         \
  switch (K_) {                                     \
    case 2: { constexpr int KK = 2; __VA_ARGS__; } break; \
    default: break;                                 \
  }
#else
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
#endif
// K and the prefetch depth (4, 8 or 12 sixteen-byte chunks per thread)
#define DISPATCH_KP(K_, PF_, ...)                                     \
  switch (PF_) {                                                      \
    case 4: { constexpr int PP = 4; DISPATCH_K(K_, __VA_ARGS__); } break;   \
    case 8: { constexpr int PP = 8; DISPATCH_K(K_, __VA_ARGS__); } break;   \
    default: { constexpr int PP = 12; DISPATCH_K(K_, __VA_ARGS__); } break; \
  }



// ------------------------------------------------------------------------------------------
// Geo::farl -- the reports of the levels beyond the LDS ones
// ------------------------------------------------------------------------------------------
// Geo::farl: every tie's reports of the levels beyond the LDS ones (row >= rows_near) are moved to the FRONT of its list -- one wave
// per step, a lane walks its tie's column and swaps a far report with the first near one -- so that only the first nf rounds of a
// step (nf = the most far reports any of its 64 ties has; stored in the high half of sy) ever take the pass' far-level code: left
// where they fall, 2 % of far reports put one into two rounds of three.
// slots read by a statistics pass that skips the rounds of level 0 only: sum over steps of 64 min(R, n1)  (vmr_kernel_bytes)
__global__ __launch_bounds__(256) void k_stat_slots(const unsigned* __restrict__ rsl, const unsigned* __restrict__ syl, size_t NS, unsigned long long* __restrict__ out) {
  unsigned long long v = 0;
  for (size_t s = (size_t)blockIdx.x * 256 + threadIdx.x; s < NS; s += (size_t)gridDim.x * 256) {
    const unsigned R = (rsl[s + 1] - rsl[s]) >> 6, n1 = syl[s] >> 16;
    v += 64ull * (unsigned long long)min(R, n1);
  }
  if (v) atomicAdd(out, v);
}
// x0 != null: also the tie's summed counts of the near reports, by position (SlArgs::x0p).
__global__ __launch_bounds__(256) void k_far_first(unsigned* __restrict__ E, const unsigned* __restrict__ rsl, unsigned* __restrict__ syl, size_t NS, unsigned rows_near,
                                                   unsigned* __restrict__ x0) {
  const int lane = threadIdx.x & 63;
  for (size_t s = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); s < NS; s += (size_t)gridDim.x * 4) {
    const unsigned ea = rsl[s], R = (rsl[s + 1] - ea) >> 6;
    unsigned* col = E + (size_t)ea + lane;
    unsigned f = 0;   // far reports found so far = the slot the next one goes to
    unsigned xs = 0;
    for (unsigned r = 0; r < R; ++r) {
      const unsigned e = col[(size_t)r * 64];
      if (e != 0u && SL_YM(e) >= rows_near) {
        if (f != r) { const unsigned o = col[(size_t)f * 64]; col[(size_t)f * 64] = e; col[(size_t)r * 64] = o; }
        ++f;
      } else xs += SL_X(e);
    }
    if (x0) x0[s * 64 + (unsigned)lane] = xs;
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) f = max(f, (unsigned)__shfl_xor((int)f, o2, 64));
    if (lane == 0) syl[s] = (syl[s] & 0xffffu) | (min(f, 0xffffu) << 16);
  }
}

// One wave per step of the sorted lists: (position, entry) of every report whose mirror count is at least yfar, appended to the
// layer's far list (order immaterial: their sums are float atomics anyway).
__global__ __launch_bounds__(256) void k_far_collect(const unsigned* __restrict__ E, const unsigned* __restrict__ rsl, size_t NS, unsigned rows_near,
                                                     unsigned* __restrict__ far_pos, unsigned* __restrict__ far_ent, unsigned long long* __restrict__ counter) {
  const int lane = threadIdx.x & 63;
  for (size_t s = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); s < NS; s += (size_t)gridDim.x * 4) {
    const unsigned ea = rsl[s], R = (rsl[s + 1] - ea) >> 6;
    for (unsigned r = 0; r < R; ++r) {
      const unsigned e = E[(size_t)ea + r * 64 + lane];
      const bool far = e != 0u && SL_X(e) != 0u && SL_YM(e) >= rows_near;
      const unsigned long long bal = __ballot(far);
      if (bal == 0ull) continue;
      unsigned long long base = 0ull;
      if (lane == 0) base = atomicAdd(counter, (unsigned long long)__popcll(bal));
      base = ((unsigned long long)(unsigned)__shfl((int)(unsigned)(base >> 32), 0, 64) << 32) | (unsigned)__shfl((int)(unsigned)base, 0, 64);
      if (far) {
        const unsigned long long at = base + (unsigned long long)__popcll(bal & ((1ull << lane) - 1ull));
        far_pos[at] = (unsigned)(s * 64 + (unsigned)lane);
        far_ent[at] = e;
      }
    }
  }
}

// The statistics of the far reports: H[y][m][k] += x rho_k (k >= 1; the deficits of ties whose rho does not sum to 1 in slot 0)
// for the reports of levels y >= hc, which the pass left out -- the next LF levels in LDS (float atomics, as the pass' own), any
// beyond those as global adds -- and their share of nu_shp - alpha = sum x rho_k w2_k (model.py:820-830).  The grid's last
// workgroup finishes nu exactly as the pass' does where it keeps every level (sweep_sl.hip).  rho is gathered by position: a few
// per cent of the reports.  count: the constants' pass of vmr_create (every tie "is" category 1).
template <int K>
__global__ __launch_bounds__(1024) void k_far_hist(const unsigned* __restrict__ far_pos, const unsigned* __restrict__ far_ent,
                                                   const unsigned long long* __restrict__ fbase, const double* __restrict__ rho, double* Hg, double* par,
                                                   double* nu_acc, double* elbo_dev, int count, int commit_nu, int Gl, int LF, Geo g) {
  extern __shared__ double Hf[];   // [K][LF][Mp], 16 doubles for the block sums
  const int Mp = g.Mp, tid = threadIdx.x, nthr = (int)blockDim.x;
  const int l = (int)blockIdx.x / Gl, gb = (int)blockIdx.x - l * Gl;
  const unsigned lfm = (unsigned)LF * (unsigned)Mp, hcm = (unsigned)g.hc * (unsigned)Mp;
  double* red = Hf + (size_t)K * lfm;
  for (unsigned q = tid; q < (unsigned)K * lfm; q += nthr) Hf[q] = 0.0;
  __syncthreads();
  const ParOff o = par_off(g.L, Mp, g.K);
  const size_t T = (size_t)g.N * g.N;
  double Gla[K];
#pragma unroll
  for (int k = 0; k < K; ++k) Gla[k] = par[o.G_la + l * K + k];
  const double gnu = par[o.sc + SC_G_NU];
  const double* gth = par + o.G_th + (size_t)l * Mp;
  const double* rl = rho + (size_t)l * T * K;
  double* Hl = Hg + ((size_t)l * NH + (gb % NH)) * g.Y * Mp * K;
  double share = 0.0;
  const unsigned long long i1 = fbase[l + 1];
  constexpr int FB = 4;   // reports in flight per thread (the gathers of rho are what this kernel waits for)
  for (unsigned long long i0 = fbase[l] + (unsigned long long)gb * nthr * FB + tid; i0 < i1; i0 += (unsigned long long)Gl * nthr * FB) {
    unsigned pos[FB], ent[FB];
    double r[FB][K];
#pragma unroll
    for (int u = 0; u < FB; ++u) {
      const unsigned long long i = i0 + (unsigned long long)u * nthr;
      const bool on = i < i1;
      pos[u] = on ? far_pos[i] : 0u;
      ent[u] = on ? far_ent[i] : 0u;
    }
#pragma unroll
    for (int u = 0; u < FB; ++u) {
#pragma unroll
      for (int k = 0; k < K; ++k) r[u][k] = count ? (k == 1 ? 1.0 : 0.0) : (ent[u] ? rl[(size_t)pos[u] * K + k] : 0.0);
    }
#pragma unroll
    for (int u = 0; u < FB; ++u) {
      if (ent[u] == 0u) continue;
      const unsigned ym = SL_YM(ent[u]), rel = ym - hcm;
      const double dx = (double)SL_X(ent[u]);
      double sm = 0.0;
#pragma unroll
      for (int k = 0; k < K; ++k) sm += r[u][k];
      double dfc = 1.0 - sm;
      if (fabs(dfc) <= 1e-14) dfc = 0.0;   // (as the pass decides it)
      if (rel < lfm) {
#pragma unroll
        for (int k = 1; k < K; ++k) atomicAdd(&Hf[(unsigned)k * lfm + rel], dx * r[u][k]);
        if (dfc != 0.0) atomicAdd(&Hf[rel], dx * dfc);
      } else {
        double* d = Hl + (size_t)ym * K;
#pragma unroll
        for (int k = 1; k < K; ++k) atomicAdd(&d[k], dx * r[u][k]);
        if (dfc != 0.0) atomicAdd(&d[0], dx * dfc);
      }
      if (nu_acc) {
        const unsigned y = ym / (unsigned)Mp, m = ym - y * (unsigned)Mp;
        const double z2 = gnu * (double)y, gt = gth[m];
        const double d0 = gt * Gla[0] + z2, w0 = d0 == 0.0 ? 0.0 : z2 / d0;
        share -= w0 * dx * dfc;
#pragma unroll
        for (int k = 1; k < K; ++k) {
          const double dk = gt * Gla[k] + z2;
          share += ((dk == 0.0 ? 0.0 : z2 / dk) - w0) * dx * r[u][k];
        }
      }
    }
  }
  __syncthreads();
  for (unsigned q = tid; q < (unsigned)K * lfm; q += nthr) {
    const double v = Hf[q];
    if (v != 0.0) { const unsigned k = q / lfm, rel = q - k * lfm; atomicAdd(&Hl[(size_t)(hcm + rel) * K + k], v); }
  }
  if (!nu_acc) return;
  share = block_sum_n(share, red);
  __shared__ int last;
  if (tid == 0) {
    atomicAdd(&nu_acc[0], share);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (this workgroup's share is performed at the memory side before the ticket is drawn)
    const double t = atomicAdd(&nu_acc[1], 1.0);
    last = (t == (double)(gridDim.x - 1u));
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      double tot = atomicAdd(&nu_acc[0], 0.0);   // the pass' workgroups' shares and this kernel's (device-scope read)
      for (int ll = 0; ll < g.L; ++ll) tot += nu_acc[2 + ll];
      nu_acc[0] = 0.0; nu_acc[1] = 0.0;
      elbo_dev[1] = tot;   // the raw piece, for fits whose layers are spread over several handles (vmr_sweep_local)
      if (commit_nu) {
        double* sc = par + o.sc;
        sc[SC_G_NU_STALE] = sc[SC_G_NU];           // what the last cache refresh held (model.py:684)
        sc[SC_NU_SHP] = sc[SC_A_ETA] + tot;
        sc[SC_G_NU] = exp(digamma_pos(sc[SC_NU_SHP]) - log(sc[SC_NU_RTE]));
        sc[SC_E_NU] = sc[SC_NU_SHP] / sc[SC_NU_RTE];
      }
    }
  }
}
// SlArgs::h0s: the level-0 statistics of the rounds that took them as per-tie products, put into copy 0 of H with the marginals the
// finalize kernels read -- sum_m H[0][m][k] = S_k (the pass' sums), sum_k H[0][m][k] = C[0][m] - deficits (rebuilt from the constant
// as for every level, h_fold_item): H[0][m][k] += C[0][m] S_k / sum_m C[0][m] for k >= 1.  One workgroup per layer; consumes h0s.
__global__ __launch_bounds__(256) void k_level0_spread(double* Hg, const double* __restrict__ Cg, double* h0s, Geo g) {
  __shared__ double Sk[KMAX];
  const int l = blockIdx.x, K = g.K, Mp = g.Mp;
  if ((int)threadIdx.x < K) {
    double s = 0.0;
    for (int sl = 0; sl < NSLOT; ++sl) { s += h0s[((size_t)l * NSLOT + sl) * K + threadIdx.x]; h0s[((size_t)l * NSLOT + sl) * K + threadIdx.x] = 0.0; }
    const double tot = h0s[(size_t)g.L * NSLOT * K + l];
    Sk[threadIdx.x] = tot > 0.0 ? s / tot : 0.0;
  }
  __syncthreads();
  double* H0 = Hg + (size_t)l * NH * g.Y * Mp * K;
  const double* C0 = Cg + (size_t)l * g.Y * Mp;
  for (int q = threadIdx.x; q < g.M * (K - 1); q += 256) {
    const int m = q / (K - 1), k = 1 + (q - m * (K - 1));
    const double c = C0[m];
    if (c != 0.0 && Sk[k] != 0.0) H0[(size_t)m * K + k] += c * Sk[k];
  }
}
// sum_m C[l][0][m] into h0s' tail (vmr_create, once)
__global__ __launch_bounds__(256) void k_level0_total(const double* __restrict__ Cg, double* h0s, Geo g) {
  __shared__ double red[16];
  const int l = blockIdx.x;
  double s = 0.0;
  for (int m = threadIdx.x; m < g.M; m += 256) s += Cg[(size_t)l * g.Y * g.Mp + m];
  s = block_sum_n(s, red);
  if (threadIdx.x == 0) h0s[(size_t)g.L * NSLOT * g.K + l] = s;
}

// nu: -1 = no nu sum in this sweep; 0 = the raw sum to elbo_dev[1]; 1 = nu committed too (as launch_hist's)
static int launch_far(vmr_ctx* h, int count, int nu) {
  const Geo& g = h->g;
  if (!g.farl || h->far_off.empty() || h->far_off.back() == 0ull) {
    if (nu >= 0 && g.farl) return fail(h, VMR_ESTATE, "far lists missing");   // (cannot happen: farl is only chosen with far reports)
    return VMR_OK;
  }
  const size_t lb = (size_t)g.Mp * g.K * 8;
  const int LF = (int)std::max<size_t>(1, std::min<size_t>((size_t)(g.Y - g.hc), ((size_t)150 * 1024 - 256) / lb));
  const size_t sm = (size_t)LF * lb + 16 * 8;
  unsigned long long most = 0;
  for (int l = 0; l < g.L; ++l) most = std::max(most, h->far_off[l + 1] - h->far_off[l]);
  const int Gl = (int)std::max<unsigned long long>(1, std::min<unsigned long long>((most + 4095) / 4096, std::max(1, h->ncu / g.L)));
  double* nu_acc = (nu >= 0 && g.mut) ? h->nu_acc : nullptr;
  Prof p(h, VMR_KERNEL_GAMMA_COUNTS);
  DISPATCH_K(g.K, if (sm > 48 * 1024) HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_far_hist<KK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
             hipLaunchKernelGGL((k_far_hist<KK>), dim3(g.L * Gl), dim3(1024), sm, h->stream, h->far_pos, h->far_ent, h->far_base, h->rho, h->Hg, h->par,
                                nu_acc, h->elbo_dev, count, nu > 0 ? 1 : 0, Gl, LF, g));
  HIPCHK(h, hipGetLastError());
  return VMR_OK;
}

// k_fin_rho: nu and/or the ELBO; folds the NH copies of H into copy 0 on its way when they are not folded yet
static int launch_fin_rho(vmr_ctx* h, int do_nu, int do_elbo, int skip_nu = 0) {
  const Geo& g = h->g;
  const int fold = (!skip_nu && h->h_valid && !h->h_reduced && !g.gen) ? 1 : 0;
  {
    Prof p(h, VMR_KERNEL_FINALIZE);
    hipLaunchKernelGGL(k_fin_rho, dim3(g.L * FR_G), dim3(TPB), 0, h->stream, h->par, h->Hg, h->Cg, h->slotR, h->elbo_dev,
                       h->elbo_dev + 4, do_nu, do_elbo, fold, skip_nu, g, g.det ? h->fr_slots : nullptr);
  }
  HIPCHK(h, hipGetLastError());
  if (fold) h->h_reduced = true;
  return VMR_OK;
}
// before k_fin_gamma / k_fin_phi read H outside the fused sweep
static int ensure_h_folded(vmr_ctx* h) {
  if (h->h_reduced) return VMR_OK;
  const Geo& g = h->g;
  const size_t n = (size_t)g.L * g.Y * g.Mp * g.K;
  hipLaunchKernelGGL(k_h_reduce, dim3((unsigned)std::min<size_t>(1024, (n + TPB - 1) / TPB)), dim3(TPB), 0, h->stream, h->Hg, h->Cg, g);
  HIPCHK(h, hipGetLastError());
  h->h_reduced = true;
  return VMR_OK;
}

// The pass about to be launched adds the mask-list sums of its rho into slotA: whatever an earlier pass left there unconsumed goes
static int begin_sum_a(vmr_ctx* h, hipStream_t st) {
  const Geo& g = h->g;
  if (!h->a_zero) HIPCHK(h, hipMemsetAsync(h->slotA, 0, (size_t)g.L * NSLOT * g.W * 64 * g.K * 8, st));
  h->a_zero = false;
  return VMR_OK;
}
// H of the current rho (start of a fit / after vmr_set_state; the rho pass keeps it current afterwards)
// Deterministic mode: the integer shadows a pass left (SlArgs::det) become the doubles the finalize kernels read -- copy 0 of H,
// slot 0 of the mask sums, of the all-ones sums and of the ELBO partials (the other copies / slots stay zero) -- and are zeroed.
__global__ __launch_bounds__(256) void k_det_fold(unsigned long long* __restrict__ d, double* __restrict__ Hg, double* __restrict__ slotA,
                                                 double* __restrict__ slotF, double* __restrict__ slotR, Geo g) {
  const size_t nH = (size_t)g.Y * g.Mp * g.K, nA = (size_t)g.W * 64 * g.K, L = (size_t)g.L, K = (size_t)g.K;
  const size_t tot = L * nH + L * nA + L * K + 4;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < tot; q += (size_t)gridDim.x * 256) {
    const unsigned long long u = d[q];
    if (u == 0ull) continue;
    d[q] = 0ull;
    if (q < L * nH) { const size_t l = q / nH, i = q - l * nH; Hg[l * NH * nH + i] += det_back(u, g.det_sh); }
    else if (q < L * nH + L * nA) { const size_t r = q - L * nH, l = r / nA, i = r - l * nA; slotA[l * NSLOT * nA + i] += det_back(u, DET_SH_A); }
    else if (q < L * nH + L * nA + L * K) { const size_t r = q - L * nH - L * nA, l = r / K, k = r - l * K; slotF[l * NSLOT * K + k] += det_back(u, DET_SH_A); }
    else { const size_t r = q - L * nH - L * nA - L * K; slotR[r] += det_back(u, g.det_shr); }
  }
}
static int det_fold(vmr_ctx* h) {
  const Geo& g = h->g;
  if (!g.det) return VMR_OK;
  const size_t tot = (size_t)g.L * g.Y * g.Mp * g.K + (size_t)g.L * g.W * 64 * g.K + (size_t)g.L * g.K + 4;
  hipLaunchKernelGGL(k_det_fold, dim3((unsigned)std::min<size_t>(1024, (tot + 255) / 256)), dim3(256), 0, h->stream, h->det_buf, h->Hg, h->slotA, h->slotF,
                     h->slotR, g);
  HIPCHK(h, hipGetLastError());
  return VMR_OK;
}

// The last sweep used its new rho without writing it (launch_rho, store = false): write it now -- the same update once more, from the
// same log prior, theta, lambda and the nu of before that sweep's commit (SC_G_NU_STALE); no statistics, no sums, nothing else
// changes.  Called by everything that reads rho; vmr_step and the fit loops leave no stale rho behind, so this only runs after a
// loop that failed half way.
static int ensure_rho(vmr_ctx* h) {
  if (!h->rho_stale) return VMR_OK;
  const SlShape shs = sl_shape(h, true, false, false);
  SlArgs as = sl_args(h, shs, 0, 0);
  as.slotF = nullptr; as.slotA = nullptr;
  as.nu_stale = h->g.mut ? 1 : 0;
  int rc = sl_launch(h, 0, shs, as);
  if (rc) return rc;
  HIPCHK(h, hipGetLastError());
  h->rho_stale = false;
  return VMR_OK;
}

// nu: -1 = the pass leaves nu alone; 0 = it leaves the raw sum in elbo_dev[1]; 1 = it also commits nu (sorted lists, see SlArgs::nu_acc)
static int launch_hist(vmr_ctx* h, int nu = -1) {
  const Geo& g = h->g;
  if (g.gen) return gen_hist(h);
  { int rce = ensure_rho(h); if (rce) return rce; }
  HIPCHK(h, hipMemsetAsync(h->Hg, 0, (size_t)g.L * NH * g.Y * g.Mp * g.K * 8, h->stream));
  if (h->sparse) {
    HIPCHK(h, hipMemsetAsync(h->slotF, 0, (size_t)g.L * NSLOT * g.K * 8, h->stream));   // (the pass sums rho over all-ones mask rows)
    h->f_valid = g.fuse_full != 0;
    if (g.ml) { int rc = begin_sum_a(h, h->stream); if (rc) return rc; h->a_valid = true; }
    Prof p(h, VMR_KERNEL_GAMMA_COUNTS);
    const SlShape shs = sl_shape(h, false, false, true);
    SlArgs as = sl_args(h, shs, 1, g.ml);
    if (nu >= 0 && g.mut) { as.nu_acc = h->nu_acc; as.elbo_dev = h->elbo_dev; as.commit_nu = nu; }
    as.det = g.det ? h->det_buf : nullptr;
    int rcs = sl_launch(h, 3, shs, as);
    if (rcs) return rcs;
    HIPCHK(h, hipGetLastError());
    if (g.farl && (rcs = launch_far(h, 0, (nu >= 0 && g.mut) ? nu : -1))) return rcs;
    if (as.h0s) {
      hipLaunchKernelGGL(k_level0_spread, dim3(g.L), dim3(256), 0, h->stream, h->Hg, h->Cg, h->h0s, g);
      HIPCHK(h, hipGetLastError());
    }
    if ((rcs = det_fold(h))) return rcs;
  } else {
    Prof p(h, VMR_KERNEL_GAMMA_COUNTS);
    HistArgs a{h->X, h->rho, h->Hg, 1};
    size_t sm = shmem_hist(g);
    int rc = VMR_OK;
    if (g.mut) {
      DISPATCH_KP(g.K, g.pf, if ((rc = grid_per_layer(h, k_hist<KK, true, PP>, sm, &a.Gl))) return rc;
                  hipLaunchKernelGGL((k_hist<KK, true, PP>), dim3(g.L * a.Gl), dim3(TPB), sm, h->stream, a, g));
    } else {
      DISPATCH_KP(g.K, g.pf, if ((rc = grid_per_layer(h, k_hist<KK, false, PP>, sm, &a.Gl))) return rc;
                  hipLaunchKernelGGL((k_hist<KK, false, PP>), dim3(g.L * a.Gl), dim3(TPB), sm, h->stream, a, g));
    }
  }
  HIPCHK(h, hipGetLastError());
  h->h_valid = true;
  h->h_zero = false;
  h->h_reduced = false;
  return VMR_OK;
}

static int launch_gamma(vmr_ctx* h, bool with_phi) {
  const Geo& g = h->g;
  if (g.gen) return gen_gamma(h, with_phi);
  if (h->sparse && !h->h_valid) {   // report lists: the statistics pass also sums rho over the all-ones mask rows (slotF)
    int rc = launch_hist(h);
    if (rc) return rc;
  }
  // fork: the mask sums A = sum_ij R rho (memory-bound) run beside the statistics pass when one is needed
  hipStream_t ms = (h->serial || h->h_valid) ? h->stream : h->stream2;
  if (ms != h->stream) {
    HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));
    HIPCHK(h, hipStreamWaitEvent(ms, h->ev_fork, 0));
  }
  // mask rows that are all ones were already summed by the last rho pass (slotF); the mask kernel then only
  // handles partial rows, and is not needed at all when R has none
  const int skip_full = (g.fuse_full && h->f_valid) ? 1 : 0;
  if (skip_full && h->n_partial > 0 && h->rq && h->a_valid) {
    // the last rho / statistics pass summed the lists on its way (SpArgs::sum_a): nothing to launch
  } else if (skip_full && h->n_partial > 0 && h->rq) {   // partial rows only, and they are short lists
    { int rc = begin_sum_a(h, ms); if (rc) return rc; }
    Prof p(h, VMR_KERNEL_GAMMA_MASK, ms);
    const size_t T_ = (size_t)g.N * g.N;
    int gl = (int)std::min<size_t>(std::max<size_t>(1, (size_t)h->ncu / g.L), (T_ + 4 * TPB - 1) / (4 * TPB));
    const size_t lsm = (size_t)g.Mp * g.K * 8;
    if (lsm > 48 * 1024 && !h->ml_attr) {
      DISPATCH_K(g.K, HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_mask_lists<KK>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lsm)));
      h->ml_attr = true;
    }
    DISPATCH_K(g.K, hipLaunchKernelGGL((k_mask_lists<KK>), dim3(g.L * gl), dim3(TPB), lsm, ms, h->rq, h->Rm, h->rbase,
                                       h->rcls, h->rho, h->slotA, gl, h->perm, g));
  } else if (!skip_full || h->n_partial > 0) {
    Prof p(h, VMR_KERNEL_GAMMA_MASK, ms);
    dim3 grid(g.L * g.Gm), blk(TPB);
    switch (g.W >= 4 ? 4 : g.W) {
      case 1: DISPATCH_K(g.K, hipLaunchKernelGGL((k_gamma_mask<KK, 1>), grid, blk, 0, ms, h->Rb, h->rho, h->rcls, h->slotA, skip_full, h->perm, g)); break;
      case 2: DISPATCH_K(g.K, hipLaunchKernelGGL((k_gamma_mask<KK, 2>), grid, blk, 0, ms, h->Rb, h->rho, h->rcls, h->slotA, skip_full, h->perm, g)); break;
      case 3: DISPATCH_K(g.K, hipLaunchKernelGGL((k_gamma_mask<KK, 3>), grid, blk, 0, ms, h->Rb, h->rho, h->rcls, h->slotA, skip_full, h->perm, g)); break;
      default: DISPATCH_K(g.K, hipLaunchKernelGGL((k_gamma_mask<KK, 4>), grid, blk, 0, ms, h->Rb, h->rho, h->rcls, h->slotA, skip_full, h->perm, g)); break;
    }
  }
  if (ms != h->stream) {
    HIPCHK(h, hipEventRecord(h->ev_join, ms));
    int rc = launch_hist(h);
    if (rc) return rc;
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));   // join
  } else if (!h->h_valid) {
    int rc = launch_hist(h);
    if (rc) return rc;
  }
  {
    Prof p(h, VMR_KERNEL_FINALIZE);
    // fused sweep: this is the last reader of H and slotF before the rho pass rebuilds them, so it leaves them zeroed.  It
    // sums the NH copies the workgroups of the pass added into on its way (nh = 1 when something folded them already).
    const int consume = (with_phi && !g.two_pass) ? 1 : 0;
    const int nh = h->h_reduced ? 1 : NH;
    const size_t fsm = (size_t)2 * ((g.M + FG_G - 1) / FG_G) * 8;
    hipLaunchKernelGGL(g.det ? k_fin_gamma_det : k_fin_gamma, dim3(g.L * FG_G), dim3(FIN_TPB), fsm, h->stream, h->par, h->Hg, h->sparse ? h->Cg : nullptr,
                       (h->sparse && skip_full && h->n_partial == 0) ? nullptr : h->slotA,   // (lists, every mask row all ones: the pass summed them into slotF)
                       h->slotF, h->lutg, nullptr, h->fin_g, h->sparse ? h->nu_acc : nullptr, nh,
                       with_phi ? 1 : 0, consume, g);
    h->a_valid = false; h->a_zero = true;   // (k_fin_gamma zeroes the slots of A as it reads them)
    if (consume) { h->h_valid = false; h->f_valid = false; h->h_zero = true; }
  }
  HIPCHK(h, hipGetLastError());
  return VMR_OK;
}

static int launch_phi(vmr_ctx* h) {
  const Geo& g = h->g;
  if (g.gen) return gen_phi(h);
  if (!g.mut) return VMR_OK;   // committed by k_fin_gamma
  if (!h->h_valid) { int rc = launch_hist(h); if (rc) return rc; }
  { int rc = ensure_h_folded(h); if (rc) return rc; }
  {
    Prof p(h, VMR_KERNEL_FINALIZE);
    hipLaunchKernelGGL(k_fin_phi, dim3(g.L), dim3(TPB), 0, h->stream, h->par, h->Hg, g);
  }
  HIPCHK(h, hipGetLastError());
  return VMR_OK;
}

// mode: 0 = rho update (+nu), 1 = rho update + fused ELBO, 2 = ELBO only
// raw_nu: leave the raw nu sum in elbo_dev[1] although nu is not committed (vmr_sweep_local)
// store = false (mode 0 on report lists in one pass only): the pass uses its new rho without writing it (vmr_ctx::rho_stale)
static int launch_rho(vmr_ctx* h, int mode, bool commit_nu, bool raw_nu = false, bool store = true) {
  const Geo& g = h->g;
  if (g.gen) return gen_rho(h, mode, commit_nu, raw_nu, [](vmr_ctx* hh, int do_nu, int do_elbo, int skip_nu) { return launch_fin_rho(hh, do_nu, do_elbo, skip_nu); });
  if (h->sparse && g.farl && h->elbo_split && mode == 1) {
    // An ELBO sweep of a handle whose fused variant would drop an LDS level: the plain update pass (with the far reports' kernel and
    // nu), then the ELBO-only pass on the new rho with the G_nu of before the commit -- what the fused pass evaluates (model.py:970).
    int rc2 = launch_rho(h, 0, commit_nu, raw_nu, true);
    if (rc2) return rc2;
    const SlShape she = sl_shape(h, false, true, false);
    SlArgs ae = sl_args(h, she, 0, 0);
    ae.elbo_cur = commit_nu ? 0 : 1;
    {
      Prof p(h, VMR_KERNEL_ELBO);
      if ((rc2 = sl_launch(h, 2, she, ae))) return rc2;
    }
    HIPCHK(h, hipGetLastError());
    return launch_fin_rho(h, 0, 1, 1);
  }
  RhoArgs a{h->X, h->Rb, h->rho, h->logpr, h->par, h->slotR, h->lutg, h->Hg, h->slotF, 1};
  size_t sm = shmem_rho(g, mode != 2, mode != 0);
  dim3 blk(TPB);
  int rc = VMR_OK;
  if (mode != 2 && !h->h_zero) {   // (after a fused k_fin_gamma both are zero already)
    if (g.fuse_full) HIPCHK(h, hipMemsetAsync(h->slotF, 0, (size_t)g.L * NSLOT * g.K * 8, h->stream));
    if (!g.two_pass) HIPCHK(h, hipMemsetAsync(h->Hg, 0, (size_t)g.L * NH * g.Y * g.Mp * g.K * 8, h->stream));   // rebuilt from the new rho
  }
  if (mode != 2) h->h_zero = false;
  // sorted lists with mutuality: the pass that builds H finishes nu itself -- no finalize launch on plain sweeps
  const bool nu_in_pass = h->sparse && g.mut && mode != 2 && (commit_nu || raw_nu);
  if (mode == 2 && (rc = ensure_rho(h))) return rc;
  if (h->sparse) {
    const int do_hist = (mode != 2 && !g.two_pass) ? 1 : 0;
    const int sum_a = (g.ml && do_hist) ? 1 : 0;   // (two passes: the statistics pass that follows sums the lists)
    if (mode != 2) h->a_valid = false;
    if (sum_a) { if ((rc = begin_sum_a(h, h->stream))) return rc; h->a_valid = true; }
    const SlShape shs = sl_shape(h, mode != 2, mode != 0, do_hist != 0);
    SlArgs as = sl_args(h, shs, do_hist, sum_a);
    if (nu_in_pass && do_hist) { as.nu_acc = h->nu_acc; as.elbo_dev = h->elbo_dev; as.commit_nu = commit_nu ? 1 : 0; }
    as.det = g.det ? h->det_buf : nullptr;
    // the rho of a sweep that the next one overwrites unread is not written: only with mutuality's nu committed inside the pass
    // (the re-write of ensure_rho takes the nu before that commit from SC_G_NU_STALE) or without mutuality, in one pass per sweep
    // -- and only where the pass itself sums rho over the mask rows: the mask kernels of launch_gamma read rho from memory
    const bool lazy = !store && mode == 0 && do_hist && (nu_in_pass ? commit_nu : !g.mut) && g.fuse_full && (h->n_partial == 0 || g.ml) &&
                      !g.farl && !getenv("VMR_ALWAYS_STORE_RHO");   // (k_far_hist reads rho)
    {
      Prof p(h, lazy ? VMR_KERNEL_RHO_NOSTORE : mode == 2 ? VMR_KERNEL_ELBO : mode == 1 ? VMR_KERNEL_RHO_ELBO : VMR_KERNEL_RHO);
      if ((rc = sl_launch(h, lazy ? 4 : mode, shs, as))) return rc;
    }
    if (mode != 2) h->rho_stale = lazy;
    if (g.farl && do_hist && (rc = launch_far(h, 0, nu_in_pass ? (commit_nu ? 1 : 0) : -1))) return rc;
    if ((rc = det_fold(h))) return rc;
  } else {
    Prof p(h, mode == 2 ? VMR_KERNEL_ELBO : mode == 1 ? VMR_KERNEL_RHO_ELBO : VMR_KERNEL_RHO);
#define LRHO(MUT_, UPD_, ELB_)                                                                  \
  DISPATCH_KP(g.K, g.pf, if ((rc = grid_per_layer(h, k_rho<KK, MUT_, UPD_, ELB_, PP>, sm, &a.Gl))) return rc; \
              hipLaunchKernelGGL((k_rho<KK, MUT_, UPD_, ELB_, PP>), dim3(g.L * a.Gl), blk, sm, h->stream, a, g))
    if (g.mut) {
      if (mode == 0) { LRHO(true, true, false); } else if (mode == 1) { LRHO(true, true, true); } else { LRHO(true, false, true); }
    } else {
      if (mode == 0) { LRHO(false, true, false); } else if (mode == 1) { LRHO(false, true, true); } else { LRHO(false, false, true); }
    }
#undef LRHO
  }
  if (mode != 2) {
    h->f_valid = g.fuse_full != 0;
    h->h_valid = !g.two_pass;
    if (!g.two_pass) h->h_reduced = false;
    if (g.two_pass && (rc = launch_hist(h, nu_in_pass ? (commit_nu ? 1 : 0) : -1))) return rc;   // wide reporter dimension: second pass rebuilds H
  }
  HIPCHK(h, hipGetLastError());
  if (nu_in_pass || (h->sparse && mode != 2 && !g.mut)) {
    // nu is done (or there is none): the finalize kernel only assembles an ELBO
    if (mode != 0 && (rc = launch_fin_rho(h, 0, 1, 1))) return rc;
  } else if (mode != 0 || commit_nu) {
    if ((rc = launch_fin_rho(h, (mode != 2 && commit_nu) ? 1 : 0, mode != 0 ? 1 : 0))) return rc;
  }
  return VMR_OK;
}

static int choose_geo(Geo& g, int ncu, std::string& err) {
  g.Mp = (g.M + 15) / 16 * 16;
  g.nchunk = g.Mp / 16;
  g.stride = (g.nchunk % 2 == 1) ? g.Mp : g.Mp + 16;
  g.W = (g.M + 63) / 64;
  const size_t budget = 48 * 1024;
  // per-reporter LDS tables of the rho pass beside the tile pair (see shmem_rho): they cap the tile edge too
  const size_t tables = (size_t)g.Mp * 8 * (2 + (g.mut ? g.K : 0)) + (size_t)g.W * 8 + shmem_q() + 256;
  auto lds = [&](int b_) { return (size_t)2 * b_ * b_ * (g.stride + (size_t)g.W * 8 + 2 * g.K * 8 + 4) + tables; };
  int b = 8;
  while (b > 1 && ((size_t)2 * b * b * g.stride > budget || lds(b) > 160 * 1024)) b >>= 1;
  if ((size_t)2 * b * b * g.stride > budget) { err = "M too large for the LDS tile (M <= ~24500 supported)"; return VMR_EINVAL; }
  g.b = b; g.lb = (b == 8) ? 3 : (b == 4) ? 2 : (b == 2) ? 1 : 0;
  g.nb = (g.N + b - 1) / b;
  g.nt = 2 * b * b;
  g.S = TPB / g.nt; if (g.S > 64) g.S = 64;
  g.lS = 0; while ((1 << g.lS) < g.S) ++g.lS;
  g.P = (long long)g.nb * (g.nb + 1) / 2;
  g.Gl = 0;   // per launch, see grid_per_layer()
  g.Y = 1;    // set by vmr_create once the largest count is known
  g.hc = 0;
  g.two_pass = 0;
  g.farl = 0;
  g.fuse_full = 0;
  long long T = (long long)g.N * g.N;
  long long gm = (long long)ncu * 8 / g.L; if (gm < 1) gm = 1;
  long long maxgm = (T + 255) / 256; if (gm > maxgm) gm = maxgm; if (gm < 1) gm = 1;
  g.Gm = (int)gm;
  int need = (g.nt * g.nchunk + TPB - 1) / TPB;
  g.pf = need <= 4 ? 4 : need <= 8 ? 8 : 12;
  const char* hv = getenv("VMR_HEAVY");
  g.heavy = hv ? atoi(hv) : 8;   // in non-zero DWORDS of the share
  const char* dbg = getenv("VMR_DEBUG");
  g.dbg = dbg ? atoi(dbg) : 0;
  return VMR_OK;
}

extern "C" {

const char* vmr_version(void) { return VMR_VERSION; }

const char* vmr_last_error(vmr_handle h) { return h ? h->err.c_str() : g_create_err.c_str(); }

}  // extern "C" (the create helpers are C++)

// ------------------------------------------------------------------------------------------
// vmr_create / vmr_create_coo
// ------------------------------------------------------------------------------------------

// context, geometry, streams and the small per-dataset arrays
static int create_ctx(vmr_ctx** out, hipDeviceProp_t* prop, int device, int L, int N, int M, int K, int mutuality, double eps) {
  if (L < 1 || N < 1 || M < 1) return fail(nullptr, VMR_EINVAL, "L, N, M must be positive");
  if (K < 2 || K > KGEN_MAX) return fail(nullptr, VMR_EINVAL, "K must be in [2, 256]");
  int ndev = 0;
  CK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(nullptr, VMR_EINVAL, "no such HIP device");
  CK(hipSetDevice(device));
  CK(hipGetDeviceProperties(prop, device));
  vmr_ctx* h = new vmr_ctx();
  *out = h;
  h->device = device;
  Geo& g = h->g;
  g.L = L; g.N = N; g.M = M; g.K = K; g.mut = mutuality ? 1 : 0; g.eps = eps;
  g.gen = K > KMAX ? 1 : 0; g.wide = 0;   // (wide entries: decided once the largest count is known)
  std::string err;
  if (choose_geo(g, prop->multiProcessorCount, err) != VMR_OK) return fail(nullptr, VMR_EINVAL, err.c_str());
  memset(h->prof_ms, 0, sizeof h->prof_ms); memset(h->prof_n, 0, sizeof h->prof_n);
  CK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
  CK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
  h->ncu = prop->multiProcessorCount;
  h->serial = getenv("VMR_SERIAL") != nullptr;
  h->use_graphs = getenv("VMR_GRAPH") ? atoi(getenv("VMR_GRAPH")) != 0 : false;   // (off by default: see vmr_ctx::graphs)
  { const char* dv = getenv("VMR_DETERMINISTIC"); g.det = (dv && atoi(dv) != 0) ? 1 : 0; g.det_sh = 0; g.det_shr = 0; }   // sorted report lists unless the older step layout is asked for
  const size_t rows = (size_t)L * N * N;
  CK(hipMalloc(&h->cov, rows));
  CK(hipMalloc(&h->rcls, rows));
  CK(hipMalloc(&h->sumx, 8));
  CK(hipMemsetAsync(h->sumx, 0, 8, h->stream));
  CK(hipMalloc(&h->xmax, 4));
  CK(hipMemsetAsync(h->xmax, 0, 4, h->stream));
  CK(hipMalloc(&h->npartial, 16));
  CK(hipMemsetAsync(h->npartial, 0, 16, h->stream));
  return VMR_OK;
}

// variational state and accumulation slots; reads back the data statistics (largest count, mask row classes)
static int create_state(vmr_ctx* h, unsigned* xm_out) {
  Geo& g = h->g;
  const size_t rows = (size_t)g.L * g.N * g.N;
  // (64 rows of zeroed slack: the sweep over the sorted lists reads whole 64-tie steps without masks)
  CK(hipMalloc(&h->rho, (rows + 64) * g.K * 8));
  CK(hipMalloc(&h->logpr, (rows + 64) * g.K * 8));
  CK(hipMemsetAsync(h->rho + rows * g.K, 0, (size_t)64 * g.K * 8, h->stream));
  CK(hipMemsetAsync(h->logpr + rows * g.K, 0, (size_t)64 * g.K * 8, h->stream));
  ParOff o = par_off(g.L, g.Mp, g.K);
  h->par_doubles = o.total;
  CK(hipMalloc(&h->par, o.total * 8));
  CK(hipMemsetAsync(h->par, 0, o.total * 8, h->stream));
  const size_t nA = (size_t)g.L * NSLOT * g.W * 64 * g.K * 8;
  CK(hipMalloc(&h->slotA, nA)); CK(hipMemsetAsync(h->slotA, 0, nA, h->stream));
  CK(hipMalloc(&h->slotR, NSLOT * 4 * 8)); CK(hipMemsetAsync(h->slotR, 0, NSLOT * 4 * 8, h->stream));
  g.fuse_full = 1;
  CK(hipMalloc(&h->slotF, (size_t)g.L * NSLOT * g.K * 8));
  CK(hipMemsetAsync(h->slotF, 0, (size_t)g.L * NSLOT * g.K * 8, h->stream));
  CK(hipStreamSynchronize(h->stream));   // the stream is non-blocking: a plain hipMemcpy does not wait for it
  unsigned xm = 0;
  unsigned long long np2[2] = {0, 0};
  CK(hipMemcpy(&xm, h->xmax, 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(np2, h->npartial, 16, hipMemcpyDeviceToHost));
  h->n_partial = np2[0];
  h->all_full = (np2[0] == 0 && np2[1] == 0) ? 1 : 0;
  g.Y = g.mut ? (int)xm + 1 : 1;   // mirror counts 0..max(X)
  *xm_out = xm;
  return VMR_OK;
}

int scan_u32(vmr_ctx* h, unsigned* a, unsigned* bsum, size_t n) {
  const unsigned nb = (unsigned)((n + 2047) / 2048);
  hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(256), 0, h->stream, a, bsum, n);
  hipLaunchKernelGGL(k_scan_bsum, dim3(1), dim3(256), 0, h->stream, bsum, (int)nb);
  hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(256), 0, h->stream, a, bsum, n);
  CK(hipGetLastError());
  return VMR_OK;
}


// mask lists for partial rows that hold few reporters (self-reporter masks: two per row), from the bit-packed words
static int mask_lists_from_words(vmr_ctx* h) {
  Geo& g = h->g;
  const int L = g.L, K = g.K;
  const size_t T = (size_t)g.N * g.N, n = T + 1, rows = (size_t)L * T;
  if (!(h->n_partial > 0) || getenv("VMR_NO_RLISTS")) return VMR_OK;
  unsigned long long* tot_dev = nullptr;
  unsigned* max_dev = nullptr;
  unsigned* bsum = nullptr;
  CK(hipMalloc(&h->rq, (size_t)L * n * 4));
  CK(hipMalloc(&tot_dev, (size_t)L * 8));
  CK(hipMalloc(&max_dev, 4));
  CK(hipMalloc(&bsum, (n + 2047) / 2048 * 4));
  CK(hipMemsetAsync(tot_dev, 0, (size_t)L * 8, h->stream));
  CK(hipMemsetAsync(max_dev, 0, 4, h->stream));
  const unsigned rgrid = (unsigned)std::min<size_t>(4096, (T + 255) / 256);
  for (int l = 0; l < L; ++l)
    hipLaunchKernelGGL(k_rm_count, dim3(rgrid), dim3(256), 0, h->stream, h->Rb + (size_t)l * T * g.W,
                       h->rcls + (size_t)l * T, h->rq + (size_t)l * n, tot_dev + l, max_dev, T, g.W);
  CK(hipGetLastError());
  CK(hipStreamSynchronize(h->stream));
  std::vector<unsigned long long> rl_(L), rb_(L);
  unsigned maxrow = 0;
  CK(hipMemcpy(rl_.data(), tot_dev, (size_t)L * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&maxrow, max_dev, 4, hipMemcpyDeviceToHost));
  h->rm_maxrow = maxrow;
  CK(hipFree(tot_dev));
  CK(hipFree(max_dev));
  bool ok32 = true;
  h->n_rm = 0;
  for (int l = 0; l < L; ++l) { rb_[l] = h->n_rm; h->n_rm += rl_[l]; ok32 = ok32 && rl_[l] < 0xffffffffull; }
  // worth it when a list row is at most a quarter of the row's mask words (and rows are short: one lane walks a row) -- and
  // for small networks whatever the bytes: the sweep's pass sums the lists on its way, a launch less per sweep where the
  // launches are what a sweep costs
  const double list_bytes = 2.0 * (double)h->n_rm + 4.0 * (double)rows, word_bytes = (double)h->n_partial * g.W * 8.0;
  if (ok32 && maxrow <= 64 && (list_bytes * 4.0 <= word_bytes || g.N <= 1024) && (size_t)g.Mp * K * 8 <= 160 * 1024) {   // (A[Mp][K] of k_mask_lists lives in LDS)
    for (int l = 0; l < L; ++l) { int rc = scan_u32(h, h->rq + (size_t)l * n, bsum, n); if (rc) return rc; }
    CK(hipMalloc(&h->rbase, (size_t)L * 8));
    CK(hipMemcpyAsync(h->rbase, rb_.data(), (size_t)L * 8, hipMemcpyHostToDevice, h->stream));
    CK(hipMalloc(&h->Rm, ((size_t)h->n_rm + 64) * 2));
    for (int l = 0; l < L; ++l)
      hipLaunchKernelGGL(k_rm_fill, dim3(rgrid), dim3(256), 0, h->stream, h->Rb + (size_t)l * T * g.W,
                         h->rcls + (size_t)l * T, h->rq + (size_t)l * n, h->Rm + rb_[l], T, g.W);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(h->stream));
  } else {
    CK(hipFree(h->rq));
    h->rq = nullptr;
    h->n_rm = 0;
  }
  CK(hipFree(bsum));
  return VMR_OK;
}

// sorted lists: the per-tie arrays the sweeps read, by position (the tie-order originals stay for the mask kernels)
// reports per level (mirror count) of the sorted lists: hist[y], y clamped to 64; empty slots (x = 0, not in R) are skipped
__global__ __launch_bounds__(256) void k_level_hist(const unsigned* __restrict__ E, unsigned long long n, int Mp, unsigned long long* __restrict__ hist) {
  __shared__ unsigned long long sh[65];
  for (int i = threadIdx.x; i < 65; i += 256) sh[i] = 0;
  __syncthreads();
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) {
    const unsigned e = E[i];
    if (e == 0u) continue;
    const unsigned y = SL_YM(e) / (unsigned)Mp;
    atomicAdd(&sh[y < 64u ? y : 64u], 1ull);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 65; i += 256) if (sh[i]) atomicAdd(&hist[i], sh[i]);
}

// Geo::farl: (position, entry) of the reports of levels >= g.hc, layer by layer (k_far_collect)
static int build_far_lists(vmr_ctx* h) {
  const Geo& g = h->g;
  const size_t T = (size_t)g.N * g.N, NS = (T + 63) / 64;
  unsigned long long* cnt = nullptr;
  unsigned long long* hd = nullptr;
  struct Tmp { unsigned long long*& a; unsigned long long*& b; ~Tmp() { if (a) (void)hipFree(a); if (b) (void)hipFree(b); } } tmp_guard{cnt, hd};   // (freed on every way out)
  CK(hipMalloc(&cnt, 8));
  std::vector<unsigned long long> hist(65, 0);
  CK(hipMalloc(&hd, 65 * 8));
  CK(hipMemsetAsync(hd, 0, 65 * 8, h->stream));
  hipLaunchKernelGGL(k_level_hist, dim3(1024), dim3(256), 0, h->stream, h->E, h->n_slots, g.Mp, hd);
  CK(hipGetLastError());
  CK(hipMemcpyAsync(hist.data(), hd, 65 * 8, hipMemcpyDeviceToHost, h->stream));
  CK(hipStreamSynchronize(h->stream));
  unsigned long long far = 0;
  for (int y = std::min(g.hc, 64); y < 65; ++y) far += hist[y];   // (an upper bound: empty slots of a level count too)
  if (getenv("VMR_VERBOSE")) {
    fprintf(stderr, "vimure_hip: far lists from level %d; reports per level:", g.hc);
    for (int y = 0; y < 65; ++y) if (hist[y]) fprintf(stderr, " %d:%llu", y, hist[y]);
    fprintf(stderr, "\n");
  }
  CK(hipMalloc(&h->far_pos, (size_t)(far + 64) * 4));
  CK(hipMalloc(&h->far_ent, (size_t)(far + 64) * 4));
  h->far_off.assign(g.L + 1, 0ull);
  std::vector<unsigned long long> eb(g.L);
  CK(hipMemcpy(eb.data(), h->ebase, (size_t)g.L * 8, hipMemcpyDeviceToHost));
  for (int l = 0; l < g.L; ++l) {
    hipLaunchKernelGGL(k_far_first, dim3((unsigned)std::min<size_t>(8192, (NS + 3) / 4)), dim3(256), 0, h->stream, h->E + eb[l], h->rs + (size_t)l * (NS + 1),
                       h->sy + (size_t)l * NS, NS, (unsigned)g.hc * (unsigned)g.Mp, (unsigned*)nullptr);
    CK(hipGetLastError());
    CK(hipMemsetAsync(cnt, 0, 8, h->stream));
    hipLaunchKernelGGL(k_far_collect, dim3((unsigned)std::min<size_t>(8192, (NS + 3) / 4)), dim3(256), 0, h->stream, h->E + eb[l], h->rs + (size_t)l * (NS + 1), NS,
                       (unsigned)g.hc * (unsigned)g.Mp, h->far_pos + h->far_off[l], h->far_ent + h->far_off[l], cnt);
    CK(hipGetLastError());
    unsigned long long n = 0;
    CK(hipMemcpyAsync(&n, cnt, 8, hipMemcpyDeviceToHost, h->stream));
    CK(hipStreamSynchronize(h->stream));
    h->far_off[l + 1] = h->far_off[l] + n;
    if (h->far_off[l + 1] > far) return fail(nullptr, VMR_EHIP, "far lists: more reports than counted");
  }
  CK(hipMalloc(&h->far_base, (size_t)(g.L + 1) * 8));
  CK(hipMemcpy(h->far_base, h->far_off.data(), (size_t)(g.L + 1) * 8, hipMemcpyHostToDevice));
  return VMR_OK;
}

// the mask lists of at most two reporters, packed by sorted position (SlArgs::rm2)
__global__ __launch_bounds__(256) void k_rm2(const unsigned* __restrict__ perm, const unsigned* __restrict__ rq, const unsigned short* __restrict__ Rm,
                                             const unsigned long long* __restrict__ rbase, unsigned* __restrict__ out, size_t T, size_t NS, int L) {
  const size_t n = (size_t)L * NS * 64;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (size_t)gridDim.x * 256) {
    const size_t l = q / (NS * 64);
    const unsigned tie = perm[q];
    unsigned w = 0xffffffffu;
    if (tie != 0xffffffffu) {
      const unsigned q0 = rq[l * (T + 1) + tie], q1 = rq[l * (T + 1) + tie + 1];
      const unsigned short* r = Rm + rbase[l];
      const unsigned m0 = q1 > q0 ? r[q0] : 0xffffu, m1 = q1 > q0 + 1 ? r[q0 + 1] : 0xffffu;
      w = m0 | (m1 << 16);
    }
    out[q] = w;
  }
}

static int sl_finish(vmr_ctx* h) {
  const Geo& g = h->g;
  const size_t rows = (size_t)g.L * g.N * g.N;
  int rc = VMR_OK;
  if (!h->all_full) {
    CK(hipMalloc(&h->cls_p, rows + 64));   // (64 entries of slack, as rho)
    CK(hipMemsetAsync(h->cls_p + rows, 0, 64, h->stream));
    if ((rc = sl_permute_u8(h, h->rcls, h->cls_p))) return rc;
  }
  if (g.mut && h->Qt) {
    CK(hipMalloc(&h->Qt_p, (rows + 64) * 4));
    CK(hipMemsetAsync(h->Qt_p + rows, 0, 64 * 4, h->stream));
    if ((rc = sl_permute_u32(h, h->Qt, h->Qt_p))) return rc;
    CK(hipStreamSynchronize(h->stream));
    CK(hipFree(h->Qt)); h->Qt = nullptr;   // (only the sweeps read it)
  }
  return VMR_OK;
}

// LDS shapes, the statistics / factor tables, scratch; for report lists the count-mode launch (constants C[l][y][m])
static int create_tail(vmr_ctx* h, const hipDeviceProp_t& prop) {
  Geo& g = h->g;
  const int L = g.L, K = g.K;
  g.ml = (h->sparse && h->rq) ? 1 : 0;
  if (g.ml && !g.gen && h->rm_maxrow <= 2 && !getenv("VMR_NO_RM2")) {   // (the self-reporter mask of survey data: lists of two)
    const size_t T_ = (size_t)g.N * g.N, NS_ = (T_ + 63) / 64, n_ = (size_t)L * NS_ * 64;
    CK(hipMalloc(&h->rm2, n_ * 4));
    hipLaunchKernelGGL(k_rm2, dim3((unsigned)std::min<size_t>(4096, (n_ + 255) / 256)), dim3(256), 0, h->stream, h->perm, h->rq, h->Rm, h->rbase, h->rm2, T_, NS_, L);
    CK(hipGetLastError());
  }
  if (g.gen) {
    // the general kernels: one copy of H with every category, nothing in LDS, no constants C (sweep_gen.h)
    if (g.det) return fail(nullptr, VMR_EINVAL, "VMR_DETERMINISTIC=1 needs the specialised kernels: K <= 8, counts <= 2047, (largest count + 1) * M <= 2^20");
    g.ml = 0; g.hc = 0; g.yt = 0; g.two_pass = 0;
    if (h->rm2) { CK(hipFree(h->rm2)); h->rm2 = nullptr; }
    const double hb = (double)L * g.Y * g.Mp * K * 8.0;
    size_t fr = 0, tot = 0;
    CK(hipMemGetInfo(&fr, &tot));
    if (hb > 0.5 * (double)fr) {
      char msg[256];
      snprintf(msg, sizeof msg, "the statistics table H[L][max count + 1][M][K] would take %.1f GB (largest count %d, M = %d, K = %d): more than half "
               "of the device memory left", hb / 1e9, g.Y - 1, g.M, K);
      return fail(nullptr, VMR_EINVAL, msg);
    }
    CK(hipMalloc(&h->Hg, (size_t)hb));
    CK(hipMemsetAsync(h->Hg, 0, (size_t)hb, h->stream));
    CK(hipMalloc(&h->gen_s1, (size_t)L * (g.Mp + K + 1) * 8));   // (zero between uses: the readers zero what they read)
    CK(hipMemsetAsync(h->gen_s1, 0, (size_t)L * (g.Mp + K + 1) * 8, h->stream));
    CK(hipMalloc(&h->elbo_dev, 8 * 8));
    CK(hipMemsetAsync(h->elbo_dev, 0, 8 * 8, h->stream));
    CK(hipMalloc(&h->nu_acc, (size_t)(3 + L) * 8));
    CK(hipMemsetAsync(h->nu_acc, 0, (size_t)(3 + L) * 8, h->stream));
    CK(hipMalloc(&h->lutg, (size_t)L * g.W * 256 * 8));
    CK(hipMemsetAsync(h->lutg, 0, (size_t)L * g.W * 256 * 8, h->stream));
    CK(hipStreamSynchronize(h->stream));
    return VMR_OK;
  }
  // LDS levels (mirror counts 0..) of the statistics H and, for report lists, of the factor table F.
  g.hc = g.Y < HC_MAX ? g.Y : HC_MAX;
  g.yt = 0;
  g.two_pass = 0;
  size_t need = 0;
  if (h->sparse) {
    // Sorted lists: the populous levels of F (read) and H (float atomics) in LDS, shared by all waves of a workgroup; no per-wave
    // LDS at all.  One pass per sweep when every level of both fits at >= 16 waves per CU (or the levels beyond hold under 0.1 %
    // of the reports), else the rho pass keeps F and a statistics pass rebuilds H (two passes over the entries).
    auto env_i = [](const char* n, int dflt) { const char* e = getenv(n); return e ? atoi(e) : dflt; };
    const int want = std::max(1, std::min(g.Y, env_i("VMR_LEVELS", 64)));   // as many levels as fit: with all of them in LDS nothing is "far"
    // a variant's largest workgroup / waves per CU (registers): the statistics-only variant is lighter than the update's
    auto cap_of = [&](bool upd) { return sl_tpb_max(K, false, h->all_full != 0, upd); };
    auto wcu_of = [&](bool upd) { return 4 * sl_wpe(K, false, h->all_full != 0, upd); };
    auto waves = [&](int tpb, int yt, int hc, bool upd, bool hist) {
      const size_t b = sl_smem(g, yt, hc, upd, false, hist);
      if (b > SP_LDS_MAX || tpb > cap_of(upd)) return 0;
      const int nw = tpb / 64, wgs = std::min((int)(SP_LDS_MAX / b), wcu_of(upd) / nw);
      return wgs * nw;
    };
    auto best = [&](bool upd, bool hist, int min_waves, int& lv_out, int& tpb_out) {
      min_waves = std::min(min_waves, wcu_of(upd));
      for (int lv = want; lv >= 1; --lv) {
        int bw = 0, bt = 256;
        for (int tpb : {1024, 768, 512, 256}) { const int w = waves(tpb, upd ? lv : 0, hist ? lv : 0, upd, hist); if (w > bw) { bw = w; bt = tpb; } }
        if (bw >= min_waves) { lv_out = lv; tpb_out = bt; return true; }
      }
      return false;
    };
    int lv1 = 0, t1 = 256, lvr = 0, tr = 256, lvh = 0, th = 256;
    bool want_farl = false;
    bool one = best(true, true, 16, lv1, t1);
    if (one && lv1 < want && g.N > 1024) {   // (small networks: one pass whatever the levels -- a sweep there costs its launches)
      // Not every level fits beside the other table.  A report of a level beyond the LDS ones costs its whole 64-tie round the
      // slow path (the factor formula, a global add), so one pass only pays while such reports are rare: count the reports
      // per level.  (BASELINE config 5 -- M = 1000, K = 3: 3 levels of each fit, 2.5 % of the reports lie beyond, four rounds
      // of five hold one; the two passes keep 6 levels of F and 9 of H.)
      std::vector<unsigned long long> hist(65, 0);
      unsigned long long* hd = nullptr;
      CK(hipMalloc(&hd, 65 * 8));
      CK(hipMemsetAsync(hd, 0, 65 * 8, h->stream));
      hipLaunchKernelGGL(k_level_hist, dim3(1024), dim3(256), 0, h->stream, h->E, h->n_slots, g.Mp, hd);
      CK(hipGetLastError());
      CK(hipMemcpyAsync(hist.data(), hd, 65 * 8, hipMemcpyDeviceToHost, h->stream));
      CK(hipStreamSynchronize(h->stream));
      CK(hipFree(hd));
      unsigned long long tot = 0, far = 0;
      for (int y = 0; y < 65; ++y) { tot += hist[y]; if (y >= lv1) far += hist[y]; }
      if ((double)far > (double)tot / 1024.0) {
        // VMR_FARL=1 (round 4; off by default): still one pass while the far reports are a few per cent -- the pass takes their factors
        // by formula and leaves their statistics to k_far_hist, which adds them from a compact list of exactly those reports, moved
        // to the front of every tie's list (k_far_first).  Measured on a BASELINE config-5 layer: 3.44 ms per sweep against 3.53
        // with two passes, and ELBO sweeps the other way round -- no gain over ten sweeps, so two passes stay.
        // Not in the deterministic mode (its sums are the integer shadows').
        if (g.mut && !g.det && env_i("VMR_FARL", 0) && (double)far <= 0.125 * (double)tot) want_farl = true;
        else one = false;
      }
    }
    if (getenv("VMR_TWO_PASS")) one = one && !env_i("VMR_TWO_PASS", 0);
    if (one) { g.yt = g.hc = lv1; h->sp_tpb = h->st_tpb = t1; g.farl = want_farl ? 1 : 0; }
    else {
      g.two_pass = 1;
      if (!best(true, false, 16, lvr, tr) && !best(true, false, 4, lvr, tr)) { lvr = 0; tr = 256; }
      if (!best(false, true, 16, lvh, th) && !best(false, true, 4, lvh, th)) { lvh = 0; th = 256; }
      g.yt = lvr; g.hc = lvh; h->sp_tpb = tr; h->st_tpb = th;
    }
    {   // small datasets: smaller workgroups, so that the steps spread over every CU
      const long long NS = ((long long)g.N * g.N + 63) / 64;
      for (int* t : {&h->sp_tpb, &h->st_tpb})
        while (*t > 64 && NS * L < (long long)h->ncu * (*t / 64)) *t = std::max(64, (*t / 2) & ~63);   // (768 -> 384 -> 192 -> 64)
    }
    g.yt = std::max(0, std::min(g.Y, env_i("VMR_YT", g.yt)));
    g.hc = std::max(0, std::min(g.Y, env_i("VMR_HC", g.hc)));
    { const int t = env_i("VMR_TPB", h->sp_tpb); if (t >= 64 && t <= 1024 && t % 64 == 0) h->sp_tpb = t; }
    { const int t = env_i("VMR_ST_TPB", h->st_tpb); if (t >= 64 && t <= 1024 && t % 64 == 0) h->st_tpb = t; }
    for (int v = 0; v < 4; ++v) need = std::max(need, sl_shape(h, v != 3, v == 1 || v == 2, v == 0 || v == 1 || v == 3).smem);
    if (g.two_pass && g.mut && !g.det && g.Y > 1 && !getenv("VMR_NO_LEVEL0")) {
      // Two passes (a wide reporter dimension): the statistics pass is bound by its LDS adds, and at mirror count 0 -- more than half
      // of the reports of a mutual network -- none is needed (SlArgs::h0s): every tie's reports of count >= 1 go first, sy's high half
      // gets the first round of a step that holds level 0 only.
      const size_t T_ = (size_t)g.N * g.N, NS_ = (T_ + 63) / 64;
      std::vector<unsigned long long> eb(L);
      CK(hipMemcpy(eb.data(), h->ebase, (size_t)L * 8, hipMemcpyDeviceToHost));
      if (!getenv("VMR_NO_X0")) CK(hipMalloc(&h->x0p, (size_t)L * NS_ * 64 * 4));
      for (int l = 0; l < L; ++l)
        hipLaunchKernelGGL(k_far_first, dim3((unsigned)std::min<size_t>(8192, (NS_ + 3) / 4)), dim3(256), 0, h->stream, h->E + eb[l], h->rs + (size_t)l * (NS_ + 1),
                           h->sy + (size_t)l * NS_, NS_, (unsigned)g.Mp, h->x0p ? h->x0p + (size_t)l * NS_ * 64 : nullptr);
      CK(hipGetLastError());
      if (h->x0p) {
        unsigned long long* ss = nullptr;
        CK(hipMalloc(&ss, 8));
        CK(hipMemsetAsync(ss, 0, 8, h->stream));
        for (int l = 0; l < L; ++l)
          hipLaunchKernelGGL(k_stat_slots, dim3(256), dim3(256), 0, h->stream, h->rs + (size_t)l * (NS_ + 1), h->sy + (size_t)l * NS_, NS_, ss);
        hipError_t es = hipMemcpyAsync(&h->stat_slots, ss, 8, hipMemcpyDeviceToHost, h->stream);
        if (es == hipSuccess) es = hipStreamSynchronize(h->stream);
        (void)hipFree(ss);
        CK(es);
      }
      const size_t n0 = (size_t)L * NSLOT * K + L;
      CK(hipMalloc(&h->h0s, n0 * 8));
      CK(hipMemsetAsync(h->h0s, 0, n0 * 8, h->stream));
    }
    if (g.farl) {
      // the far lists hold the reports of levels >= g.hc: every variant that adds to H must keep exactly the levels below in LDS
      bool same = g.hc >= 1 && g.hc < g.Y;
      for (int v : {0, 3}) same = same && sl_shape(h, v != 3, false, true).hc == g.hc;
      h->elbo_split = sl_shape(h, true, true, true).hc != g.hc || sl_shape(h, true, true, true).yt != g.yt;   // (the fused ELBO variant also holds the logarithm table)
      if (!same) g.farl = 0;   // (shapes forced through VMR_YT / VMR_HC / VMR_TPB: the pass then adds the far reports itself, global adds)
      else { int rcf = build_far_lists(h); if (rcf) return rcf; }
    }
  } else {
    if (shmem_rho(g, true, false) > 80000) {
      g.two_pass = 1;
      while (g.hc > 0 && shmem_hist(g) > 160000) --g.hc;
    }
    need = std::max(shmem_rho(g, true, true), shmem_hist(g));
  }
  // per-reporter tables live in LDS (dense tiles: E[log theta], the mutuality weights c[m,k], G_theta for the ELBO:
  // (K + 2) * 8 bytes per reporter).  160 KB per workgroup on gfx950.
  if (need > (size_t)prop.sharedMemPerBlock && need > 160 * 1024) {
    char msg[256];
    snprintf(msg, sizeof msg, "M = %d reporters with K = %d%s needs %zu bytes of LDS per workgroup (limit %d): "
             "reduce M or K", g.M, K, g.mut ? " and mutuality" : "", need, 160 * 1024);
    return fail(nullptr, VMR_EINVAL, msg);
  }
  CK(hipMalloc(&h->Hg, (size_t)L * NH * g.Y * g.Mp * K * 8));
  CK(hipMemsetAsync(h->Hg, 0, (size_t)L * NH * g.Y * g.Mp * K * 8, h->stream));
  CK(hipMalloc(&h->elbo_dev, 8 * 8));   // [0..3] results, [4..6] scratch of k_fin_rho
  CK(hipMemsetAsync(h->elbo_dev, 0, 8 * 8, h->stream));
  CK(hipMalloc(&h->nu_acc, (size_t)(3 + L) * 8));   // the nu update inside the pass (SlArgs::nu_acc)
  CK(hipMemsetAsync(h->nu_acc, 0, (size_t)(3 + L) * 8, h->stream));
  // k_fin_gamma: per layer 2 K sums and a ticket; behind them, for the deterministic mode, every workgroup's own 2 K sums
  CK(hipMalloc(&h->fin_g, ((size_t)L * (2 * KMAX + 2) + (size_t)L * FG_G * 2 * KMAX) * 8));
  CK(hipMemsetAsync(h->fin_g, 0, ((size_t)L * (2 * KMAX + 2) + (size_t)L * FG_G * 2 * KMAX) * 8, h->stream));
  if (g.det) {
    // VMR_DETERMINISTIC=1 (sorted report lists only): see SlArgs::det.  The fixed point of the count-weighted sums leaves
    // room for the sum of all counts.
    if (!h->sparse) return fail(nullptr, VMR_EINVAL, "VMR_DETERMINISTIC=1 needs the report lists (a sparse tensor with M <= 8192, counts <= 2047)");
    unsigned long long sx = 0;
    CK(hipMemcpyAsync(&sx, h->sumx, 8, hipMemcpyDeviceToHost, h->stream));
    CK(hipStreamSynchronize(h->stream));
    int bits = 1;
    while (bits < 62 && (sx >> bits) != 0ull) ++bits;
    g.det_sh = std::min(40, std::max(0, 61 - bits));    // (<= 40: the conversion trick of the sweep kernel holds |v| 2^sh < 2^51, v up to 2047)
    const unsigned long long eb = 64ull * (sx + (unsigned long long)L * g.N * g.N * K);
    int rbits = 1;
    while (rbits < 62 && (eb >> rbits) != 0ull) ++rbits;
    g.det_shr = std::min(34, std::max(0, 61 - rbits));   // (<= 34: ELBO terms up to 2047 * |log eps| < 2^17)
    const size_t nd = (size_t)L * g.Y * g.Mp * K + (size_t)L * g.W * 64 * K + (size_t)L * K + 5;
    CK(hipMalloc(&h->det_buf, nd * 8));
    CK(hipMemsetAsync(h->det_buf, 0, nd * 8, h->stream));
    CK(hipMalloc(&h->fr_slots, (size_t)L * FR_G * 2 * 8));
    CK(hipMemsetAsync(h->fr_slots, 0, (size_t)L * FR_G * 2 * 8, h->stream));
  }
  CK(hipMalloc(&h->lutg, (size_t)L * g.W * 256 * 8));
  CK(hipMemsetAsync(h->lutg, 0, (size_t)L * g.W * 256 * 8, h->stream));
  if (h->sparse) {
    // the constants C[l][y][m] = sum of the counts per (mirror count, reporter): one statistics launch in count mode
    CK(hipMalloc(&h->Cg, (size_t)L * g.Y * g.Mp * 8));
    int rc = VMR_OK;
    {
      const SlShape sh = sl_shape(h, false, false, true);
      SlArgs a = sl_args(h, sh, 2);
      rc = sl_launch(h, 3, sh, a);
      if (rc == VMR_OK && g.farl) rc = launch_far(h, 1, -1);
    }
    if (rc != VMR_OK) { g_create_err = h->err; return rc; }
    CK(hipGetLastError());
    const size_t nit = (size_t)L * g.Y * g.Mp;
    hipLaunchKernelGGL(k_take_counts, dim3((unsigned)std::min<size_t>(1024, (nit + TPB - 1) / TPB)), dim3(TPB), 0, h->stream, h->Hg, h->Cg, g);
    CK(hipGetLastError());
    if (h->h0s) {
      hipLaunchKernelGGL(k_level0_total, dim3(L), dim3(256), 0, h->stream, h->Cg, h->h0s, g);
      CK(hipGetLastError());
    }

  }
  CK(hipStreamSynchronize(h->stream));
  return VMR_OK;
}

static int create_dense(vmr_ctx* h, const hipDeviceProp_t& prop, const uint8_t* X, const uint8_t* R, int data_on_device) {
  Geo& g = h->g;
  const int L = g.L, N = g.N, M = g.M;
  const size_t T = (size_t)N * N, rows = (size_t)L * T, raw = rows * M;
  const size_t slack = (size_t)g.b * N + g.b;   // tile streams may read (never use) rows past the last tie
  CK(hipMalloc(&h->X, (rows + slack) * g.Mp));
  CK(hipMemsetAsync(h->X + rows * g.Mp, 0, slack * g.Mp, h->stream));
  CK(hipMalloc(&h->Rb, (rows + slack) * g.W * 8));
  CK(hipMemsetAsync(h->Rb + rows * g.W, 0, slack * g.W * 8, h->stream));
  {
    uint8_t* tmp = nullptr;
    const uint8_t* src = X;
    if (!data_on_device) { CK(hipMalloc(&tmp, raw)); CK(hipMemcpyAsync(tmp, X, raw, hipMemcpyHostToDevice, h->stream)); src = tmp; }
    hipLaunchKernelGGL(k_pack_x, dim3(4096), dim3(256), 0, h->stream, src, h->X, rows, M, g.Mp);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(h->stream));
    if (tmp) CK(hipFree(tmp));
    tmp = nullptr; src = R;
    if (R && !data_on_device) { CK(hipMalloc(&tmp, raw)); CK(hipMemcpyAsync(tmp, R, raw, hipMemcpyHostToDevice, h->stream)); src = tmp; }
    hipLaunchKernelGGL(k_pack_r, dim3(4096), dim3(256), 0, h->stream, src, h->Rb, rows, M, g.W);
    CK(hipGetLastError());
    hipLaunchKernelGGL(k_stats, dim3(2048), dim3(256), 0, h->stream, h->X, h->Rb, h->cov, h->rcls, h->sumx, h->xmax, h->npartial, rows, g.Mp, g.W, M);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(h->stream));
    if (tmp) CK(hipFree(tmp));
  }
  unsigned xm = 0;
  int rc = create_state(h, &xm);
  if (rc) return rc;
  // ---- data format: report lists unless X is dense enough that 1 B per (tie, reporter) is less to read ----
  const char* fmt = getenv("VMR_FORMAT");   // "dense", "sparse" or unset/"auto"
  const bool force_dense = fmt && !strcmp(fmt, "dense"), force_sparse = fmt && !strcmp(fmt, "sparse");
  // 13-bit reporter field; the sorted lists hold counts <= 2047 and (max count + 1) * Mp <= 2^20 table rows, the step layout counts <= 63
  const bool packed_ok = g.Mp <= 8192 && xm <= SL_XMAX && (size_t)(xm + 1) * g.Mp <= SL_YM_ROWS;
  // beyond KMAX categories there is no dense-tile kernel: the general kernels run on report lists, with wide entries where the
  // packed ones cannot hold the tensor (a uint8 tensor always fits: 256 levels x Mp rows < 2^32)
  const bool need_lists = g.K > KMAX;
  if (need_lists && force_dense) return fail(nullptr, VMR_EINVAL, "VMR_FORMAT=dense: the dense tile kernels hold at most 8 categories");
  // ... and so do tensors the dense tiles cannot hold at all (their per-reporter tables outgrow the LDS: M of several thousand)
  bool dense_fits = true;
  {
    Geo t = g;   // (what create_tail would settle on for the dense tiles)
    t.Y = g.Y; t.hc = t.Y < HC_MAX ? t.Y : HC_MAX;
    if (shmem_rho(t, true, false) > 80000) { t.two_pass = 1; while (t.hc > 0 && shmem_hist(t) > 160000) --t.hc; }
    dense_fits = std::max(shmem_rho(t, true, true), shmem_hist(t)) <= (size_t)160 * 1024;
  }
  const bool can_list = packed_ok || need_lists || force_sparse || !dense_fits;   // (VMR_FORMAT=sparse: wide entries for what the packed ones cannot hold)
  if (!force_dense && can_list) {
    g.wide = packed_ok ? 0 : 1;
    unsigned* rp = nullptr;   // [L][T+1] per-tie entry offsets: only needed to place the entries
    CK(hipMalloc(&rp, (size_t)L * (T + 1) * 4));
    unsigned long long* nnz_dev = nullptr;
    CK(hipMalloc(&nnz_dev, (size_t)L * 8));
    CK(hipMemsetAsync(nnz_dev, 0, (size_t)L * 8, h->stream));
    const unsigned cgrid = (unsigned)std::min<size_t>(8192, (T + 15) / 16);
    for (int l = 0; l < L; ++l)
      hipLaunchKernelGGL(k_sp_count, dim3(cgrid), dim3(256), 0, h->stream, h->X + (size_t)l * T * g.Mp,
                         rp + (size_t)l * (T + 1), nnz_dev + l, g);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(h->stream));
    std::vector<unsigned long long> nl(L);
    CK(hipMemcpy(nl.data(), nnz_dev, (size_t)L * 8, hipMemcpyDeviceToHost));
    CK(hipFree(nnz_dev));
    bool fits = true;
    h->nnz = 0;
    for (int l = 0; l < L; ++l) { h->nnz += nl[l]; fits = fits && nl[l] < 0xffffffffull; }
    const double sparse_bytes = 4.0 * (double)h->nnz + 4.0 * (double)rows, dense_bytes = (double)rows * g.Mp;
    if (need_lists && !fits) { (void)hipFree(rp); return fail(nullptr, VMR_EINVAL, "more than 2^32 reports in one layer"); }
    h->sparse = fits && (force_sparse || need_lists || !dense_fits || sparse_bytes <= 0.5 * dense_bytes);
    if (!h->sparse) g.wide = 0;
    g.gen = (h->sparse && (need_lists || g.wide)) ? 1 : 0;
    if (h->sparse) {
      CK(hipMalloc(&h->Qt, rows * 4));
      CK(hipMemsetAsync(h->Qt, 0, rows * 4, h->stream));
      auto fill_layer = [&](int l, const unsigned* rpl, unsigned* etmp, unsigned* etmp2) {
        if (g.mut)
          hipLaunchKernelGGL(k_sp_fill<true>, dim3(cgrid), dim3(256), 0, h->stream, h->X + (size_t)l * T * g.Mp,
                             h->Rb + (size_t)l * T * g.W, rpl, etmp, etmp2, h->Qt + (size_t)l * T, g);
        else
          hipLaunchKernelGGL(k_sp_fill<false>, dim3(cgrid), dim3(256), 0, h->stream, h->X + (size_t)l * T * g.Mp,
                             h->Rb + (size_t)l * T * g.W, rpl, etmp, etmp2, h->Qt + (size_t)l * T, g);
      };
      {
        const SlFill ff = fill_layer;
        rc = sl_place_entries(h, rp, nl, nullptr, nullptr, &ff);
        if (!rc) rc = sl_finish(h);
      }
      if (!rc) rc = mask_lists_from_words(h);
      if (!rc && !getenv("VMR_KEEP_X")) { CK(hipFree(h->X)); h->X = nullptr; }   // the lists replace the dense tensor
    }
    CK(hipFree(rp));
    if (rc) return rc;
  } else if (force_sparse) {
    return fail(nullptr, VMR_EINVAL, "VMR_FORMAT=sparse needs M <= 8192, counts <= 2047 and (largest count + 1) * M <= 2^20");
  }
  return create_tail(h, prop);
}

extern "C" int vmr_create(vmr_handle* out, int device, int L, int N, int M, int K, int mutuality, const uint8_t* X,
                          const uint8_t* R, int data_on_device, double eps) {
  if (!out) return fail(nullptr, VMR_EINVAL, "out is NULL");
  *out = nullptr;
  if (L < 1 || N < 1 || M < 1) return fail(nullptr, VMR_EINVAL, "L, N, M must be positive");
  if (K < 2 || K > KGEN_MAX) return fail(nullptr, VMR_EINVAL, "K must be in [2, 256]");
  if (!X) return fail(nullptr, VMR_EINVAL, "X is NULL");
  vmr_ctx* h = nullptr;
  hipDeviceProp_t prop;
  int rc = create_ctx(&h, &prop, device, L, N, M, K, mutuality, eps);
  if (!rc) rc = create_dense(h, prop, X, R, data_on_device);
  if (rc) { if (h) vmr_destroy(h); return rc; }
  *out = h;
  return VMR_OK;
}

// ------------------------------------------------------------------------------------------
// vmr_create_coo: coordinate lists (what the reference holds: sptensor subs / vals, model.py:136-171; the reader's output,
// _io.py:132-295) straight to report lists -- no dense [L,N,N,M] tensor anywhere.  A Karnataka village layer is 624
// reports against a 34 MB dense tensor; a layer of BASELINE config 5 is 5 GB of reports against 64 GB.
// Reports and mask entries are sorted by (layer, tie, reporter) (one 64-bit radix sort each); sorted order IS the tie-major
// order k_sp_round consumes, the mirror count and the mask bit of a report are binary searches in the sorted keys, and a
// sparse mask becomes the mask lists directly.
// ------------------------------------------------------------------------------------------
#define COO_KEY(tie, m) (((unsigned long long)(tie) << 13) | (unsigned long long)(m))

__global__ void k_coo_keys(const int32_t* __restrict__ sl, const int32_t* __restrict__ si, const int32_t* __restrict__ sj,
                           const int32_t* __restrict__ sm, long long n, int L, int N, int M, unsigned long long* __restrict__ keys,
                           int* bad) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    const int l = sl[e], i = si[e], j = sj[e], m = sm[e];
    if (l < 0 || l >= L || i < 0 || i >= N || j < 0 || j >= N || m < 0 || m >= M) { atomicOr(bad, 1); keys[e] = ~0ull; continue; }
    keys[e] = COO_KEY(((unsigned long long)l * N + i) * N + j, m);
  }
}
// index of `key` in the sorted keys, or -1
__device__ __forceinline__ long long coo_find(const unsigned long long* __restrict__ k, long long n, unsigned long long key) {
  long long a = 0, b = n;
  while (a < b) {
    const long long c = a + ((b - a) >> 1);
    if (k[c] < key) a = c + 1; else b = c;
  }
  return (a < n && k[a] == key) ? a : -1;
}
// per-tie counts of the sorted keys (cnt [L][T+1], layer-relative ties); duplicates flagged
__global__ void k_coo_count(const unsigned long long* __restrict__ k, long long n, size_t T, unsigned* __restrict__ cnt, int* bad) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    if (e > 0 && k[e] == k[e - 1]) atomicOr(bad, 2);
    const unsigned long long tie = k[e] >> 13, l = tie / T;
    atomicAdd(&cnt[l * (T + 1) + (tie - l * T)], 1u);
  }
}
// per tie: coverage, mask row class, statistics of the partial rows
__global__ void k_coo_class(const unsigned* __restrict__ cx, const unsigned* __restrict__ cr /*null: all ones*/, size_t T, int L, int M,
                            uint8_t* __restrict__ cov, uint8_t* __restrict__ rcls, unsigned long long* npartial, unsigned* maxrow) {
  unsigned long long lpart = 0, lempty = 0;
  unsigned mx = 0;
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < (size_t)L * T; q += (size_t)gridDim.x * blockDim.x) {
    const size_t l = q / T, t = q - l * T;
    const unsigned nX = cx[l * (T + 1) + t], nR = cr ? cr[l * (T + 1) + t] : (unsigned)M;
    cov[q] = (nX > 0 && nR > 0) ? 1 : 0;
    const unsigned c = nR == 0 ? 0u : (nR == (unsigned)M ? 1u : 2u);
    rcls[q] = (uint8_t)c;
    if (c == 2u) { ++lpart; mx = max(mx, nR); }
    if (c == 0u) ++lempty;
  }
  if (lpart) atomicAdd(npartial, lpart);
  if (lempty) atomicAdd(npartial + 1, lempty);
  if (mx) atomicMax(maxrow, mx);
}
// the reports' entries in tie-major (= sorted) order, the mirror sums Qt, sum and maximum of the counts
template <bool MUT>
__global__ void k_coo_entries(const unsigned long long* __restrict__ kx, const unsigned* __restrict__ vx, long long nx,
                              const unsigned long long* __restrict__ kr, long long nr /* < 0: all ones */, int N, int Mp,
                              unsigned* __restrict__ etmp, unsigned* __restrict__ etmp2, unsigned* __restrict__ Qt, unsigned long long* sumx, unsigned* xmax, int* bad) {
  unsigned long long s = 0;
  unsigned mx = 0;
  const unsigned long long T = (unsigned long long)N * N;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < nx; e += (long long)gridDim.x * blockDim.x) {
    const unsigned long long key = kx[e], tie = key >> 13, l = tie / T, t = tie - l * T, i = t / N, j = t - i * N;
    const unsigned m = (unsigned)(key & 0x1fffu), x = vx[e];
    if (x == 0u || x > 0x7fffffffu) { atomicOr(bad, 4); continue; }   // (int32 values: zero or negative)
    s += x; mx = max(mx, x);
    const unsigned long long tm = l * T + j * N + i;
    unsigned y = 0;
    if (MUT) {
      const long long f = coo_find(kx, nx, COO_KEY(tm, m));
      y = f >= 0 ? vx[f] : 0u;
      if (nr < 0 || coo_find(kr, nr, COO_KEY(tm, m)) >= 0) atomicAdd(&Qt[tm], x);   // R[mirror, m] X[this, m]
    }
    const unsigned inr = (nr < 0 || coo_find(kr, nr, key) >= 0) ? 1u : 0u;
    // two words per entry (sweep_sl.h, wide entries); k_coo_pack folds them into one where the packed format holds the tensor
    const unsigned long long row = (unsigned long long)y * (unsigned)Mp + m;
    if (row > 0xffffffffull) atomicOr(bad, 8);
    etmp[e] = (unsigned)row;
    etmp2[e] = (x << 1) | inr;
  }
  if (s) atomicAdd(sumx, s);
  if (mx) atomicMax(xmax, mx);
}
__global__ void k_coo_pack(unsigned* __restrict__ etmp, const unsigned* __restrict__ etmp2, long long n) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    const unsigned w = etmp2[e];
    etmp[e] = etmp[e] | ((w & 1u) << 20) | ((w >> 1) << 21);
  }
}
// first sorted key of every layer (starts[L] = n)
__global__ void k_coo_layer_starts(const unsigned long long* __restrict__ k, long long n, unsigned long long T, int L, unsigned long long* starts) {
  for (int l = threadIdx.x; l <= L; l += blockDim.x) {
    const unsigned long long k0 = ((unsigned long long)l * T) << 13;
    long long a = 0, b = n;
    while (a < b) { const long long c = a + ((b - a) >> 1); if (k[c] < k0) a = c + 1; else b = c; }
    starts[l] = (unsigned long long)a;
  }
}
// listed counts of the partial rows only
__global__ void k_coo_listed(const unsigned* __restrict__ cr, const uint8_t* __restrict__ rcls, size_t T, int L, unsigned* __restrict__ rq) {
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < (size_t)L * (T + 1); q += (size_t)gridDim.x * blockDim.x) {
    const size_t l = q / (T + 1), t = q - l * (T + 1);
    rq[q] = (t < T && rcls[l * T + t] == 2) ? cr[q] : 0u;
  }
}
// reporters of the partial rows into the mask lists (rq scanned per layer; rbase: layer offsets)
__global__ void k_coo_rm(const unsigned long long* __restrict__ kr, long long nr, size_t T, const uint8_t* __restrict__ rcls,
                         const unsigned* __restrict__ rq, const unsigned long long* __restrict__ rbase, unsigned short* __restrict__ Rm) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < nr; e += (long long)gridDim.x * blockDim.x) {
    const unsigned long long tie = kr[e] >> 13, l = tie / T, t = tie - l * T;
    if (rcls[tie] != 2) continue;
    long long a = 0, b = e;   // first entry of this tie: lower bound of tie << 13 in [0, e]
    const unsigned long long k0 = tie << 13;
    while (a < b) { const long long c = a + ((b - a) >> 1); if (kr[c] < k0) a = c + 1; else b = c; }
    Rm[rbase[l] + rq[l * (T + 1) + t] + (unsigned long long)(e - a)] = (unsigned short)(kr[e] & 0x1fffu);
  }
}
// a mask whose partial rows are long: bit-packed words as in the dense path
__global__ void k_coo_rbits(const unsigned long long* __restrict__ kr, long long nr, int W, unsigned long long* __restrict__ Rb) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < nr; e += (long long)gridDim.x * blockDim.x) {
    const unsigned long long tie = kr[e] >> 13;
    const unsigned m = (unsigned)(kr[e] & 0x1fffu);
    atomicOr(&Rb[tie * W + (m >> 6)], 1ull << (m & 63));
  }
}

// sorted keys (and values) of a coordinate list on the device; *k_out, *v_out are hipMalloc'ed
static int coo_sorted(vmr_ctx* h, long long n, const int32_t* sl, const int32_t* si, const int32_t* sj, const int32_t* sm,
                      const int32_t* sv, int on_device, unsigned long long** k_out, unsigned** v_out, int* bad_dev) {
  const Geo& g = h->g;
  *k_out = nullptr;
  if (v_out) *v_out = nullptr;
  const size_t nn = (size_t)std::max<long long>(n, 1);
  const int32_t* src[5] = {sl, si, sj, sm, sv};
  int32_t* tmp[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  const int ncol = sv ? 5 : 4;
  if (!on_device) {
    for (int c = 0; c < ncol; ++c) {
      CK(hipMalloc(&tmp[c], nn * 4));
      if (n > 0) CK(hipMemcpyAsync(tmp[c], src[c], (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
      src[c] = tmp[c];
    }
  }
  unsigned long long *k0 = nullptr, *k1 = nullptr;
  unsigned* v1 = nullptr;
  CK(hipMalloc(&k0, nn * 8));
  CK(hipMalloc(&k1, nn * 8));
  if (sv) CK(hipMalloc(&v1, nn * 4));
  const unsigned grid = (unsigned)std::min<long long>(4096, (n + 255) / 256 + 1);
  hipLaunchKernelGGL(k_coo_keys, dim3(grid), dim3(256), 0, h->stream, src[0], src[1], src[2], src[3], n, g.L, g.N, g.M, k0, bad_dev);
  CK(hipGetLastError());
  int bits = 13;
  { unsigned long long ties = (unsigned long long)g.L * g.N * g.N; while ((1ull << (bits - 13)) < ties && bits < 64) ++bits; }
  size_t tb = 0;
  void* td = nullptr;
  if (n > 0) {
    if (sv) CK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k0, k1, reinterpret_cast<const unsigned*>(src[4]), v1, (int)n, 0, bits, h->stream));
    else CK(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, k0, k1, (int)n, 0, bits, h->stream));
    CK(hipMalloc(&td, tb ? tb : 8));
    if (sv) CK(hipcub::DeviceRadixSort::SortPairs(td, tb, k0, k1, reinterpret_cast<const unsigned*>(src[4]), v1, (int)n, 0, bits, h->stream));
    else CK(hipcub::DeviceRadixSort::SortKeys(td, tb, k0, k1, (int)n, 0, bits, h->stream));
  }
  CK(hipStreamSynchronize(h->stream));
  if (td) CK(hipFree(td));
  CK(hipFree(k0));
  for (int c = 0; c < ncol; ++c) if (tmp[c]) CK(hipFree(tmp[c]));
  *k_out = k1;
  if (v_out) *v_out = v1;
  return VMR_OK;
}

static int create_coo(vmr_ctx* h, const hipDeviceProp_t& prop, long long nx, const int32_t* xl, const int32_t* xi, const int32_t* xj,
                      const int32_t* xm, const int32_t* xv, long long nr, const int32_t* rl, const int32_t* ri, const int32_t* rj,
                      const int32_t* rm, int on_device) {
  Geo& g = h->g;
  const int L = g.L, N = g.N, M = g.M, K = g.K;
  const size_t T = (size_t)N * N, rows = (size_t)L * T, n1 = (size_t)L * (T + 1);
  if (g.Mp > 8192) return fail(nullptr, VMR_EINVAL, "report lists hold 13-bit reporter indices: M <= 8192 (use vmr_create for wider tensors)");
  if (nx >= 0x7fffffffll || nr >= 0x7fffffffll) return fail(nullptr, VMR_EINVAL, "more than 2^31 coordinates in one call");
  int* bad_dev = nullptr;
  CK(hipMalloc(&bad_dev, 4));
  CK(hipMemsetAsync(bad_dev, 0, 4, h->stream));
  unsigned long long *kx = nullptr, *kr = nullptr;
  unsigned* vx = nullptr;
  int rc = coo_sorted(h, nx, xl, xi, xj, xm, xv, on_device, &kx, &vx, bad_dev);
  if (!rc && nr >= 0) rc = coo_sorted(h, nr, rl, ri, rj, rm, nullptr, on_device, &kr, nullptr, bad_dev);
  unsigned *cx = nullptr, *cr = nullptr, *etmp = nullptr, *etmp2 = nullptr, *maxrow_dev = nullptr;
  auto cleanup = [&]() { void* p[] = {bad_dev, kx, kr, vx, cx, cr, etmp, etmp2, maxrow_dev}; for (void* q : p) if (q) (void)hipFree(q); };
  if (rc) { cleanup(); return rc; }
  {   // subscripts outside the tensor must not reach the kernels below (their keys index the per-tie arrays)
    int bad0 = 0;
    hipError_t e0 = hipMemcpy(&bad0, bad_dev, 4, hipMemcpyDeviceToHost);   // (coo_sorted synchronised the stream)
    if (e0 != hipSuccess) { g_create_err = hipGetErrorString(e0); cleanup(); return VMR_EHIP; }
    if (bad0 & 1) { cleanup(); return fail(nullptr, VMR_EINVAL, "a subscript lies outside (L, N, N, M)"); }
  }
#define CKC(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { g_create_err = std::string(#call) + ": " + hipGetErrorString(e_); (void)hipGetLastError(); cleanup(); return VMR_EHIP; } } while (0)
  CKC(hipMalloc(&cx, n1 * 4));
  CKC(hipMemsetAsync(cx, 0, n1 * 4, h->stream));
  const unsigned gx = (unsigned)std::min<long long>(8192, (nx + 255) / 256 + 1), gr = (unsigned)std::min<long long>(8192, (std::max<long long>(nr, 0) + 255) / 256 + 1);
  hipLaunchKernelGGL(k_coo_count, dim3(gx), dim3(256), 0, h->stream, kx, nx, T, cx, bad_dev);
  if (nr >= 0) {
    CKC(hipMalloc(&cr, n1 * 4));
    CKC(hipMemsetAsync(cr, 0, n1 * 4, h->stream));
    hipLaunchKernelGGL(k_coo_count, dim3(gr), dim3(256), 0, h->stream, kr, nr, T, cr, bad_dev);
  }
  CKC(hipMalloc(&maxrow_dev, 4));
  CKC(hipMemsetAsync(maxrow_dev, 0, 4, h->stream));
  hipLaunchKernelGGL(k_coo_class, dim3((unsigned)std::min<size_t>(4096, (rows + 255) / 256)), dim3(256), 0, h->stream, cx, cr, T, L, M,
                     h->cov, h->rcls, h->npartial, maxrow_dev);
  CKC(hipMalloc(&etmp, ((size_t)nx + 64) * 4));
  CKC(hipMalloc(&etmp2, ((size_t)nx + 64) * 4));
  CKC(hipMalloc(&h->Qt, rows * 4));
  CKC(hipMemsetAsync(h->Qt, 0, rows * 4, h->stream));
  if (g.mut) hipLaunchKernelGGL(k_coo_entries<true>, dim3(gx), dim3(256), 0, h->stream, kx, vx, nx, kr, nr, N, g.Mp, etmp, etmp2, h->Qt, h->sumx, h->xmax, bad_dev);
  else hipLaunchKernelGGL(k_coo_entries<false>, dim3(gx), dim3(256), 0, h->stream, kx, vx, nx, kr, nr, N, g.Mp, etmp, etmp2, h->Qt, h->sumx, h->xmax, bad_dev);
  CKC(hipGetLastError());
  CKC(hipStreamSynchronize(h->stream));
  int bad = 0;
  unsigned maxrow = 0;
  CKC(hipMemcpy(&bad, bad_dev, 4, hipMemcpyDeviceToHost));
  CKC(hipMemcpy(&maxrow, maxrow_dev, 4, hipMemcpyDeviceToHost));
  h->rm_maxrow = maxrow;
  if (bad) {
    cleanup();
    return fail(nullptr, VMR_EINVAL, (bad & 1) ? "a subscript lies outside (L, N, N, M)"
                                   : (bad & 2) ? "duplicate (l, i, j, m) subscripts"
                                   : (bad & 4) ? "counts must be positive (and below 2^31)"
                                               : "(largest count + 1) * M exceeds 2^32 table rows");
  }
  unsigned xmv = 0;
  if ((rc = create_state(h, &xmv))) { cleanup(); return rc; }
  h->sparse = 1;
  h->nnz = (unsigned long long)nx;
  // reports per layer: where each layer starts in the sorted keys
  std::vector<unsigned long long> nl(L, 0);
  {
    unsigned long long* starts = nullptr;
    CKC(hipMalloc(&starts, (size_t)(L + 1) * 8));
    hipLaunchKernelGGL(k_coo_layer_starts, dim3(1), dim3(64), 0, h->stream, kx, nx, (unsigned long long)T, L, starts);
    std::vector<unsigned long long> st(L + 1);
    hipError_t e1 = hipMemcpyAsync(st.data(), starts, (size_t)(L + 1) * 8, hipMemcpyDeviceToHost, h->stream);
    if (e1 == hipSuccess) e1 = hipStreamSynchronize(h->stream);
    (void)hipFree(starts);
    CKC(e1);
    for (int l = 0; l < L; ++l) nl[l] = st[l + 1] - st[l];
  }
  // packed entries (11-bit counts, 2^20 table rows) where they hold the tensor, two words per entry otherwise; the general
  // kernels take over beyond KMAX categories or with wide entries (sweep_gen.h)
  g.wide = (xmv <= SL_XMAX && (size_t)(xmv + 1) * g.Mp <= SL_YM_ROWS) ? 0 : 1;
  g.gen = (K > KMAX || g.wide) ? 1 : 0;
  if ((unsigned long long)xmv * (unsigned long long)M > 0xffffffffull) {
    cleanup();
    return fail(nullptr, VMR_EINVAL, "largest count * M exceeds 2^32 (the per-tie mirror sums are 32-bit)");
  }
  if (!g.wide) {
    hipLaunchKernelGGL(k_coo_pack, dim3(gx), dim3(256), 0, h->stream, etmp, etmp2, (long long)nx);
    CKC(hipGetLastError());
  }
  rc = sl_place_entries(h, cx, nl, etmp, g.wide ? etmp2 : nullptr, nullptr);
  if (!rc) rc = sl_finish(h);
  if (rc) { cleanup(); return rc; }
  // the mask: all ones needs nothing; partial rows become mask lists when they are short, bit-packed words otherwise
  if (nr >= 0 && h->n_partial > 0) {
    const double list_bytes = 2.0 * (double)nr + 4.0 * (double)rows, word_bytes = (double)h->n_partial * g.W * 8.0;
    const bool lists = !getenv("VMR_NO_RLISTS") && maxrow <= 64 && (list_bytes * 4.0 <= word_bytes || g.N <= 1024) && (size_t)g.Mp * K * 8 <= 160 * 1024;
    if (lists) {
      unsigned* bs = nullptr;
      CKC(hipMalloc(&h->rq, n1 * 4));
      CKC(hipMalloc(&bs, ((T + 1) + 2047) / 2048 * 4));
      hipLaunchKernelGGL(k_coo_listed, dim3((unsigned)std::min<size_t>(4096, (n1 + 255) / 256)), dim3(256), 0, h->stream, cr, h->rcls, T, L, h->rq);
      std::vector<unsigned long long> rb_(L);
      h->n_rm = 0;
      for (int l = 0; l < L; ++l) {
        if ((rc = scan_u32(h, h->rq + (size_t)l * (T + 1), bs, T + 1))) { (void)hipFree(bs); cleanup(); return rc; }
        unsigned tot = 0;
        CKC(hipStreamSynchronize(h->stream));
        CKC(hipMemcpy(&tot, h->rq + (size_t)l * (T + 1) + T, 4, hipMemcpyDeviceToHost));
        rb_[l] = h->n_rm; h->n_rm += tot;
      }
      CKC(hipFree(bs));
      CKC(hipMalloc(&h->rbase, (size_t)L * 8));
      CKC(hipMemcpy(h->rbase, rb_.data(), (size_t)L * 8, hipMemcpyHostToDevice));
      CKC(hipMalloc(&h->Rm, ((size_t)h->n_rm + 64) * 2));
      hipLaunchKernelGGL(k_coo_rm, dim3(gr), dim3(256), 0, h->stream, kr, nr, T, h->rcls, h->rq, h->rbase, h->Rm);
      CKC(hipGetLastError());
    } else {
      CKC(hipMalloc(&h->Rb, rows * g.W * 8));
      CKC(hipMemsetAsync(h->Rb, 0, rows * g.W * 8, h->stream));
      hipLaunchKernelGGL(k_coo_rbits, dim3(gr), dim3(256), 0, h->stream, kr, nr, g.W, reinterpret_cast<unsigned long long*>(h->Rb));
      CKC(hipGetLastError());
    }
  }
  CKC(hipStreamSynchronize(h->stream));
#undef CKC
  cleanup();
  return create_tail(h, prop);
}

extern "C" int vmr_create_coo(vmr_handle* out, int device, int L, int N, int M, int K, int mutuality, int64_t nx, const int32_t* xl,
                              const int32_t* xi, const int32_t* xj, const int32_t* xm, const int32_t* xv, int64_t nr, const int32_t* rl,
                              const int32_t* ri, const int32_t* rj, const int32_t* rm, int data_on_device, double eps) {
  if (!out) return fail(nullptr, VMR_EINVAL, "out is NULL");
  *out = nullptr;
  if (nx < 0 || (nx > 0 && (!xl || !xi || !xj || !xm || !xv))) return fail(nullptr, VMR_EINVAL, "X coordinate arrays missing");
  if (nr > 0 && (!rl || !ri || !rj || !rm)) return fail(nullptr, VMR_EINVAL, "R coordinate arrays missing");
  vmr_ctx* h = nullptr;
  hipDeviceProp_t prop;
  int rc = create_ctx(&h, &prop, device, L, N, M, K, mutuality, eps);
  if (!rc) rc = create_coo(h, prop, nx, xl, xi, xj, xm, xv, nr, rl, ri, rj, rm, data_on_device);
  if (rc) { if (h) vmr_destroy(h); return rc; }
  *out = h;
  return VMR_OK;
}

extern "C" {

void vmr_destroy(vmr_handle h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (auto& e : h->evs) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  for (auto& e : h->graphs) (void)hipGraphExecDestroy(e.ex);
  void* ptrs[] = {h->x0p, h->h0s, h->far_pos, h->far_ent, h->far_base, h->EX, h->gen_s1, h->rm2, h->det_buf, h->fr_slots, h->nu_acc, h->fin_g, h->perm, h->sy, h->cls_p, h->Qt_p, h->nat, h->rho_snap, h->par_snap, h->rq, h->Rm, h->rbase, h->E, h->rs, h->Cg, h->Qt, h->ebase, h->rcls, h->X, h->Rb, h->cov, h->sumx, h->rho, h->logpr, h->par, h->slotA, h->slotR, h->elbo_dev, h->lutg, h->Hg, h->xmax, h->slotF, h->npartial};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (h->stream2) { (void)hipStreamSynchronize(h->stream2); (void)hipStreamDestroy(h->stream2); }
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int vmr_data_stats(vmr_handle h, double* sum_x, uint8_t* coverage) {
  if (!h) return VMR_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  // every copy is ordered on the handle's stream: it is a non-blocking stream, which plain hipMemcpy (null stream)
  // does not wait for
  unsigned long long v = 0;
  if (sum_x) HIPCHK(h, hipMemcpyAsync(&v, h->sumx, 8, hipMemcpyDeviceToHost, h->stream));
  if (coverage) HIPCHK(h, hipMemcpyAsync(coverage, h->cov, (size_t)h->g.L * h->g.N * h->g.N, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (sum_x) *sum_x = (double)v;
  return VMR_OK;
}

// host [L,M] -> device [L,Mp]
static int upload_lm(vmr_ctx* h, size_t off, const double* src, double pad) {
  const Geo& g = h->g;
  std::vector<double> buf((size_t)g.L * g.Mp, pad);
  for (int l = 0; l < g.L; ++l) memcpy(&buf[(size_t)l * g.Mp], src + (size_t)l * g.M, (size_t)g.M * 8);
  HIPCHK(h, hipMemcpyAsync(h->par + off, buf.data(), buf.size() * 8, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));   // buf is a pageable temporary: it must outlive the copy
  return VMR_OK;
}
// small host -> device copy ordered on the handle's stream (the stream is non-blocking: a null-stream hipMemcpy
// would not wait for kernels queued on it)
static int h2d(vmr_ctx* h, void* dst, const void* src, size_t n) {
  HIPCHK(h, hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, h->stream));
  return VMR_OK;
}
static int d2h(vmr_ctx* h, void* dst, const void* src, size_t n) {
  HIPCHK(h, hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, h->stream));
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
  if ((rc = h2d(h, h->par + o.a_la, alpha_lambda, (size_t)g.L * g.K * 8))) return rc;
  if ((rc = h2d(h, h->par + o.b_la, beta_lambda, (size_t)g.L * g.K * 8))) return rc;
  double ab[2] = {alpha_eta, beta_eta};
  if ((rc = h2d(h, h->par + o.sc + SC_A_ETA, ab, 16))) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->have_priors = true;
  return VMR_OK;
}

static void drop_graphs(vmr_ctx* h);
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
  if ((rc = h2d(h, h->par + o.p_shp, phi_shp, (size_t)g.L * g.K * 8))) return rc;
  if ((rc = h2d(h, h->par + o.p_rte, phi_rte, (size_t)g.L * g.K * 8))) return rc;
  double nu[2] = {nu_shp, nu_rte};
  if ((rc = h2d(h, h->par + o.sc + SC_NU_SHP, nu, 16))) return rc;   // (the stream is synchronised below, before nu[] dies)
  const size_t n = (size_t)g.L * g.N * g.N * g.K;
  const double* src = pr_rho;
  if (h->perm) {   // sorted lists: rho and the log prior are stored by position
    if (!pr_rho_on_device) {
      if (!h->nat) HIPCHK(h, hipMalloc(&h->nat, n * 8));
      HIPCHK(h, hipMemcpyAsync(h->nat, pr_rho, n * 8, hipMemcpyHostToDevice, h->stream));
      src = h->nat;
    }
    const size_t T_ = (size_t)g.N * g.N;
    int* flag = reinterpret_cast<int*>(h->elbo_dev + 7);   // (the spare double of the ELBO scratch)
    HIPCHK(h, hipMemsetAsync(flag, 0, 4, h->stream));
    hipLaunchKernelGGL(k_init_rho_pos, dim3(4096), dim3(256), 0, h->stream, src, h->rho, h->logpr, h->perm, T_, (T_ + 63) / 64, g.L, g.K, g.eps,
                       h->rs, flag);
    int nf = 1;
    HIPCHK(h, hipMemcpyAsync(&nf, flag, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->lp0 = nf == 0 && !getenv("VMR_NO_LP0");
  } else {
    if (!pr_rho_on_device) {
      // stage through logpr (overwritten by k_init_rho element-wise after being read)
      HIPCHK(h, hipMemcpyAsync(h->logpr, pr_rho, n * 8, hipMemcpyHostToDevice, h->stream));
      src = h->logpr;
    }
    hipLaunchKernelGGL(k_init_rho, dim3(4096), dim3(256), 0, h->stream, src, h->rho, h->logpr, n, g.eps);
  }
  HIPCHK(h, hipGetLastError());
  hipLaunchKernelGGL(k_derive_all, dim3(8), dim3(256), 0, h->stream, h->par, g);
  HIPCHK(h, hipGetLastError());
  hipLaunchKernelGGL(k_build_lut, dim3(g.L), dim3(256), 0, h->stream, h->par, h->lutg, g);
  HIPCHK(h, hipGetLastError());
  // (queued before the wait: the lockstep loop runs this handle's kernels on another handle's stream, so nothing of this call
  // may still be pending on the handle's own stream when it returns)
  HIPCHK(h, hipMemsetAsync(h->slotF, 0, (size_t)g.L * NSLOT * g.K * 8, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  drop_graphs(h);   // (a realisation's arguments may differ: SlArgs::lp0)
  h->have_state = true;
  h->restored = false;
  h->rho_stale = false;
  h->h_valid = false;
  h->f_valid = false;
  h->a_valid = false;
  h->h_zero = false; h->a_zero = false;   // (whatever an earlier, possibly failed, sweep left behind)
  if (h->h0s) HIPCHK(h, hipMemsetAsync(h->h0s, 0, (size_t)h->g.L * NSLOT * h->g.K * 8, h->stream));   // (the slots, not the constants behind them)
  if (h->gen_s1) HIPCHK(h, hipMemsetAsync(h->gen_s1, 0, (size_t)h->g.L * (h->g.Mp + h->g.K) * 8, h->stream));
  return VMR_OK;
}

static int read_elbo(vmr_ctx* h, double* out) {
  HIPCHK(h, hipMemcpyAsync(out, h->elbo_dev, 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (isnan(*out)) return fail(h, VMR_ENAN, "ELBO is NaN!!!!");
  return VMR_OK;
}

static int sweep(vmr_ctx* h, int mode, bool store = true) {
  int rc;
  if ((rc = launch_gamma(h, true))) return rc;   // gamma and phi are finished by one kernel
  return launch_rho(h, mode, true, false, store);
}

// After a committed sweep the handle is in a fixed point of its bookkeeping: the next sweep is the same two or three launches with
// the same arguments.  Such sweeps are captured once (per count, kind of last sweep and bookkeeping state) and replayed as one
// graph launch.
static bool sweep_steady(const vmr_ctx* h) {
  return h->have_state && h->h_valid && !h->h_zero && h->f_valid == (h->g.fuse_full != 0) && (!h->g.ml || h->a_valid) && !h->prof;
}
static unsigned state_sig(const vmr_ctx* h) {
  return (h->h_valid ? 1u : 0u) | (h->h_reduced ? 2u : 0u) | (h->h_zero ? 4u : 0u) | (h->f_valid ? 8u : 0u) | (h->a_valid ? 16u : 0u) | (h->a_zero ? 32u : 0u) |
         (h->rho_stale ? 64u : 0u);
}
static void state_set(vmr_ctx* h, unsigned s) {
  h->h_valid = s & 1u; h->h_reduced = s & 2u; h->h_zero = s & 4u; h->f_valid = s & 8u; h->a_valid = s & 16u; h->a_zero = s & 32u; h->rho_stale = s & 64u;
}
static void drop_graphs(vmr_ctx* h) {
  for (auto& e : h->graphs) (void)hipGraphExecDestroy(e.ex);
  h->graphs.clear();
}
// n plain sweeps from the handle's present state, the last one with (lazy_last) or without leaving rho unwritten
static int graph_for(vmr_ctx* h, int n, bool lazy_last, const vmr_ctx::GraphEntry** out) {
  const unsigned sig0 = state_sig(h);
  for (auto& e : h->graphs) if (e.n == n && e.lazy_last == lazy_last && e.sig0 == sig0) { *out = &e; return VMR_OK; }
  hipGraph_t gr = nullptr;
  HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
  int rc = VMR_OK;
  for (int i = 0; i < n && rc == VMR_OK; ++i) rc = sweep(h, 0, i == n - 1 ? !lazy_last : false);   // (queues nothing: the launches are recorded)
  const unsigned sig1 = state_sig(h);
  state_set(h, sig0);   // nothing ran yet
  hipError_t e = hipStreamEndCapture(h->stream, &gr);
  if (rc != VMR_OK || e != hipSuccess || !gr) {
    if (gr) (void)hipGraphDestroy(gr);
    (void)hipGetLastError();
    h->use_graphs = false;   // eager from now on
    if (rc == VMR_OK) { h->err = std::string("hipStreamEndCapture: ") + hipGetErrorString(e); rc = VMR_EHIP; }
    return rc;
  }
  hipGraphExec_t ex = nullptr;
  e = hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0);
  (void)hipGraphDestroy(gr);
  if (e != hipSuccess) { (void)hipGetLastError(); h->use_graphs = false; h->err = std::string("hipGraphInstantiate: ") + hipGetErrorString(e); return VMR_EHIP; }
  h->graphs.push_back({n, lazy_last, sig0, sig1, ex});
  *out = &h->graphs.back();
  return VMR_OK;
}

// n sweeps.  rho is written by the LAST sweep of the call (and by ELBO sweeps) only: the rho of a sweep inside the call is overwritten
// by the next one before anyone can read it.  store_last = false: the caller runs an ELBO sweep next (the fit loops), so not even that.
static int step_n(vmr_ctx* h, int n_iters, double* elbo_out, bool store_last);
int vmr_step(vmr_handle h, int n_iters, double* elbo_out) {
  if (!h) return VMR_EINVAL;
  if (!h->have_state) return fail(h, VMR_ESTATE, "vmr_set_state must be called before vmr_step");
  if (h->restored) return fail(h, VMR_ESTATE, "vmr_step after vmr_restore: the restored state is read-only until the next vmr_set_state");
  if (n_iters < 0) return fail(h, VMR_EINVAL, "n_iters < 0");
  // (VMR_DEBUG_LAZY_RHO=1, diagnostics / tests: leave even the call's last rho unwritten, so that every reader goes through ensure_rho)
  return step_n(h, n_iters, elbo_out, getenv("VMR_DEBUG_LAZY_RHO") == nullptr);
}
static int step_n(vmr_ctx* h, int n_iters, double* elbo_out, bool store_last) {
  HIPCHK(h, hipSetDevice(h->device));
  int rc;
  int it = 0;
  const int plain = elbo_out ? n_iters - 1 : n_iters;   // sweeps without an ELBO
  if (h->use_graphs && plain >= 2) {
    if (!sweep_steady(h)) { if ((rc = sweep(h, 0, false))) return rc; it = 1; }   // (more sweeps follow: its rho is overwritten unread)
    while (h->use_graphs && sweep_steady(h) && plain - it >= 2) {
      const int n = (plain - it >= 9) ? 9 : (plain - it);   // the fit loop asks for 9 between two ELBO checks (model.py:1036)
      const bool ends = it + n == n_iters;   // the call's last sweep is in this graph (no ELBO sweep behind it)
      const vmr_ctx::GraphEntry* ge = nullptr;
      if (graph_for(h, n, !(ends && store_last), &ge) != VMR_OK) break;   // capture failed: the eager loop below takes over
      HIPCHK(h, hipGraphLaunch(ge->ex, h->stream));
      state_set(h, ge->sig1);
      it += n;
    }
  }
  for (; it < n_iters; ++it) {
    const bool last = it == n_iters - 1;
    if ((rc = sweep(h, (last && elbo_out) ? 1 : 0, last && store_last))) return rc;
  }
  if (elbo_out) {
    if (n_iters == 0) return vmr_elbo(h, elbo_out);
    return read_elbo(h, elbo_out);
  }
  return VMR_OK;
}

// the convergence loop of `fit` for one realisation (model.py:405-426, 1021-1056), without a host-language round trip
// per iteration: one call per realisation, so several fits driven from host threads do not queue for an interpreter lock.
// (st: where the loop stands -- a fresh realisation, or one whose first iterations ran elsewhere, see vmr_fit_loop_batch)
struct LoopState { int it = 1, coincide = 0, reached = 0, rows = 0; double elbo = -1e10; /* INF of the reference (model.py:24) */ };
static int fit_loop_core(vmr_ctx* h, LoopState& st, int max_iter, double tol, int decision, int cap, int* row_iter, double* row_elbo,
                         double* row_runtime, int* row_reached) {
  int rc;
  while (!st.reached && st.it <= max_iter) {
    // the ELBO is evaluated at iteration 1, every 10th and the last (model.py:1036-1039); the sweeps in between are queued at once
    const int it = st.it;
    const int nxt = (it == 1 || it % 10 == 0 || it == max_iter) ? it : std::min(max_iter, (it / 10 + 1) * 10);
    if (nxt > it) {
      if ((rc = step_n(h, nxt - it, nullptr, false))) return rc;   // (an ELBO sweep follows: it writes rho)
      st.it = nxt;
      HIPCHK(h, hipStreamSynchronize(h->stream));   // so that the runtime below is this iteration's sweep, as in the reference
    }
    const auto t0 = std::chrono::steady_clock::now();
    const double old = st.elbo;
    if ((rc = step_n(h, 1, &st.elbo, true))) return rc;
    const double runtime = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    st.coincide = (fabs(st.elbo - old) < tol) ? st.coincide + 1 : 0;
    if (st.coincide > decision) st.reached = 1;
    ++st.it;
    if ((st.it - 1) % 10 == 0 && st.rows < cap) {
      row_iter[st.rows] = st.it - 1; row_elbo[st.rows] = st.elbo; row_runtime[st.rows] = runtime; row_reached[st.rows] = st.reached;
      ++st.rows;
    }
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return VMR_OK;
}
int vmr_fit_loop(vmr_handle h, int max_iter, double tol, int decision, int cap, int* n_rows, int* row_iter, double* row_elbo,
                 double* row_runtime, int* row_reached, double* elbo_out, int* iters_out, int* converged_out) {
  if (!h || !n_rows || !elbo_out || !iters_out || !converged_out) return VMR_EINVAL;
  if (!h->have_state) return fail(h, VMR_ESTATE, "vmr_set_state must be called before vmr_fit_loop");
  if (h->restored) return fail(h, VMR_ESTATE, "vmr_fit_loop after vmr_restore: the restored state is read-only until the next vmr_set_state");
  if (cap > 0 && (!row_iter || !row_elbo || !row_runtime || !row_reached)) return fail(h, VMR_EINVAL, "trace arrays missing");
  LoopState st;
  int rc = fit_loop_core(h, st, max_iter, tol, decision, cap, row_iter, row_elbo, row_runtime, row_reached);
  if (rc) return rc;
  *n_rows = st.rows; *elbo_out = st.elbo; *iters_out = st.it - 1; *converged_out = st.reached;
  return VMR_OK;
}

// ---- many small fits in lockstep -----------------------------------------------------------------------------------
// A sweep of a Karnataka-sized layer (N = 200-800, L = 1) is two dependent launches of 20-40 us each that leave the GPU nearly
// empty, and launches of different streams of one process hardly overlap (8 host threads: 1.4x one thread's sweeps per
// second).  Here the realisations of n handles advance together: every sweep is ONE launch of the finalize kernel and ONE of
// the pass for all of them (k_fin_gamma_b, k_sweep_sl_b; on ELBO sweeps k_fin_rho_b and one copy of n ELBOs), each handle's
// workgroups reading its own arguments from a table in device memory.  A handle that converges leaves the tables.
// Handles of another kind than the first (K, mutuality, mask kind, report lists in one pass) run their loops one by one.
static bool batch_steady(const vmr_ctx* h) {
  const Geo& g = h->g;
  return h->have_state && !h->restored && h->sparse && !g.two_pass && !g.farl && h->h_valid && !h->h_reduced && !h->h_zero &&
         h->f_valid == (g.fuse_full != 0) && (!g.ml || h->a_valid) && !h->prof;
}
static bool batch_kind(const vmr_ctx* h, const vmr_ctx* h0) {
  // (a sweep of such a handle is k_fin_gamma + the pass, + k_fin_rho with an ELBO: the pass sums rho over the mask itself)
  return h->sparse && !h->g.gen && !h->g.two_pass && !h->g.farl && !h->g.det && !h->prof && h->g.fuse_full && (h->n_partial == 0 || h->g.ml) && h->device == h0->device &&
         h->g.K == h0->g.K && h->g.mut == h0->g.mut && (h->all_full != 0) == (h0->all_full != 0);
}
namespace {
struct BatchTables {   // device copies of the unit tables of one lockstep loop
  SlUnit* su[3] = {nullptr, nullptr, nullptr};   // mode 0 / 1; [2]: the statistics pass of a realisation's first sweep
  int* smap[3] = {nullptr, nullptr, nullptr};
  FinUnit* fu = nullptr;
  int *gmap = nullptr, *rmap = nullptr;
  double* be = nullptr;                 // [n][8]: every unit's elbo_dev
  int nb[3] = {0, 0, 0}, ngb = 0, nrb = 0, tpb[3] = {0, 0, 0}, fg = FG_G;
  size_t smem[3] = {0, 0, 0}, fsm = 0;
  ~BatchTables() { for (void* p : {(void*)su[0], (void*)su[1], (void*)su[2], (void*)smap[0], (void*)smap[1], (void*)smap[2], (void*)fu, (void*)gmap, (void*)rmap, (void*)be}) if (p) (void)hipFree(p); }
};
}  // namespace
// (re)builds the tables for the units listed in `act`
static int batch_tables(vmr_ctx* const* hs, const std::vector<int>& act, int n_all, BatchTables& bt, hipStream_t st) {
  vmr_ctx* h0 = hs[act[0]];
  const int K = h0->g.K, allfull = h0->all_full != 0;
  long long steps = 0;
  for (int u : act) steps += (((long long)hs[u]->g.N * hs[u]->g.N + 63) / 64) * hs[u]->g.L;
  std::vector<FinUnit> fu(act.size());
  std::vector<int> gmap, rmap;
  const long long per_cap = getenv("VMR_BATCH_PER") ? atoll(getenv("VMR_BATCH_PER")) : 64;   // (experiments)
  for (int m = 0; m < 3; ++m) {
    const int tpb = sl_tpb_max_b(K, m == 1, allfull), nw = tpb / 64;
    // steps per wave: about one workgroup per CU in all (the tables in LDS allow few more, and a second round of workgroups
    // costs a workgroup's fixed part -- its tables, its share of nu: some ten steps' worth -- again), at least 4 steps each, at
    // most 64 (192 fits of N = 200..800, lockstep loops: cap 8 1.05 s, 16 0.88, 32 0.78, 64 0.72, none 0.96 -- a small unit
    // then is one workgroup that walks all its steps)
    long long per = std::max<long long>(4, std::min<long long>(per_cap, (steps + (long long)nw * h0->ncu - 1) / ((long long)nw * h0->ncu)));
    {
      // ... but never a FEW workgroups more than are resident at once: every workgroup of the launch gets the largest unit's LDS
      // (a village of N = 900: 130 KB, one workgroup per CU), and with 293 workgroups on 256 CUs the launch takes two rounds for
      // the sake of 37.  Then rather more steps per wave, until one round holds them all.
      size_t smem_max = 0;
      for (int u : act) smem_max = std::max(smem_max, sl_shape(hs[u], m != 2, m == 1, true).smem);
      const int wpc = std::max<int>(1, std::min<int>((int)(SP_LDS_MAX / std::max<size_t>(smem_max, 1)), 4 * sl_wpe_b(K, m == 1, allfull) / nw));
      const long long cap_wgs = (long long)h0->ncu * wpc;
      auto total = [&](long long p_) {
        long long t = 0;
        for (int u : act) { const Geo& g = hs[u]->g; const long long NS = ((long long)g.N * g.N + 63) / 64; t += (long long)g.L * ((NS + nw * p_ - 1) / (nw * p_)); }
        return t;
      };
      if (total(per) > cap_wgs && !getenv("VMR_BATCH_PER")) {
        long long p2 = per;
        while (p2 < 512 && total(p2) > cap_wgs) p2 += std::max<long long>(1, p2 / 16);
        if (total(p2) <= cap_wgs) per = p2;
      }
    }
    std::vector<SlUnit> su(act.size());
    std::vector<int> map;
    size_t smem = 0;
    for (size_t i = 0; i < act.size(); ++i) {
      vmr_ctx* h = hs[act[i]];
      const Geo& g = h->g;
      const SlShape sh = sl_shape(h, m != 2, m == 1, true);
      SlArgs a = sl_args(h, sh, 1, g.ml ? 1 : 0);
      a.elbo_dev = bt.be + (size_t)act[i] * 8;
      if (g.mut && m != 2) { a.nu_acc = h->nu_acc; a.commit_nu = 1; }   // (the statistics of the initial rho leave nu alone)
      const long long NS = ((long long)g.N * g.N + 63) / 64;
      a.Gl = (int)std::max<long long>(1, std::min<long long>(4096, (NS + nw * per - 1) / (nw * per)));
      su[i].a = a; su[i].g = g; su[i].blk0 = (int)map.size(); su[i].nblk = g.L * a.Gl;
      map.insert(map.end(), (size_t)su[i].nblk, (int)i);
      smem = std::max(smem, sh.smem);
    }
    HIPCHK(h0, hipMemcpyAsync(bt.su[m], su.data(), su.size() * sizeof(SlUnit), hipMemcpyHostToDevice, st));
    HIPCHK(h0, hipMemcpyAsync(bt.smap[m], map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(h0, hipStreamSynchronize(st));   // (the host vectors go away)
    bt.nb[m] = (int)map.size(); bt.tpb[m] = tpb; bt.smem[m] = smem;
  }
  bt.fsm = 0;
  {   // workgroups per layer of the finalize launch: about one per CU in all, 2..FG_G per layer
    long long layers = 0;
    for (int u : act) layers += hs[u]->g.L;
    bt.fg = (int)std::max<long long>(2, std::min<long long>(FG_G, (1LL * h0->ncu) / std::max<long long>(1, layers)));   // (64 units: 4 -- measured 1 / 2 / 4 / 8 / 16: 3.29 / 3.06 / 2.95 / 3.11 / 3.33 s of loops)
    if (getenv("VMR_BATCH_FG")) bt.fg = std::max(1, std::min(FG_G, atoi(getenv("VMR_BATCH_FG"))));   // (experiments)
  }
  for (size_t i = 0; i < act.size(); ++i) {
    vmr_ctx* h = hs[act[i]];
    const Geo& g = h->g;
    fu[i] = FinUnit{h->par, h->Hg, h->Cg, h->slotA, h->slotF, h->lutg, h->fin_g, h->nu_acc, h->slotR, bt.be + (size_t)act[i] * 8, g,
                    (int)gmap.size(), (int)rmap.size(), g.L * FR_G};
    gmap.insert(gmap.end(), (size_t)g.L * bt.fg, (int)i);
    rmap.insert(rmap.end(), (size_t)g.L * FR_G, (int)i);
    bt.fsm = std::max(bt.fsm, (size_t)2 * ((g.M + bt.fg - 1) / bt.fg) * 8);
  }
  HIPCHK(h0, hipMemcpyAsync(bt.fu, fu.data(), fu.size() * sizeof(FinUnit), hipMemcpyHostToDevice, st));
  HIPCHK(h0, hipMemcpyAsync(bt.gmap, gmap.data(), gmap.size() * sizeof(int), hipMemcpyHostToDevice, st));
  HIPCHK(h0, hipMemcpyAsync(bt.rmap, rmap.data(), rmap.size() * sizeof(int), hipMemcpyHostToDevice, st));
  HIPCHK(h0, hipStreamSynchronize(st));
  bt.ngb = (int)gmap.size(); bt.nrb = (int)rmap.size();
  (void)n_all;
  return VMR_OK;
}

int vmr_fit_loop_batch(vmr_handle* hs, int n, int max_iter, double tol, int decision, int cap, int* n_rows, int* row_iter, double* row_elbo,
                       double* row_runtime, int* row_reached, double* elbo_out, int* iters_out, int* converged_out, int* rc_out) {
  if (!hs || n <= 0 || !n_rows || !elbo_out || !iters_out || !converged_out || !rc_out) return VMR_EINVAL;
  if (cap > 0 && (!row_iter || !row_elbo || !row_runtime || !row_reached)) return VMR_EINVAL;
  for (int u = 0; u < n; ++u) {
    if (!hs[u]) return VMR_EINVAL;
    for (int v = 0; v < u; ++v) if (hs[v] == hs[u]) return fail(hs[u], VMR_EINVAL, "vmr_fit_loop_batch: a handle is listed twice");
    rc_out[u] = VMR_OK;
  }
  std::vector<LoopState> ls(n);
  auto rows_of = [&](int u, int*& ri, double*& re, double*& rr, int*& rq) {
    ri = cap > 0 ? row_iter + (size_t)u * cap : nullptr; re = cap > 0 ? row_elbo + (size_t)u * cap : nullptr;
    rr = cap > 0 ? row_runtime + (size_t)u * cap : nullptr; rq = cap > 0 ? row_reached + (size_t)u * cap : nullptr;
  };
  auto solo = [&](int u) {   // this unit's loop (or the rest of it) on its own
    int* ri; double* re; double* rr; int* rq;
    rows_of(u, ri, re, rr, rq);
    vmr_ctx* h = hs[u];
    if (!h->have_state) { rc_out[u] = fail(h, VMR_ESTATE, "vmr_set_state must be called before vmr_fit_loop_batch"); return; }
    if (h->restored) { rc_out[u] = fail(h, VMR_ESTATE, "vmr_fit_loop_batch after vmr_restore: the restored state is read-only until the next vmr_set_state"); return; }
    rc_out[u] = fit_loop_core(h, ls[u], max_iter, tol, decision, cap, ri, re, rr, rq);
  };
  auto finish = [&]() {
    int worst = VMR_OK;
    for (int u = 0; u < n; ++u) {
      n_rows[u] = ls[u].rows; elbo_out[u] = ls[u].elbo; iters_out[u] = ls[u].it - 1; converged_out[u] = ls[u].reached;
      if (rc_out[u] && !worst) worst = rc_out[u];
    }
    return worst;
  };
  // the units of the first batchable handle's kind advance together; the others run alone
  int lead = -1;
  for (int u = 0; u < n && lead < 0; ++u) if (hs[u]->have_state && !hs[u]->restored && batch_kind(hs[u], hs[u])) lead = u;
  std::vector<int> act;
  for (int u = 0; u < n; ++u) {
    if (lead >= 0 && hs[u]->have_state && !hs[u]->restored && batch_kind(hs[u], hs[lead]) && vmr_sl_batch_launcher(hs[u]->g.K)) act.push_back(u);
    else solo(u);
  }
  if (act.size() < 2 || max_iter < 2) { for (int u : act) solo(u); return finish(); }
  vmr_ctx* h0 = hs[act[0]];
  // Every failure from here on leaves through ONE exit: the code goes into rc_out of every unit still in the lockstep loop (their
  // traces, ELBOs and iteration counts are then meaningless) and into the return value.
  auto lockstep = [&]() -> int {
  HIPCHK(h0, hipSetDevice(h0->device));
  // the loop runs every unit's kernels on ONE stream (the first unit's): whatever a unit's own stream still holds -- the tail of
  // vmr_set_state, an earlier sweep -- is waited for here, once
  for (int u : act) HIPCHK(h0, hipStreamSynchronize(hs[u]->stream));
  // Iteration 1 (the ELBO is evaluated there, model.py:1036) also builds the statistics of the initial rho -- launches the
  // steady sweeps do not have.  Handles that come straight from vmr_set_state take it inside the tables (below); if some do
  // not, every handle runs it on its own stream.
  bool fresh = true;
  for (int u : act) fresh = fresh && !hs[u]->h_valid && !hs[u]->prof;
  if (!fresh) {
    for (int u : act) { rc_out[u] = sweep(hs[u], 1); }
    for (size_t i = 0; i < act.size();) {
      const int u = act[i];
      if (!rc_out[u]) rc_out[u] = read_elbo(hs[u], &ls[u].elbo);
      if (rc_out[u]) { (void)hipStreamSynchronize(hs[u]->stream); act.erase(act.begin() + i); continue; }
      ls[u].coincide = (fabs(ls[u].elbo - (-1e10)) < tol) ? 1 : 0;
      if (ls[u].coincide > decision) ls[u].reached = 1;
      ls[u].it = 2;
      if (ls[u].reached) { act.erase(act.begin() + i); continue; }
      if (!batch_steady(hs[u])) { solo(u); act.erase(act.begin() + i); continue; }
      ++i;
    }
  }
  if (act.size() < 2) { for (int u : act) solo(u); act.clear(); return VMR_OK; }
  h0 = hs[act[0]];
  hipStream_t st = h0->stream;
  const int K = h0->g.K, allfull = h0->all_full != 0;
  sl_launch_batch_fn pass = vmr_sl_batch_launcher(K);
  BatchTables bt;
  {
    size_t blocks = 0, gb = 0, rb = 0;
    for (int u : act) {
      const Geo& g = hs[u]->g;
      blocks += (size_t)g.L * 4096; gb += (size_t)g.L * FG_G; rb += (size_t)g.L * FR_G;
    }
    for (int m = 0; m < 3; ++m) {
      HIPCHK(h0, hipMalloc(&bt.su[m], act.size() * sizeof(SlUnit)));
      HIPCHK(h0, hipMalloc(&bt.smap[m], blocks * sizeof(int)));
    }
    HIPCHK(h0, hipMalloc(&bt.fu, act.size() * sizeof(FinUnit)));
    HIPCHK(h0, hipMalloc(&bt.gmap, gb * sizeof(int)));
    HIPCHK(h0, hipMalloc(&bt.rmap, rb * sizeof(int)));
    HIPCHK(h0, hipMalloc(&bt.be, (size_t)n * 8 * 8));
    HIPCHK(h0, hipMemsetAsync(bt.be, 0, (size_t)n * 8 * 8, st));
  }
  std::vector<double> be_host((size_t)n * 8);
  int rc = batch_tables(hs, act, n, bt, st);
  if (rc) return rc;
  const bool lazy_rho = !getenv("VMR_ALWAYS_STORE_RHO");   // plain sweeps do not write rho: the ELBO sweep every loop ends with does
  auto launch_sweeps = [&](int mode) -> int {
    hipLaunchKernelGGL(k_fin_gamma_b, dim3(bt.ngb), dim3(FIN_TPB), bt.fsm, st, bt.fu, bt.gmap, bt.fg);
    const int pm = (mode == 0 && lazy_rho) ? 4 : mode;
    int r = pass(h0, st, pm, allfull, bt.su[mode], bt.smap[mode], bt.nb[mode], bt.tpb[mode], bt.smem[mode]);
    for (int u : act) hs[u]->rho_stale = pm == 4;
    if (r) return r;
    if (mode == 1) hipLaunchKernelGGL(k_fin_rho_b, dim3(bt.nrb), dim3(TPB), 0, st, bt.fu, bt.rmap);
    HIPCHK(h0, hipGetLastError());
    return VMR_OK;
  };
  int it = fresh ? 1 : 2;   // every active unit stands at the same iteration
  while (!act.empty() && it <= max_iter) {
    const int nxt = (it == 1 || it % 10 == 0 || it == max_iter) ? it : std::min(max_iter, (it / 10 + 1) * 10);
    for (; it < nxt; ++it) if ((rc = launch_sweeps(0))) break;
    if (rc) break;
    HIPCHK(h0, hipStreamSynchronize(st));
    const auto t0 = std::chrono::steady_clock::now();
    if (it == 1) {
      // the first sweep of the realisations: H, the all-ones sums and the mask sums of the initial rho (launch_hist, for all units)
      for (int u : act) {
        vmr_ctx* h = hs[u];
        const Geo& g = h->g;
        HIPCHK(h0, hipMemsetAsync(h->Hg, 0, (size_t)g.L * NH * g.Y * g.Mp * g.K * 8, st));
        HIPCHK(h0, hipMemsetAsync(h->slotF, 0, (size_t)g.L * NSLOT * g.K * 8, st));
        if (g.ml) HIPCHK(h0, hipMemsetAsync(h->slotA, 0, (size_t)g.L * NSLOT * g.W * 64 * g.K * 8, st));
      }
      if ((rc = pass(h0, st, 3, allfull, bt.su[2], bt.smap[2], bt.nb[2], bt.tpb[2], bt.smem[2]))) break;
      for (int u : act) {   // (where a sweep leaves a handle: statistics of the current rho in place, nothing folded or zeroed)
        vmr_ctx* h = hs[u];
        h->h_valid = true; h->h_zero = false; h->h_reduced = false; h->f_valid = h->g.fuse_full != 0; h->a_valid = h->g.ml != 0; h->a_zero = false;
      }
    }
    if ((rc = launch_sweeps(1))) break;
    HIPCHK(h0, hipMemcpyAsync(be_host.data(), bt.be, (size_t)n * 8 * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(h0, hipStreamSynchronize(st));
    const double runtime = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ++it;
    bool changed = false;
    for (size_t i = 0; i < act.size();) {
      const int u = act[i];
      LoopState& s = ls[u];
      const double old = s.elbo;
      s.elbo = be_host[(size_t)u * 8];
      s.it = it;
      bool out = false;
      if (isnan(s.elbo)) { rc_out[u] = fail(hs[u], VMR_ENAN, "ELBO is NaN!!!!"); out = true; }
      else {
        s.coincide = (fabs(s.elbo - old) < tol) ? s.coincide + 1 : 0;
        if (s.coincide > decision) s.reached = 1;
        if ((it - 1) % 10 == 0 && s.rows < cap) {
          int* ri; double* re; double* rr; int* rq;
          rows_of(u, ri, re, rr, rq);
          ri[s.rows] = it - 1; re[s.rows] = s.elbo; rr[s.rows] = runtime; rq[s.rows] = s.reached;
          ++s.rows;
        }
        out = s.reached != 0;
      }
      if (out) { act.erase(act.begin() + i); changed = true; } else ++i;
    }
    if (changed && !act.empty() && it <= max_iter && (rc = batch_tables(hs, act, n, bt, st))) break;
  }
  (void)hipStreamSynchronize(st);
  return rc;
  };
  const int rc_all = lockstep();
  if (rc_all) {
    (void)hipStreamSynchronize(h0->stream);
    for (int u : act) if (!rc_out[u]) { rc_out[u] = rc_all; if (hs[u] != h0) hs[u]->err = h0->err; }
  }
  return finish();
}

int vmr_elbo(vmr_handle h, double* out) {
  if (!h || !out) return VMR_EINVAL;
  if (!h->have_state) return fail(h, VMR_ESTATE, "vmr_set_state must be called before vmr_elbo");
  if (h->restored) return fail(h, VMR_ESTATE, "vmr_elbo after vmr_restore: the restored state is read-only until the next vmr_set_state");
  HIPCHK(h, hipSetDevice(h->device));
  int rc;
  if ((rc = launch_rho(h, 2, false))) return rc;
  return read_elbo(h, out);
}

int vmr_sweep_local(vmr_handle h, int want_elbo, double* out3) {
  if (!h || !out3) return VMR_EINVAL;
  if (!h->have_state) return fail(h, VMR_ESTATE, "vmr_set_state must be called before vmr_sweep_local");
  if (h->restored) return fail(h, VMR_ESTATE, "vmr_sweep_local after vmr_restore: the restored state is read-only until the next vmr_set_state");
  HIPCHK(h, hipSetDevice(h->device));
  int rc;
  if ((rc = launch_gamma(h, true))) return rc;
  if ((rc = launch_rho(h, want_elbo ? 1 : 0, false, true))) return rc;   // rho updated, nu NOT committed
  if (!want_elbo && !h->sparse) {   // launch_rho skipped the finalize kernel: run it for the raw pieces only
    if ((rc = launch_fin_rho(h, 0, 0))) return rc;   // (sorted lists: the pass left the raw nu sum itself)
  }
  double v[4];
  HIPCHK(h, hipMemcpyAsync(v, h->elbo_dev, 32, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  out3[0] = v[1]; out3[1] = v[2]; out3[2] = v[3];
  return VMR_OK;
}

int vmr_commit_nu(vmr_handle h, double nu_partial_total) {
  if (!h) return VMR_EINVAL;
  if (!h->have_state) return fail(h, VMR_ESTATE, "vmr_set_state must be called before vmr_commit_nu");
  HIPCHK(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(k_commit_nu, dim3(1), dim3(64), 0, h->stream, h->par, nu_partial_total, (const double*)nullptr, h->g);
  HIPCHK(h, hipGetLastError());
  return VMR_OK;
}

// the same exchange with the three doubles left on the device (RCCL all-reduce on the handle's stream, see vmr_stream)
int vmr_sweep_local_dev(vmr_handle h, int want_elbo, double* out3_dev) {
  if (!h || !out3_dev) return VMR_EINVAL;
  if (!h->have_state) return fail(h, VMR_ESTATE, "vmr_set_state must be called before vmr_sweep_local_dev");
  if (h->restored) return fail(h, VMR_ESTATE, "vmr_sweep_local_dev after vmr_restore: the restored state is read-only until the next vmr_set_state");
  HIPCHK(h, hipSetDevice(h->device));
  int rc;
  if ((rc = launch_gamma(h, true))) return rc;
  if ((rc = launch_rho(h, want_elbo ? 1 : 0, false, true))) return rc;
  if (!want_elbo && !h->sparse && (rc = launch_fin_rho(h, 0, 0))) return rc;
  HIPCHK(h, hipMemcpyAsync(out3_dev, h->elbo_dev + 1, 24, hipMemcpyDeviceToDevice, h->stream));
  return VMR_OK;
}

int vmr_commit_nu_dev(vmr_handle h, const double* nu_partial_total_dev) {
  if (!h || !nu_partial_total_dev) return VMR_EINVAL;
  if (!h->have_state) return fail(h, VMR_ESTATE, "vmr_set_state must be called before vmr_commit_nu_dev");
  HIPCHK(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(k_commit_nu, dim3(1), dim3(64), 0, h->stream, h->par, 0.0, nu_partial_total_dev, h->g);
  HIPCHK(h, hipGetLastError());
  return VMR_OK;
}

void* vmr_stream(vmr_handle h) { return h ? (void*)h->stream : nullptr; }

int vmr_sub_step(vmr_handle h, int which) {
  if (!h) return VMR_EINVAL;
  if (!h->have_state) return fail(h, VMR_ESTATE, "vmr_set_state must be called before vmr_sub_step");
  if (h->restored) return fail(h, VMR_ESTATE, "vmr_sub_step after vmr_restore: the restored state is read-only until the next vmr_set_state");
  HIPCHK(h, hipSetDevice(h->device));
  if (h->g.det) return fail(h, VMR_ESTATE, "vmr_sub_step is not available with VMR_DETERMINISTIC=1 (whole sweeps only: vmr_step, vmr_fit_loop)");
  { const int rce = ensure_rho(h); if (rce) return rce; }
  switch (which) {
    case VMR_STEP_GAMMA: return launch_gamma(h, false);
    case VMR_STEP_PHI: return launch_phi(h);
    // the rho pass also rebuilds the statistics H the nu update reads; nu is committed by the NU sub-step
    case VMR_STEP_RHO: return launch_rho(h, 0, false);
    case VMR_STEP_NU: {
      if (!h->g.mut) return VMR_OK;
      if (!h->h_valid) { int rc = launch_hist(h); if (rc) return rc; }
      return launch_fin_rho(h, 1, 0);
    }
    default: return fail(h, VMR_EINVAL, "unknown sub-step");
  }
}

// the whole parameter block in ONE copy (tens of KB): the getters of the small posteriors are called once per realisation of every
// small fit, from several host threads -- six small copies with a synchronisation each cost them a millisecond per call
static int fetch_par(vmr_ctx* h, std::vector<double>& buf) {
  buf.resize(h->par_doubles);
  HIPCHK(h, hipMemcpyAsync(buf.data(), h->par, h->par_doubles * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return VMR_OK;
}
static void rows_lm(const Geo& g, const std::vector<double>& buf, size_t off, double* dst) {
  for (int l = 0; l < g.L; ++l) memcpy(dst + (size_t)l * g.M, &buf[off + (size_t)l * g.Mp], (size_t)g.M * 8);
}

int vmr_get_state(vmr_handle h, double* gamma_shp, double* gamma_rte, double* phi_shp, double* phi_rte,
                  double* nu_shp, double* nu_rte, double* rho) {
  if (!h) return VMR_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  const Geo& g = h->g;
  ParOff o = par_off(g.L, g.Mp, g.K);
  int rc;
  if (gamma_shp || gamma_rte || phi_shp || phi_rte || nu_shp || nu_rte) {
    std::vector<double> buf;
    if ((rc = fetch_par(h, buf))) return rc;
    if (gamma_shp) rows_lm(g, buf, o.g_shp, gamma_shp);
    if (gamma_rte) rows_lm(g, buf, o.g_rte, gamma_rte);
    if (phi_shp) memcpy(phi_shp, &buf[o.p_shp], (size_t)g.L * g.K * 8);
    if (phi_rte) memcpy(phi_rte, &buf[o.p_rte], (size_t)g.L * g.K * 8);
    if (nu_shp) *nu_shp = buf[o.sc + SC_NU_SHP];
    if (nu_rte) *nu_rte = buf[o.sc + SC_NU_RTE];
  }
  if (rho) {
    { const int rce = ensure_rho(h); if (rce) return rce; }
    const size_t nr = (size_t)g.L * g.N * g.N * g.K;
    const double* src = h->rho;
    if (h->perm) {   // back to tie order
      if (!h->nat) HIPCHK(h, hipMalloc(&h->nat, nr * 8));
      if ((rc = sl_permute_rows(h, h->rho, h->nat, false))) return rc;
      src = h->nat;
    }
    if ((rc = d2h(h, rho, src, nr * 8))) return rc;
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return VMR_OK;
}

int vmr_get_geometric(vmr_handle h, double* g_theta, double* g_lambda, double* g_nu, double* g_nu_cache) {
  if (!h) return VMR_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  const Geo& g = h->g;
  ParOff o = par_off(g.L, g.Mp, g.K);
  int rc;
  std::vector<double> buf;
  if ((rc = fetch_par(h, buf))) return rc;
  if (g_theta) rows_lm(g, buf, o.G_th, g_theta);
  if (g_lambda) memcpy(g_lambda, &buf[o.G_la], (size_t)g.L * g.K * 8);
  if (g_nu) *g_nu = buf[o.sc + SC_G_NU];
  if (g_nu_cache) *g_nu_cache = buf[o.sc + SC_G_NU_STALE];
  return VMR_OK;
}

// `_update_optimal_parameters` (model.py:925-942) without a host round trip: keep a device copy of the posteriors
int vmr_snapshot(vmr_handle h) {
  if (!h) return VMR_EINVAL;
  if (!h->have_state) return fail(h, VMR_ESTATE, "vmr_set_state must be called before vmr_snapshot");
  HIPCHK(h, hipSetDevice(h->device));
  { const int rce = ensure_rho(h); if (rce) return rce; }
  const Geo& g = h->g;
  const size_t nr = (size_t)g.L * g.N * g.N * g.K * 8, np_ = h->par_doubles * 8;
  if (!h->rho_snap) { HIPCHK(h, hipMalloc(&h->rho_snap, nr)); HIPCHK(h, hipMalloc(&h->par_snap, np_)); }
  HIPCHK(h, hipMemcpyAsync(h->rho_snap, h->rho, nr, hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->par_snap, h->par, np_, hipMemcpyDeviceToDevice, h->stream));
  h->have_snap = true;
  return VMR_OK;
}

// make the snapshot the current state again (rho and every parameter; the statistics are rebuilt on the next sweep)
int vmr_restore(vmr_handle h) {
  if (!h) return VMR_EINVAL;
  if (!h->have_snap) return fail(h, VMR_ESTATE, "vmr_snapshot must be called before vmr_restore");
  HIPCHK(h, hipSetDevice(h->device));
  const Geo& g = h->g;
  HIPCHK(h, hipMemcpyAsync(h->rho, h->rho_snap, (size_t)g.L * g.N * g.N * g.K * 8, hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->par, h->par_snap, h->par_doubles * 8, hipMemcpyDeviceToDevice, h->stream));
  h->h_valid = false; h->f_valid = false; h->a_valid = false; h->h_zero = false;
  h->restored = true;   // the log prior on the device is the LAST realisation's: reading is fine, sweeping is not
  h->rho_stale = false;
  HIPCHK(h, hipMemsetAsync(h->slotF, 0, (size_t)g.L * NSLOT * g.K * 8, h->stream));
  hipLaunchKernelGGL(k_build_lut, dim3(g.L), dim3(256), 0, h->stream, h->par, h->lutg, g);
  HIPCHK(h, hipGetLastError());
  return VMR_OK;
}

int vmr_readout(vmr_handle h, int method, double threshold, void* out, int out_on_device) {
  if (!h || !out) return VMR_EINVAL;
  if (!h->have_state) return fail(h, VMR_ESTATE, "vmr_set_state must be called before vmr_readout");
  if (method < VMR_READ_RHO_MAX || method > VMR_READ_THRESHOLD) return fail(h, VMR_EINVAL, "unknown read-out method");
  HIPCHK(h, hipSetDevice(h->device));
  { const int rce = ensure_rho(h); if (rce) return rce; }
  const Geo& g = h->g;
  const size_t ties = (size_t)g.L * g.N * g.N, bytes = ties * (method == VMR_READ_RHO_MEAN ? 8 : 1);
  void* dst = out;
  if (!out_on_device) HIPCHK(h, hipMalloc(&dst, bytes));
  const size_t T_ = (size_t)g.N * g.N;
  hipLaunchKernelGGL(k_readout, dim3((unsigned)std::min<size_t>(4096, (ties + 255) / 256)), dim3(256), 0, h->stream, h->rho, dst, ties,
                     g.K, method, threshold, h->perm, T_, (T_ + 63) / 64);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && !out_on_device) e = hipMemcpyAsync(out, dst, bytes, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (!out_on_device) (void)hipFree(dst);
  if (e != hipSuccess) { h->err = std::string("vmr_readout: ") + hipGetErrorString(e); (void)hipGetLastError(); return VMR_EHIP; }
  return VMR_OK;
}

int vmr_sample(vmr_handle h, uint64_t seed, int n_trials, uint8_t* out, int out_on_device) {
  if (!h || !out) return VMR_EINVAL;
  if (!h->have_state) return fail(h, VMR_ESTATE, "vmr_set_state must be called before vmr_sample");
  if (n_trials < 1) return fail(h, VMR_EINVAL, "n_trials must be positive");
  HIPCHK(h, hipSetDevice(h->device));
  { const int rce = ensure_rho(h); if (rce) return rce; }
  const Geo& g = h->g;
  const size_t T_ = (size_t)g.N * g.N, ties = (size_t)g.L * T_;
  uint8_t* dst = out;
  if (!out_on_device) HIPCHK(h, hipMalloc(&dst, ties));
  if (g.K > KMAX) { const int rcg = gen_sample(h, (unsigned long long)seed, n_trials, dst); if (rcg) { if (!out_on_device) (void)hipFree(dst); return rcg; } }
  else hipLaunchKernelGGL(k_sample, dim3((unsigned)std::min<size_t>(4096, (ties + 255) / 256)), dim3(256), 0, h->stream, h->rho, dst, ties, g.K,
                          n_trials, (unsigned long long)seed, h->perm, T_, (T_ + 63) / 64);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && !out_on_device) e = hipMemcpyAsync(out, dst, ties, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (!out_on_device) (void)hipFree(dst);
  if (e != hipSuccess) { h->err = std::string("vmr_sample: ") + hipGetErrorString(e); (void)hipGetLastError(); return VMR_EHIP; }
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
  // enable = 2: only the passes over the data (the roofline kernels); the small finalize kernels run unbracketed -- two event
  // records per launch are ~4 us on the stream, 9 % of a config-3 sweep when every kernel carries them
  h->prof_mask = enable == 2 ? ((1u << VMR_KERNEL_GAMMA_COUNTS) | (1u << VMR_KERNEL_RHO) | (1u << VMR_KERNEL_ELBO) | (1u << VMR_KERNEL_RHO_ELBO) | (1u << VMR_KERNEL_RHO_NOSTORE)) : ~0u;
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

int vmr_data_format(vmr_handle h, int* sparse, uint64_t* nnz) {
  if (!h) return VMR_EINVAL;
  if (sparse) *sparse = h->sparse;
  if (nnz) *nnz = h->nnz;
  return VMR_OK;
}

int vmr_mask_format(vmr_handle h, int* lists, uint64_t* listed) {
  if (!h) return VMR_EINVAL;
  if (lists) *lists = h->rq ? 1 : 0;
  if (listed) *listed = h->n_rm;
  return VMR_OK;
}

int vmr_sweep_shape(vmr_handle h, int* passes, int* lds_levels, uint64_t* far_reports) {
  if (!h) return VMR_EINVAL;
  if (passes) *passes = h->g.two_pass ? 2 : 1;
  if (lds_levels) *lds_levels = h->g.gen ? 0 : h->g.hc;
  if (far_reports) *far_reports = (h->g.farl && !h->far_off.empty()) ? h->far_off.back() : 0ull;
  return VMR_OK;
}

int vmr_kernel_bytes(vmr_handle h, int kernel_class, double* bytes) {
  if (!h || !bytes) return VMR_EINVAL;
  const Geo& g = h->g;
  const double V = (double)g.L * g.N * g.N * g.M;   // canonical: X 1 B/elt, R 1 bit/elt
  const double SX = V, SR = V / 8.0, Srho = 8.0 * g.L * (double)g.N * g.N * g.K;
  if (h->sparse) {   // report lists: 4 B per non-zero count + 4 B per tie; mask words only for partial rows
    const double ties = (double)g.L * g.N * g.N;
    // entries (without the rounds' padding); step pointers: two per 64 ties in the step layout, one in the sorted lists
    const double E = (g.wide ? 8.0 : 4.0) * (double)h->nnz, RP = 4.0 * (ties / 64.0 + g.L);   // (two-word entries: 8 B per report)
    const double mask = h->all_full ? 0.0 : ties + (h->rq ? 4.0 * ties + 2.0 * (double)h->n_rm : (double)h->n_partial * g.W * 8.0);
    const double Q = g.mut ? 4.0 * ties : 0.0;
    switch (kernel_class) {
      case VMR_KERNEL_GAMMA_MASK: *bytes = ties + (h->rq ? 4.0 * ties + 2.0 * (double)h->n_rm : (double)h->n_partial * g.W * 8.0) + Srho; break;
      case VMR_KERNEL_GAMMA_COUNTS:   // (x0p: the rounds of level 0 only are not read -- a count per tie instead)
        *bytes = ((h->x0p && g.two_pass) ? 4.0 * (double)h->stat_slots + 4.0 * ties : E) + RP + Srho;
        break;
      case VMR_KERNEL_RHO: *bytes = E + RP + mask + 2.0 * Srho; break;
      case VMR_KERNEL_RHO_NOSTORE: *bytes = E + RP + mask + Srho; break;   // (the log prior is read, rho is not written)
      case VMR_KERNEL_ELBO: *bytes = E + RP + mask + Q + 2.0 * Srho; break;
      case VMR_KERNEL_RHO_ELBO: *bytes = E + RP + mask + Q + 2.0 * Srho; break;
      default: *bytes = 0.0; break;
    }
    return VMR_OK;
  }
  switch (kernel_class) {
    case VMR_KERNEL_GAMMA_MASK: *bytes = SR + Srho; break;
    case VMR_KERNEL_GAMMA_COUNTS: *bytes = SX + Srho; break;   // statistics pass (once per realisation)
    case VMR_KERNEL_PHI: *bytes = 0.0; break;                  // no pass of its own: phi_shp comes from H
    case VMR_KERNEL_RHO: *bytes = SX + SR + 2.0 * Srho; break;
    case VMR_KERNEL_ELBO: *bytes = SX + SR + 2.0 * Srho; break;
    case VMR_KERNEL_RHO_ELBO: *bytes = SX + SR + 2.0 * Srho; break;
    default: *bytes = 0.0; break;
  }
  return VMR_OK;
}

}  // extern "C"
