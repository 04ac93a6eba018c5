// sweep_gen.hip -- the general CAVI kernels (sweep_gen.h): any K in [2, KGEN_MAX], packed or wide entries.
//
//   k_sweep_gen      one pass over the sorted report lists: the rho update (model.py:763-818, 889-923), the statistics
//                    H[l][y][m][k] = sum x rho_k of the new rho (what gamma / phi / nu are finished from: model.py:698-761, 820-887),
//                    the sums of rho over mask rows (model.py:704-718, 742-749) and the ELBO's data terms (model.py:948-995)
//   k_fin_gamma_gen  gamma, then phi, from H and the mask sums (model.py:698-761)
//   k_fin_phi_gen    the PHI sub-step's commit (mutuality on)
//   k_sample_gen     vmr_sample for K > KMAX
// nu and the ELBO's assembly: k_fin_rho (vimure_hip.hip), which reads this file's single-copy H directly.
//
// A tie is spread over G = min(64, 2^ceil(log2 K)) lanes, one category per lane (NCH = ceil(K / 64) per lane beyond 64): the sum
// over categories that normalises rho and the ELBO's inner sums are shuffles inside the group, the adds into H land on K
// consecutive doubles.  The factor of a report, F = (E log theta_m + E log lambda_k) w1_k(m, y), is computed where it is used.
#include "sweep_gen.h"

struct GenArgs {
  const unsigned *E, *EX, *rs;
  const unsigned long long* ebase;
  const unsigned* perm;
  const uint8_t* cls;          // by position; null: every mask row is all ones
  const unsigned* Qt;          // by position
  const uint64_t* Rb; const unsigned* rq; const unsigned short* Rm; const unsigned long long* rbase;   // partial mask rows, by tie
  double* rho; const double* logpr; const double* par;
  double *slotR, *Hg, *slotF, *slotA;
  int Gl, update, elbo, hist, sum_a;
  int lds_tab;   // the per-reporter tables (G_theta, E log theta, E theta: 24 Mp bytes) are copied to LDS
  unsigned long long* ctr;   // [L], zero at launch: the next unclaimed step of a layer (its workgroups draw their steps from it)
  int YL;        // rows y < YL of H are summed in LDS first (a workgroup walks many steps; most reports mirror a small count), then added to Hg
  int dbg;       // -DGEN_DEBUG builds: VMR_GEN_DBG bits switch phases off for timing (1: the adds into H, 2: the rounds of the update, 4: the rho store)
};
#ifdef GEN_DEBUG
#define GEN_ON(bit) (!(a.dbg & (bit)))
#else
#define GEN_ON(bit) true
#endif

template <int NCH>
__global__ __launch_bounds__(512) void k_sweep_gen(GenArgs a, Geo g) {
  __shared__ double red[16];
  __shared__ double s_tfull;
  const int K = g.K, Mp = g.Mp, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = (int)blockDim.x >> 6;
  int lg = 1;
  while ((1 << lg) < K && lg < 6) ++lg;
  const int G = 1 << lg, kk = lane & (G - 1), grp = lane >> lg, TPW = 64 >> lg;   // lanes per tie, this lane's category, its tie of the TPW in flight
  const int l = (int)blockIdx.x / a.Gl, gb = (int)blockIdx.x - l * a.Gl;
  const ParOff o = par_off(g.L, Mp, K);
  const size_t T = (size_t)g.N * g.N;
  const long long NS = (long long)((T + 63) / 64);
  extern __shared__ double gen_lds[];
  const double* Gth = a.par + o.G_th + (size_t)l * Mp;
  const double* Lth = a.par + o.l_th + (size_t)l * Mp;
  const double* Eth = a.par + o.E_th + (size_t)l * Mp;
  double* Hs = gen_lds + (a.lds_tab ? 3 * Mp : 0);   // H[y < YL][m][k]
  const unsigned hs_rows = (unsigned)a.YL * (unsigned)Mp;
  for (unsigned q = tid; q < hs_rows * (unsigned)K; q += blockDim.x) Hs[q] = 0.0;
  if (a.lds_tab) {   // (a report's factor is computed where it is used: two table reads per report, from LDS instead of L2)
    for (int m = tid; m < Mp; m += (int)blockDim.x) { gen_lds[m] = Gth[m]; gen_lds[Mp + m] = Lth[m]; gen_lds[2 * Mp + m] = Eth[m]; }
    __syncthreads();
    Gth = gen_lds; Lth = gen_lds + Mp; Eth = gen_lds + 2 * Mp;
  }
  {   // T of an all-ones mask row: sum_m E[theta_m] (model.py:766-792)
    double tf = 0.0;
    for (int m = tid; m < g.M; m += (int)blockDim.x) tf += Eth[m];
    tf = block_sum_n(tf, red);
    if (tid == 0) s_tfull = tf;
    __syncthreads();
  }
  const double Tfull = s_tfull;
  double Ela[NCH], Gla[NCH], Lla[NCH];
  bool kv[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int k = c * G + kk;
    kv[c] = k < K;
    Ela[c] = kv[c] ? a.par[o.E_la + l * K + k] : 0.0;
    Gla[c] = kv[c] ? a.par[o.G_la + l * K + k] : 0.0;
    Lla[c] = kv[c] ? a.par[o.l_la + l * K + k] : 0.0;
  }
  const double gnu_f = a.par[o.sc + SC_G_NU];                                    // the cache refresh before the rho update (model.py:651)
  const double gnu_e = a.par[o.sc + (a.update ? SC_G_NU : SC_G_NU_STALE)];      // the ELBO's: the stale one (model.py:970)
  const double eps = g.eps;
  const unsigned* rsl = a.rs + (size_t)l * (NS + 1);
  const unsigned* El = a.E + a.ebase[l];
  const unsigned* EXl = a.EX ? a.EX + a.ebase[l] : nullptr;
  const unsigned* pl = a.perm + (size_t)l * NS * 64;
  const uint8_t* cl = a.cls ? a.cls + (size_t)l * T : nullptr;
  const unsigned* Ql = a.Qt ? a.Qt + (size_t)l * T : nullptr;
  const uint64_t* Rl = a.Rb ? a.Rb + (size_t)l * T * g.W : nullptr;
  const unsigned* rql = a.rq ? a.rq + (size_t)l * (T + 1) : nullptr;
  const unsigned short* Rml = a.rq ? a.Rm + a.rbase[l] : nullptr;
  double* rl = a.rho + (size_t)l * T * K;
  const double* lpl = a.logpr + (size_t)l * T * K;
  double* Hl = a.Hg + (size_t)l * g.Y * Mp * K;
  double* Al = a.slotA + ((size_t)l * NSLOT + (gb % NSLOT)) * (size_t)g.W * 64 * K;
  // the reporters of a partial mask row: f(m) for every m with R[l, tie, m] = 1
  auto for_reporters = [&](unsigned tie, auto&& f) {
    if (rql) {
      const unsigned q0 = rql[tie], q1 = rql[tie + 1];
      for (unsigned q = q0; q < q1; ++q) f((int)Rml[q]);
    } else if (Rl) {
      const uint64_t* rw = Rl + (size_t)tie * g.W;
      for (int w = 0; w < g.W; ++w) {
        uint64_t bits = rw[w];
        while (bits) { const int b = __builtin_ctzll(bits); bits &= bits - 1; f(w * 64 + b); }
      }
    }
  };
  double e_lin = 0.0, e_q = 0.0, e_log = 0.0;
  double accF[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) accF[c] = 0.0;

  constexpr int RB = 8;   // rounds of a step held in registers, one tie's per lane (coalesced loads), handed to the tie's lanes by shuffles
  // A step's G sub-steps (TPW ties each) are shared by sw waves of the workgroup: the longest steps (a tie with many reports: one
  // round per report, in sequence) then bound the pass by 1 / sw of their length.
  // The steps are drawn from a counter, in list order -- longest first (the lists are sorted by reports, descending): a fixed
  // share per workgroup leaves the CUs with the long steps working alone at the end.
  const int sw = nw < G ? nw : G, nsb = nw / sw, sub0 = wv % sw;
  __shared__ unsigned long long s_next;
  for (;;) {
    if (tid == 0) s_next = atomicAdd(&a.ctr[l], (unsigned long long)nsb);
    __syncthreads();
    const long long s = (long long)s_next + wv / sw;
    __syncthreads();
    if (s - wv / sw >= NS) break;
    if (s >= NS) continue;
    const unsigned ea = rsl[s];
    const int R = (int)((rsl[s + 1] - ea) >> 6);
    // what the step's 64 ties need, requested at once, lane <-> position: the class of the mask row, the tie (partial rows), the
    // ELBO's mirror sum, the first RB rounds of entries -- one memory round trip per step instead of one per tie and round
    const size_t posl = (size_t)s * 64 + (unsigned)lane;
    const bool actl = posl < T;
    const unsigned cls_v = actl ? (cl ? (unsigned)cl[posl] : 1u) : 0u;
    const unsigned tie_v = (actl && cls_v == 2u) ? pl[posl] : 0u;
    const unsigned qt_v = (a.elbo && Ql && actl) ? Ql[posl] : 0u;
    unsigned ent[RB], enx[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const bool on = j < R;
      ent[j] = on ? El[(size_t)ea + (unsigned)j * 64 + (unsigned)lane] : 0u;
      enx[j] = (on && EXl) ? EXl[(size_t)ea + (unsigned)j * 64 + (unsigned)lane] : 0u;
    }
    // f(ym, x, inr) over the rounds of the tie at position pi, RB at a time: the first RB from the registers of lane pi, later
    // ones from memory -- RB independent loads in flight, not one round trip per round
    auto for_rounds = [&](int pi, auto&& f) {
      for (int r0 = 0; r0 < R; r0 += RB) {
        unsigned eb[RB], xb[RB];
        if (r0 == 0) {
#pragma unroll
          for (int j = 0; j < RB; ++j) {
            eb[j] = (unsigned)__shfl((int)ent[j], pi, 64);
            xb[j] = EXl ? (unsigned)__shfl((int)enx[j], pi, 64) : 0u;
          }
        } else {
#pragma unroll
          for (int j = 0; j < RB; ++j) {
            const bool on = r0 + j < R;
            const size_t slot = (size_t)ea + (unsigned)(r0 + j) * 64 + (unsigned)pi;
            eb[j] = on ? El[slot] : 0u;
            xb[j] = (on && EXl) ? EXl[slot] : 0u;
          }
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          if (r0 + j < R) {   // (wave-uniform)
            if (EXl) f(eb[j], xb[j] >> 1, xb[j] & 1u);
            else f(SL_YM(eb[j]), SL_X(eb[j]), SL_INR(eb[j]));
          }
        }
      }
    };
    // the log prior / current rho of the first TPW ties; the next ones are requested while these are worked on
    double lpn[NCH], rn[NCH];
    auto fetch_sub = [&](int sub_) {
      const size_t pos_ = (size_t)s * 64 + (unsigned)(sub_ * TPW + grp);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const bool on = sub_ < G && pos_ < T && kv[c];
        lpn[c] = (on && (a.update || a.elbo)) ? lpl[pos_ * K + (c * G + kk)] : 0.0;
        rn[c] = (on && !a.update) ? rl[pos_ * K + (c * G + kk)] : 0.0;
      }
    };
    fetch_sub(sub0);
    for (int sub = sub0; sub < G; sub += sw) {   // the step's 64 ties, TPW at a time, this wave's share
      const int pi = sub * TPW + grp;
      const size_t pos = (size_t)s * 64 + pi;
      const bool act = pos < T;
      const unsigned cls = (unsigned)__shfl((int)cls_v, pi, 64);
      const unsigned tie = (unsigned)__shfl((int)tie_v, pi, 64);
      const unsigned qt = (unsigned)__shfl((int)qt_v, pi, 64);   // (shuffles only where the whole wave passes)
      double lp[NCH], r[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) { lp[c] = lpn[c]; r[c] = rn[c]; }
      fetch_sub(sub + sw);
      double Tt = 0.0;   // sum_m R[tie, m] E[theta_m]
      if (a.update || a.elbo) {
        if (cls == 1u) Tt = Tfull;
        else if (cls == 2u) for_reporters(tie, [&](int m) { Tt += Eth[m]; });
      }
      if (a.update) {
        double U[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) U[c] = 0.0;
        if (GEN_ON(2)) for_rounds(pi, [&](unsigned ym, unsigned x, unsigned) {
          if (x != 0u) {
            const unsigned y = ym / (unsigned)Mp, m = ym - y * (unsigned)Mp;
            const double lt = Lth[m], gt = Gth[m], dx = (double)x;
#pragma unroll
            for (int c = 0; c < NCH; ++c)
              if (kv[c]) U[c] = fma(dx, f_entry(g.mut, lt, gt, Lla[c], Gla[c], gnu_f, (int)y), U[c]);
          }
        });
        double sl = 0.0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          r[c] = kv[c] ? exp((lp[c] + U[c]) - Tt * Ela[c]) : 0.0;   // raw exponentials, no max-subtraction (model.py:807)
          sl += r[c];
        }
        const double sum = group_sum(sl, G);
        if (sum > 0.0) {   // model.py:808-811: rows whose exponentials all underflow stay all zero
#pragma unroll
          for (int c = 0; c < NCH; ++c) r[c] /= sum;
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          if (!act) r[c] = 0.0;
          if (act && kv[c] && GEN_ON(4)) rl[pos * K + (c * G + kk)] = r[c];
        }
      }
      if ((a.update || a.hist) && act && cls == 1u) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) accF[c] += r[c];
      }
      if (a.sum_a && act && cls == 2u) {
        for_reporters(tie, [&](int m) {
#pragma unroll
          for (int c = 0; c < NCH; ++c)
            if (kv[c]) atomicAdd(&Al[(size_t)m * K + (c * G + kk)], r[c]);
        });
      }
      if (a.hist) {
        for_rounds(pi, [&](unsigned ym, unsigned x, unsigned) {
          if (x != 0u && act) {
            const double dx = (double)x;
            if (ym < hs_rows) {   // (LDS adds)
#pragma unroll
              for (int c = 0; c < NCH; ++c)
                if (kv[c] && GEN_ON(1)) atomicAdd(&Hs[(size_t)ym * K + (c * G + kk)], dx * r[c]);
            } else {
              double* hrow = Hl + (size_t)ym * K;
#pragma unroll
              for (int c = 0; c < NCH; ++c)
                if (kv[c] && GEN_ON(1)) atomicAdd(&hrow[c * G + kk], dx * r[c]);
            }
          }
        });
      }
      if (a.elbo) {
        double er[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) er[c] = kv[c] ? exp(r[c]) : 0.0;   // exp(rho), model.py:971
        for_rounds(pi, [&](unsigned ym, unsigned x, unsigned inr) {   // (every lane walks every round: the group sums are shuffles)
          const unsigned y = ym / (unsigned)Mp, m = ym - y * (unsigned)Mp;
          const double z2 = gnu_e * (double)y, gt = Gth[m];
          double il = 0.0;
#pragma unroll
          for (int c = 0; c < NCH; ++c)
            if (kv[c]) il += er[c] * (gt * Gla[c] + z2);
          const double inner = group_sum(il, G);
          if (kk == 0 && x != 0u && act) e_log += (double)x * log((inr ? inner : 0.0) + eps);   // eps alone outside R (model.py:990-994)
        });
        if (act) {
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            if (kv[c]) {
              e_lin += r[c] * lp[c] - r[c] * log(r[c] + eps) - r[c] * Ela[c] * Tt;   // model.py:1306-1313, 975-985
              if (Ql) e_q += r[c] * (double)qt;
            }
          }
        }
      }
    }
  }
  if (hs_rows) {   // the workgroup's sums of the low rows: one global add per cell it touched
    __syncthreads();
    for (unsigned q = tid; q < hs_rows * (unsigned)K; q += blockDim.x) {
      const double v = Hs[q];
      if (v != 0.0) atomicAdd(&Hl[q], v);
    }
  }
  if (a.update || a.hist) {   // rho summed over the all-ones mask rows: lanes of equal category across the wave's groups
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      double v = accF[c];
      for (int o2 = G; o2 < 64; o2 <<= 1) v += __shfl_xor(v, o2, 64);
      if (grp == 0 && kv[c] && v != 0.0) atomicAdd(&a.slotF[((size_t)l * NSLOT + (gb % NSLOT)) * K + (c * G + kk)], v);
    }
  }
  if (a.elbo) {
    const double v1 = wave_sum(e_lin), v2 = wave_sum(e_log), v3 = wave_sum(e_q);
    if (lane == 0) {
      double* out = a.slotR + (size_t)(blockIdx.x % NSLOT) * 4;
      atomicAdd(&out[1], v1); atomicAdd(&out[2], v2); atomicAdd(&out[3], v3);
    }
  }
}

// ------------------------------------------------------------------------------------------
// finalize: gamma, phi
// ------------------------------------------------------------------------------------------
// Weighted sums of one layer's H[y][m][k] by one workgroup: w = w1_k(m, y) from G_theta (gth), G_lambda (gla, LDS) and G_nu
// (mutuality off: 1; model.py:680-693):   by_m[m] += sum_{y,k} w H   (by_m != null)   or   by_k[k] += sum_{y,m} w H   (by_k != null).
// A row (y, m) is spread over G = min(64, 2^ceil(log2 K)) lanes, one category per lane, as in the pass: the sum over k is a
// shuffle inside the group and the groups of a wave add to different m; a lane keeps its categories' sums in registers.  (One
// thread per cell instead costs a same-address LDS add per lane and a division per cell.)  FU rows per group are requested at once.
// bi of nb: this workgroup's share of the rows (nb > 1: k_gen_hsum, sums in global memory); zero: the rows read are left zeroed.
__device__ void h_weighted_sums(double* __restrict__ Hl, const Geo& g, const double* gth, const double* gla, double gnu,
                                double* by_m, double* by_k, int bi = 0, int nb = 1, bool zero = false) {
  constexpr int FU = 8;
  const int K = g.K, Mp = g.Mp, M = g.M, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nwv = (int)blockDim.x >> 6;
  int lg = 1;
  while ((1 << lg) < K && lg < 6) ++lg;
  const int G = 1 << lg, kk = lane & (G - 1), grp = lane >> lg, TPW = 64 >> lg;
  const size_t rows = (size_t)g.Y * Mp;
  const int rstep = nb * nwv * TPW, r0 = (bi * nwv + wv) * TPW + grp;   // rows a trip of the layer's workgroups covers, this group's first
  int m0 = r0 % Mp, y0 = r0 / Mp;                     // (y, m) of this group's row, advanced by carries
  const int sm = rstep % Mp, sy = rstep / Mp;
  double acc[KGEN_MAX / 64];
#pragma unroll
  for (int c = 0; c < KGEN_MAX / 64; ++c) acc[c] = 0.0;
  for (size_t base = 0; base < rows; base += (size_t)rstep * FU) {
    int mu[FU], yu[FU];
#pragma unroll
    for (int u = 0; u < FU; ++u) {
      mu[u] = m0; yu[u] = y0;
      m0 += sm; const int cy = m0 >= Mp ? 1 : 0; m0 -= cy ? Mp : 0; y0 += sy + cy;
    }
#pragma unroll
    for (int c = 0; c < KGEN_MAX / 64; ++c) {
      if (c * G < K) {   // (K <= 64: one chunk)
        const int k = c * G + kk;
        double hv[FU];
#pragma unroll
        for (int u = 0; u < FU; ++u) {
          const size_t row = base + (size_t)u * rstep + (unsigned)r0;
          const bool on = row < rows && k < K;
          hv[u] = on ? Hl[row * K + k] : 0.0;
          if (zero && on) Hl[row * K + k] = 0.0;
        }
        const double gk = k < K ? gla[k] : 0.0;
#pragma unroll
        for (int u = 0; u < FU; ++u) {
          const int m = mu[u];
          double v = 0.0;
          if (hv[u] != 0.0 && m < M) v = (g.mut ? w1_of(gth[m] * gk, gnu * (double)yu[u]) : 1.0) * hv[u];
          if (by_m) {
            v = group_sum(v, G);
            if (kk == 0 && v != 0.0) atomicAdd(&by_m[m], v);
          } else {
            acc[c] += v;
          }
        }
      }
    }
  }
  if (by_k) {
#pragma unroll
    for (int c = 0; c < KGEN_MAX / 64; ++c) {
      if (c * G < K) {
        double v = acc[c];
        for (int o2 = G; o2 < 64; o2 <<= 1) v += __shfl_xor(v, o2, 64);
        if (grp == 0 && c * G + kk < K && v != 0.0) atomicAdd(&by_k[c * G + kk], v);
      }
    }
  }
}

// Large tables (one workgroup walking a layer's H is bound by memory latency): the sums by several workgroups per layer into
// global scratch, out[l][m] (by_k = 0, old G_theta: before k_fin_gamma_gen) or out[l][k] (by_k = 1, new G_theta: after it).
__global__ __launch_bounds__(1024) void k_gen_hsum(const double* __restrict__ par, double* Hg, double* out, int by_k, int zero, int nb, Geo g) {
  extern __shared__ double dyn[];   // G_lambda, K
  const int K = g.K, Mp = g.Mp, l = (int)blockIdx.x / nb, bi = (int)blockIdx.x - l * nb;
  const ParOff o = par_off(g.L, Mp, K);
  for (int k = threadIdx.x; k < K; k += (int)blockDim.x) dyn[k] = par[o.G_la + l * K + k];
  __syncthreads();
  double* Hl = Hg + (size_t)l * g.Y * Mp * K;
  const double* gth = par + o.G_th + (size_t)l * Mp;
  const double gnu = par[o.sc + SC_G_NU];
  if (by_k) h_weighted_sums(Hl, g, gth, dyn, gnu, nullptr, out + (size_t)l * K, bi, nb, zero != 0);
  else h_weighted_sums(Hl, g, gth, dyn, gnu, out + (size_t)l * Mp, nullptr, bi, nb, zero != 0);
}

// One workgroup per layer.  gamma_shp[m] = alpha + sum_{y,k} w1_k(m,y) H (old parameters; model.py:698-703, 832-859),
// gamma_rte[m] = beta + sum_k E[lambda_k] A[m,k] (model.py:704-718), then with the NEW theta: phi_rte[k] = beta + sum_m E[theta_m] A[m,k]
// (model.py:742-749) and phi_shp[k] = alpha + sum_{m,y} w1_k(m,y) H (model.py:731-733, 861-887); A[m,k] = the pass' sums of rho over
// the mask rows holding m (all-ones rows: slotF).  consume: H, slotF are left zeroed for the next pass; slotA always is.
__global__ __launch_bounds__(1024) void k_fin_gamma_gen(double* par, double* Hg, double* slotA, double* slotF, double* s1g, int s1_lds, int ext, int do_phi, int consume, Geo g) {
  extern __shared__ double dyn[];   // fk | ela_old | gla_old | prs | pss, K each; then (s1_lds) s1[Mp]
  const int K = g.K, Mp = g.Mp, M = g.M, l = blockIdx.x, tid = threadIdx.x, nthr = (int)blockDim.x, Wp = g.W * 64;
  double *fk = dyn, *ela_old = dyn + K, *gla_old = dyn + 2 * K, *prs = dyn + 3 * K, *pss = dyn + 4 * K;
  const ParOff o = par_off(g.L, Mp, K);
  double* s1 = s1_lds ? dyn + 5 * K : s1g + (size_t)l * Mp;   // (LDS while the reporters fit: the sums are then LDS adds)
  for (int k = tid; k < K; k += nthr) {
    double f = 0.0;
    for (int sl = 0; sl < NSLOT; ++sl) f += slotF[((size_t)l * NSLOT + sl) * K + k];
    fk[k] = f;
    ela_old[k] = par[o.p_shp + l * K + k] / par[o.p_rte + l * K + k];
    gla_old[k] = par[o.G_la + l * K + k];
    prs[k] = 0.0; pss[k] = 0.0;
  }
  if (!ext) for (int m = tid; m < Mp; m += nthr) s1[m] = 0.0;   // (ext: k_gen_hsum left the sums in s1g, and the phi sums follow in another one)
  // the slots of the mask sums folded into slot 0, all threads over the (m, k) cells (slotA null: no partial mask rows, all zero)
  double* A0 = slotA ? slotA + (size_t)l * NSLOT * Wp * K : nullptr;
  if (A0) {
    for (size_t q = tid; q < (size_t)M * K; q += nthr) {
      double v = A0[q];
      for (int sl = 1; sl < NSLOT; ++sl) { v += A0[(size_t)sl * Wp * K + q]; A0[(size_t)sl * Wp * K + q] = 0.0; }
      A0[q] = v;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  const double gnu = par[o.sc + SC_G_NU];
  const size_t hcs = (size_t)g.Y * Mp * K;
  double* Hl = Hg + (size_t)l * hcs;
  double* gth = par + o.G_th + (size_t)l * Mp;
  if (!ext) h_weighted_sums(Hl, g, gth, gla_old, gnu, s1, nullptr);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // no partial mask rows (A0 null): every reporter's rate sum is the same sum_k E[lambda_k] F_k, and phi's is F_k sum_m E[theta_m]
  __shared__ double red[16];
  __shared__ double s_c0, s_es;
  if (!A0) {
    double c0 = 0.0;
    for (int k = tid; k < K; k += nthr) c0 += ela_old[k] * fk[k];
    c0 = block_sum_n(c0, red);
    if (tid == 0) s_c0 = c0;
    __syncthreads();
  }
  auto finish_m = [&](int m, double rsum, bool write) -> double {   // gamma of reporter m from its two sums; returns E[theta_m]
    const size_t q = (size_t)l * Mp + m;
    const double shp = par[o.a_th + q] + ((s1_lds || ext) ? s1[m] : atomicAdd(&s1[m], 0.0));   // (global, this kernel's own adds: a device-scope read of what the atomics left at the memory side)
    const double rte = par[o.b_th + q] + rsum;
    const double e = shp / rte, lg = digamma_pos(shp) - log(rte);
    if (write) {
      if (ext) s1[m] = 0.0;   // (zero again for the next k_gen_hsum)
      par[o.g_shp + q] = shp; par[o.g_rte + q] = rte;
      par[o.E_th + q] = e; par[o.l_th + q] = lg; par[o.G_th + q] = exp(lg);
    }
    return e;
  };
  if (!A0) {
    double es = 0.0;
    for (int m = tid; m < M; m += nthr) es += finish_m(m, s_c0, true);
    es = block_sum_n(es, red);
    if (tid == 0) s_es = es;
    __syncthreads();
    for (int k = tid; k < K; k += nthr) prs[k] = fk[k] * s_es;
  } else {
    // a reporter's row A[m, :] over the G lanes of a group (as the rows of H above): its rate sum is a shuffle, phi's sums stay
    // in the lanes' registers until the end
    const int lane = tid & 63, wv = tid >> 6, nwv = nthr >> 6;
    int lg2 = 1;
    while ((1 << lg2) < K && lg2 < 6) ++lg2;
    const int G = 1 << lg2, kk = lane & (G - 1), grp = lane >> lg2, TPW = 64 >> lg2, mstep = nwv * TPW;
    double acc[KGEN_MAX / 64];
#pragma unroll
    for (int c = 0; c < KGEN_MAX / 64; ++c) acc[c] = 0.0;
    for (int mb = 0; mb < M; mb += mstep) {   // (uniform trip count: the group sums are shuffles)
      const int m = mb + wv * TPW + grp;
      const bool on = m < M;
      double ak[KGEN_MAX / 64], part = 0.0;
#pragma unroll
      for (int c = 0; c < KGEN_MAX / 64; ++c) {
        const int k = c * G + kk;
        ak[c] = 0.0;
        if (c * G < K && on && k < K) {
          ak[c] = fk[k] + A0[(size_t)m * K + k];
          A0[(size_t)m * K + k] = 0.0;   // (the slots are zero again for the next pass)
          part += ela_old[k] * ak[c];
        }
      }
      part = group_sum(part, G);
      const double e = on ? finish_m(m, part, kk == 0) : 0.0;
#pragma unroll
      for (int c = 0; c < KGEN_MAX / 64; ++c) acc[c] += e * ak[c];
    }
#pragma unroll
    for (int c = 0; c < KGEN_MAX / 64; ++c) {
      if (c * G < K) {
        double v = acc[c];
        for (int o2 = G; o2 < 64; o2 <<= 1) v += __shfl_xor(v, o2, 64);
        if (grp == 0 && c * G + kk < K) atomicAdd(&prs[c * G + kk], v);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // phi_shp's sums: H with the NEW G_theta and the old G_lambda (the cache refresh between the two updates, model.py:647);
  // mutuality off: Y = 1 and the weight is 1 (model.py:680)
  if (ext) {   // k_gen_hsum and k_fin_phi_gen finish phi
    for (int k = tid; k < K; k += nthr) par[o.p_rte_pend + l * K + k] = par[o.b_la + l * K + k] + prs[k];
    if (consume) for (int q = tid; q < NSLOT * K; q += nthr) slotF[(size_t)l * NSLOT * K + q] = 0.0;
    return;
  }
  if ((do_phi && g.mut) || !g.mut) {
    h_weighted_sums(Hl, g, gth, gla_old, gnu, nullptr, pss);
  }
  __syncthreads();
  for (int k = tid; k < K; k += nthr) {
    const int q = l * K + k;
    const double rte = par[o.b_la + q] + prs[k];
    if (g.mut && !do_phi) {
      par[o.p_rte_pend + q] = rte;   // the PHI sub-step commits (k_fin_phi_gen)
    } else {
      const double shp = par[o.a_la + q] + pss[k];
      par[o.p_shp + q] = shp; par[o.p_rte + q] = rte;
      if (g.mut) par[o.p_rte_pend + q] = rte;
      const double lg = digamma_pos(shp) - log(rte);
      par[o.E_la + q] = shp / rte; par[o.l_la + q] = lg; par[o.G_la + q] = exp(lg);
    }
  }
  if (consume) {
    for (size_t q = tid; q < hcs; q += nthr) Hl[q] = 0.0;
    for (int q = tid; q < NSLOT * K; q += nthr) slotF[(size_t)l * NSLOT * K + q] = 0.0;
  }
}

// s2g != null: the sums come from k_gen_hsum (and are zeroed here for the next one).
__global__ __launch_bounds__(1024) void k_fin_phi_gen(double* par, double* Hg, double* s2g, Geo g) {
  extern __shared__ double dyn[];   // gla_old | pss, K each
  const int K = g.K, Mp = g.Mp, M = g.M, l = blockIdx.x, tid = threadIdx.x, nthr = (int)blockDim.x;
  double *gla_old = dyn, *pss = dyn + K;
  const ParOff o = par_off(g.L, Mp, K);
  for (int k = tid; k < K; k += nthr) { gla_old[k] = par[o.G_la + l * K + k]; pss[k] = 0.0; }
  __syncthreads();
  const double gnu = par[o.sc + SC_G_NU];
  const size_t hcs = (size_t)g.Y * Mp * K;
  double* Hl = Hg + (size_t)l * hcs;
  const double* gth = par + o.G_th + (size_t)l * Mp;   // new
  if (s2g) {
    for (int k = tid; k < K; k += nthr) { pss[k] = s2g[l * K + k]; s2g[l * K + k] = 0.0; }
  } else {
    h_weighted_sums(Hl, g, gth, gla_old, gnu, nullptr, pss);
  }
  __syncthreads();
  for (int k = tid; k < K; k += nthr) {
    const int q = l * K + k;
    const double shp = par[o.a_la + q] + pss[k], rte = par[o.p_rte_pend + q];
    par[o.p_shp + q] = shp; par[o.p_rte + q] = rte;
    const double lg = digamma_pos(shp) - log(rte);
    par[o.E_la + q] = shp / rte; par[o.l_la + q] = lg; par[o.G_la + q] = exp(lg);
  }
}

// Posterior samples (vmr_sample, k_sample in vimure_hip.hip) for K > KMAX: the trial counts live in LDS, one column per thread.
__global__ __launch_bounds__(64) void k_sample_gen(const double* __restrict__ rho, uint8_t* __restrict__ out, size_t ties, int K, int n_trials,
                                                   unsigned long long seed, const unsigned* __restrict__ perm, size_t T, size_t NS) {
  extern __shared__ unsigned cnt_s[];   // [K][64]
  unsigned* cnt = cnt_s + threadIdx.x;
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < ties; q += (size_t)gridDim.x * blockDim.x) {
    const double* r = rho + q * K;
    size_t t = q;
    if (perm) { const size_t l = q / T, pos = q - l * T; t = l * T + perm[l * NS * 64 + pos]; }
    for (int k = 0; k < K; ++k) cnt[k * 64] = 0u;
    for (int n = 0; n < n_trials; n += 2) {
      unsigned c[4] = {(unsigned)t, (unsigned)((unsigned long long)t >> 32), (unsigned)(n >> 1), 0u};
      philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
      for (int i = 0; i < 2; ++i) {
        if (n + i < n_trials) {
          const double u = ((double)(c[2 * i] >> 5) * 67108864.0 + (double)(c[2 * i + 1] >> 6)) * (1.0 / 9007199254740992.0);
          int sel = 0;
          double acc = r[0];
          for (int k = 1; k < K; ++k) { if (u >= acc) sel = k; acc += r[k]; }
          cnt[sel * 64] += 1u;
        }
      }
    }
    int best = 0;
    for (int k = 1; k < K; ++k) if (cnt[k * 64] > cnt[best * 64]) best = k;
    out[t] = (uint8_t)best;
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static int gen_pass(vmr_ctx* h, int update, int elbo, int hist, int sum_a) {
  const Geo& g = h->g;
  const long long NS = ((long long)g.N * g.N + 63) / 64;
  GenArgs a{h->E, h->EX, h->rs, h->ebase, h->perm, h->cls_p, h->Qt_p, h->Rb, h->rq, h->Rm, h->rbase, h->rho, h->logpr, h->par,
            h->slotR, h->Hg, h->slotF, h->slotA, 1, update, elbo, hist, sum_a, 0,
            reinterpret_cast<unsigned long long*>(h->gen_s1 + (size_t)g.L * (g.Mp + g.K)), 0, 0};
  HIPCHK(h, hipMemsetAsync(a.ctr, 0, (size_t)g.L * 8, h->stream));
#ifdef GEN_DEBUG
  if (const char* d = getenv("VMR_GEN_DBG")) a.dbg = atoi(d);
#endif
  int lg = 1;
  while ((1 << lg) < g.K && lg < 6) ++lg;
  const int nw = 8, sw = std::min(nw, 1 << lg), nsb = nw / sw;   // (as the kernel derives them: waves sharing a step, steps per workgroup and trip)
  // One workgroup per CU is what the registers allow anyway (8 waves of ~164 VGPRs), so a workgroup walks its share of the steps in
  // a loop -- and can sum the rows of H of the small mirror counts in LDS on the way.
  a.Gl = (int)std::max<long long>(1, std::min<long long>((NS + nsb - 1) / nsb, std::max<long long>(1, (long long)h->ncu / g.L)));
  const dim3 grid((unsigned)(g.L * a.Gl)), blk(nw * 64);
  const size_t tab = (size_t)3 * g.Mp * 8;
  a.lds_tab = tab <= 48 * 1024 ? 1 : 0;   // (M <= 2048; wider reporter dimensions read the tables through L2)
  size_t sm = a.lds_tab ? tab : 0;
  if (hist) {
    const size_t row = (size_t)g.Mp * g.K * 8, budget = (size_t)128 * 1024 - sm;   // (160 KB of LDS per CU, one workgroup on it)
    a.YL = (int)std::min<size_t>((size_t)g.Y, budget / row);
    if (getenv("VMR_GEN_NO_LDS_H")) a.YL = 0;
    sm += (size_t)a.YL * row;
  }
  const void* fn = g.K <= 64 ? reinterpret_cast<const void*>(k_sweep_gen<1>) : g.K <= 128 ? reinterpret_cast<const void*>(k_sweep_gen<2>) : reinterpret_cast<const void*>(k_sweep_gen<4>);
  if (sm > 48 * 1024) HIPCHK(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
  if (g.K <= 64) hipLaunchKernelGGL(k_sweep_gen<1>, grid, blk, sm, h->stream, a, g);
  else if (g.K <= 128) hipLaunchKernelGGL(k_sweep_gen<2>, grid, blk, sm, h->stream, a, g);
  else hipLaunchKernelGGL(k_sweep_gen<4>, grid, blk, sm, h->stream, a, g);
  HIPCHK(h, hipGetLastError());
  return VMR_OK;
}
static size_t gen_h_bytes(const Geo& g) { return (size_t)g.L * g.Y * g.Mp * g.K * 8; }

int gen_hist(vmr_ctx* h) {
  const Geo& g = h->g;
  const int sum_a = h->n_partial > 0 ? 1 : 0;
  HIPCHK(h, hipMemsetAsync(h->Hg, 0, gen_h_bytes(g), h->stream));
  HIPCHK(h, hipMemsetAsync(h->slotF, 0, (size_t)g.L * NSLOT * g.K * 8, h->stream));
  if (sum_a && !h->a_zero) HIPCHK(h, hipMemsetAsync(h->slotA, 0, (size_t)g.L * NSLOT * g.W * 64 * g.K * 8, h->stream));
  {
    Prof p(h, VMR_KERNEL_GAMMA_COUNTS);
    int rc = gen_pass(h, 0, 0, 1, sum_a);
    if (rc) return rc;
  }
  h->h_valid = true; h->h_zero = false; h->h_reduced = true; h->f_valid = true; h->a_valid = true; h->a_zero = !sum_a;
  return VMR_OK;
}

// Workgroups per layer for the sums over H: one while the table is small (the finalize kernel then does everything itself), else
// one per 128 rows up to the chip (k_gen_hsum).
static int gen_hsum_blocks(const vmr_ctx* h) {
  const Geo& g = h->g;
  const size_t rows = (size_t)g.Y * g.Mp, cells = rows * g.K;
  const char* ev = getenv("VMR_GEN_HSUM");   // experiments and tests: force the number
  if (ev && atoi(ev) > 0) return atoi(ev);
  if (cells < 32768) return 1;
  int lg = 1;
  while ((1 << lg) < g.K && lg < 6) ++lg;
  const size_t per = (size_t)16 * (64 >> lg) * 8;   // rows of one trip of a 1024-thread workgroup
  return (int)std::max<size_t>(2, std::min<size_t>((rows + per - 1) / per, std::max<size_t>(2, (size_t)h->ncu * 2 / g.L)));
}

int gen_gamma(vmr_ctx* h, bool with_phi) {
  const Geo& g = h->g;
  if (!h->h_valid || !h->f_valid || !h->a_valid) { int rc = gen_hist(h); if (rc) return rc; }
  {
    Prof p(h, VMR_KERNEL_FINALIZE);
    const int nb = gen_hsum_blocks(h), ext = nb > 1 ? 1 : 0;
    const int s1_lds = (!ext && (size_t)(5 * g.K + g.Mp) * 8 <= 48 * 1024) ? 1 : 0;   // (within the default dynamic-LDS limit, the K-sized arrays included)
    double* slotA = (h->n_partial > 0 || !h->a_zero) ? h->slotA : nullptr;
    if (ext) hipLaunchKernelGGL(k_gen_hsum, dim3(g.L * nb), dim3(1024), (size_t)g.K * 8, h->stream, h->par, h->Hg, h->gen_s1, 0, 0, nb, g);
    hipLaunchKernelGGL(k_fin_gamma_gen, dim3(g.L), dim3(1024), (size_t)5 * g.K * 8 + (s1_lds ? (size_t)g.Mp * 8 : 0), h->stream, h->par, h->Hg, slotA,
                       h->slotF, h->gen_s1, s1_lds, ext, with_phi ? 1 : 0, with_phi ? 1 : 0, g);
    if (ext && (with_phi || !g.mut)) {   // phi's sums with the new theta, then its commit (mutuality off: always, model.py:680)
      double* s2 = h->gen_s1 + (size_t)g.L * g.Mp;
      hipLaunchKernelGGL(k_gen_hsum, dim3(g.L * nb), dim3(1024), (size_t)g.K * 8, h->stream, h->par, h->Hg, s2, 1, with_phi ? 1 : 0, nb, g);
      hipLaunchKernelGGL(k_fin_phi_gen, dim3(g.L), dim3(1024), (size_t)2 * g.K * 8, h->stream, h->par, h->Hg, s2, g);
    }
  }
  HIPCHK(h, hipGetLastError());
  h->a_valid = false; h->a_zero = true;
  if (with_phi) { h->h_valid = false; h->f_valid = false; h->h_zero = true; }
  return VMR_OK;
}

int gen_phi(vmr_ctx* h) {
  const Geo& g = h->g;
  if (!g.mut) return VMR_OK;   // committed by k_fin_gamma_gen
  if (!h->h_valid) { int rc = gen_hist(h); if (rc) return rc; }
  {
    Prof p(h, VMR_KERNEL_FINALIZE);
    const int nb = gen_hsum_blocks(h);
    double* s2 = nb > 1 ? h->gen_s1 + (size_t)g.L * g.Mp : nullptr;
    if (s2) hipLaunchKernelGGL(k_gen_hsum, dim3(g.L * nb), dim3(1024), (size_t)g.K * 8, h->stream, h->par, h->Hg, s2, 1, 0, nb, g);
    hipLaunchKernelGGL(k_fin_phi_gen, dim3(g.L), dim3(1024), (size_t)2 * g.K * 8, h->stream, h->par, h->Hg, s2, g);
  }
  HIPCHK(h, hipGetLastError());
  return VMR_OK;
}

int gen_rho(vmr_ctx* h, int mode, bool commit_nu, bool raw_nu, gen_fin_rho_fn fin_rho) {
  const Geo& g = h->g;
  const int upd = mode != 2, sum_a = (upd && h->n_partial > 0) ? 1 : 0;
  if (upd) {
    if (!h->h_zero) {
      HIPCHK(h, hipMemsetAsync(h->slotF, 0, (size_t)g.L * NSLOT * g.K * 8, h->stream));
      HIPCHK(h, hipMemsetAsync(h->Hg, 0, gen_h_bytes(g), h->stream));
    }
    h->h_zero = false;
    if (sum_a && !h->a_zero) HIPCHK(h, hipMemsetAsync(h->slotA, 0, (size_t)g.L * NSLOT * g.W * 64 * g.K * 8, h->stream));
    if (sum_a) h->a_zero = false;
  }
  {
    Prof p(h, mode == 2 ? VMR_KERNEL_ELBO : mode == 1 ? VMR_KERNEL_RHO_ELBO : VMR_KERNEL_RHO);
    int rc = gen_pass(h, upd, mode != 0, upd, sum_a);
    if (rc) return rc;
  }
  if (upd) { h->f_valid = true; h->h_valid = true; h->h_reduced = true; h->a_valid = true; }
  if (mode == 2) return fin_rho(h, 0, 1, 1);
  const bool nu = g.mut && (commit_nu || raw_nu);
  if (mode != 0 || nu) return fin_rho(h, (commit_nu && g.mut) ? 1 : 0, mode != 0 ? 1 : 0, g.mut ? 0 : 1);
  return VMR_OK;
}

int gen_sample(vmr_ctx* h, unsigned long long seed, int n_trials, uint8_t* out_dev) {
  const Geo& g = h->g;
  const size_t T_ = (size_t)g.N * g.N, ties = (size_t)g.L * T_, smem = (size_t)g.K * 64 * 4;
  if (smem > 48 * 1024) HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_sample_gen), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(k_sample_gen, dim3((unsigned)std::min<size_t>(4096, (ties + 63) / 64)), dim3(64), smem, h->stream, h->rho, out_dev, ties, g.K,
                     n_trials, seed, h->perm, T_, (T_ + 63) / 64);
  HIPCHK(h, hipGetLastError());
  return VMR_OK;
}
