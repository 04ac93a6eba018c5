// sweep_sl.hip -- ONE pass over the sorted report lists per CAVI sweep: the rho update (model.py:763-818, 889-923), the
// sufficient statistics H of the new rho (what gamma / phi / nu are finished from: model.py:698-761, 820-887) and the ELBO's
// data terms (model.py:948-995).  Compiled once per number of categories (-DVMR_K=2..8), one object each.
//
// A wave takes steps of 64 consecutive sorted positions, one tie per lane (layout: sweep_sl.h).  Ties are sorted by their
// number of reports, so every lane of a step walks the same number of rounds R, and R never grows from one step of a wave to
// its next.  A step is
//   prefetch   the next step's per-tie values (log prior or rho, mask class) and its rounds are requested at the top of the
//              step, after ONE wait for this step's own (requested a step ago); the slot range and the highest table level of
//              the step after next ride in three lanes of a vector register (a scalar load would be waited for with
//              lgkmcnt(0): LDS queue drained)
//   walk 1     U_k = sum_r x_r F[y_r, m_r, k]: the table reads back to back, then the FMAs
//   update     rho = exp(log prior + U - T E[lambda]) normalised where the sum is positive: raw exp, as model.py:807
//   walk 2     H[y_r, m_r, k] += x_r rho_k for k >= 1 (LDS float atomics; H_0 is rebuilt from the constant sum of x); the ELBO
//              variant takes its logarithms here
// Consecutive steps of a wave with the same R run in a loop compiled for that R (R = 0..SL_PF): straight-line code with no
// per-report branch, no scatter and no workgroup barrier, and a back edge at which the compiler knows every request's age.  The populous levels (mirror counts) of F and H live in
// LDS, shared by all waves of a workgroup; a step whose reports reach a level beyond them (known per step from the list
// build) takes the general code -- global reads and atomics -- and steps of more than SL_PF rounds stream the further ones
// through a register ring.
#include "sweep_sl.h"

#ifndef VMR_K
#error "compile with -DVMR_K=<number of categories>"
#endif
#ifndef SL_WPE   // experiments: override the waves per SIMD the K = 2 update variants are compiled for
#define SL_WPE sl_wpe(2, false, true)
#endif

template <int K>
struct StepIn {          // what is prefetched for one step
  double v[K];           // log prior (UPDATE or ELBO)
  double w[K];           // current rho (not UPDATE)
  unsigned cls, qt, tie;
  unsigned x0;           // (SlArgs::x0p) the tie's summed counts at mirror count 0
  unsigned e[sl_pf(K)];
};

template <int R> struct RC { static constexpr int value = R; };
// p + off bytes: a wave-uniform pointer plus a 32-bit per-lane offset (the global_load saddr + voffset form: no 64-bit address
// arithmetic per lane)
template <class V> __device__ __forceinline__ const V* at_bytes(const V* p, unsigned off) {
  return reinterpret_cast<const V*>(reinterpret_cast<const char*>(p) + off);
}
template <class V> __device__ __forceinline__ V* at_bytes(V* p, unsigned off) {
  return reinterpret_cast<V*>(reinterpret_cast<char*>(p) + off);
}
// 1 / d for d in [1, 1e305): v_rcp_f64 and two Newton steps (the error of the first, 2e-15, squared); the IEEE divide is 11
// dependent instructions
__device__ __forceinline__ double rcp_nr2(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  return fma(fma(-d, r, 1.0), r, r);
}
#define SL_INL __attribute__((always_inline))
// table rows read back to back (one LDS latency per batch): as many as keep a batch within 16 registers
#ifndef SL_BATCH2
#define SL_BATCH2 8   // (4: 0.191 ms, 8: 0.188 ms per config-3 launch)
#endif
template <int K> struct Batch { static constexpr int value = K <= 2 ? SL_BATCH2 : (K <= 4 ? 4 : 2); };

// The sweep over one handle's lists.  bx / gx: this workgroup's index in, and the size of, the handle's grid -- the whole launch
// (k_sweep_sl) or a handle's share of a launch that serves many small handles (k_sweep_sl_b).
// STORE = false (update without ELBO only): the new rho is used -- for the statistics H, the mask sums, nu -- and NOT written.  A rho
// that no one can read before the next sweep overwrites it is a dead store of a third of the pass' bytes: vmr_step / the fit loops
// write rho on the last sweep of a call and on ELBO sweeps only (vmr_ctx::rho_stale, ensure_rho in vimure_hip.hip).
// DET (VMR_DETERMINISTIC=1, SlArgs::det): every sum whose order would vary from run to run -- which wave takes which step, which
// atomic lands first -- is an INTEGER sum in fixed point: the LDS statistics and mask sums (64-bit LDS adds), the per-lane sums over
// ties (rho over all-ones rows, the ELBO partials), the cross-workgroup shadows.  Integer adds commute exactly, so the workgroups
// keep their 16 waves and their tickets (rounds 2-3 ran this mode with ONE wave per workgroup: 4.5 times slower).
template <int K, bool UPDATE, bool ELBO, bool ALLFULL, bool STORE = true, bool DET = false>
__device__ __forceinline__ void sweep_body(const SlArgs& a, const Geo& g, const unsigned bx, const unsigned gx) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int PFK = sl_pf(K);   // rounds prefetched one step ahead
  constexpr bool LV0 = !UPDATE && !ELBO && !DET;   // the variant that takes level-0 rounds without LDS adds (SlArgs::h0s): the statistics pass
  constexpr bool LV0R = UPDATE && K >= 3;           // the variants whose general body reads one table value per level-0 report (SlArgs::lv0r);
                                                    // not K = 2: the few registers it takes cost BASELINE config 3's variants 1 %
  const int tid = threadIdx.x, lane = tid & 63, nthr = (int)blockDim.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
  const int Mp = g.Mp;
  const unsigned ytm = UPDATE ? (unsigned)a.yt * (unsigned)Mp : 0u;     // rows (y, m) of F held in LDS
  const unsigned hcm = a.do_hist ? (unsigned)a.hc * (unsigned)Mp : 0u;  // rows of H held in LDS
  size_t off = 0;
  double* F = reinterpret_cast<double*>(smem + off); off += (size_t)ytm * K * 8;            // [yt][Mp][K]
  const int nHc = (int)hcm * K;
  double* Hc = reinterpret_cast<double*>(smem + off); off += (size_t)nHc * 8;                // [K][hc][Mp]: plane 0 = deficits, k = category k
  // per-reporter tables: G_theta (the factor table, the ELBO's inner sums, the nu weights) and E[log theta] (the factor table)
  double* Gth = reinterpret_cast<double*>(smem + off); off += (size_t)Mp * 8;
  double* Lth = reinterpret_cast<double*>(smem + off); off += UPDATE ? (size_t)Mp * 8 : 0;
  double* As = reinterpret_cast<double*>(smem + off); off += a.sum_a ? (size_t)Mp * K * 8 : 0;   // [Mp][K]
  double* wsum = reinterpret_cast<double*>(smem + off); off += (size_t)g.W * 8;
  double* red = reinterpret_cast<double*>(smem + off); off += 16 * 8;
  double* xt = reinterpret_cast<double*>(smem + off); off += 64 * 8;
  double* lt = reinterpret_cast<double*>(smem + off); off += ELBO ? 256 * 8 : 0;
  double* EthL = reinterpret_cast<double*>(smem + off); off += a.rq ? (size_t)Mp * 8 : 0;   // E[theta] for the mask lists
  const ParOff o = par_off(g.L, g.Mp, g.K);
  const int l = (int)bx / a.Gl, gb = (int)bx - l * a.Gl;
#ifdef SL_DEBUG
  unsigned long long* dbt = a.dbg_t ? a.dbg_t + ((size_t)bx * nw + wv) * 8 : nullptr;
  if (dbt && lane == 0) dbt[0] = wall_clock64();
#endif
  const size_t T = (size_t)g.N * g.N;
  const long long NS = (long long)((T + 63) / 64);
  const double gnu = a.par[o.sc + ((UPDATE || a.elbo_cur) ? SC_G_NU : SC_G_NU_STALE)];   // stand-alone ELBO: the stale one (model.py:970)
  // (lv0_off: some geometric expectation of this layer has underflowed to 0 -- the weight of a report at mirror count 0 is then the
  // reference's 0 / 0 -> 0 (model.py:685-696), not 1: no level-0 shortcut in this launch)
  bool gz = false;
  for (int m = tid; m < Mp; m += nthr) {
    const double gv = a.par[o.G_th + (size_t)l * Mp + m];
    Gth[m] = gv;
    if (LV0 || LV0R) gz = gz || (m < g.M && gv == 0.0);
    if (UPDATE) Lth[m] = a.par[o.l_th + (size_t)l * Mp + m];
  }
  for (int q = tid; q < nHc; q += nthr) Hc[q] = 0.0;
  if (a.sum_a) for (int q = tid; q < Mp * K; q += nthr) As[q] = 0.0;
  if (a.rq) for (int m = tid; m < Mp; m += nthr) EthL[m] = m < g.M ? a.par[o.E_th + (size_t)l * Mp + m] : 0.0;
  if (UPDATE || ELBO) sp_math_tables(xt, lt, tid, nthr, ELBO);
  const double* lut = a.lutg + (size_t)l * g.W * 256;
  for (int w = tid; w < g.W; w += nthr) {
    double v = 0.0;
    for (int n = 0; n < 16; ++n) v += lut[(w * 16 + n) * 16 + 15];
    wsum[w] = v;
  }
  double Ela[K], Gla[K], Lla[K];
#pragma unroll
  for (int k = 0; k < K; ++k) { Ela[k] = a.par[o.E_la + l * K + k]; Gla[k] = a.par[o.G_la + l * K + k]; Lla[k] = a.par[o.l_la + l * K + k]; if (LV0 || LV0R) gz = gz || Gla[k] == 0.0; }
  const double eps = g.eps;
  // fixed point of the deterministic mode: v * 2^sh rounded to nearest through the 1.5 * 2^52 trick (|v| 2^sh < 2^51: the host caps
  // det_sh at 40 and det_shr at 34 -- counts hold 11 bits, ELBO terms 16 -- and keeps DET_SH_A = 30 for sums of rho)
  const double sc_h = DET ? __builtin_amdgcn_ldexp(1.0, g.det_sh) : 0.0, sc_r = DET ? __builtin_amdgcn_ldexp(1.0, g.det_shr) : 0.0;
  const double sc_a = DET ? __builtin_amdgcn_ldexp(1.0, DET_SH_A) : 0.0;
  auto fxm = [](double v, double scale) SL_INL -> unsigned long long {
    return (unsigned long long)(__double_as_longlong(fma(v, scale, 6755399441055744.0)) - 0x4338000000000000ll);
  };
  auto lds_add = [&](double* p, double v, double scale) SL_INL {   // an LDS cell: a double, or (DET) a 64-bit fixed-point integer
    if (DET) atomicAdd(reinterpret_cast<unsigned long long*>(p), fxm(v, scale)); else atomicAdd(p, v);
  };
  // The level y = row / Mp of a table row (< 2^20 rows, Mp <= 8192) by ONE single-precision multiply-add and a conversion: fl((row + 1/2) / Mp)
  // is off by less than row 2^-23 / Mp < 1 / (2 Mp), the distance of (row + 1/2) / Mp from the next integer (checked for every Mp and
  // every boundary row on the host).  (A multiply, a conversion and two corrections before: five vector instructions more per report.)
  const float rcp_mp = 1.0f / (float)Mp, hrcp_mp = 0.5f * rcp_mp;
  auto row_level = [&](unsigned ym) SL_INL -> unsigned { return (unsigned)__builtin_fmaf((float)ym, rcp_mp, hrcp_mp); };
  const double lp0_0 = log(1.0 + eps), lp0_k = log(eps);   // the log prior of a tie nobody reported on (the values k_init_rho_pos stored)
  double e_lin = 0.0, e_q = 0.0, e_log = 0.0;
  double accF[K], acc0[K];   // sums over this lane's ties: rho over all-ones mask rows; (SlArgs::h0s) rho_k times the tie's counts in level-0 rounds
#pragma unroll
  for (int k = 0; k < K; ++k) { accF[k] = 0.0; acc0[k] = 0.0; }
  // (DET: the same sums as integers -- a lane's ties depend on the tickets its wave drew)
  unsigned long long ie_lin = 0ull, ie_q = 0ull, ie_log = 0ull, iaccF[K];
#pragma unroll
  for (int k = 0; k < K; ++k) iaccF[k] = 0ull;
  double* Hl = a.Hg + ((size_t)l * NH + (gb % NH)) * g.Y * Mp * K;
  // deterministic mode: the integer shadows (SlArgs::det), located where they are used (nothing of this lives across the step loop)
  auto det_H = [&]() SL_INL { return a.det + (size_t)l * g.Y * Mp * K; };
  auto det_A = [&]() SL_INL { return a.det + (size_t)g.L * g.Y * Mp * K + (size_t)l * g.W * 64 * K; };
  auto det_F = [&]() SL_INL { return a.det + (size_t)g.L * g.Y * Mp * K + (size_t)g.L * g.W * 64 * K + (size_t)l * K; };
  auto det_R = [&]() SL_INL { return a.det + (size_t)g.L * g.Y * Mp * K + (size_t)g.L * g.W * 64 * K + (size_t)g.L * K; };   // [4], then the nu share
  const unsigned* rsl = a.rs + (size_t)l * (NS + 1);
  const unsigned* syl = a.sy + (size_t)l * NS;
  const unsigned* El = a.E + a.ebase[l];
  const unsigned* pl = (a.rm2 ? a.rm2 : a.perm) + (size_t)l * NS * 64;   // (the `tie` slot of a step: the tie, or its packed list)
  const uint8_t* cl = ALLFULL ? nullptr : a.cls + (size_t)l * T;
  const unsigned* Ql = a.Qt ? a.Qt + (size_t)l * T : nullptr;
  const uint64_t* Rl = a.Rb ? a.Rb + (size_t)l * T * g.W : nullptr;
  double* rl = a.rho + (size_t)l * T * K;
  double* rho_slack = a.rho + (size_t)g.L * T * K;   // 64 rows behind the last layer (vmr_create)
  const double* lpl = a.logpr + (size_t)l * T * K;
  const unsigned* rql = a.rq ? a.rq + (size_t)l * (T + 1) : nullptr;
  const unsigned short* Rml = a.rq ? a.Rm + a.rbase[l] : nullptr;
  // a step is "far" when one of its reports lies in a level beyond the LDS copies this launch holds: it takes the general body,
  // which decides per group of reports (lim1 / lim2: first level beyond the copies of F / H, none when every level is held)
  const unsigned lim_y = UPDATE ? (a.do_hist ? (unsigned)min(a.yt, a.hc) : (unsigned)a.yt) : (a.do_hist ? (unsigned)a.hc : 0xffffffffu);
  const unsigned lim1 = (UPDATE && a.yt < g.Y) ? (unsigned)a.yt : 0xffffffffu;
  const unsigned lim2 = (a.do_hist && a.hc < g.Y) ? (unsigned)a.hc : 0xffffffffu;

  // Steps of this workgroup: gb, gb + Gl, gb + 2 Gl, ... -- every workgroup sees the whole range of the sorted order -- handed to
  // its waves in that order by a ticket counter in LDS: the order is by falling number of rounds, so a wave that drew a long step
  // draws fewer.  (A fixed share per wave left a few waves with all the longest steps: the ties of BASELINE config 5's 1.9 % true
  // edges carry 1000 reports against 19 for the rest, and an eighth of the waves ran five times as long as the others.)  The
  // first two tickets of a wave are fixed -- wv and nw + wv -- so that its first loads go out before the tables are built.
  const long long Gl_ = a.Gl;
  unsigned* tick = reinterpret_cast<unsigned*>(smem + off); off += 16;
  if (tid == 0) { tick[0] = 2u * (unsigned)nw; tick[1] = 0u; }   // ([1]: some step of this workgroup added a deficit)
  auto step_of = [&](unsigned t) SL_INL -> long long { return (long long)gb + (long long)t * Gl_; };
  // (deterministic mode: a workgroup is one wave, so the counter hands it 0, 1, 2, ... -- a fixed share, in order)
  auto draw = [&]() SL_INL -> long long {   // this wave's next ticket (one LDS atomic by lane 0)
    unsigned t = 0u;
    if (lane == 0) t = atomicAdd(tick, 1u);
    return step_of((unsigned)__builtin_amdgcn_readfirstlane((int)t));
  };
  long long s = step_of((unsigned)wv), sn = step_of((unsigned)(nw + wv));   // the current step and the next
  unsigned ea = 0;    // first slot of the current step
  int R = 0;          // its rounds
  unsigned ymax = 0;  // its sy word: the highest mirror-count level (low half); Geo::farl: the leading rounds that may hold reports of the levels beyond the LDS ones (high half)
  unsigned rgv = 0;   // lanes 0..2: rs[s2], rs[s2 + 1], sy[s2] of the NEXT step (loaded one step earlier)
  StepIn<K> P;        // the prefetched step

  const unsigned lane4 = (unsigned)lane * 4u, laneK8 = (unsigned)lane * (unsigned)(K * 8);
  const unsigned T32 = (unsigned)T;   // (< 2^31: checked when the lists are built)
  // (every lane loads: no mask, no branch around a load -- the compiler counts the step's requests exactly)
  auto fetch_range = [&](long long st) SL_INL {
    const unsigned ra = rsl[st + (lane < 1 ? 0 : 1)], ry = syl[st];
    rgv = lane == 2 ? ry : ra;
  };
  // loads of one step's per-tie values: a scalar base per step plus a constant per-lane offset, no masks (the arrays carry
  // 64 rows of slack; positions past the last tie read them and are switched off through `cls`) ...
  auto fetch_tie = [&](StepIn<K>& d, long long st, const bool lp_const = false) SL_INL {   // lp_const: the step's log prior is the one-hot one (SlArgs::lp0)
    const size_t row0 = (size_t)st * 64;
    const bool ok = (unsigned)row0 + (unsigned)lane < T32;
    d.qt = 0u; d.tie = 0u;
    if (ALLFULL) d.cls = ok ? 1u : 0u;
    else {
      const unsigned c = (unsigned)*at_bytes(cl + row0, (unsigned)lane);
      d.cls = ok ? c : 0u;
      d.tie = *at_bytes(pl + row0, lane4);   // (partial mask rows are found by tie)
    }
    if (ELBO && Ql) d.qt = *at_bytes(Ql + row0, lane4);
    d.x0 = 0u;
    if (LV0 && a.x0p) d.x0 = *at_bytes(a.x0p + (size_t)l * NS * 64 + row0, lane4);
#pragma unroll
    for (int k = 0; k < K; ++k) { d.v[k] = 0.0; d.w[k] = 0.0; }
    if (UPDATE || ELBO) {
      if (lp_const) {
#pragma unroll
        for (int k = 0; k < K; ++k) d.v[k] = k == 0 ? lp0_0 : lp0_k;
      } else load_k<K>(at_bytes(lpl + row0 * K, laneK8), d.v);
    }
    if (!UPDATE && a.do_hist != 2) load_k<K>(at_bytes(static_cast<const double*>(rl) + row0 * K, laneK8), d.w);
  };
  // ... and of its first CNT rounds (rounds it does not have read later steps' slots, unused; in bounds: SL_SLACK)
  auto fetch_ent = [&](StepIn<K>& d, unsigned base, auto cntc) SL_INL {
    constexpr int CNT = decltype(cntc)::value;
    const unsigned* pe = El + base;
#pragma unroll
    for (int j = 0; j < CNT; ++j) d.e[j] = *at_bytes(pe + j * 64, lane4);
  };

  if (s < NS) {   // prologue: this wave's first step
    ea = rsl[s];
    R = (int)((rsl[s + 1] - ea) >> 6);
    ymax = syl[s];
    fetch_range(sn < NS ? sn : s);
    fetch_tie(P, s);
    fetch_ent(P, ea, RC<PFK>{});
  }
#ifdef SL_DEBUG
  if (dbt && lane == 0) dbt[4] = wall_clock64();
#endif
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the first step finds nothing of its own outstanding (see the wait in `body`)
  bool lv0_off = false;
  if (LV0 || LV0R) lv0_off = __syncthreads_or(gz ? 1 : 0) != 0;   // per-reporter tables (the barrier), and whether one of them holds a zero
  else __syncthreads();
#ifdef SL_DEBUG
  if (dbt && lane == 0) dbt[5] = wall_clock64();
#endif
  // The factor table of the rho update, F[y][m][k] = (E log theta_m + E log lambda_k) w1_k(m, y) (model.py:685-693, 911-921), for
  // the levels this launch keeps in LDS: a few divides per thread from the per-reporter tables -- no global table, no kernel
  // that builds one.  Reports of levels beyond take the same formula on the fly (f_far).
  const double gnu_f = a.par[o.sc + (a.nu_stale ? SC_G_NU_STALE : SC_G_NU)];   // (nu_stale: rho of the LAST sweep once more, ensure_rho)
  if (UPDATE) {
    for (int q = tid; q < (int)ytm; q += nthr) {
      const int y = q / Mp, m = q - y * Mp;
#pragma unroll
      for (int k = 0; k < K; ++k) F[q * K + k] = (m < g.M) ? f_entry(g.mut, Lth[m], Gth[m], Lla[k], Gla[k], gnu_f, y) : 0.0;
    }
    __syncthreads();
  }
  double Tfull = 0.0;
  for (int w = 0; w < g.W; ++w) Tfull += wsum[w];
#ifdef SL_DEBUG
  if (dbt && lane == 0) dbt[1] = wall_clock64();
#endif

  // ---- pieces of a step -------------------------------------------------------------------------------------------
  // A[m][k] += rho_k of this lane's tie over its listed reporters (partial mask rows; model.py:704-718, 742-749)
  auto add_lists = [&](unsigned t, bool on, const double (&rr)[K]) SL_INL {
    if (a.rm2) {   // t = m0 | m1 << 16 (SlArgs::rm2)
#pragma unroll
      for (int h_ = 0; h_ < 2; ++h_) {
        const unsigned m = h_ ? (t >> 16) : (t & 0xffffu);
        const bool v = on && m != 0xffffu;
        const unsigned long long vm = __ballot(v);
        if (vm == 0ull) continue;
        const int m0 = __builtin_amdgcn_readlane((int)m, __builtin_ctzll(vm));
        if (__all(!v || (int)m == m0)) {   // one reporter for the whole wave (ties (i, j..j+63)): one add instead of 64 on one address
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const double sm_ = wave_sum(v ? rr[k] : 0.0);   // (fixed lanes, fixed ties: the same sum every run)
            if (lane == 0) lds_add(&As[m0 * K + k], sm_, sc_a);
          }
        } else if (v) {
#pragma unroll
          for (int k = 0; k < K; ++k) lds_add(&As[m * K + k], rr[k], sc_a);
        }
      }
      return;
    }
    unsigned q0 = 0, q1 = 0;
    if (on) { q0 = rql[t]; q1 = rql[t + 1]; }
    for (unsigned i = 0;; ++i) {
      const bool v = q0 + i < q1;
      const unsigned long long vm = __ballot(v);
      if (vm == 0ull) break;
      const int m = v ? (int)Rml[q0 + i] : -1;
      const int m0 = __builtin_amdgcn_readlane(m, __builtin_ctzll(vm));
      if (__all(!v || m == m0)) {   // one reporter for the whole wave (ties (i, j..j+63) of a self-reporter mask): one add
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const double sm_ = wave_sum(v ? rr[k] : 0.0);
          if (lane == 0) lds_add(&As[m0 * K + k], sm_, sc_a);
        }
      } else if (v) {
#pragma unroll
        for (int k = 0; k < K; ++k) lds_add(&As[m * K + k], rr[k], sc_a);
      }
    }
  };
  // T = sum_m R E[theta_m] of this lane's tie (model.py:766-792)
  auto mask_sum = [&](unsigned cls, unsigned tie) SL_INL -> double {
    if (ALLFULL) return Tfull;
    double Tt = 0.0;
    if (cls == 1u) Tt = Tfull;
    else if (cls == 2u && a.rm2) {   // (tie = m0 | m1 << 16)
      const unsigned m0 = tie & 0xffffu, m1 = tie >> 16;
      Tt = (m0 != 0xffffu ? EthL[m0] : 0.0) + (m1 != 0xffffu ? EthL[m1] : 0.0);
    } else if (cls == 2u && rql) {
      const unsigned q0 = rql[tie], q1 = rql[tie + 1];
      for (unsigned q = q0; q < q1; ++q) Tt += EthL[Rml[q]];
    } else if (cls == 2u) {
      const uint64_t* rwt = Rl + (size_t)tie * g.W;
      for (int w = 0; w < g.W; ++w) {
        uint64_t bits = rwt[w];
        if (bits == ~0ull) { Tt += wsum[w]; continue; }
        for (int n = 0; bits != 0; ++n, bits >>= 4) Tt += lut[((w * 16 + n) << 4) + (unsigned)(bits & 15u)];
      }
    }
    return Tt;
  };
  // w2_k(m, y) = z2 / (z1 + z2), z1 = G_theta_m G_lambda_k, z2 = G_nu y (model.py:694-696; 0 at y = 0 and where both vanish)
  const double gnu_cur = a.par[o.sc + SC_G_NU];
  auto w2_at = [&](unsigned ym, int k) SL_INL -> double {
    const unsigned y = row_level(ym);
    const double z2 = gnu_cur * (double)y, den = Gth[ym - y * (unsigned)Mp] * Gla[k] + z2;
    return (y == 0u || den == 0.0) ? 0.0 : z2 / den;
  };
  // the K factors of a (y, m) row that may lie beyond the LDS levels
  // the factors of a row beyond the LDS levels: the formula of the table, from the per-reporter tables (K divides, no memory)
  auto f_far = [&](unsigned ym, double (&f)[K]) SL_INL {
    const unsigned y = row_level(ym);
    const unsigned m = ym - y * (unsigned)Mp;
    const double lt = Lth[m], gt = Gth[m];
#pragma unroll
    for (int k = 0; k < K; ++k) f[k] = f_entry(g.mut, lt, gt, Lla[k], Gla[k], gnu_f, (int)y);
  };
  // sum_k e^rho_k (G_theta G_lambda_k + G_nu y) + eps, eps alone outside R  (model.py:967-995)
  // (er: the tie's two sums  sum_k e^rho_k G_lambda_k  and  sum_k e^rho_k -- the sum over k of a report is G_theta_m er[0] + G_nu y er[1]:
  // two multiply-adds per report whatever K, and K - 2 registers fewer across the walk)
  auto elbo_inner = [&](unsigned ent, const double (&er)[2]) SL_INL -> double {
    const unsigned ym = SL_YM(ent);
    const unsigned y = row_level(ym);
    const double z2 = gnu * (double)y, gt = Gth[ym - y * (unsigned)Mp];
    const double inner = fma(gt, er[0], z2 * er[1]);
    return (SL_INR(ent) ? inner : 0.0) + eps;
  };
  // U += x F[row] over NP entries whose rows are all in LDS: the table reads in batches, back to back
  auto walk1_near = [&](const unsigned* e, auto npc, double (&U)[K]) SL_INL {
    constexpr int NP = decltype(npc)::value, BT = Batch<K>::value;
#pragma unroll
    for (int j0 = 0; j0 < NP; j0 += BT) {
      double f[BT][K];
#pragma unroll
      for (int u = 0; u < BT; ++u) {
        if (j0 + u < NP) {
#pragma unroll
          for (int k = 0; k < K; ++k) f[u][k] = F[SL_YM(e[j0 + u]) * K + k];
        }
      }
#pragma unroll
      for (int u = 0; u < BT; ++u) {
        if (j0 + u < NP) {
          const double dx = (double)SL_X(e[j0 + u]);
#pragma unroll
          for (int k = 0; k < K; ++k) U[k] = fma(dx, f[u][k], U[k]);
        }
      }
    }
  };
  // H += x rho (+ the ELBO's log terms) over NP entries whose rows are all in LDS; an empty slot adds 0
  auto walk2_near = [&](const unsigned* e, auto npc, const double (&r)[K], const double (&er)[2]) SL_INL {
#ifndef SL_LG
#define SL_LG 2
#endif
    constexpr int NP = decltype(npc)::value, LG = SL_LG;   // logarithms side by side (their chains interleave; more would spill)
#pragma unroll
    for (int j0 = 0; j0 < NP; j0 += LG) {
      double in_[LG];
#pragma unroll
      for (int u = 0; u < LG; ++u) {
        if (j0 + u < NP) {
          const unsigned ym = SL_YM(e[j0 + u]);
          const double dx = (double)SL_X(e[j0 + u]);
          if (a.do_hist) {
#pragma unroll
            for (int k = 1; k < K; ++k) lds_add(&Hc[(unsigned)k * hcm + ym], dx * r[k], sc_h);
          }
          if (ELBO) in_[u] = elbo_inner(e[j0 + u], er);
        }
      }
      if (ELBO) {
#pragma unroll
        for (int u = 0; u < LG; ++u) if (j0 + u < NP) {
          const double term = (double)SL_X(e[j0 + u]) * log_tab(in_[u], lt);
          if (DET) ie_log += fxm(term, sc_r); else e_log += term;
        }
        __builtin_amdgcn_sched_barrier(0);   // (keeps the scheduler from hoisting every group's table reads and chains to the top: spills)
      }
    }
  };
  // the deficits x (1 - sum_k rho_k) of ties whose rho does not sum to 1 go to plane 0 of H.  (In LDS like the rest: the true
  // edges of BASELINE config 5 -- 1000 reports each, every exponential underflows, the reference's all-zero rows -- sent 40 M
  // adds per pass to a few thousand addresses of the global table when this was a global add.)
  auto deficit = [&](unsigned ent, double dfc) SL_INL {   // (a step whose rows all lie in the LDS levels)
    lds_add(&Hc[SL_YM(ent)], (double)SL_X(ent) * dfc, sc_h);
  };
  // ---- one step: RCT = its rounds (0..SL_PF, compile time: straight-line walks), -1 = general (far levels, or more rounds) --
  // ---- the work of one step on its loaded values: RCT = its rounds (0..SL_PF, compile time: straight-line walks), -1 = general
  // (far levels, or more rounds).  row0 / act: the step's first position, this lane's tie exists; ea_c, Rr: its slots and rounds.
  // nf: only the step's rounds 0 .. nf - 1 can hold reports of the levels beyond the LDS ones (Geo::farl: the lists are ordered that way;
  // otherwise every round can)
  auto compute = [&](auto rct, const StepIn<K>& cur, const size_t row0, const bool act, const unsigned ea_c, const int Rr, const unsigned sy_word) SL_INL {
    const unsigned ymax = sy_word & 0xffffu, nf = a.farl ? sy_word >> 16 : 0x7fffffffu;
    const unsigned n1 = (LV0 && a.h0s && !lv0_off) ? sy_word >> 16 : 0x7fffffffu;   // (SlArgs::h0s) from this round on every report of the step has mirror count 0
    constexpr int RCT = decltype(rct)::value;
    constexpr int NP = RCT < 0 ? PFK : RCT;   // rounds held in registers
    const unsigned cls = cur.cls, qt = cur.qt, tie = cur.tie;
    // The rounds of a general step, ONE loop for all of them so that the per-round code exists PFK times and no more (a
    // body of tens of kilobytes does not stay in the instruction cache: a cut with the prefetched rounds and the ring walked
    // by separate code, far levels handled in line, was 68-183 KB per kernel and spent a third of its wave time waiting to
    // issue).  The ring starts from the prefetched rounds and keeps PFK loads in flight over the further ones: no branch around
    // a load -- past the step's last round it loads empty entries (x = 0, row 0) from the zeroed slack behind the lists, through
    // a wave-uniform pointer (scalar base + one per-lane offset register: no vector instruction per load but the load, no select
    // per entry; round 4 -- the general body issued 57 vector instructions per round of a config-5 layer).
    // use(c, j): GB entries of rounds j .. j + GB - 1, in order.
    const unsigned* pe = El + ea_c;
    auto rounds = [&](auto gbc, auto&& use, const int Rr) SL_INL {   // (Rr: the rounds to walk -- the step's, or fewer: SlArgs::x0p)
      constexpr int GB = decltype(gbc)::value;
      unsigned rg[PFK];   // the ring: PFK loads in flight, starting from the prefetched rounds
#pragma unroll
      for (int i = 0; i < PFK; ++i) rg[i] = i < Rr ? cur.e[i] : 0u;   // (a general step may be a short one: its prefetch then read later steps' slots)
      for (int j0 = 0; j0 < Rr; j0 += PFK) {
#pragma unroll
        for (int gi = 0; gi < PFK; gi += GB) {
          unsigned c[GB];
#pragma unroll
          for (int u = 0; u < GB; ++u) {
            c[u] = rg[gi + u];
            const int jn = j0 + gi + u + PFK;
            rg[gi + u] = *at_bytes(jn < Rr ? pe + (size_t)jn * 64 : a.Ez, lane4);
          }
          if (j0 + gi < Rr) use(c, (unsigned)(j0 + gi));   // (wave-uniform)
        }
      }
    };
    double r[K], er[2] = {0.0, 0.0};   // er: sum_k e^rho_k G_lambda_k, sum_k e^rho_k (elbo_inner)
#pragma unroll
    for (int k = 0; k < K; ++k) r[k] = act ? cur.w[k] : 0.0;   // (positions past the last tie read slack rows)
    if (a.do_hist == 2) {   // count mode: every tie "is" category 1 with certainty, so slot 1 of H collects sum x
#pragma unroll
      for (int k = 0; k < K; ++k) r[k] = (k == 1) ? 1.0 : 0.0;
    }
    double Tt = 0.0;
    if (UPDATE || ELBO) Tt = mask_sum(cls, tie);
    double dfc = 0.0;
    bool irr = false;

    if (UPDATE) {
      // ---- walk 1
      double U[K];
#pragma unroll
      for (int k = 0; k < K; ++k) U[k] = 0.0;
#ifdef SL_DEBUG   // timing experiments (results are wrong): VMR_DEBUG bit 16 = no walk 1, 32 = no walk 2, 8 = no H flush
      if (g.dbg & 16) {} else
#endif
      if (RCT >= 0) walk1_near(cur.e, RC<NP>{}, U);
      else {
        // A general step.  Kept SMALL: every round's code exists once per position of a group, and a body of tens of
        // kilobytes does not stay in the instruction cache (a cut with 24 prefetched rounds: 68-183 KB per kernel, a third
        // of the wave time waiting to issue).  Groups of GB rounds read the table for the lanes whose rows are in LDS (the
        // others read row 0 with x = 0); lanes whose row lies beyond take the table's formula, per entry and only where some
        // lane needs it.  Rounds past the step's last are empty entries.
        constexpr int GB = Batch<K>::value < 4 ? Batch<K>::value : 4;
        const unsigned n1u = (LV0R && a.lv0r && !lv0_off) ? sy_word >> 16 : 0x7fffffffu;   // (SlArgs::lv0r) from this round on every report of the step has mirror count 0
        double sl0 = 0.0, sx1 = 0.0;
        rounds(RC<GB>{}, [&](const unsigned (&c)[GB], const unsigned j) SL_INL {
          if (LV0R && j >= n1u) {   // level 0: w1 = 1, the factor is E log theta_m + E log lambda_k (the row index is m)
            double l0[GB];
#pragma unroll
            for (int u = 0; u < GB; ++u) l0[u] = Lth[SL_YM(c[u])];
#pragma unroll
            for (int u = 0; u < GB; ++u) { const double dx = (double)SL_X(c[u]); sl0 = fma(dx, l0[u], sl0); sx1 += dx; }
            return;
          }
          if (ymax < lim1 || j >= nf) { walk1_near(c, RC<GB>{}, U); return; }   // (these reports all lie in the LDS levels of F: known per step, and per round where the far ones come first)
          double f[GB][K];
          bool fr[GB];
#pragma unroll
          for (int u = 0; u < GB; ++u) {
            const unsigned ym = SL_YM(c[u]);
            fr[u] = ym >= ytm;
#pragma unroll
            for (int k = 0; k < K; ++k) f[u][k] = F[(fr[u] ? 0u : ym) * K + k];
          }
#pragma unroll
          for (int u = 0; u < GB; ++u) {
            const double dx = fr[u] ? 0.0 : (double)SL_X(c[u]);
#pragma unroll
            for (int k = 0; k < K; ++k) U[k] = fma(dx, f[u][k], U[k]);
          }
#pragma unroll
          for (int u = 0; u < GB; ++u) {
            if (__any(fr[u])) {   // (some lane's row lies beyond the LDS levels: the table's formula for those lanes)
              if (fr[u]) {
                double ff[K];
                f_far(SL_YM(c[u]), ff);
                const double dx = (double)SL_X(c[u]);
#pragma unroll
                for (int k = 0; k < K; ++k) U[k] = fma(dx, ff[k], U[k]);
              }
            }
          }
        }, Rr);
        if (LV0R && a.lv0r && !lv0_off) {
#pragma unroll
          for (int k = 0; k < K; ++k) U[k] += fma(Lla[k], sx1, sl0);
        }
      }
      // ---- per-tie update from the finished sums
      double aa[K];
#pragma unroll
      for (int k = 0; k < K; ++k) aa[k] = (cur.v[k] + U[k]) - Tt * Ela[k];
      bool done = false;
      if (K == 2) {
        // two categories: rho_0 = 1 / (1 + e^(a1-a0)) -- one exp, one reciprocal -- wherever the reference's raw exponentials
        // neither overflow nor underflow (then equal to exp(a_k) / sum up to rounding); other ties below
        const double d = aa[1] - aa[0];
        const bool safe = fabs(aa[0]) < 700.0 && fabs(aa[1]) < 700.0 && fabs(d) < 700.0;
        if (__all(safe)) {
          const double e = exp_tab(d, xt);   // (a table-free degree-13 polynomial was measured 17 % slower per launch)
          r[0] = rcp_nr2(1.0 + e);   // (1 + e in [1, e^700]: no scaling needed)
          r[1] = e * r[0];
          done = true;
        }
      }
      if (!done) {
        double sum = 0.0;
        bool tame = true;   // every a_k of the wave's ties where exp neither overflows nor underflows (else: the library's)
#pragma unroll
        for (int k = 0; k < K; ++k) tame = tame && fabs(aa[k]) < 700.0;
        tame = __all(tame);
        if (tame) {
#pragma unroll
          for (int k = 0; k < K; ++k) r[k] = exp_tab(aa[k], xt);
        } else {
#pragma unroll
          for (int k = 0; k < K; ++k) r[k] = exp(aa[k]);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) sum += r[k];   // no max-subtraction, as model.py:807
        if (sum > 0.0) {   // model.py:808-811; a true divide: 1/sum overflows when sum is subnormal
#pragma unroll
          for (int k = 0; k < K; ++k) r[k] /= sum;
        }
        if (a.do_hist) {   // does every tie's rho sum to 1?  (the K = 2 formula does by construction)
          double sm = 0.0;
#pragma unroll
          for (int k = 0; k < K; ++k) sm += r[k];
          dfc = 1.0 - sm;
          if (fabs(dfc) <= 1e-14 || !act) dfc = 0.0;
          irr = __any(dfc != 0.0);
          if (irr && lane == 0) tick[1] = 1u;
        }
      }
      // exactly NST store instructions on every path, no mask (the wait at the bottom of the step counts on it): positions
      // past the last tie write their (finite) values into the slack rows behind the array
      if (STORE) store_k<K>(act ? at_bytes(rl + row0 * K, laneK8) : at_bytes(rho_slack, laneK8), r);
      if (act && (ALLFULL || cls == 1u)) {
#pragma unroll
        for (int k = 0; k < K; ++k) { if (DET) iaccF[k] += fxm(r[k], sc_a); else accF[k] += r[k]; }
      }
      if (!ALLFULL && a.sum_a) add_lists(tie, act && cls == 2u, r);
    } else if (a.do_hist) {
      // the current rho: does it sum to 1 (a user-supplied prior may not)?
      double sm = 0.0;
#pragma unroll
      for (int k = 0; k < K; ++k) sm += r[k];
      dfc = 1.0 - sm;
      if (fabs(dfc) <= 1e-14 || !act) dfc = 0.0;
      irr = __any(dfc != 0.0);
      if (irr && lane == 0) tick[1] = 1u;
      if (!ELBO && a.do_hist == 1 && act && (ALLFULL || cls == 1u)) {   // all-ones mask rows are summed here too
#pragma unroll
        for (int k = 0; k < K; ++k) { if (DET) iaccF[k] += fxm(r[k], sc_a); else accF[k] += r[k]; }
      }
      if (!ALLFULL && !ELBO && a.sum_a) add_lists(tie, act && cls == 2u, r);
    }
    if (ELBO) {
#pragma unroll
      for (int k = 0; k < K; ++k) { const double ek = exp_tab(r[k], xt); er[0] = fma(ek, Gla[k], er[0]); er[1] += ek; }   // exp(rho), model.py:971
    }
    // ---- walk 2: H of the (new) rho, the ELBO's log terms
#ifdef SL_DEBUG
    if (g.dbg & 32) {} else
#endif
    if (a.do_hist || ELBO) {
      if (RCT >= 0) {
        walk2_near(cur.e, RC<NP>{}, r, er);
        if (irr) {
#pragma unroll
          for (int j = 0; j < NP; ++j) deficit(cur.e[j], dfc);
        }
      } else {
        // a general step, as in walk 1: the lanes whose row is in the LDS levels add there (the others add 0 to row 0); rows
        // beyond the levels go to global memory, per entry and only where some lane needs it
        // (SlArgs::x0p: the rounds of level 0 only are not even read -- the tie's counts there are a constant of the data -- unless a tie
        // of the step is irregular: its deficits are per reporter)
        const bool skip0 = LV0 && a.x0p != nullptr && a.do_hist == 1 && n1 != 0x7fffffffu && !irr;
        const int Rw = skip0 ? min(Rr, (int)n1) : Rr;
        double sx0 = skip0 ? (double)cur.x0 : 0.0;   // (SlArgs::h0s) this tie's counts in the rounds of level 0 only
        rounds(RC<1>{}, [&](const unsigned (&c1)[1], const unsigned j) SL_INL {
          const unsigned c = c1[0], ym = SL_YM(c), x = SL_X(c);
          const bool far_round = j < nf && lim2 != 0xffffffffu;   // (wave-uniform) only such rounds can hold a report beyond the LDS levels
          const bool fr = far_round && ym >= hcm;
          const double dx = (double)x;
          if (LV0 && a.do_hist && j >= n1) {   // a round of level 0 only: no LDS add but an irregular tie's deficit
            sx0 += dx;
            if (irr) lds_add(&Hc[ym], dx * dfc, sc_h);
          } else if (a.do_hist) {
            if (!fr) {   // (far lanes add nothing here: zeros added to one common row would serialise them)
              if (!(skip0 && ym < (unsigned)Mp)) {   // (SlArgs::x0p: a level-0 report of a mixed round is in the tie's constant too)
#pragma unroll
                for (int k = 1; k < K; ++k) lds_add(&Hc[(unsigned)k * hcm + ym], dx * r[k], sc_h);
              }
              if (irr) lds_add(&Hc[ym], dx * dfc, sc_h);   // (wave-uniform: some tie of the step does not sum to 1)
            }
            if (far_round && !a.farl && __any(fr)) {   // (rare) rows beyond the LDS levels: global adds.  Their share of nu is taken from the global
              if (fr && x != 0u) {       // table by the grid's last workgroup (nu_far).  (farl: k_far_hist adds them.)
                if (DET) {
                  unsigned long long* d0 = a.det;
                  asm volatile("" : "+s"(d0));   // (keeps this address arithmetic inside the rare branch: hoisted, it cost every variant ten registers)
                  unsigned long long* d = d0 + (size_t)l * g.Y * Mp * K + (size_t)ym * K;
#pragma unroll
                  for (int k = 1; k < K; ++k) atomicAdd(&d[k], fxm((double)x * r[k], sc_h));
                  if (dfc != 0.0) atomicAdd(&d[0], fxm((double)x * dfc, sc_h));
                } else {
                  double* d = Hl + (size_t)ym * K;
#pragma unroll
                  for (int k = 1; k < K; ++k) atomicAdd(&d[k], (double)x * r[k]);
                  if (dfc != 0.0) atomicAdd(&d[0], (double)x * dfc);
                }
              }
              asm volatile("" ::: "memory");
            }
          }
          if (ELBO) { const double term = (double)x * log_tab(elbo_inner(c, er), lt); if (DET) ie_log += fxm(term, sc_r); else e_log += term; }
        }, Rw);
        if (LV0 && a.h0s && a.do_hist == 1 && act) {
#pragma unroll
          for (int k = 1; k < K; ++k) acc0[k] = fma(r[k], sx0, acc0[k]);
        }
      }
    }
    if (ELBO && act) {
      double sr = 0.0, se = 0.0, en = 0.0;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        sr += r[k]; se += r[k] * Ela[k];
        en += r[k] * cur.v[k] - r[k] * log_tab(r[k] + eps, lt);   // model.py:1306-1313
      }
      if (DET) { ie_lin += det_fx(en - se * Tt, g.det_shr); if (Ql) ie_q += det_fx(sr * (double)qt, g.det_shr); }   // (per tie: the exact conversion, whatever the magnitude)
      else { e_lin += en - se * Tt; if (Ql) e_q += sr * (double)qt; }
    }
  };
  // ---- one step: prefetch of the next, the work (compute), one wait
  auto body = [&](auto rct) SL_INL {
    constexpr int RCT = decltype(rct)::value;
    constexpr int NP = RCT < 0 ? PFK : RCT;   // rounds held in registers
    const size_t row0 = (size_t)s * 64;
    const bool act = (unsigned)row0 + (unsigned)lane < T32;
    const unsigned ea_c = ea;
    const int Rr = RCT < 0 ? R : RCT;
    const StepIn<K> cur = P;
    // the next step: its range was loaded a step ago; its per-tie values and rounds now.  It has at most as many rounds as
    // this one (sorted order), so NP loads cover them; surplus loads read later steps' slots, unused.  After the wave's last
    // step the same loads are issued once more, of this step's own (valid) addresses: no branch around a load.
    const long long s2 = sn, s3 = draw();
    const bool more = s2 < NS;
    const unsigned ea2 = more ? (unsigned)__builtin_amdgcn_readlane((int)rgv, 0) : ea_c;
    const int R2 = more ? (int)(((unsigned)__builtin_amdgcn_readlane((int)rgv, 1) - ea2) >> 6) : 0;
    const unsigned ym2 = more ? (unsigned)__builtin_amdgcn_readlane((int)rgv, 2) : 0u;
    fetch_range(s3 < NS ? s3 : s);
    fetch_tie(P, more ? s2 : s, RCT == 0 && a.lp0 != 0);   // (sorted order: after a step without reports come only such steps)
    if (RCT >= 0 || PFK <= 8) fetch_ent(P, ea2, RC<NP>{});
    else {   // a general step of a wide-prefetch variant: as many blocks of 8 rounds as the next step has
      const unsigned* pe = El + ea2;
#pragma unroll
      for (int j0 = 0; j0 < PFK; j0 += 8) {
        if (j0 == 0 || j0 < R2) {   // (wave-uniform)
#pragma unroll
          for (int j = j0; j < j0 + 8; ++j) P.e[j] = *at_bytes(pe + j * 64, lane4);
        }
      }
    }
    compute(rct, cur, row0, act, ea_c, Rr, ymax);
    // Everything the NEXT step needs was requested at the top of this one: wait for it HERE, before the next step's own
    // requests go out.  (Left to itself the compiler waits at the first use, after those requests -- and where it cannot tell
    // how many requests are younger than the ones it needs it drains them all: memory latency in every step.)  This step's rho
    // store is younger than the requests and need not have landed: the counter is in order, so "all but the NST youngest"
    // covers exactly the loads -- every path through a step issues exactly NST store instructions (see there).
    constexpr int NST = (UPDATE && STORE) ? (K % 2 == 0 ? K / 2 : K) : 0;
    static_assert(NST <= 8, "vmcnt immediate below");
    if (RCT >= 0) __builtin_amdgcn_s_waitcnt(0x0F70 | NST);   // vmcnt(NST)
    else __builtin_amdgcn_s_waitcnt(0x0F70);                  // (general steps may add global atomics: vmcnt(0))
    // advance
    s = s2; sn = s3; ea = ea2; R = R2; ymax = ym2;
  };
  // consecutive steps of this wave with the same number of rounds run in one straight-line loop
  auto run = [&](auto rct) SL_INL {
    constexpr int RCT = decltype(rct)::value;
    do { body(rct); } while (s < NS && R == RCT && (ymax & 0xffffu) < lim_y);
  };
  while (s < NS) {
    if (R > SL_PF || (ymax & 0xffffu) >= lim_y) { body(RC<-1>{}); continue; }
    switch (R) {
      case 0: run(RC<0>{}); break;
      case 1: run(RC<1>{}); break;
      case 2: run(RC<2>{}); break;
      case 3: run(RC<3>{}); break;
      case 4: run(RC<4>{}); break;
      case 5: run(RC<5>{}); break;
      case 6: run(RC<6>{}); break;
      case 7: run(RC<7>{}); break;
      default: run(RC<8>{}); break;
    }
  }

#ifdef SL_DEBUG
  if (dbt && lane == 0) dbt[2] = wall_clock64();
#endif
  const bool nu_here = a.nu_acc && a.do_hist == 1;
  if (nu_here) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's global adds (far levels, deficits) are performed: see nu_far
  __syncthreads();
  if (nu_here) {
    // this workgroup's share of nu_shp - alpha: sum_{y>0,m} [sum_{k>0} (w2_k - w2_0) H_k - w2_0 D] over its LDS levels (D: the
    // deficits, plane 0); summed BEFORE the flush, whose float atomics the ticket below must not wait for
    double a0p = 0.0;
    for (int q = (tick[1] ? 0 : (int)hcm) + tid; q < nHc; q += nthr) {   // (plane 0 only when something was added to it)
      const double v = DET ? det_back(reinterpret_cast<const unsigned long long*>(Hc)[q], g.det_sh) : Hc[q];
      if (v != 0.0) {
        const int k = q / (int)hcm;
        const unsigned ym = (unsigned)(q - k * (int)hcm);
        a0p += ((k ? w2_at(ym, k) : 0.0) - w2_at(ym, 0)) * v;
      }
    }
    a0p = block_sum_n(a0p, red);   // (a fixed thread <-> cell map and a fixed tree: the same double every run)
    if (tid == 0) {
      if (DET) atomicAdd(&det_R()[4], det_fx(a0p, g.det_sh)); else atomicAdd(&a.nu_acc[0], a0p);
    }
  }
#ifdef SL_DEBUG
  if (dbt && lane == 0) dbt[6] = wall_clock64();
#endif
  // the flush: every wave but the first when the first draws the ticket (it then has nothing else of its own outstanding)
  const int f0 = (nu_here && nw > 1) ? 64 : 0;
#ifdef SL_DEBUG
  if (g.dbg & 8) {} else
#endif
  if (a.do_hist && tid >= f0) {   // the LDS levels ([K][hc][Mp]) into this workgroup's copy of H ([Y][Mp][K])
    for (int q = (tick[1] ? 0 : (int)hcm) + tid - f0; q < nHc; q += nthr - f0) {
      if (DET) {   // (the cell IS the fixed-point sum: added to the shadow as it stands)
        const unsigned long long iv = reinterpret_cast<const unsigned long long*>(Hc)[q];
        if (iv != 0ull) { const int k = q / (int)hcm, ym = q - k * (int)hcm; atomicAdd(&det_H()[(size_t)ym * K + k], iv); }
        continue;
      }
      const double v = Hc[q];
      if (v != 0.0) {
        const int k = q / (int)hcm, ym = q - k * (int)hcm;
        atomicAdd(&Hl[(size_t)ym * K + k], v);
      }
    }
  }
  if (a.sum_a && tid >= f0) {   // this workgroup's mask-list sums into its slot of slotA ([l][slot][W*64][K])
    double* out = a.slotA + ((size_t)l * NSLOT + (gb % NSLOT)) * (size_t)g.W * 64 * K;
    for (int q = tid - f0; q < g.M * K; q += nthr - f0) {
      if (DET) { const unsigned long long iv = reinterpret_cast<const unsigned long long*>(As)[q]; if (iv != 0ull) atomicAdd(&det_A()[q], iv); continue; }
      const double v = As[q];
      if (v != 0.0) atomicAdd(&out[q], v);
    }
  }
#ifdef SL_DEBUG
  if (dbt && lane == 0) dbt[7] = wall_clock64();
#endif
  if (nu_here) {
    // the ticket; the grid's last workgroup finishes nu (model.py:820-830)
    __shared__ int nu_last;
    if (tid == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (this workgroup's share is performed at the memory side before the ticket is drawn)
      const double t = atomicAdd(&a.nu_acc[1], 1.0);
      nu_last = (t == (double)(gx - 1u));
      if (nu_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (nu_last && a.farl) {   // k_far_hist adds the far levels' share and finishes nu: the sum stays, the ticket is reset
      if (tid == 0) a.nu_acc[1] = 0.0;
    } else if (nu_last) {
      // nu_far: what went to the global table directly -- reports of levels beyond the LDS copies (rows y >= hc: categories
      // k > 0 and, in slot 0, the deficits of irregular ties) -- weighted like the rest; every workgroup's adds were performed
      // before its ticket.  Usually all zero: the loads are the cost ((Y - hc) Mp K NH values over the workgroup).
      double far = 0.0;
      const size_t hcs = (size_t)g.Y * Mp * K;
      const int y0 = max(1, a.hc);   // first row that can hold anything
      if (a.hc < g.Y) {
        for (int ll = 0; ll < g.L; ++ll) {
          const double* H0 = a.Hg + (size_t)ll * NH * hcs;
          const double* gthl = a.par + o.G_th + (size_t)ll * Mp;
          for (int it = y0 * Mp + tid; it < g.Y * Mp; it += nthr) {
            const int y = it / Mp, m = it - y * Mp;
            double v0[NH], vk[NH][K > 1 ? K - 1 : 1];
#pragma unroll
            for (int c = 0; c < NH; ++c) {   // loads first, all in flight
              v0[c] = H0[(size_t)c * hcs + (size_t)it * K];
#pragma unroll
              for (int k = 1; k < K; ++k) vk[c][k - 1] = H0[(size_t)c * hcs + (size_t)it * K + k];
            }
            if (DET) {   // (deterministic mode: the far rows are in the integer shadow)
              const unsigned long long* hi = a.det + (size_t)ll * hcs + (size_t)it * K;
              v0[0] += det_back(hi[0], g.det_sh);
#pragma unroll
              for (int k = 1; k < K; ++k) vk[0][k - 1] += det_back(hi[k], g.det_sh);
            }
            double h_[K];
#pragma unroll
            for (int k = 0; k < K; ++k) h_[k] = 0.0;
#pragma unroll
            for (int c = 0; c < NH; ++c) {
              h_[0] += v0[c];
#pragma unroll
              for (int k = 1; k < K; ++k) h_[k] += vk[c][k - 1];
            }
            bool any = h_[0] != 0.0;
#pragma unroll
            for (int k = 1; k < K; ++k) any = any || h_[k] != 0.0;
            if (any && m < g.M) {
              const double z2 = gnu_cur * (double)y, gt = gthl[m];
              const double d0 = gt * a.par[o.G_la + ll * K] + z2, w0 = d0 == 0.0 ? 0.0 : z2 / d0;
              far -= w0 * h_[0];
#pragma unroll
              for (int k = 1; k < K; ++k) {
                const double dk = gt * a.par[o.G_la + ll * K + k] + z2;
                far += ((dk == 0.0 ? 0.0 : z2 / dk) - w0) * h_[k];
              }
            }
          }
        }
      }
      far = block_sum_n(far, red);
      if (tid == 0) {
        double tot = far;   // + every workgroup's share (device-scope read)
        if (DET) { unsigned long long* dR = det_R(); tot += det_back(atomicAdd(&dR[4], 0ull), g.det_sh); dR[4] = 0ull; } else tot += atomicAdd(&a.nu_acc[0], 0.0);
        for (int ll = 0; ll < g.L; ++ll) tot += a.nu_acc[2 + ll];
        a.nu_acc[0] = 0.0; a.nu_acc[1] = 0.0;
        a.elbo_dev[1] = tot;   // the raw piece, for fits whose layers are spread over several handles (vmr_sweep_local)
        if (a.commit_nu) {
          double* sc = const_cast<double*>(a.par) + o.sc;
          sc[SC_G_NU_STALE] = sc[SC_G_NU];           // what the last cache refresh held (model.py:684)
          sc[SC_NU_SHP] = sc[SC_A_ETA] + tot;
          sc[SC_G_NU] = exp(digamma_pos(sc[SC_NU_SHP]) - log(sc[SC_NU_RTE]));
          sc[SC_E_NU] = sc[SC_NU_SHP] / sc[SC_NU_RTE];
        }
      }
    }
  }
  // a wave's integer sum (DET): 64-bit adds through two 32-bit shuffles
  auto wave_isum = [&](unsigned long long v) SL_INL -> unsigned long long {
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) {
      const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, o2, 64), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), o2, 64);
      v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
  };
  if ((UPDATE || (!ELBO && a.do_hist == 1)) && a.slotF) {   // (slotF null: a pass that only re-writes rho, ensure_rho)
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (DET) { const unsigned long long v = wave_isum(iaccF[k]); if (lane == 0 && v != 0ull) atomicAdd(&det_F()[k], v); continue; }
      double v = block_sum_n(accF[k], red);
      if (tid == 0) atomicAdd(&a.slotF[((size_t)l * NSLOT + (gb % NSLOT)) * K + k], v);
    }
  }
  if (LV0 && a.h0s && a.do_hist == 1) {
#pragma unroll
    for (int k = 1; k < K; ++k) {
      const double v = block_sum_n(acc0[k], red);
      if (tid == 0 && v != 0.0) atomicAdd(&a.h0s[((size_t)l * NSLOT + (gb % NSLOT)) * K + k], v);
    }
  }
  if (ELBO) {
    if (DET) {
      const unsigned long long v1 = wave_isum(ie_lin), v2 = wave_isum(ie_log), v3 = wave_isum(ie_q);
      if (lane == 0) { unsigned long long* dR = det_R(); atomicAdd(&dR[1], v1); atomicAdd(&dR[2], v2); atomicAdd(&dR[3], v3); }
    } else {
      double v1 = block_sum_n(e_lin, red);
      double v2 = block_sum_n(e_log, red);
      double v3 = block_sum_n(e_q, red);
      if (tid == 0) {
        double* out = a.slotR + (size_t)(bx % NSLOT) * 4;
        atomicAdd(&out[1], v1); atomicAdd(&out[2], v2); atomicAdd(&out[3], v3);
      }
    }
  }
#ifdef SL_DEBUG
  if (dbt && lane == 0) dbt[3] = wall_clock64();
#endif
}

#define SL_BOUNDS(K, UPDATE, ELBO, ALLFULL) \
  __launch_bounds__(sl_tpb_max(K, ELBO, ALLFULL, UPDATE), (K == 2 && !ELBO && ALLFULL) ? SL_WPE : sl_wpe(K, ELBO, ALLFULL, UPDATE))
template <int K, bool UPDATE, bool ELBO, bool ALLFULL, bool STORE = true, bool DET = false>
__global__ SL_BOUNDS(K, UPDATE, ELBO, ALLFULL) void k_sweep_sl(SlArgs a, Geo g) {
  sweep_body<K, UPDATE, ELBO, ALLFULL, STORE, DET>(a, g, blockIdx.x, gridDim.x);
}
// One launch for many small handles in lockstep (vmr_fit_loop_batch): workgroup -> unit through `blk_unit`, the unit's
// arguments from device memory.  A sweep of a Karnataka-sized layer is two dependent 20-40 us launches that leave the GPU
// nearly empty; here every unit's sweep shares them.
template <int K, bool UPDATE, bool ELBO, bool ALLFULL, bool STORE = true>
__global__ __launch_bounds__(sl_tpb_max_b(K, ELBO, ALLFULL), sl_wpe_b(K, ELBO, ALLFULL)) void k_sweep_sl_b(const SlUnit* __restrict__ units, const int* __restrict__ blk_unit) {
  const SlUnit& u = units[blk_unit[blockIdx.x]];
  sweep_body<K, UPDATE, ELBO, ALLFULL, STORE>(u.a, u.g, blockIdx.x - (unsigned)u.blk0, (unsigned)u.nblk);
}

// ------------------------------------------------------------------------------------------
// launcher of this object's K
// ------------------------------------------------------------------------------------------
template <bool UPDATE, bool ELBO, bool ALLFULL, bool STORE = true, bool DET = false>
static int sl_launch_one(vmr_ctx* h, const SlShape& sh, SlArgs& a) {
  constexpr int K = VMR_K;
  const Geo& g = h->g;
  const long long NS = ((long long)g.N * g.N + 63) / 64, nw = sh.tpb / 64;
  int rc = grid_per_layer(h, k_sweep_sl<K, UPDATE, ELBO, ALLFULL, STORE, DET>, sh.smem, &a.Gl, (NS + nw - 1) / nw, sh.tpb);
  if (rc) return rc;
#ifdef SL_DEBUG
  // VMR_DEBUG_TIMES=<file>: every launch appends "<update><elbo> <waves>" and one line of four clock readings per wave
  static unsigned long long* dbg_buf = nullptr;
  const char* tf = getenv("VMR_DEBUG_TIMES");
  const size_t nwv = (size_t)g.L * a.Gl * nw;
  a.dbg_t = nullptr;
  if (tf) {
    if (!dbg_buf) (void)hipMalloc(&dbg_buf, (size_t)1 << 22);
    if (nwv * 64 <= ((size_t)1 << 22)) a.dbg_t = dbg_buf;
  }
#endif
  hipLaunchKernelGGL((k_sweep_sl<K, UPDATE, ELBO, ALLFULL, STORE, DET>), dim3(g.L * a.Gl), dim3(sh.tpb), sh.smem, h->stream, a, g);
#ifdef SL_DEBUG
  if (a.dbg_t) {
    std::vector<unsigned long long> t(nwv * 8);
    (void)hipStreamSynchronize(h->stream);
    (void)hipMemcpy(t.data(), dbg_buf, nwv * 64, hipMemcpyDeviceToHost);
    if (FILE* f = fopen(tf, "a")) {
      fprintf(f, "launch %d%d hist %d waves %zu tpb %d\n", (int)UPDATE, (int)ELBO, a.do_hist, nwv, sh.tpb);
      for (size_t i = 0; i < nwv; ++i) fprintf(f, "%llu %llu %llu %llu %llu %llu %llu %llu\n", t[i * 8], t[i * 8 + 1], t[i * 8 + 2], t[i * 8 + 3], t[i * 8 + 4], t[i * 8 + 5], t[i * 8 + 6], t[i * 8 + 7]);
      fclose(f);
    }
  }
#endif
  return VMR_OK;
}

#define SL_CAT2(a, b) a##b
#define SL_CAT(a, b) SL_CAT2(a, b)
int SL_CAT(vmr_sl_launch_k, VMR_K)(vmr_ctx* h, int mode, const SlShape& sh, SlArgs& a) {
  if (a.det) {   // deterministic mode: the integer-sum variants
    if (a.cls == nullptr) {
      switch (mode) {
        case 0: return sl_launch_one<true, false, true, true, true>(h, sh, a);
        case 4: return sl_launch_one<true, false, true, false, true>(h, sh, a);
        case 1: return sl_launch_one<true, true, true, true, true>(h, sh, a);
        case 2: return sl_launch_one<false, true, true, true, true>(h, sh, a);
        default: return sl_launch_one<false, false, true, true, true>(h, sh, a);
      }
    }
    switch (mode) {
      case 0: return sl_launch_one<true, false, false, true, true>(h, sh, a);
      case 4: return sl_launch_one<true, false, false, false, true>(h, sh, a);
      case 1: return sl_launch_one<true, true, false, true, true>(h, sh, a);
      case 2: return sl_launch_one<false, true, false, true, true>(h, sh, a);
      default: return sl_launch_one<false, false, false, true, true>(h, sh, a);
    }
  }
  if (a.cls == nullptr) {   // every mask row is all ones
    switch (mode) {
      case 0: return sl_launch_one<true, false, true>(h, sh, a);
      case 1: return sl_launch_one<true, true, true>(h, sh, a);
      case 2: return sl_launch_one<false, true, true>(h, sh, a);
      case 4: return sl_launch_one<true, false, true, false>(h, sh, a);   // (rho update, rho not written)
      default: return sl_launch_one<false, false, true>(h, sh, a);
    }
  }
  switch (mode) {
    case 0: return sl_launch_one<true, false, false>(h, sh, a);
    case 1: return sl_launch_one<true, true, false>(h, sh, a);
    case 2: return sl_launch_one<false, true, false>(h, sh, a);
    case 4: return sl_launch_one<true, false, false, false>(h, sh, a);
    default: return sl_launch_one<false, false, false>(h, sh, a);
  }
}

// the rho update (mode 0), rho update + ELBO data terms (mode 1) or the statistics of the current rho (mode 3: a realisation's
// first sweep) of `nblocks` workgroups' worth of units in one launch
template <bool UPDATE, bool ELBO, bool ALLFULL, bool STORE = true>
static int sl_launch_batch_one(vmr_ctx* h, hipStream_t st, const SlUnit* units, const int* blk_unit, int nblocks, int tpb, size_t smem) {
  constexpr int K = VMR_K;
  // (the leave for > 48 KB of dynamic LDS is per device and this may be any thread's first launch there: asked for every time --
  // a call that costs nothing beside a batch's launches -- instead of remembered in a process-wide static)
  if (smem > 48 * 1024)
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep_sl_b<K, UPDATE, ELBO, ALLFULL, STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL((k_sweep_sl_b<K, UPDATE, ELBO, ALLFULL, STORE>), dim3(nblocks), dim3(tpb), smem, st, units, blk_unit);
  return VMR_OK;
}
int SL_CAT(vmr_sl_launch_batch_k, VMR_K)(vmr_ctx* h, hipStream_t st, int mode, int allfull, const SlUnit* units, const int* blk_unit, int nblocks,
                                          int tpb, size_t smem) {
  if (mode == 3) return allfull ? sl_launch_batch_one<false, false, true>(h, st, units, blk_unit, nblocks, tpb, smem)
                                : sl_launch_batch_one<false, false, false>(h, st, units, blk_unit, nblocks, tpb, smem);
  if (mode == 4) return allfull ? sl_launch_batch_one<true, false, true, false>(h, st, units, blk_unit, nblocks, tpb, smem)
                                : sl_launch_batch_one<true, false, false, false>(h, st, units, blk_unit, nblocks, tpb, smem);
  if (allfull) return mode ? sl_launch_batch_one<true, true, true>(h, st, units, blk_unit, nblocks, tpb, smem)
                           : sl_launch_batch_one<true, false, true>(h, st, units, blk_unit, nblocks, tpb, smem);
  return mode ? sl_launch_batch_one<true, true, false>(h, st, units, blk_unit, nblocks, tpb, smem)
              : sl_launch_batch_one<true, false, false>(h, st, units, blk_unit, nblocks, tpb, smem);
}
