// sweep_sl.h -- the sorted report lists (the default data format of a sparse X) and the sweep kernel over them.
//
// Ties of a layer are SORTED by their number of reports (descending, stable) and taken in steps of 64 consecutive sorted
// positions -- one wave, one tie per lane.  After the sort the ties of a step hold (almost) the same number of reports, so a
// step is R_s full rounds of 64 slots and nothing else:
//     E[ebase[l] + rs[l][s] + r * 64 + lane]   the r-th report of the tie at position s * 64 + lane (0 = none)
//     (wide entries, Geo::wide -- counts beyond 2047 or more than 2^20 table rows; the general kernels of sweep_gen.hip only:
//      E[slot] = y * Mp + m in 32 bits, EX[slot] = x << 1 | R[l,i,j,m])
//     entry   bits 0..19  y * Mp + m  (mirror count y = X[l,j,i,m], 0 when mutuality is off; reporter m): the row of the
//                         per-(y, m) tables F and H        bit 20  R[l,i,j,m]        bits 21..31  the count x (<= 2047)
//     rs[l][s] (u32, NS + 1 per layer)   first slot of step s;  R_s = (rs[s+1] - rs[s]) / 64, non-increasing in s
//     perm[l][pos] (u32)                 the tie (i * N + j) at sorted position pos (0xffffffff beyond the last tie)
//     sy[l][s] (u32)                     bits 0..15: the highest mirror count y among the step's reports (which table levels it needs);
//                                        bits 16..31 (Geo::farl): the step's reports of the levels beyond the LDS ones are its ties' FIRST
//                                        ones -- rounds 0 .. nf - 1 -- and this is nf (k_far_first)
// Padding only appears in the ~max-count steps per layer where the count changes.  Every per-tie array the sweeps touch is
// stored BY POSITION (rho, log prior, mask-row class, the ELBO's mirror sums Qt), so all of a wave's accesses are contiguous;
// the boundary functions (vmr_set_state, vmr_get_state, vmr_readout, vmr_sample) translate through perm.
#ifndef VMR_SWEEP_SL_H
#define VMR_SWEEP_SL_H

#include "vmr_internal.h"
#include <functional>

#define SL_YM(e) ((e) & 0xfffffu)
#define SL_INR(e) (((e) >> 20) & 1u)
#define SL_X(e) ((e) >> 21)
#define SL_XMAX 2047u          // largest count an entry holds
#define SL_YM_ROWS (1u << 20)  // (max count + 1) * Mp must not exceed this
#define SL_PF 8                // steps of up to this many rounds run in straight-line code compiled for their round count
// rounds of a step that are prefetched one step ahead (registers) = the depth of the ring that streams the further rounds of a
// longer step: what a wave keeps in flight.  8 for every K: with 16 (tried for K >= 3, whose budgets have the registers) a
// 19-report step of a config-5 layer issues 32 loads for 19 reports, and the over-fetch costs more than the depth hides.
#ifndef SL_PFW
#define SL_PFW 8    // (a multiple of 8; 16: config-5 sweep 3.63 instead of 3.53 ms)
#endif
constexpr int sl_pf(int K) { return K <= 2 ? SL_PF : SL_PFW; }
#define SL_SLACK (64 * 64)     // entry slots past the last one that prefetches may read (never use)

struct SlArgs {
  const unsigned* E; const unsigned* rs; const unsigned long long* ebase; const unsigned* perm;
  const unsigned* sy;    // [L][NS] highest mirror-count level among the reports of a step
  const uint8_t* cls;    // [L][T] mask-row class by position: 0 empty, 1 all ones, 2 partial (null: every row is all ones)
  const unsigned* Qt;    // [L][T] by position: sum_m R[t,m] X[mirror(t),m]
  const uint64_t* Rb; const unsigned* rq; const unsigned short* Rm; const unsigned long long* rbase;   // mask rows, by TIE
  // Mask lists of at most two reporters (the self-reporter mask of survey data: R[l,i,j,m] = 1 iff m is i or j), BY POSITION:
  // m0 | m1 << 16, 0xffff = none.  Prefetched with the tie's other values one step ahead, so a partial row costs two LDS reads
  // of E[theta] instead of three dependent global loads (tie -> list range -> reporters -> E[theta]) inside the step -- which was
  // what a step of a Karnataka-sized layer took its time for.  null: the lists by tie (rq / Rm).
  const unsigned* rm2;
  double* rho; const double* logpr;   // [L][T][K] by position
  const double* par; double* slotR; const double* lutg; double* Hg; double* slotF; double* slotA;
  int Gl;        // workgroups per layer
  int do_hist;   // 1: accumulate the statistics H; 2: count mode (vmr_create): every tie is category 1, slot 1 gets sum x
  int yt, hc;    // levels (mirror counts 0..) of F / of H held in LDS
  int sum_a;     // also sum the (new) rho over the listed reporters of the partial mask rows into slotA
  // The nu update inside the pass (model.py:820-830): nu_shp - alpha = sum x rho_k w2_k is a linear functional of the
  // statistics H the pass holds, so every workgroup adds its share to nu_acc[0] and the grid's last one (ticket nu_acc[1])
  // finishes nu -- no finalize launch on plain sweeps.  nu_acc[2 + l] = sum_{y,m} w2_0 C[l][y][m], the part that comes from
  // the constant C (H_0 = C - D - sum_{k>0} H_k, D: the deficits of ties whose rho does not sum to 1), left there by
  // k_fin_gamma.  null: the pass does not touch nu.
  double* nu_acc; double* elbo_dev; int commit_nu;
  int nu_stale;   // the factor table from the nu BEFORE its last update (SC_G_NU_STALE): the pass that re-writes the rho of the last sweep (ensure_rho)
  // Deterministic mode (Geo::det): every sum that crosses workgroups is added as a 64-bit integer in fixed point (integer adds
  // commute exactly) into these shadows -- [L][Y][Mp][K] H | [L][W*64][K] mask sums | [L][K] rho over all-ones rows | 4 ELBO
  // partials | 1 nu share -- which k_det_fold turns into the doubles the finalize kernels read.  Inside a workgroup nothing
  // needs it: a workgroup is ONE wave there and walks a fixed share of the steps in order.  null: floating-point atomics.
  unsigned long long* det;
  // Every tie of a step WITHOUT reports carries the reference's one-hot prior (1, 0, .., 0) (model.py:536-556; found at vmr_set_state):
  // such steps -- 99.99 % of a survey layer's -- take log(1 + eps), log(eps) from registers instead of 8 K bytes per tie from memory.
  int lp0;
  // Geo::farl: reports of levels beyond the LDS ones add nothing to H here (k_far_hist does, from the compact far lists, and
  // finishes nu): no global adds in the pass, and the grid's last workgroup leaves the nu sum standing.
  int farl;
  // Rounds of level 0 only (two-pass handles, whose statistics pass is bound by its LDS adds): every tie's reports of mirror count
  // >= 1 come first in its list (k_far_first with the level-0 rows as "near"), and sy's high half holds n1, the first round of a
  // step in which EVERY tie's report has mirror count 0.  At level 0 the weights are w1 = 1, w2 = 0, so the finalize kernels need
  // of H[0][m][k] the marginals only: sum_k = C[0][m] - deficits (the constant), sum_m = sum over ties of rho_k times the tie's
  // counts -- a per-tie product.  From round n1 on the general body adds nothing to the LDS table (but the deficits of irregular
  // ties) and sums those products into h0s [L][NSLOT][K]; k_level0_spread puts them into H with exactly those marginals.
  // null: level 0 like every level.  (Masking the level-0 LANES of mixed rounds instead was measured: an LDS instruction costs
  // the same with half its lanes off.)
  double* h0s;
  // ... and the statistics pass does not even READ those rounds: what it needs of them is the tie's summed count at mirror count 0,
  // a constant of the data kept by position (4 bytes per tie instead of 4 per report).  Steps with an irregular tie (a rho that does
  // not sum to 1: its deficits are per reporter) walk every round as before.  null: the rounds are walked for their counts.
  const unsigned* x0p;
  int lv0r;   // sy's high half holds that round (the handle's lists are ordered for it): the rho update takes a level-0 round's factors
              // as E log theta_m + E log lambda_k -- one table read per report instead of K
  const unsigned* Ez;   // 64 empty entries (the zeroed slack behind E): what the ring of a long step loads past the step's last round
  int elbo_cur;   // ELBO-only pass: the CURRENT G_nu, not the stale one -- nu was not committed since the rho it evaluates (split ELBO sweep of vmr_sweep_local)
#ifdef SL_DEBUG
  unsigned long long* dbg_t;   // [waves][8]: a wave's start, end of prologue, end of step loop, end; first loads issued, tables' barrier, nu share done, flush done (100 MHz clock)
#endif
};

struct SlShape { int tpb, yt, hc; size_t smem; };
// one handle's share of a launch that serves many (k_sweep_sl_b): its arguments, its first workgroup and how many it has
struct SlUnit { SlArgs a; Geo g; int blk0, nblk; };

// Register budget of a variant: the per-tie state grows with K (log prior, sums, rho, table rows: ~14 K registers) and the
// ELBO variants carry the logarithms' on top, so the kernels are compiled for fewer, fatter waves as K grows -- no variant
// spills (profiles/r03_kernel_resources.md).  Largest workgroup / waves per SIMD the kernel<K, ., ELBO, .> is compiled for:
// (allfull: every mask row is all ones -- the variants without the mask code need a few registers fewer)
#ifndef SL_LIGHT_UPD_K   // largest K whose all-ones update variant is compiled for 128 registers (4 waves per SIMD, 1024 threads)
#define SL_LIGHT_UPD_K 2
#endif
constexpr bool sl_light(int K, bool elbo, bool allfull, bool update) { return !elbo && allfull && (K <= SL_LIGHT_UPD_K || (!update && K <= 4)); }   // fits 128 registers
#ifndef SL_ELBO3_K   // ... and the largest K whose ELBO variants are
#define SL_ELBO3_K 2
#endif
#ifndef SL_ELBO3   // K = 2 ELBO variants at 3 waves per SIMD (168 registers, ~20 dwords spilled): 0.259 against 0.282 ms per config-3 launch at 2
#define SL_ELBO3 1
#endif
constexpr int sl_tpb_max(int K, bool elbo, bool allfull, bool update = true) {
  return elbo ? ((SL_ELBO3 && K <= SL_ELBO3_K) ? 768 : (K <= 4 ? 512 : 256)) : (sl_light(K, elbo, allfull, update) ? 1024 : (K <= 3 ? 768 : (K <= 7 ? 512 : 256)));
}
// ... and of the lockstep entry k_sweep_sl_b (its arguments come from memory: a few registers more than the same variant's own launch)
constexpr int sl_tpb_max_b(int K, bool elbo, bool allfull) { return elbo ? sl_tpb_max(K, true, allfull) : (K <= 2 ? 768 : (K <= 7 ? 512 : 256)); }
constexpr int sl_wpe_b(int K, bool elbo, bool allfull) { return elbo ? ((SL_ELBO3 && K <= SL_ELBO3_K) ? 3 : (K <= 4 ? 2 : 1)) : (K <= 2 ? 3 : (K <= 7 ? 2 : 1)); }
constexpr int sl_wpe(int K, bool elbo, bool allfull, bool update = true) {
  return elbo ? ((SL_ELBO3 && K <= SL_ELBO3_K) ? 3 : (K <= 4 ? 2 : 1)) : (sl_light(K, elbo, allfull, update) ? 4 : (K <= 3 ? 3 : (K <= 7 ? 2 : 1)));
}

// LDS bytes of one workgroup of the sweep kernel
static inline size_t sl_smem(const Geo& g, int yt, int hc, bool update, bool elbo, bool hist) {
  const size_t lb = (size_t)g.Mp * g.K * 8;
  return (update ? (size_t)yt * lb : 0) + (hist ? (size_t)hc * lb : 0) + (size_t)g.Mp * 8 * (update ? 2 : 1) +
         (size_t)g.W * 8 + 128 + (g.ml ? lb + (size_t)g.Mp * 8 : 0) + (size_t)(64 + (elbo ? 256 : 0)) * 8 + 16;   // (the logarithm table: ELBO variants only)
}

// mode: 0 = rho update (+ H), 1 = rho update + ELBO data terms, 2 = ELBO only, 3 = statistics only (do_hist 1 or 2),
//       4 = rho update (+ H) without writing rho (sweep_body's STORE = false)
typedef int (*sl_launch_fn)(vmr_ctx* h, int mode, const SlShape& sh, SlArgs& a);
sl_launch_fn vmr_sl_launcher(int K);   // null when K was not compiled in
// mode 0 / 1 (rho update / + ELBO data terms) of the units' sweeps in one launch on `st`; units / blk_unit: device memory
typedef int (*sl_launch_batch_fn)(vmr_ctx* h, hipStream_t st, int mode, int allfull, const SlUnit* units, const int* blk_unit, int nblocks,
                                  int tpb, size_t smem);
sl_launch_batch_fn vmr_sl_batch_launcher(int K);

// sorted_lists.hip
// Builds perm, rs, E (and ebase, n_slots) from tie-major entries.  rp [L][T+1]: per-tie report counts (overwritten by their
// exclusive scan); nl[l]: reports of layer l; etmp_all: every layer's entries tie-major (layer l at offset sum nl[<l]) or null,
// then fill(l, rp_l scanned, etmp) writes layer l's.
// Wide entries (Geo::wide): every entry is two words -- etmp2 / etmp2_all hold the second ones -- and h->EX receives them.
typedef std::function<void(int l, const unsigned* rpl, unsigned* etmp, unsigned* etmp2)> SlFill;
int sl_place_entries(vmr_ctx* h, unsigned* rp, const std::vector<unsigned long long>& nl, unsigned* etmp_all, unsigned* etmp2_all, const SlFill* fill);
// out[l][pos] = in[l][perm[l][pos]] for the per-tie arrays the sweeps read by position
int sl_permute_u8(vmr_ctx* h, const uint8_t* in, uint8_t* out);
int sl_permute_u32(vmr_ctx* h, const unsigned* in, unsigned* out);
// [L][T][K] doubles between tie order and position order (to_pos: out[pos] = in[perm[pos]]; else out[perm[pos]] = in[pos])
int sl_permute_rows(vmr_ctx* h, const double* in, double* out, bool to_pos);

#endif
