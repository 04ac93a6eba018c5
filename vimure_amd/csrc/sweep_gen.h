// sweep_gen.h -- the general CAVI kernels: any number of categories K (2..KGEN_MAX) and entries of any width.
//
// The specialised sweep (sweep_sl.hip) keeps a tie's K categories in one lane's registers and is compiled per K <= KMAX = 8, with
// counts packed into 11 bits of a 32-bit entry.  The reference takes more: without `K` it sets K = max(X) + 1 (model.py:179-197;
// its tests go that way, test/test_model.py:59-115) and its counts are int64 (utils.py:241-242).  Handles outside the specialised
// range (Geo::gen) run these kernels on the same sorted report lists (sweep_sl.h): a tie is spread over G = min(64, 2^ceil(log2 K))
// lanes, ONE CATEGORY PER LANE (K > 64: a few per lane), so K is a run-time number; table rows come from the formula instead of LDS
// copies, the statistics go to the global table directly.  Slower than the specialised kernels by design -- these inputs must be
// fitted right, not fast.
#ifndef VMR_SWEEP_GEN_H
#define VMR_SWEEP_GEN_H

#include "sweep_sl.h"

// statistics H, sums over all-ones mask rows (slotF) and over partial rows (slotA) of the CURRENT rho
int gen_hist(vmr_ctx* h);
// gamma (and, with_phi, phi) from them (model.py:698-761)
int gen_gamma(vmr_ctx* h, bool with_phi);
// the PHI sub-step's commit with mutuality on (model.py:729-761)
int gen_phi(vmr_ctx* h);
// mode 0: rho update (model.py:763-818) + statistics of the new rho; 1: + the ELBO's data terms (model.py:948-995); 2: ELBO only.
// Then k_fin_rho for nu (commit_nu / raw_nu) and the ELBO's assembly -- launched by the caller through `fin_rho`.
typedef int (*gen_fin_rho_fn)(vmr_ctx* h, int do_nu, int do_elbo, int skip_nu);
int gen_rho(vmr_ctx* h, int mode, bool commit_nu, bool raw_nu, gen_fin_rho_fn fin_rho);
// vmr_sample for K > KMAX
int gen_sample(vmr_ctx* h, unsigned long long seed, int n_trials, uint8_t* out_dev);

#endif
