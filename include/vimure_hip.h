/*
 * vimure_hip.h -- C-ABI of libvimure_hip.so, the MI355X (gfx950) CAVI engine for the
 * VIMuRe latent-network model.
 *
 * The reference (latentnetworks/vimure) has no FFI: its hot path is a set of private
 * methods of `VimureModel` that mutate NumPy arrays held in `self`
 * (src/python/vimure/model.py).  This header is the seam a maintainer would bind with
 * ctypes; every entry point names the reference code it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; all floating point is IEEE double, as in the reference;
 *   - arrays are C-order: X,R [L,N,N,M] (layer, ego, alter, reporter), rho/pr_rho [L,N,N,K],
 *     gamma_* [L,M], phi_* [L,K];
 *   - every function returns 0 on success and a negative VMR_E* code on failure;
 *     vmr_last_error() gives the message (pass NULL for errors of vmr_create);
 *   - one handle <-> one device <-> one HIP stream; a handle is not re-entrant, distinct
 *     handles are independent (that is the multi-GPU model: one process per GPU);
 *   - host pointers are only read/written during the call; the library owns all device memory.
 */
#ifndef VIMURE_HIP_H
#define VIMURE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vmr_ctx* vmr_handle;

enum {
  VMR_OK = 0,
  VMR_EINVAL = -1,   /* bad argument / unsupported size (ValueError in the host class) */
  VMR_EHIP = -2,     /* HIP runtime error (RuntimeError) */
  VMR_ENAN = -3,     /* ELBO is NaN (model.py:1015-1016 raises ValueError("ELBO is NaN!!!!")) */
  VMR_ESTATE = -4    /* call order violated (priors/state not set) */
};

/* sub-steps of one sweep, for vmr_sub_step (model.py:643-656) */
enum { VMR_STEP_GAMMA = 0, VMR_STEP_PHI = 1, VMR_STEP_RHO = 2, VMR_STEP_NU = 3 };

/* read-out methods for vmr_readout (model.py:1099-1188) */
enum { VMR_READ_RHO_MAX = 0, VMR_READ_RHO_MEAN = 1, VMR_READ_THRESHOLD = 2 };

/* kernel classes for vmr_profile_read */
enum {
  VMR_KERNEL_GAMMA_MASK = 0,   /* masked reduction over R: A[l,m,k] = sum_ij R rho           */
  VMR_KERNEL_GAMMA_COUNTS = 1, /* sweep over X for gamma_shp (and phi_shp when mutuality off) */
  VMR_KERNEL_PHI = 2,          /* sweep over X for phi_shp (mutuality on)                     */
  VMR_KERNEL_RHO = 3,          /* sweep over X,R: rho update (+ nu partial)                   */
  VMR_KERNEL_ELBO = 4,         /* stand-alone ELBO sweep                                      */
  VMR_KERNEL_FINALIZE = 5,     /* the small reduce/parameter kernels                          */
  VMR_KERNEL_RHO_ELBO = 6,     /* rho update with the ELBO data terms reduced in the same pass */
  VMR_KERNEL_RHO_NOSTORE = 7,  /* rho update whose rho is used (statistics, nu) but not written: the inner sweeps of a vmr_step call, whose rho
                                  the next sweep overwrites unread */
  VMR_KERNEL_COUNT = 8
};

/*
 * Dataset handle.  Replaces the data set-up half of `__check_fit_params`
 * (model.py:134-213).  The dense count tensor is turned into REPORT LISTS on the device -- one 4-byte entry per non-zero
 * count carrying the mirrored count X[l,j,i,m], so X^T (model.py:141-161 `data_T`, `data_T_vals`) is never
 * materialised -- and then freed; tensors that are not sparse enough (or with M > 8192) stay as
 * dense uint8 tiles with R packed to one bit per (l,i,j,m).  vmr_data_format tells which.
 *   X  [L,N,N,M] uint8 counts (values <= 255).
 *   R  [L,N,N,M] uint8 0/1, or NULL = every reporter may report on every tie
 *      (model.py:206-211 default).
 *   data_on_device != 0: X and R are device pointers on `device` (e.g. torch tensors).
 *   mutuality: 0/1 (model.py:60-65; the host class forces 0 for undirected networks).
 *   eps: the EPS white-noise constant (model.py:215-218), normally 1e-12.
 * The handle is reusable across realisations and seeds (vmr_set_state restarts it).
 * Environment, read here: VMR_DETERMINISTIC=1 makes the handle's sweeps bit-reproducible run to run (the reference's
 * single-threaded NumPy, model.py:623-660, is): cross-workgroup sums as 64-bit integers in fixed point, one wave per workgroup
 * with a fixed share of the work; report lists only (VMR_EINVAL otherwise), about a fifth of the default speed, vmr_sub_step
 * returns VMR_ESTATE.
 */
int vmr_create(vmr_handle* out, int device, int L, int N, int M, int K, int mutuality,
               const uint8_t* X, const uint8_t* R, int data_on_device, double eps);

/*
 * The same dataset handle from COORDINATE LISTS -- what the reference actually holds: `X.subs` / `X.vals` of the
 * sptensor built by `read_from_edgelist` (_io.py:132-295) or `preprocess` (utils.py:220-248), and `R.subs` of a sparse
 * reporter mask; `__check_fit_params` derives `data_T_vals` from them with an O(nnz^2) lookup (model.py:148-161).  No dense
 * [L,N,N,M] tensor is built anywhere: the lists are sorted on the device and become the report lists directly.
 *   nx reports: xl, xi, xj, xm (int32 subscripts) and xv (counts, 1..2047); no duplicates.
 *   nr mask entries rl, ri, rj, rm (R = 1 there, 0 elsewhere); nr < 0: every reporter may report on every tie
 *   (model.py:206-211).
 *   data_on_device != 0: all index arrays are device pointers on `device`.
 * Needs M <= 8192 and (largest count + 1) * M <= 2^20; wider tensors go through vmr_create.
 */
int vmr_create_coo(vmr_handle* out, int device, int L, int N, int M, int K, int mutuality,
                   int64_t nx, const int32_t* xl, const int32_t* xi, const int32_t* xj, const int32_t* xm, const int32_t* xv,
                   int64_t nr, const int32_t* rl, const int32_t* ri, const int32_t* rj, const int32_t* rm,
                   int data_on_device, double eps);

void vmr_destroy(vmr_handle h);

const char* vmr_last_error(vmr_handle h);

/* Data statistics the host-side initialisation needs:
 *   sum_x     = X.vals.sum()  (model.py:174, enters nu_rte at model.py:593-595)
 *   coverage  [L,N,N] uint8, 1 iff the tie has at least one R entry AND at least one
 *             non-zero report; ties with 0 get the one-hot rho prior (model.py:508-556).
 *             May be NULL. */
int vmr_data_stats(vmr_handle h, double* sum_x, uint8_t* coverage);

/* Hyper-parameters, already broadcast to full arrays by the host (model.py:238-317):
 * alpha/beta_theta [L,M], alpha/beta_lambda [L,K], eta prior scalars. */
int vmr_set_priors(vmr_handle h, const double* alpha_theta, const double* beta_theta,
                   const double* alpha_lambda, const double* beta_lambda,
                   double alpha_eta, double beta_eta);

/* Start of a realisation: `_initialize_priors` + `_initialize_old_variables`
 * (model.py:561-617).  The host draws the values from RandomState (model.py:470-482,
 * 570-592) so fixed-seed fits match the reference; the library sets rho <- pr_rho and
 * logpr_rho <- log(pr_rho + eps) (model.py:559, 602).
 * pr_rho_on_device != 0: pr_rho is a device pointer. */
int vmr_set_state(vmr_handle h, const double* gamma_shp, const double* gamma_rte,
                  const double* phi_shp, const double* phi_rte, double nu_shp, double nu_rte,
                  const double* pr_rho, int pr_rho_on_device);

/* n_iters full sweeps of `_update_CAVI` (model.py:623-660): gamma -> phi -> rho -> nu, each
 * with the cache refresh of model.py:662-696 fused in.  Asynchronous on the handle's stream
 * unless elbo_out != NULL, in which case the ELBO (`__ELBO`, model.py:948-1019) is reduced
 * inside the last sweep's rho pass and returned (this synchronises). */
int vmr_step(vmr_handle h, int n_iters, double* elbo_out);

/* The convergence loop of `fit` for the realisation set up by vmr_set_state (model.py:405-426 with the stop rule of
 * `_check_for_convergence`, model.py:1021-1056): sweeps until the ELBO -- evaluated at iteration 1, every 10th and
 * max_iter -- has changed by less than `tol` on more than `decision` consecutive evaluations, or max_iter is reached.
 * Trace rows as the reference appends them (iterations that are multiples of 10, model.py:423-426): iteration, ELBO,
 * wall time of that iteration's sweep in seconds, reached flag; at most `cap` rows.  *elbo = last ELBO, *iters =
 * iterations run, *converged = the stop rule fired.  One call per realisation instead of one per iteration: host
 * threads that drive several small fits at once then meet in the GPU, not in the caller's interpreter. */
int vmr_fit_loop(vmr_handle h, int max_iter, double tol, int decision, int cap, int* n_rows, int* row_iter, double* row_elbo,
                 double* row_runtime, int* row_reached, double* elbo, int* iters, int* converged);

/* The same loop for n handles in lockstep -- the realisations of many small fits (the reference's Karnataka experiment,
 * notebooks/python/experiments/karnataka.py:170-191: a village layer of N = 200-800 is two dependent 20-40 us launches per
 * sweep that leave the GPU nearly empty).  Every handle must have had its vmr_set_state.  The handles of the first one's kind
 * (report lists in one pass, same K / mutuality / mask kind, same device) share ONE launch of each kernel per sweep, their
 * ELBOs come back in one copy, a handle that converges leaves the launch; handles of another kind run their loops one after the
 * other.  Each handle's results are exactly those of vmr_fit_loop.  Arrays are per handle: n_rows[n], row_*[n * cap] (handle u
 * at u * cap), elbo[n], iters[n], converged[n], rc[n] (a handle's own return code; its message via vmr_last_error).  Returns the
 * first non-zero rc, else VMR_OK.  Replaces n calls of model.py:405-426 running side by side. */
int vmr_fit_loop_batch(vmr_handle* hs, int n, int max_iter, double tol, int decision, int cap, int* n_rows, int* row_iter,
                       double* row_elbo, double* row_runtime, int* row_reached, double* elbo, int* iters, int* converged, int* rc);

/* Stand-alone ELBO of the current state (model.py:948-1019 incl. the stale G_exp_nu of
 * model.py:970).  Synchronises.  Returns VMR_ENAN when the value is NaN. */
int vmr_elbo(vmr_handle h, double* out);

/* Fits whose LAYERS are spread over several handles / GPUs (BASELINE config 5).  nu is one scalar shared by
 * all layers (model.py:589-596, 822-825) and the ELBO stop rule is joint (model.py:1039-1047), so the owners
 * exchange three doubles per sweep:
 *   vmr_sweep_local runs gamma, phi, rho on the local layers, does NOT commit nu, synchronises and returns
 *     out3[0] = local sum x w2 rho (the local part of nu_shp - alpha_eta),
 *     out3[1] = local ELBO terms that do not involve nu (only when want_elbo; data, entropy, theta/lambda Gamma terms),
 *     out3[2] = local sum_t (sum_k rho_k) Q_t  (enters the ELBO as -E[nu] * total; only when want_elbo);
 *   the caller sums the three over all owners (2-3 doubles all-reduce, latency-bound) and gives every owner
 *   vmr_commit_nu(total of out3[0]).  ELBO = total out3[1] - E[nu] * total out3[2] + Gamma term of nu
 *   (model.py:1006-1011), with nu_rte = beta_eta + sum of X over ALL layers set through vmr_set_state. */
int vmr_sweep_local(vmr_handle h, int want_elbo, double* out3);
int vmr_commit_nu(vmr_handle h, double nu_partial_total);

/* The same exchange without a host hop per sweep (RCCL on GPUs): vmr_sweep_local_dev queues the sweep and leaves the three
 * doubles in the caller's DEVICE buffer out3_dev (asynchronous on the handle's stream); the caller all-reduces that buffer
 * ON THE HANDLE'S STREAM (vmr_stream returns the hipStream_t; e.g. torch.cuda.ExternalStream) and vmr_commit_nu_dev reads the
 * total from device memory, again on that stream.  Only ELBO evaluations (every 10th sweep) bring numbers to the host. */
int vmr_sweep_local_dev(vmr_handle h, int want_elbo, double* out3_dev);
int vmr_commit_nu_dev(vmr_handle h, const double* nu_partial_total_dev);
void* vmr_stream(vmr_handle h);

/* One update of a sweep (test hook for step-level parity with model.py:643-656). */
int vmr_sub_step(vmr_handle h, int which);

/* Copy the posteriors to the host: what `_update_optimal_parameters` snapshots
 * (model.py:925-942).  Any pointer may be NULL.  Synchronises. */
int vmr_get_state(vmr_handle h, double* gamma_shp, double* gamma_rte, double* phi_shp,
                  double* phi_rte, double* nu_shp, double* nu_rte, double* rho);

/* Best-realisation bookkeeping on the device: vmr_snapshot keeps a copy of rho and of every parameter -- what
 * `_update_optimal_parameters` copies into the `*_f` attributes (model.py:925-942), without moving rho (8 L N^2 K bytes)
 * to the host after every realisation; vmr_restore makes the snapshot the current state again, so that
 * vmr_get_state / vmr_get_geometric read the best realisation once, at the end of `fit`.  Asynchronous on the
 * handle's stream.  The snapshot does not hold the log prior of its realisation: after vmr_restore the state can be read
 * (vmr_get_*, vmr_readout, vmr_sample) but the sweeping entry points return VMR_ESTATE until the next vmr_set_state. */
int vmr_snapshot(vmr_handle h);
int vmr_restore(vmr_handle h);

/* Posterior read-out of the CURRENT rho on the device -- `get_inferred_model` (model.py:1099-1188) and
 * `apply_rho_threshold` (utils.py:207-217) -- so that a 16 MB answer crosses PCIe instead of the 256 MB of rho:
 *   VMR_READ_RHO_MAX    out uint8 [L,N,N]   argmax_k rho (first maximum, as np.argmax)
 *   VMR_READ_RHO_MEAN   out double [L,N,N]  sum_k k rho_k
 *   VMR_READ_THRESHOLD  out uint8 [L,N,N]   rho[...,1] >= threshold
 * out_on_device != 0: `out` is a device pointer.  Synchronises. */
int vmr_readout(vmr_handle h, int method, double threshold, void* out, int out_on_device);

/* Posterior samples of Y on the device -- `sample_inferred_model` (model.py:1062-1096; also the draw of
 * `PosteriorSyntheticNetwork`, synthetic.py:964-1177): per tie, n_trials categorical trials from the CURRENT rho and the most
 * frequent category (first maximum), i.e. Generator.multinomial(n_trials, rho).argmax(-1).  out uint8 [L,N,N].  The uniforms
 * are Philox4x32-10 with key = seed and counter = (tie index, trial): reproducible for a seed, independent of the data
 * layout, NOT NumPy's PCG64 stream (the host class keeps that exact mode).  rho (8 L N^2 K bytes) stays on the device.
 * out_on_device != 0: `out` is a device pointer.  Synchronises. */
int vmr_sample(vmr_handle h, uint64_t seed, int n_trials, uint8_t* out, int out_on_device);

/* exp(E[log .]) of theta [L,M], lambda [L,K], nu from the current shape/rate parameters
 * (model.py:676-684), plus g_nu_cache = the G_exp_nu the last cache refresh held, i.e. the
 * value computed BEFORE the last nu update -- what `model.G_exp_nu` reads after `fit` and what
 * the ELBO uses (model.py:684 vs :822, :970).  Any pointer may be NULL. */
int vmr_get_geometric(vmr_handle h, double* g_theta, double* g_lambda, double* g_nu, double* g_nu_cache);

/* Wait for all work queued on the handle's stream. */
int vmr_sync(vmr_handle h);

/* Per-kernel-class timing with HIP events on the handle's stream (for bench.py's roofline
 * line).  enable=1 starts recording and clears totals; enable=2 records only the passes over the data (classes
 * GAMMA_COUNTS, RHO, ELBO, RHO_ELBO), not the small finalize kernels.  vmr_profile_read synchronises. */
int vmr_profile(vmr_handle h, int enable);
int vmr_profile_read(vmr_handle h, int kernel_class, double* total_ms, int64_t* launches);

/* Algorithmic bytes one launch of a kernel class moves under the handle's data format, see DESIGN.md:
 * report lists (4 B per non-zero count, 4 B per tie, mask words of partial rows, rho/logpr_rho 8 B) or the
 * dense encoding (X 1 B/elt, R 1 bit/elt, rho/logpr_rho 8 B). */
int vmr_kernel_bytes(vmr_handle h, int kernel_class, double* bytes);

/* Data format chosen by vmr_create: *sparse = 1 for report lists (default whenever they are at most half the
 * dense bytes; env VMR_FORMAT=dense|sparse overrides), 0 for the dense tiles.  *nnz = non-zero counts in X
 * (0 when the dense format was forced before counting).  Either pointer may be NULL. */
int vmr_data_format(vmr_handle h, int* sparse, uint64_t* nnz);

/* Mask layout: *lists = 1 when the partial rows of R are also held as short reporter lists (rows with few
 * reporters, e.g. the self-reporter mask of _io.py:230-242; env VMR_NO_RLISTS=1 disables), *listed = reporters in
 * those lists.  The bit-packed mask is kept either way.  Either pointer may be NULL. */
int vmr_mask_format(vmr_handle h, int* lists, uint64_t* listed);

/* Shape of a sweep over the report lists (no reference equivalent; what `_update_CAVI`, model.py:623-660, costs here):
 * *passes = passes over the entries per sweep (1, or 2 where the tables of a wide reporter dimension do not fit in LDS side
 * by side), *lds_levels = mirror-count levels of the statistics H the pass keeps in LDS, *far_reports = reports of the
 * levels beyond that a one-pass handle built with env VMR_FARL=1 also keeps as a compact list (k_far_hist adds their
 * statistics after the pass; default: two passes instead).  Dense tiles: 1 (2), the LDS levels, 0.  Any pointer may be NULL. */
int vmr_sweep_shape(vmr_handle h, int* passes, int* lds_levels, uint64_t* far_reports);

/* Synthetic generators on the device (no handle: they produce the inputs of vmr_create).
 * vmr_generate_y replaces the ground-truth draw of the reference's StandardSBM (synthetic.py:548-571, 639-667):
 * Y[l,i,j] ~ Poisson(w[grp[i]][grp[j]]) clipped to K - 1, zero diagonal.  w [C][C] and grp [N] are host arrays, Y_dev a
 * uint8 [L][N][N] array in device memory.
 * vmr_generate_x replaces `_build_X` (synthetic.py:63-352; the pair-wise draw :159-231): for every unordered pair a fair coin
 * picks the direction drawn first, first ~ Poisson((own + eta * mirror) / (1 - eta^2)), second ~ Poisson(own + eta * first), with
 * own = lambda * theta[l,m] in float64; lambda from Y_dev (0.01 where Y = 0, else Y, or 0.01 + lambda_diff when lambda_diff > 0;
 * synthetic.py:140-157) or given as lam_dev (double [L][N][N], device; Y_dev may then be NULL).  theta [L][M] is a host array.
 * self_reporter != 0: only a tie's own two nodes report (`_io.py:230-242`; M == N) and X_dev must come zeroed.
 * X_dev: uint8 [L][N][N][M] in device memory, counts clamped to 255 -- what vmr_create(data_on_device = 1) takes.
 * Counter-based stream (Philox4x32-10 keyed by seed; counter = layer, pair, reporter): a draw depends on (seed, l, i, j, m) only.
 * NOT the reference's RandomState stream: the host classes keep that exact mode (vimure_amd/synthetic.py).  Both synchronise. */
int vmr_generate_y(int device, int L, int N, int K, int C, const double* w, const int32_t* grp, uint64_t seed, uint8_t* Y_dev);
int vmr_generate_x(int device, int L, int N, int M, const uint8_t* Y_dev, const double* lam_dev, const double* theta, double eta,
                   double lambda_diff, uint64_t seed, int self_reporter, uint8_t* X_dev);

/* Library/version string. */
const char* vmr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VIMURE_HIP_H */
