/*
 * cavi_ref.c -- plain-C CPU oracle of the VIMuRe CAVI hot path.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * A second restatement (besides oracle/vimure_oracle.py) of the reference algorithm
 * latentnetworks/vimure src/python/vimure/model.py, used (a) by tests/ as an independent
 * checker and (b) by bench.py's `cpu_baseline` leg, because the NumPy restatement cannot
 * stream a config-3 sized layer (the reference itself cannot run it at all).
 * Pinned to the golden vectors dumped from the real reference: tests/test_c_oracle.py.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load this library.
 *
 * Layout: X uint8 [L,N,N,M] counts, R uint8 [L,N,N,M] 0/1 or NULL (= all ones),
 * rho/logpr double [L,N,N,K], gamma_* [L,M], phi_* [L,K].  OpenMP over ego rows i.
 *
 * Each function cites the reference lines it restates.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define KMAX 256

typedef struct {
  int L, N, M, K, mut;
  const uint8_t* X;
  const uint8_t* R; /* may be NULL */
  double eps;
  const double *a_th, *b_th; /* [L,M] */
  const double *a_la, *b_la; /* [L,K] */
  double a_eta, b_eta;
  double *gamma_shp, *gamma_rte; /* [L,M] */
  double *phi_shp, *phi_rte;     /* [L,K] */
  double nu_shp, nu_rte;
  double* rho;         /* [L,N,N,K] */
  const double* logpr; /* [L,N,N,K] */
  double g_nu_cache;   /* exp(E log nu) of the last cache refresh (model.py:684) */
} ref_state;

static double digamma(double x) {
  double r = 0.0;
  while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
  double f = 1.0 / (x * x);
  double t = f * (-1.0 / 12.0 + f * (1.0 / 120.0 + f * (-1.0 / 252.0 + f * (1.0 / 240.0 +
             f * (-1.0 / 132.0 + f * (691.0 / 32760.0 + f * (-1.0 / 12.0)))))));
  return r + log(x) - 0.5 / x + t;
}

int ref_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void ref_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* cache refresh, model.py:662-696: geometric expectations for one layer */
static void layer_cache(ref_state* s, int l, double* Gth, double* lth, double* Eth, double* Gla, double* lla,
                        double* Ela) {
  for (int m = 0; m < s->M; ++m) {
    double a = s->gamma_shp[l * s->M + m], b = s->gamma_rte[l * s->M + m];
    lth[m] = digamma(a) - log(b); Gth[m] = exp(lth[m]); Eth[m] = a / b;
  }
  for (int k = 0; k < s->K; ++k) {
    double a = s->phi_shp[l * s->K + k], b = s->phi_rte[l * s->K + k];
    lla[k] = digamma(a) - log(b); Gla[k] = exp(lla[k]); Ela[k] = a / b;
  }
}

static double refresh_gnu(ref_state* s) {
  if (s->mut) s->g_nu_cache = exp(digamma(s->nu_shp) - log(s->nu_rte)); /* model.py:684 */
  else s->g_nu_cache = 0.0;                                              /* model.py:600 */
  return s->g_nu_cache;
}

/* x * z1/(z1+z2) weight, model.py:685-693 */
static inline double w1(double z1, double z2) {
  double den = z1 + z2;
  if (den == 0.0) den = 1.0;
  return z1 / den;
}

/* model.py:698-727 + :832-859 */
void ref_update_gamma(ref_state* s) {
  const int L = s->L, N = s->N, M = s->M, K = s->K;
  const double gnu = refresh_gnu(s);
  for (int l = 0; l < L; ++l) {
    double *Gth = malloc(sizeof(double) * M * 3), *lth = Gth + M, *Eth = lth + M;
    double Gla[KMAX], lla[KMAX], Ela[KMAX];
    layer_cache(s, l, Gth, lth, Eth, Gla, lla, Ela);
    double* shp = calloc(M, sizeof(double));
    double* rte = calloc(M, sizeof(double));
#pragma omp parallel
    {
      double* shp_t = calloc(M, sizeof(double));
      double* rte_t = calloc(M, sizeof(double));
#pragma omp for schedule(static)
      for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
          const size_t t = ((size_t)l * N + i) * N + j, tT = ((size_t)l * N + j) * N + i;
          const uint8_t* x = s->X + t * M;
          const uint8_t* y = s->X + tT * M;
          const uint8_t* r = s->R ? s->R + t * M : NULL;
          const double* rho = s->rho + t * K;
          double e = 0.0;
          for (int k = 0; k < K; ++k) e += rho[k] * Ela[k];
          for (int m = 0; m < M; ++m) {
            if (!r || r[m]) rte_t[m] += e;
            if (x[m]) {
              double acc = 0.0;
              for (int k = 0; k < K; ++k)
                acc += rho[k] * (s->mut ? w1(Gth[m] * Gla[k], gnu * (double)y[m]) : 1.0);
              shp_t[m] += (double)x[m] * acc;
            }
          }
        }
#pragma omp critical
      for (int m = 0; m < M; ++m) { shp[m] += shp_t[m]; rte[m] += rte_t[m]; }
      free(shp_t); free(rte_t);
    }
    for (int m = 0; m < M; ++m) {
      s->gamma_shp[l * M + m] = s->a_th[l * M + m] + shp[m];
      s->gamma_rte[l * M + m] = s->b_th[l * M + m] + rte[m];
    }
    free(shp); free(rte); free(Gth);
  }
}

/* model.py:729-761 + :861-887 */
void ref_update_phi(ref_state* s) {
  const int L = s->L, N = s->N, M = s->M, K = s->K;
  const double gnu = refresh_gnu(s);
  for (int l = 0; l < L; ++l) {
    double *Gth = malloc(sizeof(double) * M * 3), *lth = Gth + M, *Eth = lth + M;
    double Gla[KMAX], lla[KMAX], Ela[KMAX];
    layer_cache(s, l, Gth, lth, Eth, Gla, lla, Ela);
    double shp[KMAX] = {0}, rte[KMAX] = {0};
#pragma omp parallel
    {
      double shp_t[KMAX] = {0}, rte_t[KMAX] = {0};
#pragma omp for schedule(static)
      for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
          const size_t t = ((size_t)l * N + i) * N + j, tT = ((size_t)l * N + j) * N + i;
          const uint8_t* x = s->X + t * M;
          const uint8_t* y = s->X + tT * M;
          const uint8_t* r = s->R ? s->R + t * M : NULL;
          const double* rho = s->rho + t * K;
          double T = 0.0;
          for (int m = 0; m < M; ++m) {
            if (!r || r[m]) T += Eth[m];
            if (x[m])
              for (int k = 0; k < K; ++k)
                shp_t[k] += rho[k] * (double)x[m] * (s->mut ? w1(Gth[m] * Gla[k], gnu * (double)y[m]) : 1.0);
          }
          for (int k = 0; k < K; ++k) rte_t[k] += rho[k] * T;
        }
#pragma omp critical
      for (int k = 0; k < K; ++k) { shp[k] += shp_t[k]; rte[k] += rte_t[k]; }
    }
    for (int k = 0; k < K; ++k) {
      s->phi_shp[l * K + k] = s->a_la[l * K + k] + shp[k];
      s->phi_rte[l * K + k] = s->b_la[l * K + k] + rte[k];
    }
    free(Gth);
  }
}

/* model.py:763-818 + :889-923, and :820-830 when do_nu (the nu update reads the new rho) */
static void rho_nu(ref_state* s, int do_rho, int do_nu) {
  const int L = s->L, N = s->N, M = s->M, K = s->K;
  const double gnu = refresh_gnu(s);
  double nu_acc = 0.0;
  for (int l = 0; l < L; ++l) {
    double *Gth = malloc(sizeof(double) * M * 3), *lth = Gth + M, *Eth = lth + M;
    double Gla[KMAX], lla[KMAX], Ela[KMAX];
    layer_cache(s, l, Gth, lth, Eth, Gla, lla, Ela);
#pragma omp parallel for schedule(static) reduction(+ : nu_acc)
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) {
        const size_t t = ((size_t)l * N + i) * N + j, tT = ((size_t)l * N + j) * N + i;
        const uint8_t* x = s->X + t * M;
        const uint8_t* y = s->X + tT * M;
        const uint8_t* r = s->R ? s->R + t * M : NULL;
        double* rho = s->rho + t * K;
        if (do_rho) {
          double T = 0.0, U[KMAX] = {0};
          for (int m = 0; m < M; ++m) {
            if (!r || r[m]) T += Eth[m];
            if (x[m])
              for (int k = 0; k < K; ++k)
                U[k] += (lth[m] + lla[k]) * ((double)x[m] * (s->mut ? w1(Gth[m] * Gla[k], gnu * (double)y[m]) : 1.0));
          }
          double sum = 0.0;
          for (int k = 0; k < K; ++k) { rho[k] = exp((s->logpr[t * K + k] + U[k]) - T * Ela[k]); sum += rho[k]; }
          if (sum > 0.0) for (int k = 0; k < K; ++k) rho[k] /= sum;
        }
        if (do_nu && s->mut)
          for (int m = 0; m < M; ++m)
            if (x[m] && y[m])
              for (int k = 0; k < K; ++k) {
                double z1 = Gth[m] * Gla[k], z2 = gnu * (double)y[m];
                nu_acc += (double)x[m] * (z2 / (z1 + z2)) * rho[k];
              }
      }
    free(Gth);
  }
  if (do_nu && s->mut) s->nu_shp = s->a_eta + nu_acc;
}

void ref_update_rho(ref_state* s) { rho_nu(s, 1, 0); }
void ref_update_nu(ref_state* s) { rho_nu(s, 0, 1); }

/* one sweep, model.py:623-660 (rho and nu share one pass over X: same arithmetic) */
void ref_cavi_step(ref_state* s) {
  ref_update_gamma(s);
  ref_update_phi(s);
  rho_nu(s, 1, 1);
}

static double gamma_term(double pa, double pb, double qa, double qb) { /* model.py:1300-1303 */
  return lgamma(qa) - pa * log(qb) + (pa - qa) * digamma(qa) + qa * (1.0 - pb / qb);
}

/* model.py:948-1019, 1220-1313 -- uses g_nu_cache (stale), exp(rho), log(eps) outside R */
double ref_elbo(ref_state* s) {
  const int L = s->L, N = s->N, M = s->M, K = s->K;
  const double gnu = s->g_nu_cache, Enu = s->nu_shp / s->nu_rte;
  double total = 0.0;
  for (int l = 0; l < L; ++l) {
    double *Gth = malloc(sizeof(double) * M * 3), *lth = Gth + M, *Eth = lth + M;
    double Gla[KMAX], lla[KMAX], Ela[KMAX];
    layer_cache(s, l, Gth, lth, Eth, Gla, lla, Ela);
    double acc = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : acc)
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) {
        const size_t t = ((size_t)l * N + i) * N + j, tT = ((size_t)l * N + j) * N + i;
        const uint8_t* x = s->X + t * M;
        const uint8_t* y = s->X + tT * M;
        const uint8_t* r = s->R ? s->R + t * M : NULL;
        const double* rho = s->rho + t * K;
        double T = 0.0, Q = 0.0, er[KMAX], sr = 0.0, se = 0.0, ent = 0.0;
        for (int k = 0; k < K; ++k) {
          er[k] = exp(rho[k]); sr += rho[k]; se += rho[k] * Ela[k];
          ent += rho[k] * s->logpr[t * K + k] - rho[k] * log(rho[k] + s->eps);
        }
        for (int m = 0; m < M; ++m) {
          int in_r = (!r || r[m]);
          double yt = s->mut ? (double)y[m] : 0.0; /* mutuality off: X^T is all zero, model.py:145 */
          if (in_r) { T += Eth[m]; Q += yt; }
          if (x[m]) {
            double inner = 0.0;
            if (in_r) for (int k = 0; k < K; ++k) inner += er[k] * (Gth[m] * Gla[k] + gnu * yt);
            acc += (double)x[m] * log(inner + s->eps);
          }
        }
        acc += ent - se * T - Enu * sr * Q;
      }
    total += acc;
    for (int m = 0; m < M; ++m)
      total += gamma_term(s->a_th[l * M + m], s->b_th[l * M + m], s->gamma_shp[l * M + m], s->gamma_rte[l * M + m]);
    for (int k = 0; k < K; ++k)
      total += gamma_term(s->a_la[l * K + k], s->b_la[l * K + k], s->phi_shp[l * K + k], s->phi_rte[l * K + k]);
    free(Gth);
  }
  total += gamma_term(s->a_eta, s->b_eta, s->nu_shp, s->nu_rte);
  return total;
}
