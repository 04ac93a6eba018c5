/*
 * cavi_coo.c -- plain-C CPU oracle of the VIMuRe CAVI hot path over COORDINATE LISTS.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Third restatement of latentnetworks/vimure src/python/vimure/model.py (besides oracle/vimure_oracle.py, dense
 * NumPy, and oracle/cavi_ref.c, dense C).  The reference itself works on coordinate lists: `subs_nz` / `X.vals` /
 * `data_T_vals` for X (model.py:136-171) and `R.subs` for a sparse reporter mask (model.py:712-718, 737-749,
 * 772-792); this file keeps that formulation, so it can check the engine at sizes where a dense [L,N,N,M] tensor
 * on the host is out of reach (BASELINE configs[2] at full size, the config-5 regime) and the engine's COO entry
 * point (vmr_create_coo).  Pinned to the golden vectors dumped from the real reference and to cavi_ref.c:
 * tests/test_coo_oracle.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load this library.
 *
 * Inputs: X as (tie, reporter, count) triples sorted by (tie, reporter), tie = (l*N + i)*N + j; R either "all ones"
 * (the reference's default mask, model.py:206-211) or (tie, reporter) pairs sorted the same way.
 * rho/logpr double [L,N,N,K], gamma_* [L,M], phi_* [L,K].  OpenMP over ties.
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
  int64_t nx;
  const int64_t* xt; /* tie of every non-zero count */
  const int32_t* xm; /* its reporter */
  const int32_t* xv; /* the count */
  int r_all;         /* 1: every reporter may report on every tie */
  int64_t nr;
  const int64_t* rt; /* tie of every mask entry (r_all == 0) */
  const int32_t* rm;
  double eps;
  const double *a_th, *b_th, *a_la, *b_la;
  double a_eta, b_eta;
  double *gamma_shp, *gamma_rte, *phi_shp, *phi_rte;
  double nu_shp, nu_rte;
  double* rho;
  const double* logpr;
  double g_nu_cache;
  /* derived by coo_prepare (owned by the library, freed by coo_release) */
  int64_t* xp;  /* [T+1] first X entry of every tie */
  int64_t* rpn; /* [T+1] first mask entry of every tie (r_all == 0) */
  int32_t* xy;  /* data_T_vals: X[l,j,i,m] for every X entry (model.py:141-161) */
  uint8_t* xin; /* the X entry lies inside R */
  int64_t* q;   /* [T] sum over the mask entries m of tie t of X[l,j,i,m]: the ELBO's X^T sum (model.py:1257-1291) */
} coo_state;

static double digamma(double x) {
  double r = 0.0;
  while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
  double f = 1.0 / (x * x);
  double t = f * (-1.0 / 12.0 + f * (1.0 / 120.0 + f * (-1.0 / 252.0 + f * (1.0 / 240.0 +
             f * (-1.0 / 132.0 + f * (691.0 / 32760.0 + f * (-1.0 / 12.0)))))));
  return r + log(x) - 0.5 / x + t;
}

int coo_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static int64_t find_m(const int32_t* m, int64_t a, int64_t b, int32_t key) { /* index of key in the sorted run [a,b) or -1 */
  while (a < b) {
    int64_t c = a + (b - a) / 2;
    if (m[c] < key) a = c + 1; else b = c;
  }
  return a;
}

static int64_t* csr(const int64_t* t, int64_t n, int64_t T) {
  int64_t* p = calloc((size_t)T + 1, sizeof(int64_t));
  for (int64_t e = 0; e < n; ++e) p[t[e] + 1]++;
  for (int64_t u = 0; u < T; ++u) p[u + 1] += p[u];
  return p;
}

/* data set-up: row pointers, data_T_vals, mask membership, mirror sums (model.py:136-171, :977-992, :1257-1291) */
int coo_prepare(coo_state* s) {
  const int64_t N = s->N, T = (int64_t)s->L * N * N;
  for (int64_t e = 1; e < s->nx; ++e)
    if (s->xt[e] < s->xt[e - 1] || (s->xt[e] == s->xt[e - 1] && s->xm[e] <= s->xm[e - 1])) return -1;
  for (int64_t e = 1; e < s->nr; ++e)
    if (s->rt[e] < s->rt[e - 1] || (s->rt[e] == s->rt[e - 1] && s->rm[e] <= s->rm[e - 1])) return -2;
  s->xp = csr(s->xt, s->nx, T);
  s->rpn = s->r_all ? NULL : csr(s->rt, s->nr, T);
  s->xy = calloc((size_t)(s->nx ? s->nx : 1), sizeof(int32_t));
  s->xin = calloc((size_t)(s->nx ? s->nx : 1), 1);
  s->q = calloc((size_t)T, sizeof(int64_t));
#pragma omp parallel for schedule(dynamic, 4096)
  for (int64_t t = 0; t < T; ++t) {
    const int64_t l = t / (N * N), ij = t - l * N * N, i = ij / N, j = ij - i * N, tT = (l * N + j) * N + i;
    for (int64_t e = s->xp[t]; e < s->xp[t + 1]; ++e) {
      const int64_t f = find_m(s->xm, s->xp[tT], s->xp[tT + 1], s->xm[e]);
      s->xy[e] = (f < s->xp[tT + 1] && s->xm[f] == s->xm[e]) ? s->xv[f] : 0;
      if (s->r_all) s->xin[e] = 1;
      else {
        const int64_t g = find_m(s->rm, s->rpn[t], s->rpn[t + 1], s->xm[e]);
        s->xin[e] = (g < s->rpn[t + 1] && s->rm[g] == s->xm[e]) ? 1 : 0;
      }
    }
    int64_t acc = 0;
    if (s->r_all) {
      for (int64_t e = s->xp[tT]; e < s->xp[tT + 1]; ++e) acc += s->xv[e];
    } else {
      for (int64_t g = s->rpn[t]; g < s->rpn[t + 1]; ++g) {
        const int64_t f = find_m(s->xm, s->xp[tT], s->xp[tT + 1], s->rm[g]);
        if (f < s->xp[tT + 1] && s->xm[f] == s->rm[g]) acc += s->xv[f];
      }
    }
    s->q[t] = acc;
  }
  return 0;
}

void coo_release(coo_state* s) {
  free(s->xp); free(s->rpn); free(s->xy); free(s->xin); free(s->q);
  s->xp = s->rpn = NULL; s->xy = NULL; s->xin = NULL; s->q = NULL;
}

typedef struct { double *Gth, *lth, *Eth; double Gla[KMAX], lla[KMAX], Ela[KMAX]; double Tfull; } cache_t;

/* cache refresh, model.py:662-696 */
static void layer_cache(const coo_state* s, int l, cache_t* c) {
  c->Gth = malloc(sizeof(double) * s->M * 3); c->lth = c->Gth + s->M; c->Eth = c->lth + s->M;
  c->Tfull = 0.0;
  for (int m = 0; m < s->M; ++m) {
    double a = s->gamma_shp[l * s->M + m], b = s->gamma_rte[l * s->M + m];
    c->lth[m] = digamma(a) - log(b); c->Gth[m] = exp(c->lth[m]); c->Eth[m] = a / b;
    c->Tfull += c->Eth[m];
  }
  for (int k = 0; k < s->K; ++k) {
    double a = s->phi_shp[l * s->K + k], b = s->phi_rte[l * s->K + k];
    c->lla[k] = digamma(a) - log(b); c->Gla[k] = exp(c->lla[k]); c->Ela[k] = a / b;
  }
}

static double refresh_gnu(coo_state* s) {
  s->g_nu_cache = s->mut ? exp(digamma(s->nu_shp) - log(s->nu_rte)) : 0.0; /* model.py:684, :600 */
  return s->g_nu_cache;
}

static inline double w1(double z1, double z2) { /* model.py:685-693 */
  double den = z1 + z2;
  if (den == 0.0) den = 1.0;
  return z1 / den;
}

/* sum over the mask entries of tie t of v[m] (all-ones mask: the precomputed total) */
static inline double mask_sum(const coo_state* s, int64_t t, const double* v, double vfull) {
  if (s->r_all) return vfull;
  double a = 0.0;
  for (int64_t g = s->rpn[t]; g < s->rpn[t + 1]; ++g) a += v[s->rm[g]];
  return a;
}

/* model.py:698-727 + :832-859 */
void coo_update_gamma(coo_state* s) {
  const int M = s->M, K = s->K;
  const int64_t NN = (int64_t)s->N * s->N;
  const double gnu = refresh_gnu(s);
  for (int l = 0; l < s->L; ++l) {
    cache_t c; layer_cache(s, l, &c);
    double* shp = calloc(M, sizeof(double));
    double* rte = calloc(M, sizeof(double));
    double eall = 0.0;
#pragma omp parallel
    {
      double* shp_t = calloc(M, sizeof(double));
      double* rte_t = calloc(M, sizeof(double));
      double e_t = 0.0;
#pragma omp for schedule(static)
      for (int64_t u = 0; u < NN; ++u) {
        const int64_t t = (int64_t)l * NN + u;
        const double* rho = s->rho + t * K;
        double e = 0.0;
        for (int k = 0; k < K; ++k) e += rho[k] * c.Ela[k];
        if (s->r_all) e_t += e;
        else for (int64_t g = s->rpn[t]; g < s->rpn[t + 1]; ++g) rte_t[s->rm[g]] += e;
        for (int64_t q = s->xp[t]; q < s->xp[t + 1]; ++q) {
          const int m = s->xm[q];
          double acc = 0.0;
          for (int k = 0; k < K; ++k) acc += rho[k] * (s->mut ? w1(c.Gth[m] * c.Gla[k], gnu * (double)s->xy[q]) : 1.0);
          shp_t[m] += (double)s->xv[q] * acc;
        }
      }
#pragma omp critical
      { for (int m = 0; m < M; ++m) { shp[m] += shp_t[m]; rte[m] += rte_t[m]; } eall += e_t; }
      free(shp_t); free(rte_t);
    }
    for (int m = 0; m < M; ++m) {
      s->gamma_shp[l * M + m] = s->a_th[l * M + m] + shp[m];
      s->gamma_rte[l * M + m] = s->b_th[l * M + m] + rte[m] + eall;
    }
    free(shp); free(rte); free(c.Gth);
  }
}

/* model.py:729-761 + :861-887 */
void coo_update_phi(coo_state* s) {
  const int K = s->K;
  const int64_t NN = (int64_t)s->N * s->N;
  const double gnu = refresh_gnu(s);
  for (int l = 0; l < s->L; ++l) {
    cache_t c; layer_cache(s, l, &c);
    double shp[KMAX] = {0}, rte[KMAX] = {0};
#pragma omp parallel
    {
      double shp_t[KMAX] = {0}, rte_t[KMAX] = {0};
#pragma omp for schedule(static)
      for (int64_t u = 0; u < NN; ++u) {
        const int64_t t = (int64_t)l * NN + u;
        const double* rho = s->rho + t * K;
        const double T = mask_sum(s, t, c.Eth, c.Tfull);
        for (int64_t q = s->xp[t]; q < s->xp[t + 1]; ++q) {
          const int m = s->xm[q];
          for (int k = 0; k < K; ++k)
            shp_t[k] += rho[k] * (double)s->xv[q] * (s->mut ? w1(c.Gth[m] * c.Gla[k], gnu * (double)s->xy[q]) : 1.0);
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
    free(c.Gth);
  }
}

/* model.py:763-818 + :889-923, and :820-830 when do_nu (the nu update reads the new rho) */
static void rho_nu(coo_state* s, int do_rho, int do_nu) {
  const int K = s->K;
  const int64_t NN = (int64_t)s->N * s->N;
  const double gnu = refresh_gnu(s);
  double nu_acc = 0.0;
  for (int l = 0; l < s->L; ++l) {
    cache_t c; layer_cache(s, l, &c);
#pragma omp parallel for schedule(static) reduction(+ : nu_acc)
    for (int64_t u = 0; u < NN; ++u) {
      const int64_t t = (int64_t)l * NN + u;
      double* rho = s->rho + t * K;
      if (do_rho) {
        const double T = mask_sum(s, t, c.Eth, c.Tfull);
        double U[KMAX] = {0};
        for (int64_t q = s->xp[t]; q < s->xp[t + 1]; ++q) {
          const int m = s->xm[q];
          for (int k = 0; k < K; ++k)
            U[k] += (c.lth[m] + c.lla[k]) * ((double)s->xv[q] * (s->mut ? w1(c.Gth[m] * c.Gla[k], gnu * (double)s->xy[q]) : 1.0));
        }
        double sum = 0.0;
        for (int k = 0; k < K; ++k) { rho[k] = exp((s->logpr[t * K + k] + U[k]) - T * c.Ela[k]); sum += rho[k]; }
        if (sum > 0.0) for (int k = 0; k < K; ++k) rho[k] /= sum;
      }
      if (do_nu && s->mut)
        for (int64_t q = s->xp[t]; q < s->xp[t + 1]; ++q)
          if (s->xy[q]) {
            const int m = s->xm[q];
            for (int k = 0; k < K; ++k) {
              double z1 = c.Gth[m] * c.Gla[k], z2 = gnu * (double)s->xy[q];
              nu_acc += (double)s->xv[q] * (z2 / (z1 + z2)) * rho[k];
            }
          }
    }
    free(c.Gth);
  }
  if (do_nu && s->mut) s->nu_shp = s->a_eta + nu_acc;
}

void coo_update_rho(coo_state* s) { rho_nu(s, 1, 0); }
void coo_update_nu(coo_state* s) { rho_nu(s, 0, 1); }

/* one sweep, model.py:623-660 */
void coo_cavi_step(coo_state* s) {
  coo_update_gamma(s);
  coo_update_phi(s);
  rho_nu(s, 1, 1);
}

static double gamma_term(double pa, double pb, double qa, double qb) { /* model.py:1300-1303 */
  return lgamma(qa) - pa * log(qb) + (pa - qa) * digamma(qa) + qa * (1.0 - pb / qb);
}

/* model.py:948-1019, 1220-1313 -- uses g_nu_cache (stale), exp(rho), log(eps) for reports outside R */
double coo_elbo(coo_state* s) {
  const int M = s->M, K = s->K;
  const int64_t NN = (int64_t)s->N * s->N;
  const double gnu = s->g_nu_cache, Enu = s->nu_shp / s->nu_rte;
  double total = 0.0;
  for (int l = 0; l < s->L; ++l) {
    cache_t c; layer_cache(s, l, &c);
    double acc = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : acc)
    for (int64_t u = 0; u < NN; ++u) {
      const int64_t t = (int64_t)l * NN + u;
      const double* rho = s->rho + t * K;
      double er[KMAX], sr = 0.0, se = 0.0, ent = 0.0;
      for (int k = 0; k < K; ++k) {
        er[k] = exp(rho[k]); sr += rho[k]; se += rho[k] * c.Ela[k];
        ent += rho[k] * s->logpr[t * K + k] - rho[k] * log(rho[k] + s->eps);
      }
      const double T = mask_sum(s, t, c.Eth, c.Tfull);
      const double Q = s->mut ? (double)s->q[t] : 0.0; /* mutuality off: X^T is all zero, model.py:145 */
      for (int64_t q = s->xp[t]; q < s->xp[t + 1]; ++q) {
        const int m = s->xm[q];
        const double yt = s->mut ? (double)s->xy[q] : 0.0;
        double inner = 0.0;
        if (s->xin[q]) for (int k = 0; k < K; ++k) inner += er[k] * (c.Gth[m] * c.Gla[k] + gnu * yt);
        acc += (double)s->xv[q] * log(inner + s->eps);
      }
      acc += ent - se * T - Enu * sr * Q;
    }
    total += acc;
    for (int m = 0; m < M; ++m)
      total += gamma_term(s->a_th[l * M + m], s->b_th[l * M + m], s->gamma_shp[l * M + m], s->gamma_rte[l * M + m]);
    for (int k = 0; k < K; ++k)
      total += gamma_term(s->a_la[l * K + k], s->b_la[l * K + k], s->phi_shp[l * K + k], s->phi_rte[l * K + k]);
    free(c.Gth);
  }
  total += gamma_term(s->a_eta, s->b_eta, s->nu_shp, s->nu_rte);
  return total;
}
