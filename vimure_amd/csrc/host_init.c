/*
 * host_init.c -- host-side initial state of a realisation: the rho prior of `_set_rho_prior`
 * (latentnetworks/vimure src/python/vimure/model.py:470-482, 536-556) in ONE pass, straight into the (pinned) upload
 * buffer, bit-identical to the NumPy statements it replaces:
 *
 *     pr_rho = 1 + 0.01 * prng.rand(L, N, N, K)          (RandomState = MT19937, 53-bit doubles, C order)
 *     pr_rho[..., 0] += bias0
 *     pr_rho /= pr_rho.sum(axis=-1)[..., None]           (K < 8: a left-to-right sum)
 *     pr_rho[tie] = one-hot(0) for ties no reporter covers / nobody reports   (coverage == 0)
 *
 * NumPy spends ~0.1 s per 32 M doubles on the draw and more on the four array passes after it; a CAVI fit of BASELINE
 * config 3 converges in 0.02 s on the GPU, so the draw IS the fit time.  vmr_host_mt_skip advances a generator without
 * producing numbers (the recurrence alone, vectorised), which gives the seed of realisation r + 1 (reference
 * model.py:432-437: seed + randint drawn AFTER the prior) without drawing realisation r: the realisations' priors can
 * then be drawn by parallel host threads.  This is host glue, not CAVI arithmetic: the generator is the reference's
 * (numpy/random/src/legacy + mt19937: genrand_res53), restated.
 * Compile with -ffp-contract=off: `1 + 0.01 * u` must round twice, as NumPy does.
 */
#include <stdint.h>
#include <stddef.h>

#define MT_N 624
#define MT_M 397
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define FAST __attribute__((target_clones("avx2", "default")))
#else
#define FAST
#endif

FAST static void mt_refill(uint32_t* restrict key) {
  int i;
  for (i = 0; i < MT_N - MT_M; ++i) {
    const uint32_t y = (key[i] & 0x80000000u) | (key[i + 1] & 0x7fffffffu);
    key[i] = key[i + MT_M] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
  }
  for (; i < MT_N - 1; ++i) {
    const uint32_t y = (key[i] & 0x80000000u) | (key[i + 1] & 0x7fffffffu);
    key[i] = key[i - (MT_N - MT_M)] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
  }
  {
    const uint32_t y = (key[MT_N - 1] & 0x80000000u) | (key[0] & 0x7fffffffu);
    key[MT_N - 1] = key[MT_M - 1] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
  }
}

/* advance the generator by n 32-bit outputs without producing them */
void vmr_host_mt_skip(uint32_t* key, int* pos, int64_t n) {
  while (n > 0) {
    if (*pos >= MT_N) { mt_refill(key); *pos = 0; }
    int64_t m = MT_N - *pos;
    if (m > n) m = n;
    *pos += (int)m;
    n -= m;
  }
}

/* the next n tempered 32-bit outputs */
FAST static void mt_words(uint32_t* restrict key, int* pos, uint32_t* restrict out, int64_t n) {
  while (n > 0) {
    if (*pos >= MT_N) { mt_refill(key); *pos = 0; }
    int64_t m = MT_N - *pos;
    if (m > n) m = n;
    const uint32_t* src = key + *pos;
    for (int64_t i = 0; i < m; ++i) {
      uint32_t y = src[i];
      y ^= y >> 11;
      y ^= (y << 7) & 0x9d2c5680u;
      y ^= (y << 15) & 0xefc60000u;
      y ^= y >> 18;
      out[i] = y;
    }
    *pos += (int)m; out += m; n -= m;
  }
}

#define CHUNK 2048 /* ties per block */

/* key[624], *pos: the RandomState's MT19937 state (get_state()[1], [2]); advanced by ties*K doubles.
 * cov[ties] (may be NULL = all covered); out[ties*K]. */
FAST void vmr_host_draw_pr_rho(uint32_t* key, int* pos, int64_t ties, int K, double bias0, const uint8_t* cov, double* out) {
  static __thread uint32_t w[2 * 64 * CHUNK];
  for (int64_t t0 = 0; t0 < ties; t0 += CHUNK) {
    const int64_t nt = ties - t0 < CHUNK ? ties - t0 : CHUNK, nd = nt * K;
    double* o = out + t0 * K;
    mt_words(key, pos, w, 2 * nd);
    for (int64_t i = 0; i < nd; ++i) {   /* genrand_res53, as RandomState.random_sample; then 1 + 0.01 u */
      const uint32_t a = w[2 * i] >> 5, b = w[2 * i + 1] >> 6;
      o[i] = 1.0 + 0.01 * ((a * 67108864.0 + b) / 9007199254740992.0);
    }
    for (int64_t t = 0; t < nt; ++t) {
      double* v = o + t * K;
      double sum;
      v[0] += bias0;
      sum = v[0];
      for (int k = 1; k < K; ++k) sum += v[k];
      if (cov && !cov[t0 + t]) {
        v[0] = 1.0;
        for (int k = 1; k < K; ++k) v[k] = 0.0;
      } else {
        for (int k = 0; k < K; ++k) v[k] /= sum;
      }
    }
  }
}
