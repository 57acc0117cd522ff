/*
 * tfhe_oracle.c -- CPU restatement of the rs-tfhe gate-bootstrapping hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (rs-tfhe_amd/, the
 * C-ABI library libtfhe_hip.so) links, loads or calls this file.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as
 * the checker / the reported CPU baseline.
 *
 * Every function cites the reference file:line (paths relative to
 * /root/reference/) whose algorithm it restates.  The reference is a Rust
 * crate that is bound at compile time to SECURITY_128_BIT (src/params.rs:426-465);
 * this restatement takes the same formulas with (n, l, bgbit, basebit, t) as
 * run-time values so that the other parameter sets of src/params.rs:91-404 can
 * be exercised.  N is 1024 in every set.
 *
 * Third-party arithmetic: the reference's 512-point complex FFT is rustfft ^6.1
 * (Cargo.toml:21, un-vendored, no lockfile).  It computes the standard
 * un-normalised DFT; orc_cfft512() below is an independent radix-2
 * implementation of that same published definition.
 *
 * Pinning status: see DESIGN.md "Oracle pinning".  The reference holds no
 * golden vectors; this file is pinned against (i) the deterministic KAT inputs
 * of the reference's own tests with the exact schoolbook product as expected
 * value, (ii) the reference's SPQLIOS C++/asm negacyclic FFT compiled from
 * /root/reference into oracle/_ref/ (products, a whole external product, the
 * rotation with its MAX - x quirk, a CMUX chain, sample_extract_index at every
 * index), (iii) the known constants of the reference, (iv) decrypt-equality
 * property tests that mirror the reference's randomized unit tests, (v) closed
 * forms that need no key (tests/closed_forms.py): the key switch under a
 * noise-free key, and trivial ciphertexts through the composed bootstrap (the
 * lookup table's entry exactly; the ten gate preps as linear forms).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_N 1024
#define ORC_N2 512
#define ORC_MAX_L 4

/* ------------------------------------------------------------------------- */
/* Parameters: src/params.rs:53-84 (SecurityParams / TrgswParams)             */
/* ------------------------------------------------------------------------- */
typedef struct {
  int32_t n;        /* tlwe_lv0.n                                   */
  int32_t l;        /* trgsw_lv1.l                                  */
  int32_t bgbit;    /* trgsw_lv1.bgbit                              */
  int32_t basebit;  /* trgsw_lv1.basebit                            */
  int32_t t;        /* trgsw_lv1.iks_t                              */
  double alpha_lv0; /* tlwe_lv0.alpha  (KSK_ALPHA, params.rs:468)   */
  double alpha_lv1; /* tlwe_lv1.alpha  (BSK_ALPHA, params.rs:469)   */
} orc_params;

/* ------------------------------------------------------------------------- */
/* Torus helpers: src/utils.rs:9-16                                           */
/* ------------------------------------------------------------------------- */
uint32_t orc_f64_to_torus(double d) {
  /* (d % 1.0) * 2^32, `as i64` (truncate toward zero), `as u32` (wrap) */
  double torus = fmod(d, 1.0) * 4294967296.0;
  return (uint32_t)(int64_t)torus;
}

double orc_torus_to_f64(uint32_t t) { return (double)t / 4294967296.0; }

/* ------------------------------------------------------------------------- */
/* Seeded PRNG.  The reference uses unseeded rand::thread_rng (tlwe.rs:38,     */
/* key.rs:32); a seeded generator is a harness choice, distributionally the   */
/* same: uniform u32 / uniform bits / N(0, alpha).                            */
/* ------------------------------------------------------------------------- */
typedef struct {
  uint64_t s[4];
  int have_spare;
  double spare;
} orc_rng;

static uint64_t splitmix64(uint64_t *x) {
  uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

void orc_rng_seed(orc_rng *r, uint64_t seed) {
  uint64_t x = seed;
  for (int i = 0; i < 4; i++) r->s[i] = splitmix64(&x);
  r->have_spare = 0;
  r->spare = 0.0;
}

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

static inline uint64_t rng_next(orc_rng *r) { /* xoshiro256** */
  uint64_t *s = r->s;
  uint64_t result = rotl64(s[1] * 5, 7) * 9;
  uint64_t t = s[1] << 17;
  s[2] ^= s[0];
  s[3] ^= s[1];
  s[1] ^= s[2];
  s[0] ^= s[3];
  s[2] ^= t;
  s[3] = rotl64(s[3], 45);
  return result;
}

static inline uint32_t rng_u32(orc_rng *r) { return (uint32_t)(rng_next(r) >> 32); }

static inline double rng_unit(orc_rng *r) { /* (0,1] */
  return ((double)(rng_next(r) >> 11) + 1.0) * (1.0 / 9007199254740992.0);
}

static double rng_normal(orc_rng *r, double sigma) { /* Box-Muller */
  if (r->have_spare) {
    r->have_spare = 0;
    return r->spare * sigma;
  }
  double u1 = rng_unit(r), u2 = rng_unit(r);
  double rad = sqrt(-2.0 * log(u1));
  double ang = 6.283185307179586476925286766559 * u2;
  r->spare = rad * sin(ang);
  r->have_spare = 1;
  return rad * cos(ang) * sigma;
}

/* utils.rs:22-38: gaussian_f64(mu) = f64_to_torus(sample) + f64_to_torus(mu) */
static inline uint32_t gaussian_f64(orc_rng *r, double mu, double alpha) {
  double sample = rng_normal(r, alpha);
  return orc_f64_to_torus(sample) + orc_f64_to_torus(mu);
}

/* ------------------------------------------------------------------------- */
/* 512-point complex DFT (stand-in for rustfft, klemsa.rs:62-65,105-107,131-133)
 * forward: X[k] = sum_j x[j] exp(-2 pi i jk/512); inverse: +sign, both
 * un-normalised.  Radix-2 DIT on split re/im arrays.                         */
/* ------------------------------------------------------------------------- */
static double g_tw_re[ORC_N2]; /* stage tables, concatenated: half=1,2,4,...,256 */
static double g_tw_im[ORC_N2];
static int g_brev[ORC_N2];
static double g_twist_re[ORC_N2]; /* klemsa.rs:49-58 exp(i*pi*k/N) */
static double g_twist_im[ORC_N2];
static int g_init_done = 0;

void orc_init(void) {
  if (g_init_done) return;
#ifdef _OPENMP
#pragma omp critical(orc_init_lock)
#endif
  {
    if (!g_init_done) {
      for (int i = 0; i < ORC_N2; i++) {
        int r = 0;
        for (int b = 0; b < 9; b++)
          if (i & (1 << b)) r |= 1 << (8 - b);
        g_brev[i] = r;
      }
      int off = 0;
      for (int half = 1; half < ORC_N2; half <<= 1) {
        for (int j = 0; j < half; j++) {
          double ang = -M_PI * (double)j / (double)half;
          g_tw_re[off + j] = cos(ang);
          g_tw_im[off + j] = sin(ang);
        }
        off += half;
      }
      double unit = M_PI / (double)ORC_N;
      for (int i = 0; i < ORC_N2; i++) {
        double ang = (double)i * unit;
        g_twist_re[i] = cos(ang);
        g_twist_im[i] = sin(ang);
      }
      g_init_done = 1;
    }
  }
}

/* sign = -1 forward, +1 inverse; in-place on natural-order data */
static void cfft512(double *re, double *im, int inverse) {
  for (int i = 0; i < ORC_N2; i++) {
    int j = g_brev[i];
    if (j > i) {
      double t = re[i];
      re[i] = re[j];
      re[j] = t;
      t = im[i];
      im[i] = im[j];
      im[j] = t;
    }
  }
  int off = 0;
  for (int half = 1; half < ORC_N2; half <<= 1) {
    const double *wr = g_tw_re + off;
    const double *wi = g_tw_im + off;
    for (int base = 0; base < ORC_N2; base += 2 * half) {
      double *ar = re + base, *ai = im + base;
      double *br = re + base + half, *bi = im + base + half;
      if (!inverse) {
        for (int j = 0; j < half; j++) {
          double tr = br[j] * wr[j] - bi[j] * wi[j];
          double ti = br[j] * wi[j] + bi[j] * wr[j];
          br[j] = ar[j] - tr;
          bi[j] = ai[j] - ti;
          ar[j] += tr;
          ai[j] += ti;
        }
      } else {
        for (int j = 0; j < half; j++) {
          double tr = br[j] * wr[j] + bi[j] * wi[j];
          double ti = bi[j] * wr[j] - br[j] * wi[j];
          br[j] = ar[j] - tr;
          bi[j] = ai[j] - ti;
          ar[j] += tr;
          ai[j] += ti;
        }
      }
    }
    off += half;
  }
}

/* exported for the DFT-definition unit test */
void orc_cfft512(double *re, double *im, int inverse) {
  orc_init();
  cfft512(re, im, inverse);
}

/* ------------------------------------------------------------------------- */
/* KlemsaProcessor: src/fft/klemsa.rs:88-174                                  */
/* ------------------------------------------------------------------------- */
/* klemsa.rs:88-117 `ifft`: torus -> spectrum (re[0..512] || im[0..512])      */
void orc_klemsa_ifft(const uint32_t *input, double *result) {
  orc_init();
  double re[ORC_N2], im[ORC_N2];
  for (int i = 0; i < ORC_N2; i++) {
    double in_re = (double)(int32_t)input[i];          /* `as i32 as f64` :96 */
    double in_im = (double)(int32_t)input[i + ORC_N2]; /* :97 */
    double w_re = g_twist_re[i], w_im = g_twist_im[i];
    re[i] = in_re * w_re - in_im * w_im; /* :100 */
    im[i] = in_re * w_im + in_im * w_re;
  }
  cfft512(re, im, 0);
  for (int i = 0; i < ORC_N2; i++) { /* :111-114 */
    result[i] = re[i] * 2.0;
    result[i + ORC_N2] = im[i] * 2.0;
  }
}

static inline uint32_t round_to_torus(double x) {
  /* klemsa.rs:145-146: `.round() as i64 as u32`; f64::round = half away from 0 */
  return (uint32_t)(int64_t)round(x);
}

/* klemsa.rs:119-150 `fft`: spectrum -> torus */
void orc_klemsa_fft(const double *input, uint32_t *result) {
  orc_init();
  double re[ORC_N2], im[ORC_N2];
  for (int i = 0; i < ORC_N2; i++) { /* :125-127 */
    re[i] = input[i] * 0.5;
    im[i] = input[i + ORC_N2] * 0.5;
  }
  cfft512(re, im, 1);
  const double normalization = 1.0 / (double)ORC_N2; /* :136 */
  for (int i = 0; i < ORC_N2; i++) {
    double w_re = g_twist_re[i], w_im = g_twist_im[i];
    double f_re = re[i], f_im = im[i];
    double tmp_re = (f_re * w_re + f_im * w_im) * normalization; /* :143 */
    double tmp_im = (f_im * w_re - f_re * w_im) * normalization; /* :144 */
    result[i] = round_to_torus(tmp_re);
    result[i + ORC_N2] = round_to_torus(tmp_im);
  }
}

/* klemsa.rs:152-174 `poly_mul` */
void orc_klemsa_poly_mul(const uint32_t *a, const uint32_t *b, uint32_t *out) {
  double a_fft[ORC_N], b_fft[ORC_N], r[ORC_N];
  orc_klemsa_ifft(a, a_fft);
  orc_klemsa_ifft(b, b_fft);
  for (int i = 0; i < ORC_N2; i++) {
    double ar = a_fft[i], ai = a_fft[i + ORC_N2];
    double br = b_fft[i], bi = b_fft[i + ORC_N2];
    r[i] = (ar * br - ai * bi) * 0.5;
    r[i + ORC_N2] = (ar * bi + ai * br) * 0.5;
  }
  orc_klemsa_fft(r, out);
}

/* src/fft/mod.rs:240-255: O(N^2) negacyclic product, wrapping u32 (the
 * reference's own ground truth in its FFT tests; exact mod 2^32)             */
void orc_negacyclic_schoolbook(const uint32_t *a, const uint32_t *b, uint32_t *res) {
  memset(res, 0, ORC_N * sizeof(uint32_t));
  for (int i = 0; i < ORC_N; i++) {
    uint32_t ai = a[i];
    if (ai == 0) continue;
    for (int j = 0; j < ORC_N - i; j++) res[i + j] += ai * b[j];
    for (int j = ORC_N - i; j < ORC_N; j++) res[i + j - ORC_N] -= ai * b[j];
  }
}

/* ------------------------------------------------------------------------- */
/* Keys: src/key.rs                                                           */
/* ------------------------------------------------------------------------- */
/* key.rs:78-89 */
uint32_t orc_gen_decomposition_offset(int l, int bgbit) {
  uint32_t offset = 0;
  uint32_t bg = 1u << bgbit;
  for (int i = 0; i < l; i++) offset += (bg / 2) * (1u << (32 - (i + 1) * bgbit));
  return offset;
}

/* key.rs:91-100; layout a[0..N] || b[0..N] */
void orc_gen_testvec(uint32_t *tv) {
  uint32_t b_torus = orc_f64_to_torus(0.125);
  for (int i = 0; i < ORC_N; i++) {
    tv[i] = 0;
    tv[ORC_N + i] = b_torus;
  }
}

/* key.rs:39-46 : uniform bits */
void orc_gen_secret_key(uint64_t seed, int n, uint32_t *key_lv0, uint32_t *key_lv1) {
  orc_rng r;
  orc_rng_seed(&r, seed);
  for (int i = 0; i < n; i++) key_lv0[i] = rng_u32(&r) & 1u;
  for (int i = 0; i < ORC_N; i++) key_lv1[i] = rng_u32(&r) & 1u;
}

/* tlwe.rs:37-53 (lv0) and :232-249 (lv1): generic dimension `dim` */
static void tlwe_encrypt_f64_rng(orc_rng *r, double p, double alpha, const uint32_t *key, int dim,
                                 uint32_t *out) {
  uint32_t inner = 0;
  for (int i = 0; i < dim; i++) {
    uint32_t a = rng_u32(r);
    inner += key[i] * a;
    out[i] = a;
  }
  out[dim] = inner + gaussian_f64(r, p, alpha);
}

void orc_tlwe_encrypt_f64(uint64_t seed, double p, double alpha, const uint32_t *key, int dim,
                          uint32_t *out) {
  orc_rng r;
  orc_rng_seed(&r, seed);
  tlwe_encrypt_f64_rng(&r, p, alpha, key, dim, out);
}

/* batch helper for the harness: count ciphertexts, messages as f64 torus fractions */
void orc_tlwe_encrypt_f64_batch(uint64_t seed, const double *p, int count, double alpha,
                                const uint32_t *key, int dim, uint32_t *out) {
#pragma omp parallel for schedule(static)
  for (int c = 0; c < count; c++) {
    orc_rng r;
    orc_rng_seed(&r, seed + 0x632BE59BD9B4E019ull * (uint64_t)(c + 1));
    tlwe_encrypt_f64_rng(&r, p[c], alpha, key, dim, out + (size_t)c * (dim + 1));
  }
}

/* tlwe.rs:60-68 decrypt_bool */
int orc_tlwe_decrypt_bool(const uint32_t *ct, const uint32_t *key, int dim) {
  uint32_t inner = 0;
  for (int i = 0; i < dim; i++) inner += ct[i] * key[i];
  int32_t res = (int32_t)(ct[dim] - inner);
  return res >= 0;
}

/* phase b - <a,s> (tlwe.rs:61-66) */
uint32_t orc_tlwe_phase(const uint32_t *ct, const uint32_t *key, int dim) {
  uint32_t inner = 0;
  for (int i = 0; i < dim; i++) inner += ct[i] * key[i];
  return ct[dim] - inner;
}

/* tlwe.rs:111-126 decrypt_lwe_message */
int orc_tlwe_decrypt_lwe_message(const uint32_t *ct, int message_modulus, const uint32_t *key,
                                 int dim) {
  uint32_t res_torus = orc_tlwe_phase(ct, key, dim);
  double res_f64 = orc_torus_to_f64(res_torus);
  double scale = 1.0 / (2.0 * (double)message_modulus);
  uint64_t message = (uint64_t)(res_f64 / scale + 0.5);
  return (int)(message % (uint64_t)message_modulus);
}

/* tlwe.rs:84-98: encoded value of encrypt_lwe_message */
double orc_lwe_message_encoding(int message, int message_modulus) {
  int m = message % message_modulus;
  double scale = 1.0 / (2.0 * (double)message_modulus);
  return (double)m * scale;
}

/* trlwe.rs:30-52 encrypt_f64 with p == 0 (the only use on the keygen path) */
static void trlwe_encrypt_zero(orc_rng *r, double alpha, const uint32_t *key1, uint32_t *a,
                               uint32_t *b) {
  uint32_t poly_res[ORC_N];
  for (int i = 0; i < ORC_N; i++) a[i] = rng_u32(r);
  for (int i = 0; i < ORC_N; i++) b[i] = gaussian_f64(r, 0.0, alpha);
  orc_klemsa_poly_mul(a, key1, poly_res);
  for (int i = 0; i < ORC_N; i++) b[i] += poly_res[i];
}

/* trgsw.rs:29-49 encrypt_torus + trgsw.rs:58-68 / trlwe.rs:91-96 (FFT form).
 * out_time: [2l][2][N] u32 (a then b), out_fft: [2l][2][N] f64 (may be NULL) */
static void trgsw_encrypt(orc_rng *r, uint32_t p, const orc_params *P, const uint32_t *key1,
                          uint32_t *out_time, double *out_fft) {
  const int l = P->l;
  uint32_t p_torus[ORC_MAX_L];
  for (int i = 0; i < l; i++) {
    double bg = (double)(1u << P->bgbit);
    p_torus[i] = orc_f64_to_torus(pow(bg, -(double)(1 + i))); /* :33 */
  }
  for (int row = 0; row < 2 * l; row++)
    trlwe_encrypt_zero(r, P->alpha_lv1, key1, out_time + (size_t)row * 2 * ORC_N,
                       out_time + (size_t)row * 2 * ORC_N + ORC_N);
  for (int i = 0; i < l; i++) {
    out_time[(size_t)i * 2 * ORC_N + 0] += p * p_torus[i];               /* row i, a[0]   :45 */
    out_time[(size_t)(i + l) * 2 * ORC_N + ORC_N + 0] += p * p_torus[i]; /* row i+l, b[0] :46 */
  }
  if (out_fft) {
    for (int row = 0; row < 2 * l; row++) {
      orc_klemsa_ifft(out_time + (size_t)row * 2 * ORC_N, out_fft + (size_t)row * 2 * ORC_N);
      orc_klemsa_ifft(out_time + (size_t)row * 2 * ORC_N + ORC_N,
                      out_fft + (size_t)row * 2 * ORC_N + ORC_N);
    }
  }
}

/* key.rs:124-156 gen_bootstrapping_key: TRGSW(key_lv0[i]) for i<n, alpha = BSK_ALPHA.
 * bsk_fft: [n][2l][2][N] f64 (reference TRGSWLv1FFT layout, re||im halves)
 * bsk_time: [n][2l][2][N] u32 or NULL (harness extra: time-domain rows for the
 * exact-integer ground truth)                                                */
void orc_gen_bootstrapping_key(uint64_t seed, const orc_params *P, const uint32_t *key_lv0,
                               const uint32_t *key_lv1, double *bsk_fft, uint32_t *bsk_time) {
  orc_init();
  const size_t per = (size_t)2 * P->l * 2 * ORC_N;
#pragma omp parallel for schedule(dynamic, 4)
  for (int i = 0; i < P->n; i++) {
    orc_rng r;
    orc_rng_seed(&r, seed ^ (0xA0761D6478BD642Full * (uint64_t)(i + 1)));
    uint32_t *tmp = (uint32_t *)malloc(per * sizeof(uint32_t));
    trgsw_encrypt(&r, key_lv0[i], P, key_lv1, tmp, bsk_fft + (size_t)i * per);
    if (bsk_time) memcpy(bsk_time + (size_t)i * per, tmp, per * sizeof(uint32_t));
    free(tmp);
  }
}

/* key.rs:102-122 gen_key_switching_key: [N][t][base][n+1], k==0 slots zero   */
void orc_gen_key_switching_key(uint64_t seed, const orc_params *P, const uint32_t *key_lv0,
                               const uint32_t *key_lv1, uint32_t *ksk) {
  const int base = 1 << P->basebit;
  const int n = P->n;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < ORC_N; i++) {
    orc_rng r;
    orc_rng_seed(&r, seed ^ (0xE7037ED1A0B428DBull * (uint64_t)(i + 1)));
    for (int j = 0; j < P->t; j++) {
      for (int k = 0; k < base; k++) {
        size_t idx = ((size_t)base * P->t * i) + ((size_t)base * j) + k;
        uint32_t *row = ksk + idx * (size_t)(n + 1);
        if (k == 0) {
          memset(row, 0, (size_t)(n + 1) * sizeof(uint32_t));
          continue;
        }
        double p = (double)((uint32_t)k * key_lv1[i]) / (double)(1u << ((j + 1) * P->basebit));
        tlwe_encrypt_f64_rng(&r, p, P->alpha_lv0, key_lv0, n, row);
      }
    }
  }
}

/* ------------------------------------------------------------------------- */
/* TRGSW / blind rotation: src/trgsw.rs                                       */
/* ------------------------------------------------------------------------- */
/* trgsw.rs:144-171 decomposition: trlwe = a||b, out = [2l][N] (wrapped u32)  */
void orc_decomposition(const uint32_t *trlwe, int l, int bgbit, uint32_t offset, uint32_t *out) {
  const uint32_t mask = (1u << bgbit) - 1u;
  const uint32_t half_bg = 1u << (bgbit - 1);
  const uint32_t *a = trlwe, *b = trlwe + ORC_N;
  for (int j = 0; j < ORC_N; j++) {
    uint32_t tmp0 = a[j] + offset;
    uint32_t tmp1 = b[j] + offset;
    for (int i = 0; i < l; i++)
      out[(size_t)i * ORC_N + j] = ((tmp0 >> (32 - (i + 1) * bgbit)) & mask) - half_bg;
    for (int i = 0; i < l; i++)
      out[(size_t)(i + l) * ORC_N + j] = ((tmp1 >> (32 - (i + 1) * bgbit)) & mask) - half_bg;
  }
}

/* trgsw.rs:118-142 fma_in_fd_1024 (operation order kept) */
static void fma_in_fd_1024(double *res, const double *a, const double *b) {
  for (int i = 0; i < ORC_N2; i++) {
    res[i] = (a[i + ORC_N2] * b[i + ORC_N2]) * 0.5 - res[i];
    res[i] = (a[i] * b[i]) * 0.5 - res[i];
    res[i + ORC_N2] += (a[i] * b[i + ORC_N2] + a[i + ORC_N2] * b[i]) * 0.5;
  }
}

/* trgsw.rs:77-116 external_product_with_fft.
 * trgsw_fft: [2l][2][N] f64; trlwe: a||b; out: a||b                           */
void orc_external_product_fft(const double *trgsw_fft, const uint32_t *trlwe, int l, int bgbit,
                              uint32_t offset, uint32_t *out) {
  uint32_t dec[2 * ORC_MAX_L * ORC_N];
  double dec_fft[ORC_N];
  double out_a_fft[ORC_N], out_b_fft[ORC_N];
  orc_decomposition(trlwe, l, bgbit, offset, dec);
  memset(out_a_fft, 0, sizeof(out_a_fft));
  memset(out_b_fft, 0, sizeof(out_b_fft));
  for (int i = 0; i < 2 * l; i++) {
    orc_klemsa_ifft(dec + (size_t)i * ORC_N, dec_fft);
    fma_in_fd_1024(out_a_fft, dec_fft, trgsw_fft + (size_t)i * 2 * ORC_N);
    fma_in_fd_1024(out_b_fft, dec_fft, trgsw_fft + (size_t)i * 2 * ORC_N + ORC_N);
  }
  orc_klemsa_fft(out_a_fft, out);
  orc_klemsa_fft(out_b_fft, out + ORC_N);
}

/* Exact-integer ground truth for the same external product: sum_r dec_r (*) row_r
 * in Z_{2^32}[X]/(X^N+1), wrapping u32 arithmetic (exact mod 2^32).
 * trgsw_time: [2l][2][N] u32.                                                 */
static void negacyclic_mac_signed(uint32_t *res, const uint32_t *d, const uint32_t *b) {
  for (int i = 0; i < ORC_N; i++) {
    uint32_t di = d[i];
    if (di == 0) continue;
    uint32_t *r0 = res + i;
    for (int j = 0; j < ORC_N - i; j++) r0[j] += di * b[j];
    uint32_t *r1 = res + i - ORC_N;
    for (int j = ORC_N - i; j < ORC_N; j++) r1[j] -= di * b[j];
  }
}

void orc_external_product_exact(const uint32_t *trgsw_time, const uint32_t *trlwe, int l,
                                int bgbit, uint32_t offset, uint32_t *out) {
  uint32_t dec[2 * ORC_MAX_L * ORC_N];
  orc_decomposition(trlwe, l, bgbit, offset, dec);
  memset(out, 0, 2 * ORC_N * sizeof(uint32_t));
  for (int i = 0; i < 2 * l; i++) {
    negacyclic_mac_signed(out, dec + (size_t)i * ORC_N, trgsw_time + (size_t)i * 2 * ORC_N);
    negacyclic_mac_signed(out + ORC_N, dec + (size_t)i * ORC_N,
                          trgsw_time + (size_t)i * 2 * ORC_N + ORC_N);
  }
}

/* trgsw.rs:307-330 poly_mul_with_x_k (k in [0, 2N]); note Torus::MAX - a[i] */
void orc_poly_mul_with_x_k(const uint32_t *a, int k, uint32_t *res) {
  const int N = ORC_N;
  if (k < N) {
    for (int i = 0; i < N - k; i++) res[i + k] = a[i];
    for (int i = N - k; i < N; i++) res[i + k - N] = 0xFFFFFFFFu - a[i];
  } else {
    for (int i = 0; i < 2 * N - k; i++) res[i + k - N] = 0xFFFFFFFFu - a[i];
    for (int i = 2 * N - k; i < N; i++) res[i - (2 * N - k)] = a[i];
  }
}

/* trgsw.rs:174-196 cmux: in1 + ExtProd(cond, in2 - in1).  use_exact selects the
 * exact-integer product (cond_time) instead of the f64 FFT one (cond_fft).   */
static void cmux_any(const uint32_t *in1, const uint32_t *in2, const double *cond_fft,
                     const uint32_t *cond_time, int l, int bgbit, uint32_t offset,
                     uint32_t *res) {
  uint32_t tmp[2 * ORC_N], tmp2[2 * ORC_N];
  for (int i = 0; i < 2 * ORC_N; i++) tmp[i] = in2[i] - in1[i];
  if (cond_time)
    orc_external_product_exact(cond_time, tmp, l, bgbit, offset, tmp2);
  else
    orc_external_product_fft(cond_fft, tmp, l, bgbit, offset, tmp2);
  for (int i = 0; i < 2 * ORC_N; i++) res[i] = tmp2[i] + in1[i];
}

void orc_cmux(const uint32_t *in1, const uint32_t *in2, const double *cond_fft, int l, int bgbit,
              uint32_t offset, uint32_t *res) {
  cmux_any(in1, in2, cond_fft, NULL, l, bgbit, offset, res);
}

/* trgsw.rs:198-226 blind_rotate / :242-274 blind_rotate_with_testvec.
 * src: [n+1]; testvec: a||b; bsk_fft: [n][2l][2][N]; out: a||b.
 * If bsk_time != NULL the exact-integer external product is used instead.    */
static void blind_rotate_any(const uint32_t *src, const uint32_t *testvec, const double *bsk_fft,
                             const uint32_t *bsk_time, const orc_params *P, uint32_t offset,
                             uint32_t *out) {
  const int N = ORC_N, NBIT = 10;
  const size_t per = (size_t)2 * P->l * 2 * ORC_N;
  /* :202-203 -- `src.b() as usize + (1 << 20)` is a 64-bit add: no u32 wrap */
  uint64_t bt = ((uint64_t)src[P->n] + (1ull << (32 - 1 - NBIT - 1))) >> (32 - NBIT - 1);
  int b_tilda = (int)(2 * N - (int)bt);
  uint32_t res[2 * ORC_N], res2[2 * ORC_N], nxt[2 * ORC_N];
  orc_poly_mul_with_x_k(testvec, b_tilda, res);
  orc_poly_mul_with_x_k(testvec + N, b_tilda, res + N);
  for (int i = 0; i < P->n; i++) {
    /* :210-211 wrapping_add in u32 */
    int a_tilda = (int)((uint32_t)(src[i] + (1u << (32 - 1 - NBIT - 1))) >> (32 - NBIT - 1));
    orc_poly_mul_with_x_k(res, a_tilda, res2);
    orc_poly_mul_with_x_k(res + N, a_tilda, res2 + N);
    cmux_any(res, res2, bsk_fft ? bsk_fft + (size_t)i * per : NULL,
             bsk_time ? bsk_time + (size_t)i * per : NULL, P->l, P->bgbit, offset, nxt);
    memcpy(res, nxt, sizeof(res));
  }
  memcpy(out, res, sizeof(res));
}

void orc_blind_rotate(const uint32_t *src, const uint32_t *testvec, const double *bsk_fft,
                      const orc_params *P, uint32_t offset, uint32_t *out) {
  orc_init();
  blind_rotate_any(src, testvec, bsk_fft, NULL, P, offset, out);
}

void orc_blind_rotate_exact(const uint32_t *src, const uint32_t *testvec, const uint32_t *bsk_time,
                            const orc_params *P, uint32_t offset, uint32_t *out) {
  orc_init();
  blind_rotate_any(src, testvec, NULL, bsk_time, P, offset, out);
}

/* ------------------------------------------------------------------------- */
/* Sample extraction: src/trlwe.rs:106-136                                    */
/* ------------------------------------------------------------------------- */
/* trlwe.rs:106-120: out [N+1] */
void orc_sample_extract_index(const uint32_t *trlwe, int k, uint32_t *out) {
  const int N = ORC_N;
  for (int i = 0; i < N; i++) {
    if (i <= k)
      out[i] = trlwe[k - i];
    else
      out[i] = 0xFFFFFFFFu - trlwe[N + k - i];
  }
  out[N] = trlwe[ORC_N + k];
}

/* trlwe.rs:122-136: same formula with N := tlwe_lv0::N (= n); out [n+1] */
void orc_sample_extract_index_2(const uint32_t *trlwe, int k, int n, uint32_t *out) {
  for (int i = 0; i < n; i++) {
    if (i <= k)
      out[i] = trlwe[k - i];
    else
      out[i] = 0xFFFFFFFFu - trlwe[n + k - i];
  }
  out[n] = trlwe[ORC_N + k];
}

/* ------------------------------------------------------------------------- */
/* Identity key switching: src/trgsw.rs:332-360                               */
/* src: [N+1]; ksk: [N][t][base][n+1]; out: [n+1]                             */
/* ------------------------------------------------------------------------- */
void orc_identity_key_switching(const uint32_t *src, const uint32_t *ksk, const orc_params *P,
                                uint32_t *out) {
  const int N = ORC_N, n = P->n, basebit = P->basebit, t = P->t;
  const int base = 1 << basebit;
  memset(out, 0, (size_t)(n + 1) * sizeof(uint32_t));
  out[n] = src[N];
  const uint32_t prec_offset = 1u << (32 - (1 + basebit * t));
  for (int i = 0; i < N; i++) {
    uint32_t a_bar = src[i] + prec_offset;
    for (int j = 0; j < t; j++) {
      uint32_t k = (a_bar >> (32 - (j + 1) * basebit)) & ((1u << basebit) - 1u);
      if (k != 0) {
        size_t idx = ((size_t)base * t * i) + ((size_t)base * j) + k;
        const uint32_t *row = ksk + idx * (size_t)(n + 1);
        for (int x = 0; x <= n; x++) out[x] -= row[x];
      }
    }
  }
}

/* ------------------------------------------------------------------------- */
/* Gates: src/gates.rs:54-150 (linear prep), op codes shared with the C ABI   */
/* ------------------------------------------------------------------------- */
enum {
  ORC_GATE_NAND = 0,
  ORC_GATE_OR = 1,
  ORC_GATE_AND = 2,
  ORC_GATE_XOR = 3,
  ORC_GATE_XNOR = 4,
  ORC_GATE_NOR = 5,
  ORC_GATE_ANDNY = 6,
  ORC_GATE_ANDYN = 7,
  ORC_GATE_ORNY = 8,
  ORC_GATE_ORYN = 9,
  ORC_GATE_COPY = 10 /* no prep: bootstrap(a) */
};

int orc_gate_prep(int op, const uint32_t *a, const uint32_t *b, int n, uint32_t *out) {
  uint32_t c;
  switch (op) {
    case ORC_GATE_NAND: /* gates.rs:54-58 */
      for (int i = 0; i <= n; i++) out[i] = 0u - (a[i] + b[i]);
      c = orc_f64_to_torus(0.125);
      break;
    case ORC_GATE_OR: /* :62-66 */
      for (int i = 0; i <= n; i++) out[i] = a[i] + b[i];
      c = orc_f64_to_torus(0.125);
      break;
    case ORC_GATE_AND: /* :70-74 */
      for (int i = 0; i <= n; i++) out[i] = a[i] + b[i];
      c = orc_f64_to_torus(-0.125);
      break;
    case ORC_GATE_XOR: /* :78-82 add_mul(b, 2) */
      for (int i = 0; i <= n; i++) out[i] = a[i] + b[i] * 2u;
      c = orc_f64_to_torus(0.25);
      break;
    case ORC_GATE_XNOR: /* :86-90 sub_mul(b, 2) */
      for (int i = 0; i <= n; i++) out[i] = a[i] - b[i] * 2u;
      c = orc_f64_to_torus(-0.25);
      break;
    case ORC_GATE_NOR: /* :94-98 */
      for (int i = 0; i <= n; i++) out[i] = 0u - (a[i] + b[i]);
      c = orc_f64_to_torus(-0.125);
      break;
    case ORC_GATE_ANDNY: /* :102-111  -a + b */
      for (int i = 0; i <= n; i++) out[i] = (0u - a[i]) + b[i];
      c = orc_f64_to_torus(-0.125);
      break;
    case ORC_GATE_ANDYN: /* :115-124  a - b */
      for (int i = 0; i <= n; i++) out[i] = a[i] - b[i];
      c = orc_f64_to_torus(-0.125);
      break;
    case ORC_GATE_ORNY: /* :128-137 */
      for (int i = 0; i <= n; i++) out[i] = (0u - a[i]) + b[i];
      c = orc_f64_to_torus(0.125);
      break;
    case ORC_GATE_ORYN: /* :141-150 */
      for (int i = 0; i <= n; i++) out[i] = a[i] - b[i];
      c = orc_f64_to_torus(0.125);
      break;
    case ORC_GATE_COPY:
      for (int i = 0; i <= n; i++) out[i] = a[i];
      return 0;
    default:
      return -1;
  }
  out[n] += c;
  return 0;
}

/* ------------------------------------------------------------------------- */
/* Bootstrap strategies: src/bootstrap/vanilla.rs:40-63, src/bootstrap/lut.rs:79-99
 * A "cloud key" for the oracle is the tuple (P, offset, testvec, bsk_fft, ksk).*/
/* ------------------------------------------------------------------------- */
typedef struct {
  orc_params P;
  uint32_t decomposition_offset;
  const uint32_t *testvec;  /* [2][N] */
  const double *bsk_fft;    /* [n][2l][2][N] */
  const uint32_t *bsk_time; /* optional */
  const uint32_t *ksk;      /* [N][t][base][n+1] */
} orc_cloud_key;

/* vanilla.rs:40-52 / lut.rs:79-99 (custom testvec) */
void orc_bootstrap(const orc_cloud_key *ck, const uint32_t *ct, const uint32_t *testvec_or_null,
                   uint32_t *out) {
  uint32_t trlwe[2 * ORC_N], lv1[ORC_N + 1];
  orc_blind_rotate(ct, testvec_or_null ? testvec_or_null : ck->testvec, ck->bsk_fft, &ck->P,
                   ck->decomposition_offset, trlwe);
  orc_sample_extract_index(trlwe, 0, lv1);
  orc_identity_key_switching(lv1, ck->ksk, &ck->P, out);
}

/* vanilla.rs:54-63 */
void orc_bootstrap_without_key_switch(const orc_cloud_key *ck, const uint32_t *ct, uint32_t *out) {
  uint32_t trlwe[2 * ORC_N];
  orc_blind_rotate(ct, ck->testvec, ck->bsk_fft, &ck->P, ck->decomposition_offset, trlwe);
  orc_sample_extract_index_2(trlwe, 0, ck->P.n, out);
}

/* gates.rs:357-383 batch_nand_with_railgun and siblings (:388-547); Rayon
 * par_iter -> OpenMP parallel for (rayon_impl.rs:40-47).  nthreads<=0: all.   */
int orc_batch_gate(const orc_cloud_key *ck, int op, const uint32_t *a, const uint32_t *b,
                   uint32_t *out, int count, int nthreads) {
  orc_init();
  const int n = ck->P.n;
  int rc = 0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
  for (int c = 0; c < count; c++) {
    uint32_t prep[2048];
    if (orc_gate_prep(op, a + (size_t)c * (n + 1), b ? b + (size_t)c * (n + 1) : NULL, n, prep)) {
      rc = -1;
      continue;
    }
    orc_bootstrap(ck, prep, NULL, out + (size_t)c * (n + 1));
  }
  return rc;
}

/* batch LUT bootstrap (lut.rs:79-99 applied per ciphertext; shared or per-ct testvec) */
void orc_batch_bootstrap(const orc_cloud_key *ck, const uint32_t *in, const uint32_t *testvec,
                         int per_ct_testvec, int keyswitch, uint32_t *out, int count,
                         int nthreads) {
  orc_init();
  const int n = ck->P.n;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
  for (int c = 0; c < count; c++) {
    const uint32_t *tv = testvec ? (per_ct_testvec ? testvec + (size_t)c * 2 * ORC_N : testvec)
                                 : ck->testvec;
    if (keyswitch) {
      orc_bootstrap(ck, in + (size_t)c * (n + 1), tv, out + (size_t)c * (n + 1));
    } else {
      uint32_t trlwe[2 * ORC_N];
      orc_blind_rotate(in + (size_t)c * (n + 1), tv, ck->bsk_fft, &ck->P,
                       ck->decomposition_offset, trlwe);
      orc_sample_extract_index_2(trlwe, 0, n, out + (size_t)c * (n + 1));
    }
  }
}

void orc_batch_blind_rotate(const orc_cloud_key *ck, const uint32_t *in, const uint32_t *testvec,
                            uint32_t *out, int count, int nthreads) {
  orc_init();
  const int n = ck->P.n;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
  for (int c = 0; c < count; c++)
    orc_blind_rotate(in + (size_t)c * (n + 1), testvec ? testvec : ck->testvec, ck->bsk_fft,
                     &ck->P, ck->decomposition_offset, out + (size_t)c * 2 * ORC_N);
}

/* gates.rs:157-183 Gates::mux (the reference formula) */
void orc_mux(const orc_cloud_key *ck, const uint32_t *a, const uint32_t *b, const uint32_t *c,
             uint32_t *out) {
  const int n = ck->P.n;
  uint32_t t1[2048], t2[2048], u1[2048], u2[2048], na[2048];
  orc_gate_prep(ORC_GATE_AND, a, b, n, t1);
  orc_bootstrap_without_key_switch(ck, t1, u1);
  for (int i = 0; i <= n; i++) na[i] = 0u - a[i]; /* gates.rs:202-204 not */
  orc_gate_prep(ORC_GATE_AND, na, c, n, t2);
  orc_bootstrap_without_key_switch(ck, t2, u2);
  orc_gate_prep(ORC_GATE_OR, u1, u2, n, t1);
  orc_bootstrap(ck, t1, NULL, out);
}

/* gates.rs:189-199 Gates::mux_naive */
void orc_mux_naive(const orc_cloud_key *ck, const uint32_t *a, const uint32_t *b,
                   const uint32_t *c, uint32_t *out) {
  const int n = ck->P.n;
  uint32_t t[2048], a_and_b[2048], nand_a_c[2048], na[2048];
  orc_gate_prep(ORC_GATE_AND, a, b, n, t);
  orc_bootstrap(ck, t, NULL, a_and_b);
  for (int i = 0; i <= n; i++) na[i] = 0u - a[i];
  orc_gate_prep(ORC_GATE_AND, na, c, n, t);
  orc_bootstrap(ck, t, NULL, nand_a_c);
  orc_gate_prep(ORC_GATE_OR, a_and_b, nand_a_c, n, t);
  orc_bootstrap(ck, t, NULL, out);
}

void orc_batch_mux(const orc_cloud_key *ck, int naive, const uint32_t *a, const uint32_t *b,
                   const uint32_t *c, uint32_t *out, int count, int nthreads) {
  orc_init();
  const int n = ck->P.n;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
  for (int i = 0; i < count; i++) {
    size_t o = (size_t)i * (n + 1);
    if (naive)
      orc_mux_naive(ck, a + o, b + o, c + o, out + o);
    else
      orc_mux(ck, a + o, b + o, c + o, out + o);
  }
}

/* ------------------------------------------------------------------------- */
/* LUT construction: src/lut/encoder.rs:29-73,96-105; src/lut/generator.rs:89-137,264-266
 * ------------------------------------------------------------------------- */
static size_t div_round(size_t a, size_t b) { return (a + b / 2) / b; } /* generator.rs:264-266 */

size_t orc_div_round(size_t a, size_t b) { return div_round(a, b); }

/* encoder.rs:66-73 with scale = 1/(2m) (:29-41) */
uint32_t orc_lut_encode(int message, int message_modulus) {
  int m = message % message_modulus;
  double scale = 1.0 / (2.0 * (double)message_modulus);
  return orc_f64_to_torus((double)m * scale);
}

/* encoder.rs:96-105 */
int orc_lut_decode(uint32_t value, int message_modulus) {
  double scale = 1.0 / (2.0 * (double)message_modulus);
  double f = orc_torus_to_f64(value);
  uint64_t message = (uint64_t)(f / scale + 0.5);
  return (int)(message % (uint64_t)message_modulus);
}

/* generator.rs:89-137: fvals[x] = f(x) for x < m; out testvec a||b */
void orc_lut_generate(const int *fvals, int message_modulus, uint32_t *testvec) {
  const size_t size = ORC_N;
  uint32_t raw[ORC_N], rot[ORC_N];
  memset(raw, 0, sizeof(raw));
  for (int x = 0; x < message_modulus; x++) {
    size_t start = div_round((size_t)x * size, (size_t)message_modulus);
    size_t end = div_round((size_t)(x + 1) * size, (size_t)message_modulus);
    uint32_t enc = orc_lut_encode(fvals[x], message_modulus);
    for (size_t i = start; i < end && i < size; i++) raw[i] = enc;
  }
  size_t offset = div_round(size, (size_t)(2 * message_modulus));
  for (size_t i = 0; i < size; i++) rot[i] = raw[(i + offset) % size];
  for (size_t i = size - offset; i < size; i++) rot[i] = 0u - rot[i]; /* wrapping_neg */
  for (size_t i = 0; i < size; i++) {
    testvec[i] = 0;
    testvec[ORC_N + i] = rot[i];
  }
}

/* ------------------------------------------------------------------------- */
/* Proxy re-encryption: src/proxy_reenc.rs (feature `proxy-reenc`)             */
/* ------------------------------------------------------------------------- */
/* PublicKeyLv0::new_with_params (proxy_reenc.rs:144-153): `size` encryptions of zero; pk [size][n+1] */
void orc_gen_public_key_lv0(uint64_t seed, const uint32_t *key_lv0, int n, int size, double alpha,
                            uint32_t *pk) {
#pragma omp parallel for schedule(static)
  for (int c = 0; c < size; c++) {
    orc_rng r;
    orc_rng_seed(&r, seed + 0xA24BAED4963EE407ull * (uint64_t)(c + 1));
    tlwe_encrypt_f64_rng(&r, 0.0, alpha, key_lv0, n, pk + (size_t)c * (n + 1));
  }
}

/* PublicKeyLv0::encrypt_f64 (proxy_reenc.rs:168-200): b = f64_to_torus(p); every encryption of zero joins with
 * probability 1/2 (gen_bool(0.5), :178), added or subtracted with probability 1/2 (:180-188); fresh noise on b (:193-197) */
static void public_key_encrypt_f64_rng(orc_rng *r, const uint32_t *pk, int size, int n, double p,
                                       double alpha, uint32_t *out) {
  memset(out, 0, (size_t)(n + 1) * sizeof(uint32_t));
  out[n] = orc_f64_to_torus(p);
  for (int e = 0; e < size; e++) {
    const uint32_t *enc = pk + (size_t)e * (n + 1);
    if (rng_next(r) >> 63) {
      if (rng_next(r) >> 63) {
        for (int i = 0; i <= n; i++) out[i] += enc[i];
      } else {
        for (int i = 0; i <= n; i++) out[i] -= enc[i];
      }
    }
  }
  out[n] += gaussian_f64(r, 0.0, alpha);
}

void orc_public_key_encrypt_f64_batch(uint64_t seed, const uint32_t *pk, int size, int n, const double *p,
                                      int count, double alpha, uint32_t *out) {
#pragma omp parallel for schedule(static)
  for (int c = 0; c < count; c++) {
    orc_rng r;
    orc_rng_seed(&r, seed + 0x9FB21C651E98DF25ull * (uint64_t)(c + 1));
    public_key_encrypt_f64_rng(&r, pk, size, n, p[c], alpha, out + (size_t)c * (n + 1));
  }
}

/* ProxyReencryptionKey::new_symmetric_with_params (proxy_reenc.rs:389-425) when pk == NULL,
 * ::new_asymmetric_with_params (:294-330) otherwise.  key [n][t][base][n+1], the k == 0 entries stay zero (:311-313) */
void orc_gen_reenc_key(uint64_t seed, const orc_params *P, const uint32_t *key_from, const uint32_t *key_to,
                       const uint32_t *pk, int pk_size, double alpha, uint32_t *key) {
  const int base = 1 << P->basebit, n = P->n;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) {
    orc_rng r;
    orc_rng_seed(&r, seed ^ (0xD6E8FEB86659FD93ull * (uint64_t)(i + 1)));
    for (int j = 0; j < P->t; j++) {
      for (int k = 0; k < base; k++) {
        size_t idx = ((size_t)base * P->t * i) + ((size_t)base * j) + k;
        uint32_t *row = key + idx * (size_t)(n + 1);
        if (k == 0) {
          memset(row, 0, (size_t)(n + 1) * sizeof(uint32_t));
          continue;
        }
        double p = (double)((uint32_t)k * key_from[i]) / (double)(1u << ((j + 1) * P->basebit));
        if (pk)
          public_key_encrypt_f64_rng(&r, pk, pk_size, n, p, alpha, row);
        else
          tlwe_encrypt_f64_rng(&r, p, alpha, key_to, n, row);
      }
    }
  }
}

/* reencrypt_tlwe_lv0 (proxy_reenc.rs:468-510): src [n+1]; key [n][t][base][n+1]; out [n+1] */
void orc_reencrypt_tlwe_lv0(const uint32_t *src, const uint32_t *key, const orc_params *P, uint32_t *out) {
  const int n = P->n, basebit = P->basebit, t = P->t;
  const int base = 1 << basebit;
  memset(out, 0, (size_t)(n + 1) * sizeof(uint32_t));
  out[n] = src[n];
  const uint32_t prec_offset = 1u << (32 - (1 + basebit * t));
  for (int i = 0; i < n; i++) {
    uint32_t a_bar = src[i] + prec_offset;
    for (int j = 0; j < t; j++) {
      uint32_t k = (a_bar >> (32 - (j + 1) * basebit)) & ((1u << basebit) - 1u);
      if (k != 0) {
        size_t idx = ((size_t)base * t * i) + ((size_t)base * j) + k;
        const uint32_t *row = key + idx * (size_t)(n + 1);
        for (int x = 0; x <= n; x++) out[x] -= row[x];
      }
    }
  }
}

void orc_batch_reencrypt(const uint32_t *in, const uint32_t *key, const orc_params *P, uint32_t *out, size_t count) {
#pragma omp parallel for schedule(dynamic, 16)
  for (long c = 0; c < (long)count; c++)
    orc_reencrypt_tlwe_lv0(in + (size_t)c * (P->n + 1), key, P, out + (size_t)c * (P->n + 1));
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
