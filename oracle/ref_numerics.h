/*
 * ref_numerics.h -- the ORACLE'S OWN implementation of the numerics + RNG contract (test infrastructure, see azref.c).
 *
 * include/az_numerics.h is the contract the product compiles (host side and HIP kernels).  Until round 3 the oracle included
 * that very header, so for the random streams and the transcendental functions a HIP-vs-oracle test compared the header with
 * itself.  This file restates the contract from its written definition -- same constants and the same
 * ORDER of IEEE-754 operations, because that is what the contract fixes bit for bit -- in separate code: a different Philox
 * round structure, an explicit stream object, table-driven Horner evaluation.  A slip in either implementation (a counter word
 * in the wrong place, a draw consumed twice, a coefficient, an operation order) now shows up as a parity failure, and
 * tests/test_numerics_independent.py compares the two on random inputs bit by bit on the CPU.
 *
 * The contract, in prose:
 *  R1  philox4x32-10 (Salmon et al., SC'11): multipliers 0xD2511F53 / 0xCD9E8D57, Weyl key increments 0x9E3779B9 / 0xBB67AE85.
 *  R2  A stream is (64-bit seed; game id, move index, purpose, draw index): key = seed (low word first), counter =
 *      (game, move, purpose, draw).  EVERY draw consumes one block; draw index starts at 0 and counts up by one.
 *  R3  f64 uniform in (0,1): ((w0 << 32 | w1) >> 12 + 0.5) * 2^-52;  f32 uniform in [0,1): (w2 >> 8) * 2^-24.
 *  R4  normal: Marsaglia polar, two f64 uniforms per attempt, accept 0 < s < 1, return a * sqrt(-2 log s / s).
 *  R5  Gamma(alpha): Marsaglia-Tsang; alpha < 1: boost = U^(1/alpha) drawn FIRST, alpha += 1; d = alpha - 1/3,
 *      c = 1 / sqrt(9 d); loop: x normal, v = 1 + c x (retry if <= 0), v = v^3, u uniform,
 *      accept when log u < 0.5 x^2 + d (1 - v + log v); result boost * d * v.
 *  R6  Dirichlet(n, alpha): n Gammas in order, sum accumulated left to right, each divided by the sum.
 *  R7  categorical on Float32 p with Float32 u: cumulative Float32 sum, first index with cp > u, capped at n - 1.
 *  F1  expf: x <= -86 (or NaN) -> 0; clamp at 88; k = rint(x * log2 e); r = fma(k, -0.693359375, x); r = fma(k, 2.12194440e-4, r);
 *      Horner in fma with the six Cephes coefficients; e = fma(p, r^2, r) + 1; times 2^k.
 *  F2  tanhf: |x| > 9 (or NaN) -> +-1; else 1 - 2 / (expf(2|x|) + 1), sign copied.
 *  F3  log (f64, fdlibm e_log.c reduction and coefficients), log2 = exponent + log(mantissa) * (1 / ln 2), logf = (float)log.
 *  F4  exp (f64, fdlibm e_exp.c), clamped to [-708, 709], 2^k applied in two factors k/2 and k - k/2;  pow(x, y) = exp(y log x),
 *      0 for x <= 0.
 *  H1  mix64 = splitmix64 finaliser; hash_key(a, b) = mix64(a ^ mix64(b + 0x9e3779b97f4a7c15)).
 * Compile with -ffp-contract=off.
 */
#ifndef REF_NUMERICS_H
#define REF_NUMERICS_H
#include <stdint.h>
#include <string.h>

static inline float rn_bits_f(uint32_t u) { float f; memcpy(&f, &u, sizeof f); return f; }
static inline double rn_bits_d(uint64_t u) { double d; memcpy(&d, &u, sizeof d); return d; }
static inline uint64_t rn_dbits(double d) { uint64_t u; memcpy(&u, &d, sizeof u); return u; }

/* ------------------------------------------------------------------------------------------ R1: philox4x32-10 */
static inline void rn_philox_round(uint32_t c[4], uint32_t ka, uint32_t kb) {
  const uint64_t pa = (uint64_t)c[0] * 0xD2511F53ull, pb = (uint64_t)c[2] * 0xCD9E8D57ull;
  const uint32_t a_hi = (uint32_t)(pa >> 32), a_lo = (uint32_t)pa, b_hi = (uint32_t)(pb >> 32), b_lo = (uint32_t)pb;
  const uint32_t c1 = c[1], c3 = c[3];
  c[0] = b_hi ^ c1 ^ ka; c[1] = b_lo; c[2] = a_hi ^ c3 ^ kb; c[3] = a_lo;
}
static inline void rn_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
  for (uint32_t round = 0; round < 10; ++round)
    rn_philox_round(c, key[0] + round * 0x9E3779B9u, key[1] + round * 0xBB67AE85u);   /* the Weyl sequence of round keys */
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

/* ------------------------------------------------------------------------------------------ R2, R3: streams */
enum { RN_NOISE = 1, RN_MOVE = 2, RN_FLIP = 3, RN_ROLLOUT = 4, RN_SHUFFLE = 5 };   /* purposes: mcts.jl:228-232 | util.jl:87-90 | play.jl:305-307 | mcts.jl:41-50 | learning.jl:114-119 */
typedef struct { uint64_t seed; uint32_t game, move, purpose, draw; } rn_stream;
static inline rn_stream rn_open(uint64_t seed, uint32_t game, uint32_t move, uint32_t purpose) {
  rn_stream s; s.seed = seed; s.game = game; s.move = move; s.purpose = purpose; s.draw = 0; return s;
}
static inline void rn_next(rn_stream* s, uint32_t w[4]) {
  const uint32_t ctr[4] = {s->game, s->move, s->purpose, s->draw}, key[2] = {(uint32_t)(s->seed & 0xffffffffu), (uint32_t)(s->seed >> 32)};
  rn_philox(ctr, key, w);
  s->draw += 1u;
}
static inline double rn_u64(rn_stream* s) {
  uint32_t w[4]; rn_next(s, w);
  const uint64_t m = ((uint64_t)w[0] << 32 | (uint64_t)w[1]) >> 12;
  return ((double)m + 0.5) * 0x1p-52;
}
static inline float rn_u32(rn_stream* s) {
  uint32_t w[4]; rn_next(s, w);
  return (float)(w[2] >> 8) * 0x1p-24f;
}

/* ------------------------------------------------------------------------------------------ F1, F2: expf, tanhf */
static inline float rn_expf(float x) {
  static const float C[6] = {1.9875691500e-4f, 1.3981999507e-3f, 8.3334519073e-3f, 4.1665795894e-2f, 1.6666665459e-1f, 5.0000001201e-1f};
  if (!(x > -86.0f)) return 0.0f;
  if (x > 88.0f) x = 88.0f;
  const float k = __builtin_rintf(x * 1.44269504088896341f);
  float r = __builtin_fmaf(k, -0.693359375f, x);
  r = __builtin_fmaf(k, 2.12194440e-4f, r);
  float p = C[0];
  for (int i = 1; i < 6; ++i) p = __builtin_fmaf(p, r, C[i]);
  const float e = __builtin_fmaf(p, r * r, r) + 1.0f;
  return e * rn_bits_f((uint32_t)((int)k + 127) << 23);
}
static inline float rn_tanhf(float x) {
  const float a = __builtin_fabsf(x);
  float r = 1.0f;
  if (a <= 9.0f) r = 1.0f - 2.0f / (rn_expf(2.0f * a) + 1.0f);
  return __builtin_copysignf(r, x);
}

/* ------------------------------------------------------------------------------------------ F3: log */
static inline double rn_log(double x) {
  static const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
  static const double Lg[7] = {6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01, 2.222219843214978396e-01,
                               1.818357216161805012e-01, 1.531383769920937332e-01, 1.479819860511658591e-01};
  uint64_t bits = rn_dbits(x);
  int k = 0;
  if ((bits >> 52) == 0) { x *= 0x1p54; bits = rn_dbits(x); k = -54; }            /* subnormal */
  k += (int)((bits >> 52) & 0x7ff) - 1023;
  const uint64_t frac = bits & 0x000fffffffffffffull;
  double m;                                                                       /* mantissa moved to [sqrt 2 / 2, sqrt 2) */
  if (frac >= 0x6a09e667f3bcdull) { k += 1; m = rn_bits_d(frac | 0x3fe0000000000000ull); }
  else m = rn_bits_d(frac | 0x3ff0000000000000ull);
  const double f = m - 1.0, dk = (double)k;
  const double s = f / (2.0 + f), z = s * s, w = z * z;
  const double t1 = w * (Lg[1] + w * (Lg[3] + w * Lg[5]));
  const double t2 = z * (Lg[0] + w * (Lg[2] + w * (Lg[4] + w * Lg[6])));
  const double R = t2 + t1, hfsq = 0.5 * f * f;
  return dk * LN2_HI - ((hfsq - (s * (hfsq + R) + dk * LN2_LO)) - f);
}
static inline double rn_log2(double x) {
  const uint64_t bits = rn_dbits(x);
  const int k = (int)((bits >> 52) & 0x7ff) - 1023;
  return (double)k + rn_log(rn_bits_d((bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull)) * 1.44269504088896338700e+00;
}
static inline float rn_logf(float x) { return (float)rn_log((double)x); }

/* ------------------------------------------------------------------------------------------ F4: exp, pow */
static inline double rn_exp(double x) {
  static const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10, INV_LN2 = 1.44269504088896338700e+00;
  static const double P[5] = {1.66666666666666019037e-01, -2.77777777770155933842e-03, 6.61375632143793436117e-05,
                              -1.65339022054652515390e-06, 4.13813679705723846039e-08};
  if (!(x > -708.0)) return 0.0;
  if (x > 709.0) x = 709.0;
  const double kf = __builtin_rint(x * INV_LN2);
  const double hi = x - kf * LN2_HI, lo = kf * LN2_LO, r = hi - lo, t = r * r;
  const double c = r - t * (P[0] + t * (P[1] + t * (P[2] + t * (P[3] + t * P[4]))));
  double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
  const int k = (int)kf, ka = k / 2, kb = k - ka;
  y *= rn_bits_d((uint64_t)(ka + 1023) << 52);
  y *= rn_bits_d((uint64_t)(kb + 1023) << 52);
  return y;
}
static inline double rn_pow(double x, double y) { return x > 0.0 ? rn_exp(y * rn_log(x)) : 0.0; }

/* ------------------------------------------------------------------------------------------ R4 .. R7 */
static inline double rn_normal(rn_stream* s) {
  double a, b, q;
  do {
    a = 2.0 * rn_u64(s) - 1.0;
    b = 2.0 * rn_u64(s) - 1.0;
    q = a * a + b * b;
  } while (!(q < 1.0 && q > 0.0));
  return a * __builtin_sqrt(-2.0 * rn_log(q) / q);
}
static inline double rn_gamma(rn_stream* s, double alpha) {
  double boost = 1.0;
  if (alpha < 1.0) { boost = rn_pow(rn_u64(s), 1.0 / alpha); alpha += 1.0; }
  const double d = alpha - 1.0 / 3.0, c = 1.0 / __builtin_sqrt(9.0 * d);
  for (;;) {
    const double x = rn_normal(s);
    double v = 1.0 + c * x;
    if (v <= 0.0) continue;
    v = v * v * v;
    const double u = rn_u64(s);
    if (rn_log(u) < 0.5 * x * x + d * (1.0 - v + rn_log(v))) return boost * d * v;
  }
}
static inline void rn_dirichlet(rn_stream* s, int n, double alpha, double* eta) {
  double total = 0.0;
  for (int i = 0; i < n; ++i) { eta[i] = rn_gamma(s, alpha); total += eta[i]; }
  for (int i = 0; i < n; ++i) eta[i] = eta[i] / total;
}
static inline int rn_categorical(const float* p, int n, float u) {
  float cp = p[0];
  int i = 0;
  while (i < n - 1 && cp <= u) { ++i; cp += p[i]; }
  return i;
}

/* ------------------------------------------------------------------------------------------ H1 */
static inline uint64_t rn_mix64(uint64_t x) {
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}
static inline uint64_t rn_hash_key(uint64_t a, uint64_t b) { return rn_mix64(a ^ rn_mix64(b + 0x9e3779b97f4a7c15ull)); }

/* 0 when the translation unit neither contracts a*b+c nor has a double-rounding fma */
static inline int rn_selftest(void) {
  volatile float a = 1.0f + 0x1p-23f, b = 1.0f - 0x1p-23f, c = -1.0f;
  const float plain = a * b + c, fused = __builtin_fmaf(a, b, c);
  return (plain != 0.0f ? 1 : 0) | (fused != -0x1p-46f ? 2 : 0);
}
#endif /* REF_NUMERICS_H */
