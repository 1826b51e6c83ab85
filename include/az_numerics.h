/*
 * az_numerics.h -- the numerics + RNG contract of the self-play hot path.
 *
 * Why this header exists.  The reference (AlphaZero.jl) draws its randomness from
 * Julia's task-local RNG through Distributions.jl (`rand(Dirichlet(n, α))`,
 * src/mcts.jl:228-232; `rand(Categorical(π))`, src/util.jl:87-90) and evaluates
 * `exp`/`tanh`/`^` through Julia's libm.  None of those streams or last-bit
 * roundings can be reproduced outside Julia, so "fixed RNG" and
 * "same transcendental" are defined HERE, once, in plain C that compiles
 * unchanged for gcc (the CPU oracle), the C++ host side and HIP device code.
 * Everything is built from IEEE-754 add/mul/div/sqrt/fma only, so CPU and gfx950
 * produce bit-identical results PROVIDED the translation unit is compiled with
 * floating-point contraction OFF (-ffp-contract=off); az_numerics_selftest()
 * detects a build that contracts.
 *
 * Contents
 *   az_expf / az_tanhf            f32, used by the network heads (softmax, tanh)
 *   az_log / az_exp / az_pow      f64, used by temperatures and the Gamma sampler
 *   philox4x32-10                 counter-based generator (Salmon et al., SC'11)
 *   az_rng                        stream keyed by (seed, game id, move, purpose)
 *   az_randn / az_rand_gamma / az_dirichlet / az_categorical_f32
 *   az_mix64                      integer hash used by the synthetic test oracle
 */
#ifndef AZ_NUMERICS_H
#define AZ_NUMERICS_H

#include <stdint.h>

#if defined(__HIPCC__)
#define AZ_HD __host__ __device__ static inline
#else
#define AZ_HD static inline
#endif

/* ---------------------------------------------------------------- bit casts */
AZ_HD uint32_t az_f2u(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
AZ_HD float az_u2f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
AZ_HD uint64_t az_d2u(double d) { uint64_t u; __builtin_memcpy(&u, &d, 8); return u; }
AZ_HD double az_u2d(uint64_t u) { double d; __builtin_memcpy(&d, &u, 8); return d; }

/* single-rounding fused multiply-add (v_fma_f32 / v_fma_f64 on gfx950, vfmadd on x86) */
AZ_HD float az_fmaf(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
AZ_HD double az_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

/* ------------------------------------------------------------------ f32 exp */
/* e^x, ~1 ulp.  Range reduction x = k ln2 + r, |r| <= ln2/2, degree-6 polynomial.
 * Results below 2^-124 are flushed to 0 (callers only use it on softmax / tanh). */
AZ_HD float az_expf(float x) {
  if (!(x > -86.0f)) return 0.0f;        /* also catches NaN -> 0 (deterministic) */
  if (x > 88.0f) x = 88.0f;
  float kf = __builtin_rintf(x * 1.44269504088896341f);
  float r = az_fmaf(kf, -0.693359375f, x);
  r = az_fmaf(kf, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = az_fmaf(p, r, 1.3981999507e-3f);
  p = az_fmaf(p, r, 8.3334519073e-3f);
  p = az_fmaf(p, r, 4.1665795894e-2f);
  p = az_fmaf(p, r, 1.6666665459e-1f);
  p = az_fmaf(p, r, 5.0000001201e-1f);
  float r2 = r * r;
  float e = az_fmaf(p, r2, r) + 1.0f;
  int k = (int)kf;                        /* in [-125, 127] */
  return e * az_u2f((uint32_t)(k + 127) << 23);
}

/* tanh(x) = sign(x) (1 - 2 / (e^{2|x|} + 1)); absolute error ~1e-7 */
AZ_HD float az_tanhf(float x) {
  float a = __builtin_fabsf(x);
  float r;
  if (!(a <= 9.0f)) r = 1.0f;
  else {
    float t = az_expf(2.0f * a);
    r = 1.0f - 2.0f / (t + 1.0f);
  }
  return __builtin_copysignf(r, x);
}

/* ------------------------------------------------------------------ f64 log */
/* natural log for finite x > 0 (callers guarantee it); classic argument reduction
 * x = 2^k (1+f), sqrt(2)/2 < 1+f < sqrt(2), log(1+f) = 2s + s R(s^2), s = f/(2+f). */
AZ_HD double az_log(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01,
               L3 = 2.857142874366239149e-01, L4 = 2.222219843214978396e-01,
               L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
               L7 = 1.479819860511658591e-01;
  uint64_t u = az_d2u(x);
  int k = 0;
  if ((u >> 52) == 0) { x *= 18014398509481984.0; u = az_d2u(x); k -= 54; } /* subnormal */
  k += (int)((u >> 52) & 0x7ff) - 1023;
  uint64_t m = u & 0x000fffffffffffffULL;
  /* mantissa in [1,2); move to [sqrt(2)/2, sqrt(2)) */
  if (m >= 0x6a09e667f3bcdULL) { k += 1; u = m | 0x3fe0000000000000ULL; }
  else u = m | 0x3ff0000000000000ULL;
  double f = az_u2d(u) - 1.0;
  double dk = (double)k;
  double s = f / (2.0 + f);
  double z = s * s;
  double w = z * z;
  double t1 = w * (L2 + w * (L4 + w * L6));
  double t2 = z * (L1 + w * (L3 + w * (L5 + w * L7)));
  double R = t2 + t1;
  double hfsq = 0.5 * f * f;
  return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

/* log2 for finite x >= 1: exponent + log(mantissa) / ln 2 -- exact on powers of two (LOG_WEIGHT, learning.jl:25) */
AZ_HD double az_log2(double x) {
  uint64_t u = az_d2u(x);
  int k = (int)((u >> 52) & 0x7ff) - 1023;
  double m = az_u2d((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
  return (double)k + az_log(m) * 1.44269504088896338700e+00;
}
/* Float32 log of the loss terms (learning.jl:63-65): the f64 log rounded once */
AZ_HD float az_logf(float x) { return (float)az_log((double)x); }

/* ------------------------------------------------------------------ f64 exp */
AZ_HD double az_exp(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
               invln2 = 1.44269504088896338700e+00;
  const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
               P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
               P5 = 4.13813679705723846039e-08;
  if (!(x > -708.0)) return 0.0;
  if (x > 709.0) x = 709.0;
  double kf = __builtin_rint(x * invln2);
  double hi = x - kf * ln2_hi;
  double lo = kf * ln2_lo;
  double r = hi - lo;
  double t = r * r;
  double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
  int k = (int)kf;                        /* in [-1021, 1023] */
  /* scale in two steps so 2^k never leaves the normal range */
  int k1 = k / 2, k2 = k - k1;
  y *= az_u2d((uint64_t)(k1 + 1023) << 52);
  y *= az_u2d((uint64_t)(k2 + 1023) << 52);
  return y;
}

/* x^y for x >= 0 (probabilities): exp(y log x); 0^y = 0 for y > 0 */
AZ_HD double az_pow(double x, double y) {
  if (!(x > 0.0)) return 0.0;
  return az_exp(y * az_log(x));
}

/* ------------------------------------------------------------ philox4x32-10 */
AZ_HD void az_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int i = 0; i < 10; ++i) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Purposes (third counter word): one independent stream per use site. */
#define AZ_RNG_NOISE 1u   /* Dirichlet noise of explore!   (src/mcts.jl:228-232,240) */
#define AZ_RNG_MOVE 2u    /* categorical move sampling     (src/play.jl:311, src/util.jl:87-90) */
#define AZ_RNG_FLIP 3u    /* play_game's random symmetry    (src/play.jl:305-307, src/game.jl:329-336):
                             draw 0: rand() < flip_probability, draw 1: floor(u * #symmetries) */

#define AZ_RNG_ROLLOUT 4u /* MCTS.RolloutOracle's random playout (src/mcts.jl:41-50): simulation i of an explore!
                             owns draws i*1024 .. i*1024+1023; ply k picks available action floor(u_k * n) */

#define AZ_RNG_SHUFFLE 5u /* DataLoader(shuffle = true) of the Trainer (src/learning.jl:114-119): "game" word = epoch,
                             Fisher-Yates from the last index down, draw k swaps i with floor(u_k * (i + 1)) */

typedef struct {
  uint32_t key[2];   /* 64-bit seed */
  uint32_t ctr[4];   /* game id, move index, purpose, draw index */
} az_rng;

AZ_HD az_rng az_rng_make(uint64_t seed, uint32_t game, uint32_t move, uint32_t purpose) {
  az_rng r;
  r.key[0] = (uint32_t)seed; r.key[1] = (uint32_t)(seed >> 32);
  r.ctr[0] = game; r.ctr[1] = move; r.ctr[2] = purpose; r.ctr[3] = 0;
  return r;
}
/* every draw consumes one philox block (4 words); words 0,1 make a double, word 2 a float */
AZ_HD void az_rng_block(az_rng* r, uint32_t out[4]) {
  az_philox4x32_10(r->ctr, r->key, out);
  r->ctr[3] += 1;
}
/* uniform in (0,1), 53 bits, never 0 or 1 */
AZ_HD double az_rng_f64(az_rng* r) {
  uint32_t o[4]; az_rng_block(r, o);
  uint64_t x = (((uint64_t)o[0] << 32) | o[1]) >> 12;          /* 52 bits */
  return ((double)x + 0.5) * 2.220446049250313e-16;            /* 2^-52 */
}
/* uniform in [0,1), 24 bits */
AZ_HD float az_rng_f32(az_rng* r) {
  uint32_t o[4]; az_rng_block(r, o);
  return (float)(o[2] >> 8) * 5.9604644775390625e-8f;          /* 2^-24 */
}

/* standard normal, Marsaglia polar method (log + sqrt only) */
AZ_HD double az_randn(az_rng* r) {
  for (;;) {
    double a = 2.0 * az_rng_f64(r) - 1.0;
    double b = 2.0 * az_rng_f64(r) - 1.0;
    double s = a * a + b * b;
    if (s < 1.0 && s > 0.0) return a * __builtin_sqrt(-2.0 * az_log(s) / s);
  }
}

/* Gamma(alpha, 1), Marsaglia & Tsang (2000); alpha < 1 boosted by U^(1/alpha) */
AZ_HD double az_rand_gamma(az_rng* r, double alpha) {
  double boost = 1.0;
  if (alpha < 1.0) {
    boost = az_pow(az_rng_f64(r), 1.0 / alpha);
    alpha += 1.0;
  }
  double d = alpha - 1.0 / 3.0;
  double c = 1.0 / __builtin_sqrt(9.0 * d);
  for (;;) {
    double x = az_randn(r);
    double v = 1.0 + c * x;
    if (v <= 0.0) continue;
    v = v * v * v;
    double u = az_rng_f64(r);
    if (az_log(u) < 0.5 * x * x + d * (1.0 - v + az_log(v))) return boost * d * v;
  }
}

/* eta ~ Dirichlet(n, alpha): n Gamma draws in action-rank order, divided by their sum
 * (sum accumulated left to right). */
AZ_HD void az_dirichlet(az_rng* r, int n, double alpha, double* eta) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) { eta[i] = az_rand_gamma(r, alpha); s += eta[i]; }
  for (int i = 0; i < n; ++i) eta[i] = eta[i] / s;
}

/* Categorical draw on a Float32 probability vector with a Float32 uniform:
 * walk the Float32 cumulative sum, stop at the first cp > u or at the last index
 * (Distributions.jl's DiscreteNonParametric sampler as restated in SURVEY.md §8c). */
AZ_HD int az_categorical_f32(const float* p, int n, float u) {
  float cp = p[0];
  int i = 0;
  while (cp <= u && i < n - 1) { i += 1; cp += p[i]; }
  return i;
}

/* --------------------------------------------------------------- int hashing */
AZ_HD uint64_t az_mix64(uint64_t x) {  /* splitmix64 finaliser */
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
  x ^= x >> 27; x *= 0x94d049bb133111ebULL;
  x ^= x >> 31;
  return x;
}
AZ_HD uint64_t az_hash_key(uint64_t a, uint64_t b) {
  return az_mix64(a ^ az_mix64(b + 0x9e3779b97f4a7c15ULL));
}

/* Returns 0 when the translation unit was compiled without contraction and with a
 * single-rounding fma; non-zero otherwise. */
AZ_HD int az_numerics_selftest(void) {
  volatile float a = 1.0f + 5.9604644775390625e-8f * 2.0f;   /* 1 + 2^-23 */
  volatile float b = 1.0f - 5.9604644775390625e-8f * 2.0f;   /* 1 - 2^-23 */
  volatile float c = -1.0f;
  float unfused = a * b + c;                 /* a*b rounds to 1 -> 0 */
  float fused = az_fmaf(a, b, c);            /* exact -2^-46 */
  int bad = 0;
  if (unfused != 0.0f) bad |= 1;             /* compiler contracted a*b+c */
  if (fused != -1.4210854715202004e-14f) bad |= 2;  /* fma is not single-rounding */
  return bad;
}

#endif /* AZ_NUMERICS_H */
