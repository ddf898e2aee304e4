/* go2sim_detmath.h -- bit-reproducible single-precision elementary functions.
 *
 * Numeric contract of the go2sim C ABI: every transcendental function used on the hot path is
 * evaluated with the fixed sequences of IEEE-754 binary32 add/mul/div/sqrt below (no libm, no
 * device intrinsics, no FMA contraction: build with -ffp-contract=off).  The same header is compiled
 * by gcc for the CPU oracle and by hipcc for gfx950, so CPU and GPU results are bit-identical and the
 * parity tests can demand exact contact counts / done masks.
 *
 * The reference evaluates these through its JIT's libm / fast-math lowering (quadrants,
 * genesis/__init__.py:289 `fast_math=not debug`); the polynomials here (classic Cephes single
 * precision forms) are within 2 ulp of a correctly rounded result on the ranges the path uses, i.e.
 * tighter than the reference's own GPU fast-math path.  tests/test_detmath.py checks this.
 */
#ifndef GO2SIM_DETMATH_H
#define GO2SIM_DETMATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define DM_FN __host__ __device__ static inline
#else
#define DM_FN static inline
#endif

DM_FN float dm_bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
DM_FN uint32_t dm_f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

DM_FN float dm_sqrt(float x) { return sqrtf(x); } /* correctly rounded on both targets */
DM_FN float dm_abs(float x) { return dm_bits2f(dm_f2bits(x) & 0x7fffffffu); }
/* Arrow shape of an 18-dof floating-base quadruped (FAST ORDER factorisation of the Newton Hessian): dofs 0..5 are the base, the other twelve form four
 * chains of three that meet only in the base.  `mask` is mass_parent_mask (row-major nd x nd, lower triangle read).  Returns how the chains are numbered:
 * 1 = breadth-first (leg l holds dofs 6 + l, 10 + l, 14 + l: the order Genesis gives Go2 and ANYmal), 2 = depth-first (6 + 3 l .. 8 + 3 l), 0 = neither. */
DM_FN int dm_arrow_leg(int mode, int d) { return mode == 1 ? ((d - 6) & 3) : ((d - 6) / 3); }
DM_FN int dm_arrow_dof(int mode, int leg, int t) { return mode == 1 ? (6 + leg + 4 * t) : (6 + 3 * leg + t); }
static inline int dm_arrow_mode(const float* mask, int nd) {
  if (nd != 18) return 0;
  for (int mode = 1; mode <= 2; ++mode) {
    int ok = 1;
    for (int i = 0; i < nd && ok; ++i)
      for (int j = 0; j <= i && ok; ++j) {
        const int expect = (i < 6 || j < 6) ? 1 : (dm_arrow_leg(mode, i) == dm_arrow_leg(mode, j));
        ok = ((mask[i * nd + j] != 0.0f) ? 1 : 0) == expect;
      }
    if (ok) return mode;
  }
  return 0;
}
DM_FN float dm_floor(float x) { return floorf(x); }
DM_FN float dm_ceil(float x) { return ceilf(x); }

/* sin and cos of x, |x| < ~8000 (Cephes sinf/cosf, 3-term Cody-Waite reduction by pi/4) */
DM_FN void dm_sincos(float xx, float* s, float* c) {
  const float DP1 = 0.78515625f, DP2 = 2.4187564849853515625e-4f, DP3 = 3.77489497744594108e-8f;
  const float FOPI = 1.27323954473516f;
  float x = dm_abs(xx);
  int sign_s = (xx < 0.0f) ? -1 : 1;
  int sign_c = 1;
  int j = (int)(FOPI * x);
  float y = (float)j;
  if (j & 1) { j += 1; y += 1.0f; }
  j &= 7;
  if (j > 3) { sign_s = -sign_s; sign_c = -sign_c; j -= 4; }
  if (j > 1) sign_c = -sign_c;
  x = ((x - y * DP1) - y * DP2) - y * DP3;
  float z = x * x;
  float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * x + x;
  float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
  float rs, rc;
  if (j == 1 || j == 2) { rs = pc; rc = ps; } else { rs = ps; rc = pc; }
  *s = (sign_s < 0) ? -rs : rs;
  *c = (sign_c < 0) ? -rc : rc;
}
DM_FN float dm_sin(float x) { float s, c; dm_sincos(x, &s, &c); return s; }
DM_FN float dm_cos(float x) { float s, c; dm_sincos(x, &s, &c); return c; }

DM_FN float dm_atan(float xx) {
  float x = dm_abs(xx), y;
  if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
  else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
  else y = 0.0f;
  float z = x * x;
  y += (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
  return (xx < 0.0f) ? -y : y;
}

DM_FN float dm_atan2(float y, float x) {
  const float PI = 3.14159265358979323846f, PIO2 = 1.5707963267948966f;
  if (x == 0.0f) {
    if (y > 0.0f) return PIO2;
    if (y < 0.0f) return -PIO2;
    return 0.0f;
  }
  if (y == 0.0f) return (x < 0.0f) ? PI : 0.0f;
  float z = dm_atan(y / x);
  if (x < 0.0f) z = (y < 0.0f) ? (z - PI) : (z + PI);
  return z;
}

DM_FN float dm_asin_core(float a) { /* 0 <= a <= 0.5 polynomial */
  float z = a * a;
  return ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * a + a;
}
/* acos with the argument clamped to [-1,1] (directions are unit vectors up to rounding) */
DM_FN float dm_acos(float x) {
  const float PI = 3.14159265358979323846f, PIO2 = 1.5707963267948966f;
  if (x >= 1.0f) return 0.0f;
  if (x <= -1.0f) return PI;
  if (x < -0.5f) return PI - 2.0f * dm_asin_core(dm_sqrt(0.5f * (1.0f + x)));
  if (x > 0.5f) return 2.0f * dm_asin_core(dm_sqrt(0.5f * (1.0f - x)));
  float a = dm_abs(x);
  float r = dm_asin_core(a);
  return PIO2 - ((x < 0.0f) ? -r : r);
}

DM_FN float dm_ldexp(float x, int n) { /* x * 2^n, n in [-126,127] after clamping by caller */
  return x * dm_bits2f((uint32_t)(n + 127) << 23);
}

DM_FN float dm_exp(float xx) {
  const float LOG2EF = 1.44269504088896341f, C1 = 0.693359375f, C2 = -2.12194440e-4f;
  float x = xx;
  if (x > 88.0f) return dm_bits2f(0x7f800000u);
  if (x < -87.0f) return 0.0f;
  float z = dm_floor(LOG2EF * x + 0.5f);
  x -= z * C1;
  x -= z * C2;
  int n = (int)z;
  float z2 = x * x;
  float p = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x + 1.6666665459e-1f) * x + 5.0000001201e-1f) * z2 + x + 1.0f;
  return dm_ldexp(p, n);
}

/* natural log, x > 0 (normal numbers) */
DM_FN float dm_log(float xx) {
  const float SQRTHF = 0.707106781186547524f;
  uint32_t u = dm_f2bits(xx);
  int e = (int)((u >> 23) & 0xff) - 126;
  float x = dm_bits2f((u & 0x007fffffu) | 0x3f000000u); /* [0.5,1) */
  if (x < SQRTHF) { e -= 1; x = x + x - 1.0f; } else { x = x - 1.0f; }
  float z = x * x;
  float y = ((((((((7.0376836292e-2f * x - 1.1514610310e-1f) * x + 1.1676998740e-1f) * x - 1.2420140846e-1f) * x + 1.4249322787e-1f) * x - 1.6668057665e-1f) * x + 2.0000714765e-1f) * x - 2.4999993993e-1f) * x + 3.3333331174e-1f) * x * z;
  float fe = (float)e;
  y += -2.12194440e-4f * fe;
  y += -0.5f * z;
  z = x + y;
  z += 0.693359375f * fe;
  return z;
}

/* x^p for the solver-impedance curve (geom.py:405-422); exact repeated product for p == 2 */
DM_FN float dm_pow(float x, float p) {
  if (p == 2.0f) return x * x;
  if (p == 1.0f) return x;
  if (x <= 0.0f) return 0.0f;
  return dm_exp(p * dm_log(x));
}

/* ---- counter-based RNG: Philox4x32-10 ------------------------------------------------------- */
typedef struct { uint32_t v[4]; } dm_u4;
DM_FN dm_u4 dm_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  dm_u4 o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3; return o;
}
#ifdef GO2SIM_RNG_CONST
/* DIAGNOSTIC BUILD ONLY (tools/make_ref_env_fixtures.py "rng" cases, tests/test_ref_env_rng_fixtures.py; never the product): the generator is
   replaced by a four-entry schedule of constants so that the reference's env files -- run with torch.rand / randn_like / randint / randperm
   replaced by the same schedule -- and this library draw identical numbers.  A "random word" is then the stream key (env step for the
   per-step draws, reset-call number for the reset draws); every uniform of that key is DM_RNG_CONST_U[key & 3], every normal
   DM_RNG_CONST_Z[key & 3], an integer in [lo, hi] is lo + (int)(u * (hi - lo + 1)) and a permutation is the identity.  All table entries are
   dyadic, so every product with them is the same in float32, float64 and torch. */
DM_FN float dm_rng_const_u(uint32_t key) { return (key & 3u) == 0u ? 0.25f : (key & 3u) == 1u ? 0.75f : (key & 3u) == 2u ? 0.0625f : 0.5f; }
DM_FN float dm_rng_const_z(uint32_t key) { return (key & 3u) == 0u ? 0.5f : (key & 3u) == 1u ? -1.0f : (key & 3u) == 2u ? 0.25f : -0.5f; }
DM_FN float dm_u01(uint32_t r) { return dm_rng_const_u(r); }
DM_FN void dm_normal2(uint32_t r0, uint32_t r1, float* n0, float* n1) { (void)r1; *n0 = dm_rng_const_z(r0); *n1 = dm_rng_const_z(r0); }
#define DM_NORMAL2_DEFINED 1
#else
/* uniform in [0,1) with 24 random bits (same granularity as torch.rand for float32) */
DM_FN float dm_u01(uint32_t r) { return (float)(r >> 8) * 5.9604644775390625e-8f; }
#endif
#ifndef DM_NORMAL2_DEFINED
/* two standard normals from two words (Box-Muller on (0,1] x [0,1)) */
DM_FN void dm_normal2(uint32_t r0, uint32_t r1, float* n0, float* n1) {
  float u = 1.0f - dm_u01(r0);
  float v = dm_u01(r1);
  float rad = dm_sqrt(-2.0f * dm_log(u));
  float s, c;
  dm_sincos(6.283185307179586f * v, &s, &c);
  *n0 = rad * c; *n1 = rad * s;
}
#endif

#endif /* GO2SIM_DETMATH_H */
