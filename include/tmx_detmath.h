/*
 * tmx_detmath.h — sin / cos / atan2 with a FIXED sequence of IEEE-754 binary64 operations.
 *
 * Why: trajopt's term values (forward kinematics, tesseract's calcTransformError) go through libm's sin / cos /
 * atan2.  The reference inherits whatever libm its host has; glibc's and the device's ocml versions differ in the
 * last bit, and sco::exprToEigen keeps every entry that is not EXACTLY zero (trajopt_sco/src/solver_utils.cpp:111-144),
 * so a last-bit difference creates or drops 1e-17 entries of A, changes nnz(A) and — through the reference's own
 * warm-start test (trajopt_sco/src/osqp_interface.cpp:186-201) — the path of the whole SQP run.  Both the device
 * kernels (trajopt_amd/csrc/tmx_terms.h) and the CPU oracle (oracle/trajprob.hpp) therefore call THESE functions: only
 * + - * / on doubles and integer conversions, no FMA contraction (both are built with -ffp-contract=off), no table
 * lookups that depend on the platform.  The same inputs give the same bits on x86-64 (g++) and on gfx950 (hipcc).
 *
 * Accuracy: argument reduction by a three-part pi/2 (Cody–Waite with an exact error term, good for |x| < 1e6; larger
 * arguments are first folded with fmod, deterministically but with the accuracy of the double 2*pi) followed by the
 * classic degree-13 / degree-14 minimax kernels on [-pi/4, pi/4]; atan by breakpoint reduction (7/16, 11/16, 19/16,
 * 39/16) and a degree-11 odd/even split polynomial.  Measured against numpy / glibc over 2e6 points: sin / cos <= 1 ulp,
 * atan2 <= 2 ulp (tests/test_detmath.py).  The reference itself is specified only up to its libm, so this is as faithful a
 * restatement of "libm" as glibc is.
 */
#ifndef TMX_DETMATH_H_
#define TMX_DETMATH_H_

#if defined(__HIPCC__)
#define TMX_DM_FN __host__ __device__ static inline
#else
#define TMX_DM_FN static inline
#endif

/* round to nearest integer (ties to even) for |x| < 2^51, by the 1.5 * 2^52 shift */
TMX_DM_FN double tmx_dm_rint(double x)
{
  const double big = 6755399441055744.0;
  const double t = x + big; /* two roundings: never built with value-unsafe math */
  return t - big;
}

TMX_DM_FN double tmx_dm_abs(double x) { return __builtin_fabs(x); }

/* reduce x to r = y0 + y1 in [-pi/4, pi/4] with x = n * pi/2 + r; returns n mod 4 (0..3) */
TMX_DM_FN int tmx_dm_rem_pio2(double x, double* y0, double* y1)
{
  const double invpio2 = 6.36619772367581382433e-01;
  const double p1 = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
  const double p2 = 6.07710050630396597660e-11;  /* next 33 bits */
  const double p2t = 2.02226624879595063154e-21; /* pi/2 - (p1 + p2) */
  const double fn = tmx_dm_rint(x * invpio2);
  const double t = x - fn * p1; /* fn * p1 is exact (33 + 20 bits) */
  const double w = fn * p2;     /* exact */
  /* TwoSum(t, -w) */
  const double r = t - w;
  const double bb = r - t;
  const double e = (t - (r - bb)) + (-w - bb);
  const double c = e - fn * p2t;
  const double h = r + c;
  *y0 = h;
  *y1 = (r - h) + c;
  /* fn is an integer with |fn| < 2^21 */
  const long long k = (long long)fn;
  return (int)(k & 3LL);
}

TMX_DM_FN double tmx_dm_ksin(double x, double y)
{
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double z = x * x;
  const double w = z * z;
  const double r = (S2 + z * (S3 + z * S4)) + (z * w) * (S5 + z * S6);
  const double v = z * x;
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

TMX_DM_FN double tmx_dm_kcos(double x, double y)
{
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double z = x * x;
  const double w = z * z;
  const double r = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
  const double hz = 0.5 * z;
  const double o = 1.0 - hz;
  return o + (((1.0 - o) - hz) + (z * r - x * y));
}

/* s = sin(x), c = cos(x) */
TMX_DM_FN void tmx_sincos(double x, double* s, double* c)
{
  if (!(x == x) || x - x != 0.0) /* NaN or +-inf */
  {
    *s = *c = x - x;
    return;
  }
  if (tmx_dm_abs(x) >= 1.0e6)
  {
    /* deterministic fold; accuracy limited by the double 2*pi (never reached by joint values) */
    const double two_pi = 6.28318530717958647692528676655900577;
    const double q = x / two_pi;
    const double qi = (tmx_dm_abs(q) < 4503599627370496.0) ? tmx_dm_rint(q) : q;
    x = x - qi * two_pi;
  }
  double y0 = x, y1 = 0.0;
  int n = 0;
  if (tmx_dm_abs(x) > 0.78539816339744830962)
    n = tmx_dm_rem_pio2(x, &y0, &y1);
  const double ks = tmx_dm_ksin(y0, y1), kc = tmx_dm_kcos(y0, y1);
  switch (n)
  {
    case 0:
      *s = ks;
      *c = kc;
      break;
    case 1:
      *s = kc;
      *c = -ks;
      break;
    case 2:
      *s = -ks;
      *c = -kc;
      break;
    default:
      *s = -kc;
      *c = ks;
      break;
  }
}
TMX_DM_FN double tmx_sin(double x)
{
  double s, c;
  tmx_sincos(x, &s, &c);
  return s;
}
TMX_DM_FN double tmx_cos(double x)
{
  double s, c;
  tmx_sincos(x, &s, &c);
  return c;
}

/* atan for x >= 0 (finite or +inf) */
TMX_DM_FN double tmx_dm_atan_pos(double x)
{
  const double hi0 = 4.63647609000806093515e-01, lo0 = 2.26987774529616870924e-17; /* atan(0.5) */
  const double hi1 = 7.85398163397448278999e-01, lo1 = 3.06161699786838301793e-17; /* atan(1.0) */
  const double hi2 = 9.82793723247329054082e-01, lo2 = 1.39033110312309984516e-17; /* atan(1.5) */
  const double hi3 = 1.57079632679489655800e+00, lo3 = 6.12323399573676603587e-17; /* atan(inf) */
  const double a0 = 3.33333333333329318027e-01, a1 = -1.99999999998764832476e-01, a2 = 1.42857142725034663711e-01,
               a3 = -1.11111104054623557880e-01, a4 = 9.09088713343650656196e-02, a5 = -7.69187620504482999495e-02,
               a6 = 6.66107313738753120669e-02, a7 = -5.83357013379057348645e-02, a8 = 4.97687799461593236017e-02,
               a9 = -3.65315727442169155270e-02, a10 = 1.62858201153657823623e-02;
  if (x >= 7.3786976294838206464e19) /* 2^66: atan(x) == pi/2 in double */
    return hi3 + lo3;
  int id;
  double t;
  if (x < 0.4375)
  {
    if (x < 3.725290298461914e-09) /* 2^-28 */
      return x;
    id = -1;
    t = x;
  }
  else if (x < 1.1875)
  {
    if (x < 0.6875)
    {
      id = 0;
      t = (2.0 * x - 1.0) / (2.0 + x);
    }
    else
    {
      id = 1;
      t = (x - 1.0) / (x + 1.0);
    }
  }
  else if (x < 2.4375)
  {
    id = 2;
    t = (x - 1.5) / (1.0 + 1.5 * x);
  }
  else
  {
    id = 3;
    t = -1.0 / x;
  }
  const double z = t * t;
  const double w = z * z;
  const double s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
  const double s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
  if (id < 0)
    return t - t * (s1 + s2);
  const double hi = (id == 0) ? hi0 : (id == 1) ? hi1 : (id == 2) ? hi2 : hi3;
  const double lo = (id == 0) ? lo0 : (id == 1) ? lo1 : (id == 2) ? lo2 : lo3;
  return hi - ((t * (s1 + s2) - lo) - t);
}

TMX_DM_FN double tmx_atan2(double y, double x)
{
  const double pi = 3.1415926535897931160e+00, pi_lo = 1.2246467991473531772e-16, pio2 = 1.5707963267948965580e+00;
  if (!(x == x) || !(y == y))
    return x + y;
  const int sy = (y < 0.0) || (y == 0.0 && 1.0 / y < 0.0);
  const int sx = (x < 0.0) || (x == 0.0 && 1.0 / x < 0.0);
  const double ay = tmx_dm_abs(y), ax = tmx_dm_abs(x);
  if (ay == 0.0)
    return sx ? (sy ? -pi : pi) : y;
  if (ax == 0.0)
    return sy ? -(pio2 + 0.5 * pi_lo) : (pio2 + 0.5 * pi_lo);
  const int xinf = (ax - ax != 0.0), yinf = (ay - ay != 0.0);
  double z;
  if (xinf && yinf)
    z = 0.25 * pi;
  else if (xinf)
    z = 0.0;
  else if (yinf)
    z = pio2 + 0.5 * pi_lo;
  else
    z = tmx_dm_atan_pos(ay / ax);
  if (xinf && yinf && sx)
    z = 3.0 * (0.25 * pi);
  else if (sx)
    z = pi - (z - pi_lo);
  return sy ? -z : z;
}

#endif /* TMX_DETMATH_H_ */
