// tc_trig.h -- deterministic double-precision sin / cos / tan / atan2.
//
// Why this exists: the reference computes its transcendentals with the host libm (CPython
// `math.*`, car.py:92-122, layer.py:122,140-141,181).  A GPU has no glibc; AMD's OCML versions
// differ from glibc by an ulp here and there, and those ulps feed integer decisions (argmin over
// edge orientations, `<= pi/2`, float->int32 pixel truncation).  To make the HIP path testable
// bit-for-bit we use ONE algorithm, written with only IEEE-754 basic operations (+ - * / and
// integer bit tests, no FMA contraction: build with -ffp-contract=off), that compiles to the
// same results on the host (gcc) and on gfx950 (hipcc).  The algorithms are the classic
// argument-reduction + minimax-polynomial kernels published with Sun's freely distributable
// libm (Cody-Waite pi/2 reduction in up to three steps; sin/cos degree-13/14 kernels; tangent
// degree-27 kernel; arctangent with 4 break points).  Accuracy: < 1 ulp, checked against glibc
// in tests/test_trig.py.
//
// All functions are `static inline` and usable from C, C++ and HIP (TC_HD expands to
// `__host__ __device__` under hipcc).
#ifndef TC_TRIG_H
#define TC_TRIG_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define TC_HD __host__ __device__ static inline
#else
#define TC_HD static inline
#endif

TC_HD uint64_t tc_d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
TC_HD double tc_u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
TC_HD int32_t tc_hi(double x) { return (int32_t)(tc_d2u(x) >> 32); }
TC_HD uint32_t tc_lo(double x) { return (uint32_t)tc_d2u(x); }
TC_HD double tc_fabs(double x) { return tc_u2d(tc_d2u(x) & 0x7fffffffffffffffULL); }
TC_HD int tc_isnan(double x) { return x != x; }

// ---------------------------------------------------------------- pi/2 argument reduction
// Returns n (mod 4 matters) and y0+y1 = x - n*pi/2 with |y0| <= pi/4 (+ a hair).
// Valid for |x| < 2^20*pi/2; tinycarlo angles stay within a few multiples of pi.
TC_HD int tc_rem_pio2(double x, double* y0, double* y1) {
  const double invpio2 = 6.36619772367581382433e-01;
  const double pio2_1 = 1.57079632673412561417e+00;   // first 33 bits of pi/2
  const double pio2_1t = 6.07710050650619224932e-11;  // pi/2 - pio2_1
  const double pio2_2 = 6.07710050630396597660e-11;   // second 33 bits
  const double pio2_2t = 2.02226624879595063154e-21;  // pi/2 - (pio2_1+pio2_2)
  const double pio2_3 = 2.02226624871116645580e-21;   // third 33 bits
  const double pio2_3t = 8.47842766036889956997e-32;  // pi/2 - (pio2_1+pio2_2+pio2_3)
  int32_t hx = tc_hi(x);
  int32_t ix = hx & 0x7fffffff;
  double t = tc_fabs(x);
  int n = (int)(t * invpio2 + 0.5);
  double fn = (double)n;
  double r = t - fn * pio2_1;
  double w = fn * pio2_1t;
  double a = r - w;
  int j = ix >> 20;
  int i = j - ((tc_hi(a) >> 20) & 0x7ff);
  if (i > 16) {  // cancellation: second step
    double tt = r;
    w = fn * pio2_2;
    r = tt - w;
    w = fn * pio2_2t - ((tt - r) - w);
    a = r - w;
    i = j - ((tc_hi(a) >> 20) & 0x7ff);
    if (i > 49) {  // third step, covers every double
      tt = r;
      w = fn * pio2_3;
      r = tt - w;
      w = fn * pio2_3t - ((tt - r) - w);
      a = r - w;
    }
  }
  double b = (r - a) - w;
  if (hx < 0) {
    *y0 = -a;
    *y1 = -b;
    return -n;
  }
  *y0 = a;
  *y1 = b;
  return n;
}

// ---------------------------------------------------------------- kernels on [-pi/4, pi/4]
TC_HD double tc_ksin(double x, double y, int iy) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  int32_t ix = tc_hi(x) & 0x7fffffff;
  if (ix < 0x3e400000) {  // |x| < 2^-27
    if ((int)x == 0) return x;
  }
  double z = x * x;
  double v = z * x;
  double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  if (iy == 0) return x + v * (S1 + z * r);
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

TC_HD double tc_kcos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  int32_t ix = tc_hi(x) & 0x7fffffff;
  if (ix < 0x3e400000) {
    if ((int)x == 0) return 1.0;
  }
  double z = x * x;
  double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  if (ix < 0x3FD33333) return 1.0 - (0.5 * z - (z * r - x * y));
  double qx;
  if (ix > 0x3fe90000) {
    qx = 0.28125;
  } else {
    qx = tc_u2d((uint64_t)(uint32_t)(ix - 0x00200000) << 32);  // ~x/4
  }
  double hz = 0.5 * z - qx;
  double a = 1.0 - qx;
  return a - (hz - (z * r - x * y));
}

TC_HD double tc_ktan(double x, double y, int iy) {
  const double T0 = 3.33333333333334091986e-01, T1 = 1.33333333333201242699e-01,
               T2 = 5.39682539762260521377e-02, T3 = 2.18694882948595424599e-02,
               T4 = 8.86323982359930005737e-03, T5 = 3.59207910759131235356e-03,
               T6 = 1.45620945432529025516e-03, T7 = 5.88041240820264096874e-04,
               T8 = 2.46463134818469906812e-04, T9 = 7.81794442939557092300e-05,
               T10 = 7.14072491382608190305e-05, T11 = -1.85586374855275456654e-05,
               T12 = 2.59073051863633712884e-05;
  const double pio4 = 7.85398163397448278999e-01, pio4lo = 3.06161699786838301793e-17;
  int32_t hx = tc_hi(x);
  int32_t ix = hx & 0x7fffffff;
  if (ix < 0x3e300000) {  // |x| < 2^-28
    if ((int)x == 0) {
      if (((ix | tc_lo(x)) | (uint32_t)(iy + 1)) == 0) return 1.0 / tc_fabs(x);
      if (iy == 1) return x;
      return -1.0 / x;
    }
  }
  int big = ix >= 0x3FE59428;  // |x| >= 0.6744
  if (big) {
    if (hx < 0) {
      x = -x;
      y = -y;
    }
    double z0 = pio4 - x;
    double w0 = pio4lo - y;
    x = z0 + w0;
    y = 0.0;
  }
  double z = x * x;
  double w = z * z;
  double r = T1 + w * (T3 + w * (T5 + w * (T7 + w * (T9 + w * T11))));
  double v = z * (T2 + w * (T4 + w * (T6 + w * (T8 + w * (T10 + w * T12)))));
  double s = z * x;
  r = y + z * (s * (r + v) + y);
  r += T0 * s;
  w = x + r;
  if (big) {
    v = (double)iy;
    return (double)(1 - ((hx >> 30) & 2)) * (v - 2.0 * (x - (w * w / (w + v) - r)));
  }
  if (iy == 1) return w;
  // -1/(x+r), accurately
  double zz = tc_u2d(tc_d2u(w) & 0xffffffff00000000ULL);
  v = r - (zz - x);
  double a = -1.0 / w;
  double t = tc_u2d(tc_d2u(a) & 0xffffffff00000000ULL);
  s = 1.0 + t * zz;
  return t + a * (s + t * v);
}

// ---------------------------------------------------------------- public functions
TC_HD double tc_sin(double x) {
  int32_t ix = tc_hi(x) & 0x7fffffff;
  if (ix <= 0x3fe921fb) return tc_ksin(x, 0.0, 0);
  if (ix >= 0x7ff00000) return x - x;  // inf/nan -> nan
  double y0, y1;
  int n = tc_rem_pio2(x, &y0, &y1);
  switch (n & 3) {
    case 0: return tc_ksin(y0, y1, 1);
    case 1: return tc_kcos(y0, y1);
    case 2: return -tc_ksin(y0, y1, 1);
    default: return -tc_kcos(y0, y1);
  }
}

TC_HD double tc_cos(double x) {
  int32_t ix = tc_hi(x) & 0x7fffffff;
  if (ix <= 0x3fe921fb) return tc_kcos(x, 0.0);
  if (ix >= 0x7ff00000) return x - x;
  double y0, y1;
  int n = tc_rem_pio2(x, &y0, &y1);
  switch (n & 3) {
    case 0: return tc_kcos(y0, y1);
    case 1: return -tc_ksin(y0, y1, 1);
    case 2: return -tc_kcos(y0, y1);
    default: return tc_ksin(y0, y1, 1);
  }
}

TC_HD double tc_tan(double x) {
  int32_t ix = tc_hi(x) & 0x7fffffff;
  if (ix <= 0x3fe921fb) return tc_ktan(x, 0.0, 1);
  if (ix >= 0x7ff00000) return x - x;
  double y0, y1;
  int n = tc_rem_pio2(x, &y0, &y1);
  return tc_ktan(y0, y1, 1 - ((n & 1) << 1));
}

TC_HD double tc_atan(double x) {
  const double atanhi0 = 4.63647609000806093515e-01, atanhi1 = 7.85398163397448278999e-01,
               atanhi2 = 9.82793723247329054082e-01, atanhi3 = 1.57079632679489655800e+00;
  const double atanlo0 = 2.26987774529616870924e-17, atanlo1 = 3.06161699786838301793e-17,
               atanlo2 = 1.39033110312309984516e-17, atanlo3 = 6.12323399573676603587e-17;
  const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01,
               aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
               aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
               aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
               aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02,
               aT10 = 1.62858201153657823623e-02;
  int32_t hx = tc_hi(x);
  int32_t ix = hx & 0x7fffffff;
  int id;
  double hi = 0.0, lo = 0.0;
  if (ix >= 0x44100000) {  // |x| >= 2^66
    if (tc_isnan(x)) return x + x;
    return hx > 0 ? atanhi3 + atanlo3 : -atanhi3 - atanlo3;
  }
  if (ix < 0x3fdc0000) {  // |x| < 0.4375
    if (ix < 0x3e200000) return x;  // |x| < 2^-29
    id = -1;
  } else {
    x = tc_fabs(x);
    if (ix < 0x3ff30000) {    // |x| < 1.1875
      if (ix < 0x3fe60000) {  // 7/16 <= |x| < 11/16
        id = 0; hi = atanhi0; lo = atanlo0;
        x = (2.0 * x - 1.0) / (2.0 + x);
      } else {                // 11/16 <= |x| < 19/16
        id = 1; hi = atanhi1; lo = atanlo1;
        x = (x - 1.0) / (x + 1.0);
      }
    } else {
      if (ix < 0x40038000) {  // |x| < 2.4375
        id = 2; hi = atanhi2; lo = atanlo2;
        x = (x - 1.5) / (1.0 + 1.5 * x);
      } else {                // 2.4375 <= |x| < 2^66
        id = 3; hi = atanhi3; lo = atanlo3;
        x = -1.0 / x;
      }
    }
  }
  double z = x * x;
  double w = z * z;
  double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
  double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
  if (id < 0) return x - x * (s1 + s2);
  z = hi - ((x * (s1 + s2) - lo) - x);
  return hx < 0 ? -z : z;
}

TC_HD double tc_atan2(double y, double x) {
  const double tiny = 1.0e-300;
  const double pi_o_4 = 7.8539816339744827900E-01, pi_o_2 = 1.5707963267948965580E+00,
               pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16;
  int32_t hx = tc_hi(x), hy = tc_hi(y);
  uint32_t lx = tc_lo(x), ly = tc_lo(y);
  int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
  if (tc_isnan(x) || tc_isnan(y)) return x + y;
  if ((((uint32_t)hx - 0x3ff00000u) | lx) == 0) return tc_atan(y);  // x == 1.0
  int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);                    // 2*sign(x) + sign(y)
  if ((iy | ly) == 0) {  // y == 0
    switch (m) {
      case 0:
      case 1: return y;
      case 2: return pi + tiny;
      default: return -pi - tiny;
    }
  }
  if ((ix | lx) == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;  // x == 0
  if (ix == 0x7ff00000) {  // x == inf
    if (iy == 0x7ff00000) {
      switch (m) {
        case 0: return pi_o_4 + tiny;
        case 1: return -pi_o_4 - tiny;
        case 2: return 3.0 * pi_o_4 + tiny;
        default: return -3.0 * pi_o_4 - tiny;
      }
    } else {
      switch (m) {
        case 0: return 0.0;
        case 1: return -0.0;
        case 2: return pi + tiny;
        default: return -pi - tiny;
      }
    }
  }
  if (iy == 0x7ff00000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  int k = (iy - ix) >> 20;
  double z;
  if (k > 60) {
    z = pi_o_2 + 0.5 * pi_lo;
  } else if (hx < 0 && k < -60) {
    z = 0.0;
  } else {
    z = tc_atan(tc_fabs(y / x));
  }
  switch (m) {
    case 0: return z;
    case 1: return -z;
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}

#endif  // TC_TRIG_H
