// Float math shared by the gfx950 kernels and the host-side self-checks.
//
// The CPU reference evaluates atan2f through glibc (2.35 on the reference's
// ubuntu:22.04 CI image and on this image).  glibc 2.35's atan2f/atanf are the
// fdlibm single-precision algorithms (sysdeps/ieee754/flt-32/e_atan2f.c,
// s_atanf.c): a fixed sequence of IEEE float operations.  Restating that
// sequence here - with FP contraction off - makes the GPU polar gradients
// bit-identical to the CPU path instead of merely close, which in turn keeps
// the hard orientation binning (Orientation.hpp:121-125) identical.
// tests/test_host_math.py checks the restatement against libm on the host.
#pragma once

#include <cstdint>
#include <cstring>

#if defined(__HIPCC__) || defined(__HIP__)
#  include <hip/hip_runtime.h>
#  define SARA_HD __host__ __device__ inline __attribute__((always_inline))
#else
#  define SARA_HD inline __attribute__((always_inline))
#endif

namespace sara_hip {

  SARA_HD int32_t float_as_int(float f)
  {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_int(f);
#else
    int32_t i;
    std::memcpy(&i, &f, 4);
    return i;
#endif
  }

  SARA_HD float int_as_float(int32_t i)
  {
#if defined(__HIP_DEVICE_COMPILE__)
    return __int_as_float(i);
#else
    float f;
    std::memcpy(&f, &i, 4);
    return f;
#endif
  }

  //! fdlibm atanf for finite |x| < 2^25 (the caller handles the rest).
  SARA_HD float fdlibm_atanf(float x)
  {
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f,
                             9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f,
                             3.4473217170e-08f, 7.5497894159e-08f};
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f,
                aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f,
                aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f,
                aT10 = 1.6285819933e-02f;

    const int32_t hx = float_as_int(x);
    const int32_t ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000)  // |x| >= 2^25 (or NaN)
    {
      if (ix > 0x7f800000)
        return x + x;
      return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000)  // |x| < 0.4375
    {
      if (ix < 0x31000000)  // |x| < 2^-29
        return x;
      id = -1;
    }
    else
    {
      x = int_as_float(ix);  // fabsf
      if (ix < 0x3f980000)   // |x| < 1.1875
      {
        if (ix < 0x3f300000)  // 7/16 <= |x| < 11/16
        {
          id = 0;
          x = (2.0f * x - 1.0f) / (2.0f + x);
        }
        else  // 11/16 <= |x| < 19/16
        {
          id = 1;
          x = (x - 1.0f) / (x + 1.0f);
        }
      }
      else
      {
        if (ix < 0x401c0000)  // |x| < 2.4375
        {
          id = 2;
          x = (x - 1.5f) / (1.0f + 1.5f * x);
        }
        else  // 2.4375 <= |x| < 2^25
        {
          id = 3;
          x = -1.0f / x;
        }
      }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 =
        z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0)
      return x - x * (s1 + s2);
    const float hi = id == 0 ? atanhi[0]
                             : (id == 1 ? atanhi[1]
                                        : (id == 2 ? atanhi[2] : atanhi[3]));
    const float lo = id == 0 ? atanlo[0]
                             : (id == 1 ? atanlo[1]
                                        : (id == 2 ? atanlo[2] : atanlo[3]));
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -r : r;
  }

  //! fdlibm atan2f(y, x) for finite arguments.
  SARA_HD float fdlibm_atan2f(float y, float x)
  {
    const float tiny = 1.0e-30f;
    const float pi_o_2 = 1.5707963705e+00f;
    const float pi = 3.1415927410e+00f;
    const float pi_lo = -8.7422776573e-08f;

    const int32_t hx = float_as_int(x);
    const int32_t ix = hx & 0x7fffffff;
    const int32_t hy = float_as_int(y);
    const int32_t iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000)
      return x + y;
    if (hx == 0x3f800000)
      return fdlibm_atanf(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);

    if (iy == 0)
    {
      switch (m)
      {
      case 0:
      case 1:
        return y;
      case 2:
        return pi + tiny;
      default:
        return -pi - tiny;
      }
    }
    if (ix == 0)
      return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000)
    {
      const float pi_o_4 = 7.8539818525e-01f;
      if (iy == 0x7f800000)
      {
        switch (m)
        {
        case 0:
          return pi_o_4 + tiny;
        case 1:
          return -pi_o_4 - tiny;
        case 2:
          return 3.0f * pi_o_4 + tiny;
        default:
          return -3.0f * pi_o_4 - tiny;
        }
      }
      switch (m)
      {
      case 0:
        return 0.0f;
      case 1:
        return -0.0f;
      case 2:
        return pi + tiny;
      default:
        return -pi - tiny;
      }
    }
    if (iy == 0x7f800000)
      return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;

    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60)
      z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60)
      z = 0.0f;
    else
    {
      const float q = y / x;
      z = fdlibm_atanf(int_as_float(float_as_int(q) & 0x7fffffff));
    }
    switch (m)
    {
    case 0:
      return z;
    case 1:
      return int_as_float(float_as_int(z) ^ (int32_t) 0x80000000);
    case 2:
      return pi - (z - pi_lo);
    default:
      return (z - pi_lo) - pi;
    }
  }


  //! Branch-free restatement of fdlibm_atanf for a NON-NEGATIVE finite
  //! argument: the four argument reductions share ONE division (numerator and
  //! denominator are selected per range), the |x| < 2^-29 early-out is
  //! dropped (x - x*(s1+s2) rounds to x there) and the |x| >= 2^25 case is a
  //! final select.  Bit-identical to fdlibm_atanf (tests/test_host_math.py).
  SARA_HD float atanf_nonneg_select(float x)
  {
    const int32_t ix = float_as_int(x);
    const bool r0 = ix < 0x3ee00000;   // |x| < 0.4375      -> id = -1
    const bool r1 = ix < 0x3f300000;   // < 11/16           -> id = 0
    const bool r2 = ix < 0x3f980000;   // < 19/16           -> id = 1
    const bool r3 = ix < 0x401c0000;   // < 2.4375          -> id = 2, else 3
    const float n0 = 2.0f * x - 1.0f, d0 = 2.0f + x;
    const float n1 = x - 1.0f, d1 = x + 1.0f;
    const float n2 = x - 1.5f, d2 = 1.0f + 1.5f * x;
    const float num = r0 ? x : (r1 ? n0 : (r2 ? n1 : (r3 ? n2 : -1.0f)));
    const float den = r0 ? 1.0f : (r1 ? d0 : (r2 ? d1 : (r3 ? d2 : x)));
    const float hi = r1 ? 4.6364760399e-01f
                        : (r2 ? 7.8539812565e-01f
                              : (r3 ? 9.8279368877e-01f : 1.5707962513e+00f));
    const float lo = r1 ? 5.0121582440e-09f
                        : (r2 ? 3.7748947079e-08f
                              : (r3 ? 3.4473217170e-08f : 7.5497894159e-08f));
    const float xr = num / den;
    const float z = xr * xr;
    const float w = z * z;
    const float s1 =
        z * (3.3333334327e-01f +
             w * (1.4285714924e-01f +
                  w * (9.0908870101e-02f +
                       w * (6.6610731184e-02f +
                            w * (4.9768779427e-02f + w * 1.6285819933e-02f)))));
    const float s2 =
        w * (-2.0000000298e-01f +
             w * (-1.1111110449e-01f +
                  w * (-7.6918758452e-02f +
                       w * (-5.8335702866e-02f + w * -3.6531571299e-02f))));
    const float small = xr - xr * (s1 + s2);
    const float big = hi - ((xr * (s1 + s2) - lo) - xr);
    float r = r0 ? small : big;
    if (ix >= 0x4c000000)  // |x| >= 2^25
      r = 1.5707962513e+00f + 7.5497894159e-08f;
    return r;
  }

  //! The same reduction driven by a table instead of selects: row `id` (number
  //! of range thresholds the argument has reached, 0..4) holds (a, b, c, d,
  //! atanhi, atanlo) with numerator a*x + b and denominator c*x + d.  Every
  //! row reproduces the select version bit for bit: multiplications by 0, 1
  //! and 2 are exact, x + 0 == x for x >= 0, and with atanhi = atanlo = 0 the
  //! combination hi - ((xr*s - lo) - xr) equals xr - xr*s (negation commutes
  //! with rounding).  The gradient kernel keeps the table in LDS: two loads
  //! replace 15 selects and 5 compares per pixel (v_cndmask / v_cmp issue at
  //! about half the rate of plain arithmetic on gfx950).
  constexpr int kAtanTableFloats = 40;
#define SARA_ATAN_TABLE_INIT                                                    \
  {                                                                            \
    1.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f,                                    \
    2.f, -1.f, 1.f, 2.f, 4.6364760399e-01f, 5.0121582440e-09f, 0.f, 0.f,       \
    1.f, -1.f, 1.f, 1.f, 7.8539812565e-01f, 3.7748947079e-08f, 0.f, 0.f,       \
    1.f, -1.5f, 1.5f, 1.f, 9.8279368877e-01f, 3.4473217170e-08f, 0.f, 0.f,     \
    0.f, -1.f, 1.f, 0.f, 1.5707962513e+00f, 7.5497894159e-08f, 0.f, 0.f        \
  }

  SARA_HD float atanf_nonneg_table(float x, const float* tab)
  {
    const int32_t ix = float_as_int(x);
    const int id = int(ix >= 0x3ee00000) + int(ix >= 0x3f300000) +
                   int(ix >= 0x3f980000) + int(ix >= 0x401c0000);
    const float* t = tab + 8 * id;
    const float a = t[0], b = t[1], c = t[2], d = t[3], hi = t[4], lo = t[5];
    const float num = a * x + b;
    const float den = c * x + d;
    const float xr = num / den;
    const float z = xr * xr;
    const float w = z * z;
    const float s1 =
        z * (3.3333334327e-01f +
             w * (1.4285714924e-01f +
                  w * (9.0908870101e-02f +
                       w * (6.6610731184e-02f +
                            w * (4.9768779427e-02f + w * 1.6285819933e-02f)))));
    const float s2 =
        w * (-2.0000000298e-01f +
             w * (-1.1111110449e-01f +
                  w * (-7.6918758452e-02f +
                       w * (-5.8335702866e-02f + w * -3.6531571299e-02f))));
    float r = hi - ((xr * (s1 + s2) - lo) - xr);
    r = ix >= 0x4c000000 ? 1.5707962513e+00f + 7.5497894159e-08f : r;  // >= 2^25
    return r;
  }

  //! fdlibm_atan2f with the common path branch-free (finite inputs); the
  //! non-finite inputs take the reference implementation above.
  SARA_HD float fdlibm_atan2f_fast(float y, float x)
  {
    const int32_t hx = float_as_int(x);
    const int32_t ix = hx & 0x7fffffff;
    const int32_t hy = float_as_int(y);
    const int32_t iy = hy & 0x7fffffff;
#if !defined(__HIP_DEVICE_COMPILE__)
    // Non-finite inputs: the reference implementation (host self-check only;
    // pyramid values on the device are finite).
    if (ix >= 0x7f800000 || iy >= 0x7f800000)
      return fdlibm_atan2f(y, x);
#endif
    const float pi = 3.1415927410e+00f;
    const float pi_lo = -8.7422776573e-08f;
    const float pi_o_2 = 1.5707963705e+00f;
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    const int k = (iy - ix) >> 23;
    // general path: z = atan(|y/x|)
    const float q = y / x;
    float z = atanf_nonneg_select(int_as_float(float_as_int(q) & 0x7fffffff));
    // Every candidate below is computed unconditionally and then picked with
    // plain two-way selects of ready values: written as nested conditionals
    // with the arithmetic inside the arms, the compiler turns each arm into a
    // divergent branch (7 exec-mask branches per pixel in the gradient kernel).
    z = k > 60 ? pi_o_2 + 0.5f * pi_lo : z;
    const bool tiny = (hx < 0) & (k < -60) & !(k > 60);
    z = tiny ? 0.0f : z;
    // m = 0: z, 1: -z, 2: pi - (z - pi_lo), 3: (z - pi_lo) - pi.  Rounding
    // commutes with negation, so case 2 is -(case 3) (z - pi_lo never equals
    // pi, no signed-zero issue): one select and a sign flip for m in {1, 2}.
    const float q3 = (z - pi_lo) - pi;
    const float base = (m & 2) ? q3 : z;
    const int32_t flip = (int32_t) ((uint32_t) ((m ^ (m >> 1)) & 1) << 31);
    float r = int_as_float(float_as_int(base) ^ flip);
    // fdlibm's x == 1 shortcut (atanf(y)) and its y == 0 cases for x != 0 are
    // reproduced by the general path bit for bit (y/1 == y; atan(0) == 0 and
    // pi - (0 - pi_lo) rounds to pi), so only x == +-0 needs a select.
    const int32_t sy = hy & (int32_t) 0x80000000;
    const float pi_sy = int_as_float(float_as_int(pi) | sy);
    const float pi_o_2_sy = int_as_float(float_as_int(pi_o_2) | sy);
    const float x0y0 = (m & 2) ? pi_sy : y;
    const float x0 = iy == 0 ? x0y0 : pi_o_2_sy;
    r = ix == 0 ? x0 : r;
    return r;
  }

  //! fdlibm_atan2f_fast with the table-driven reduction.
  SARA_HD float fdlibm_atan2f_table(float y, float x, const float* tab)
  {
    const int32_t hx = float_as_int(x);
    const int32_t ix = hx & 0x7fffffff;
    const int32_t hy = float_as_int(y);
    const int32_t iy = hy & 0x7fffffff;
#if !defined(__HIP_DEVICE_COMPILE__)
    // Non-finite inputs: the reference implementation (host self-check only;
    // pyramid values on the device are finite).
    if (ix >= 0x7f800000 || iy >= 0x7f800000)
      return fdlibm_atan2f(y, x);
#endif
    const float pi = 3.1415927410e+00f;
    const float pi_lo = -8.7422776573e-08f;
    const float pi_o_2 = 1.5707963705e+00f;
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    const int k = (iy - ix) >> 23;
    // general path: z = atan(|y/x|)
    const float q = y / x;
    float z = atanf_nonneg_table(int_as_float(float_as_int(q) & 0x7fffffff), tab);
    // Every candidate below is computed unconditionally and then picked with
    // plain two-way selects of ready values: written as nested conditionals
    // with the arithmetic inside the arms, the compiler turns each arm into a
    // divergent branch (7 exec-mask branches per pixel in the gradient kernel).
    z = k > 60 ? pi_o_2 + 0.5f * pi_lo : z;
    const bool tiny = (hx < 0) & (k < -60) & !(k > 60);
    z = tiny ? 0.0f : z;
    // m = 0: z, 1: -z, 2: pi - (z - pi_lo), 3: (z - pi_lo) - pi.  Rounding
    // commutes with negation, so case 2 is -(case 3) (z - pi_lo never equals
    // pi, no signed-zero issue): one select and a sign flip for m in {1, 2}.
    const float q3 = (z - pi_lo) - pi;
    const float base = (m & 2) ? q3 : z;
    const int32_t flip = (int32_t) ((uint32_t) ((m ^ (m >> 1)) & 1) << 31);
    float r = int_as_float(float_as_int(base) ^ flip);
    // fdlibm's x == 1 shortcut (atanf(y)) and its y == 0 cases for x != 0 are
    // reproduced by the general path bit for bit (y/1 == y; atan(0) == 0 and
    // pi - (0 - pi_lo) rounds to pi), so only x == +-0 needs a select.
    const int32_t sy = hy & (int32_t) 0x80000000;
    const float pi_sy = int_as_float(float_as_int(pi) | sy);
    const float pi_o_2_sy = int_as_float(float_as_int(pi_o_2) | sy);
    const float x0y0 = (m & 2) ? pi_sy : y;
    const float x0 = iy == 0 ? x0y0 : pi_o_2_sy;
    r = ix == 0 ? x0 : r;
    return r;
  }

  // ------------------------------------------------------------------------ //
  // Device-only short forms.  The compiler's IEEE sqrt and division cost 16 and
  // 11 VALU instructions (range scaling, fix-ups); on the operand ranges below
  // a hardware seed plus one or two FMA corrections already rounds correctly.
  // Both were compared with the IEEE forms on every float of their domain on
  // the device (tools/ubench/fast_math_check.hip, and sara_hip_selfcheck_
  // device_math at test time).
  // ------------------------------------------------------------------------ //
#if defined(__HIPCC__)
  //! sqrtf(x), correctly rounded for x == 0 and 2^-102 <= x <= FLT_MAX (below,
  //! the residual g*g - x underflows; +inf gives 0): callers test
  //! sqrt_short_ok() and fall back to sqrtf().
  __device__ inline float sqrt_rn_short(float x)
  {
    const float y = __builtin_amdgcn_rsqf(x);  // +inf for x == 0
    const float g = x * y;                     // NaN for x == 0
    const float h = 0.5f * y;
    const float d = __builtin_fmaf(-g, g, x);
    return fmaxf(__builtin_fmaf(d, h, g), 0.f);  // maxNum: NaN -> 0
  }

  //! frexp exponent of x (0 for x == 0): the short form is exact from -101 up,
  //! -96 leaves a margin.
  __device__ inline int sqrt_short_exponent(float x)
  {
    return __builtin_amdgcn_frexp_expf(x);
  }
  constexpr int kSqrtShortMinExponent = -96;

  //! n / d, correctly rounded for the quotients of the atanf reduction
  //! ((a x + b) / (c x + d) of SARA_ATAN_TABLE_INIT for 0 <= x < 2^126): no
  //! operand scaling, one reciprocal refinement, one quotient refinement.
  __device__ inline float div_rn_short(float n, float d)
  {
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    const float q = n * r;
    const float rem = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(rem, r, q);
  }
#endif

  //! Look-up form of the table: one row per 2^18-aligned block of float bit
  //! patterns between the first and the last range threshold (all four
  //! thresholds are multiples of 2^18), so that the row address is a shift, a
  //! clamp and a scaled add of the argument's bits instead of four compares.
  constexpr int kAtanLutLo = 0x3ee00000 >> 18;  // first threshold block: 0xFB8
  constexpr int kAtanLutHi = 0x401c0000 >> 18;  // last threshold block: 0x1007
  constexpr int kAtanLutRows = kAtanLutHi - kAtanLutLo + 2;  // + "below" row
  constexpr int kAtanLutFloats = 8 * kAtanLutRows;

  //! Row of the 5-row table that serves LUT row j.
  SARA_HD int atan_lut_source_row(int j)
  {
    const int32_t first = (kAtanLutLo - 1 + j) << 18;  // smallest pattern of the block
    return int(first >= 0x3ee00000) + int(first >= 0x3f300000) +
           int(first >= 0x3f980000) + int(first >= 0x401c0000);
  }

  SARA_HD float atanf_nonneg_lut(float x, const float* lut)
  {
    const int32_t ix = float_as_int(x);
    // block index relative to the "below" row, clamped to the table
    int v = (ix >> 18) - (kAtanLutLo - 1);
    v = v < 0 ? 0 : v;
    v = v > kAtanLutRows - 1 ? kAtanLutRows - 1 : v;
    const float* t = lut + 8 * v;
    const float a = t[0], b = t[1], c = t[2], d = t[3], hi = t[4], lo = t[5];
    const float num = a * x + b;
    const float den = c * x + d;
#if defined(__HIP_DEVICE_COMPILE__)
    const float xr = div_rn_short(num, den);
#else
    const float xr = num / den;
#endif
    const float z = xr * xr;
    const float w = z * z;
    const float s1 =
        z * (3.3333334327e-01f +
             w * (1.4285714924e-01f +
                  w * (9.0908870101e-02f +
                       w * (6.6610731184e-02f +
                            w * (4.9768779427e-02f + w * 1.6285819933e-02f)))));
    const float s2 =
        w * (-2.0000000298e-01f +
             w * (-1.1111110449e-01f +
                  w * (-7.6918758452e-02f +
                       w * (-5.8335702866e-02f + w * -3.6531571299e-02f))));
    float r = hi - ((xr * (s1 + s2) - lo) - xr);
    r = ix >= 0x4c000000 ? 1.5707962513e+00f + 7.5497894159e-08f : r;  // >= 2^25
    return r;
  }

  //! atan2f(y, x) for finite y and x != +-0 with the look-up reduction.
  //! Compared with fdlibm_atan2f_fast two more of fdlibm's shortcuts are
  //! dropped because the general path already returns their values:
  //!  * k = exponent(y) - exponent(x) > 60 (result pi/2): then |y/x| >= 2^60
  //!    (or overflows to +inf), which the reduction's own |x| >= 2^25 case
  //!    maps to atanhi[3] + atanlo[3] - the same float as pi/2 + pi_lo/2;
  //!  * x < 0 and k < -60 (z = 0): then z = atanf(|y/x|) <= 2^-59 while
  //!    half an ulp of pi_lo is 2^-48, so z - pi_lo == -pi_lo exactly and the
  //!    quadrant formulas give the same result as with z = 0.
  //! (tests/test_host_math.py holds the cases; x == +-0 is not covered here:
  //! see atan2f_zero_x.)
  SARA_HD float atan2f_lut_nonzero_x(float y, float x, const float* lut)
  {
    const int32_t hx = float_as_int(x);
    const int32_t hy = float_as_int(y);
    const float pi = 3.1415927410e+00f;
    const float pi_lo = -8.7422776573e-08f;
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    const float q = y / x;
    const float z = atanf_nonneg_lut(int_as_float(float_as_int(q) & 0x7fffffff), lut);
    // m = 0: z, 1: -z, 2: pi - (z - pi_lo), 3: (z - pi_lo) - pi
    const float q3 = (z - pi_lo) - pi;
    const float base = (m & 2) ? q3 : z;
    const int32_t flip = (int32_t) ((uint32_t) ((m ^ (m >> 1)) & 1) << 31);
    return int_as_float(float_as_int(base) ^ flip);
  }

  //! atan2f(y, +-0): +-pi/2 by the sign of y; for y == +-0, y itself (x = +0)
  //! or +-pi (x = -0).
  SARA_HD float atan2f_zero_x(float y, float x)
  {
    const int32_t hx = float_as_int(x);
    const int32_t hy = float_as_int(y);
    const float pi = 3.1415927410e+00f;
    const float pi_o_2 = 1.5707963705e+00f;
    const int32_t sy = hy & (int32_t) 0x80000000;
    const float pi_sy = int_as_float(float_as_int(pi) | sy);
    const float pi_o_2_sy = int_as_float(float_as_int(pi_o_2) | sy);
    const float x0y0 = hx < 0 ? pi_sy : y;
    return (hy & 0x7fffffff) == 0 ? x0y0 : pi_o_2_sy;
  }

  //! fdlibm_atan2f_fast with the look-up reduction (finite inputs).
  SARA_HD float fdlibm_atan2f_lut(float y, float x, const float* lut)
  {
    const int32_t ix = float_as_int(x) & 0x7fffffff;
#if !defined(__HIP_DEVICE_COMPILE__)
    const int32_t iy = float_as_int(y) & 0x7fffffff;
    if (ix >= 0x7f800000 || iy >= 0x7f800000)
      return fdlibm_atan2f(y, x);
#endif
    const float r = atan2f_lut_nonzero_x(y, x, lut);
    return ix == 0 ? atan2f_zero_x(y, x) : r;
  }

  // ------------------------------------------------------------------------ //
  // cos / sin in double for the descriptor's rotation (SIFT.hpp:84-89 evaluates
  // std::cos / std::sin on double(theta) and rounds to float).  theta is a
  // refined histogram peak in (-pi, pi], so a three-term Cody-Waite reduction
  // by pi/2 and fdlibm's kernel polynomials (__kernel_sin / __kernel_cos
  // coefficients, error < 1 ulp of double) are enough: about 30 FMA-class
  // double operations instead of the general-range library routines (two
  // calls of a few hundred instructions with a Payne-Hanek branch).  Only
  // float(result) is consumed: it differs from the correctly rounded value
  // only when the true cosine lies within 1e-16 of a float rounding boundary
  // (checked against libm over every float in [-4, 4] on the host,
  // tests/test_host_math.py).  |x| > 4 is left to the caller's fallback.
  // ------------------------------------------------------------------------ //
  SARA_HD double fma_f64(double a, double b, double c)
  {
#if defined(__HIP_DEVICE_COMPILE__)
    return __fma_rn(a, b, c);
#else
    return __builtin_fma(a, b, c);
#endif
  }

  //! Constants of sincos_reduced_f64.  The device reads them through a
  //! __constant__ copy (scalar loads at the point of use): as literals the
  //! compiler materialises them in VGPR pairs, hoists those out of the
  //! per-keypoint loop and - in the descriptor kernel, which is compiled for
  //! 80 VGPRs - spills them, so that every keypoint started with a chain of
  //! scratch reloads (23 % of the kernel's wave time, measured).
  //! [0] 2/pi, [1..3] pi/2 = P1 + P2 + P3 (33 + 33 + 53 bits),
  //! [4..9] S6..S1 of __kernel_sin, [10..15] C6..C1 of __kernel_cos.
#define SARA_SINCOS_COEF_INIT                                                  \
  {6.36619772367581382433e-01,  1.57079632673412561417e+00,                    \
   6.07710050630396597660e-11,  2.02226624871116645580e-21,                    \
   1.58969099521155010221e-10,  -2.50507602534068634195e-08,                   \
   2.75573137070700676789e-06,  -1.98412698298579493134e-04,                   \
   8.33333333332248946124e-03,  -1.66666666666666324348e-01,                   \
   -1.13596475577881948265e-11, 2.08757232129817482790e-09,                    \
   -2.75573143513906633035e-07, 2.48015872894767294178e-05,                    \
   -1.38888888888741095749e-03, 4.16666666666666019037e-02}
  constexpr int kSincosCoefCount = 16;

  SARA_HD void sincos_reduced_f64(double x, double& s, double& c,
                                  const double* __restrict__ K)
  {
    // k = nearest integer to x * 2/pi, |k| <= 3
    const double kf = __builtin_rint(x * K[0]);
    const int k = int(kf);
    // r = x - k * pi/2 with pi/2 = P1 + P2 + P3 (33 + 33 + 53 bits)
    double r = fma_f64(-kf, K[1], x);
    r = fma_f64(-kf, K[2], r);
    r = fma_f64(-kf, K[3], r);
    r = k == 0 ? x : r;  // keeps sin(-0) = -0
    const double z = r * r;
    // __kernel_sin: r + r^3 (S1 + z (S2 + ... z S6))
    double ps = fma_f64(z, K[4], K[5]);
    ps = fma_f64(z, ps, K[6]);
    ps = fma_f64(z, ps, K[7]);
    ps = fma_f64(z, ps, K[8]);
    ps = fma_f64(z, ps, K[9]);
    double sr = fma_f64(z * r, ps, r);
    sr = z < 5.5e-17 ? r : sr;  // |r| < 2^-27: sin r = r (and keeps -0)
    // __kernel_cos: 1 - z/2 + z^2 (C1 + z (C2 + ... z C6))
    double pc = fma_f64(z, K[10], K[11]);
    pc = fma_f64(z, pc, K[12]);
    pc = fma_f64(z, pc, K[13]);
    pc = fma_f64(z, pc, K[14]);
    pc = fma_f64(z, pc, K[15]);
    const double cr = fma_f64(z * z, pc, fma_f64(z, -0.5, 1.0));
    switch (k & 3)
    {
    case 0:
      s = sr;
      c = cr;
      break;
    case 1:
      s = cr;
      c = -sr;
      break;
    case 2:
      s = -sr;
      c = -cr;
      break;
    default:
      s = -cr;
      c = sr;
      break;
    }
  }

  //! Host form (and any caller without a constant table at hand).
  inline void sincos_reduced_f64_host(double x, double& s, double& c)
  {
    static const double K[kSincosCoefCount] = SARA_SINCOS_COEF_INIT;
    sincos_reduced_f64(x, s, c, K);
  }

}  // namespace sara_hip
