// The operator-level seams (the reference's DO_SARA_USE_HALIDE hooks:
// LinearFiltering.cpp:50-54, Resize.cpp:42-43,105-108, GaussianPyramid.cpp:38-42,
// Differential.cpp:72-79, LocalExtremum.cpp:23-37) and the device self-checks.
#include "sift_host.hpp"

using namespace sara_hip;
using namespace sara_hip::host;

extern "C" {

// ---- operator-level seams --------------------------------------------------

sara_hip_status sara_hip_apply_gaussian_filter(const float* src, float* dst,
                                               int w, int h, float sigma,
                                               float gauss_truncate, int device)
{
  if (!src || !dst || w < 1 || h < 1)
    return fail(SARA_HIP_SIZE_MISMATCH,
                "Source and destination image sizes are not equal!");
  Taps taps;
  if (!to_taps(gaussian_taps(sigma, gauss_truncate), taps))
    return fail(SARA_HIP_INVALID_PARAMS, "Gaussian needs more than 113 taps");
  const sara_hip_status st = select_device(device);
  if (st != SARA_HIP_OK)
    return st;
  DeviceScratch sc;
  float *ds = nullptr, *dd = nullptr;
  const size_t n = size_t(w) * h;
  HIP_TRY(sc.get(ds, n));
  HIP_TRY(sc.get(dd, n));
  HIP_TRY(hipMemcpy(ds, src, n * sizeof(float), hipMemcpyHostToDevice));
  launch_gaussian_blur(ds, n, dd, n, nullptr, 0, w, h, 1, taps, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(dst, dd, n * sizeof(float), hipMemcpyDeviceToHost));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_scale(const float* src, int sw, int sh, float* dst,
                               int dw, int dh, int device)
{
  if (!src || !dst || sw < 1 || sh < 1 || dw < 1 || dh < 1)
    return fail(SARA_HIP_INVALID_PARAMS, "bad image sizes");
  const sara_hip_status st = select_device(device);
  if (st != SARA_HIP_OK)
    return st;
  DeviceScratch sc;
  float *ds = nullptr, *dd = nullptr;
  HIP_TRY(sc.get(ds, size_t(sw) * sh));
  HIP_TRY(sc.get(dd, size_t(dw) * dh));
  HIP_TRY(hipMemcpy(ds, src, size_t(sw) * sh * sizeof(float),
                    hipMemcpyHostToDevice));
  launch_scale(ds, 0, sw, sh, dd, 0, dw, dh, 1, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(dst, dd, size_t(dw) * dh * sizeof(float),
                    hipMemcpyDeviceToHost));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_enlarge(const float* src, int sw, int sh, float* dst,
                                 int dw, int dh, int device)
{
  if (!src || !dst)
    return fail(SARA_HIP_INVALID_PARAMS, "null image");
  if (dw < sw || dh < sh)
    return fail(SARA_HIP_OUT_OF_RANGE,
                "The destination image must have smaller sizes than the source "
                "image!");
  if (std::min(dw, dh) <= 0 || sw < 1 || sh < 1)
    return fail(SARA_HIP_OUT_OF_RANGE,
                "The sizes of the destination image must be positive!");
  const sara_hip_status st = select_device(device);
  if (st != SARA_HIP_OK)
    return st;
  DeviceScratch sc;
  float *ds = nullptr, *dd = nullptr;
  HIP_TRY(sc.get(ds, size_t(sw) * sh));
  HIP_TRY(sc.get(dd, size_t(dw) * dh));
  HIP_TRY(hipMemcpy(ds, src, size_t(sw) * sh * sizeof(float),
                    hipMemcpyHostToDevice));
  launch_enlarge(ds, 0, sw, sh, dd, 0, dw, dh, 1, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(dst, dd, size_t(dw) * dh * sizeof(float),
                    hipMemcpyDeviceToHost));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_subtract(const float* a, const float* b, float* out,
                                  size_t count, int device)
{
  if (!a || !b || !out)
    return fail(SARA_HIP_INVALID_PARAMS, "null operand");
  const sara_hip_status st = select_device(device);
  if (st != SARA_HIP_OK)
    return st;
  DeviceScratch sc;
  float *da = nullptr, *db = nullptr, *dout = nullptr;
  HIP_TRY(sc.get(da, count));
  HIP_TRY(sc.get(db, count));
  HIP_TRY(sc.get(dout, count));
  HIP_TRY(hipMemcpy(da, a, count * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(db, b, count * sizeof(float), hipMemcpyHostToDevice));
  launch_subtract(da, db, dout, count, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, dout, count * sizeof(float), hipMemcpyDeviceToHost));
  return SARA_HIP_OK;
}

static sara_hip_status u8_to_gray(const uint8_t* src, float* gray, int w, int h,
                                  int channels, int device)
{
  if (!src || !gray || w < 1 || h < 1)
    return fail(SARA_HIP_SIZE_MISMATCH,
                "Color conversion error: image sizes are not equal!");
  const sara_hip_status st = select_device(device);
  if (st != SARA_HIP_OK)
    return st;
  DeviceScratch sc;
  unsigned char* ds = nullptr;
  float* dd = nullptr;
  const size_t n = size_t(w) * h;
  HIP_TRY(sc.get(ds, n * channels));
  HIP_TRY(sc.get(dd, n));
  HIP_TRY(hipMemcpy(ds, src, n * channels, hipMemcpyHostToDevice));
  launch_u8_to_gray32f(ds, 0, channels, dd, 0, n, 1, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(gray, dd, n * sizeof(float), hipMemcpyDeviceToHost));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_from_rgb8_to_gray32f(const uint8_t* rgb, float* gray,
                                              int w, int h, int device)
{
  return u8_to_gray(rgb, gray, w, h, 3, device);
}

sara_hip_status sara_hip_from_gray8_to_gray32f(const uint8_t* src, float* gray,
                                               int w, int h, int device)
{
  return u8_to_gray(src, gray, w, h, 1, device);
}

sara_hip_status sara_hip_root_sift(float* desc, int n, int dim, int on_device,
                                   int device)
{
  if (!desc || n < 0 || dim < 1)
    return fail(SARA_HIP_INVALID_PARAMS, "null pointer, negative count or empty rows");
  if (n == 0)
    return SARA_HIP_OK;
  const sara_hip_status st = select_device(device);
  if (st != SARA_HIP_OK)
    return st;
  DeviceScratch sc;
  float* d = desc;
  const size_t bytes = size_t(n) * dim * sizeof(float);
  if (!on_device)
  {
    HIP_TRY(sc.get(d, size_t(n) * dim));
    HIP_TRY(hipMemcpy(d, desc, bytes, hipMemcpyHostToDevice));
  }
  launch_root_sift(d, n, dim, nullptr);
  HIP_TRY(hipGetLastError());
  if (!on_device)
    HIP_TRY(hipMemcpy(desc, d, bytes, hipMemcpyDeviceToHost));
  else
    HIP_TRY(hipStreamSynchronize(nullptr));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_gradient_polar_coordinates(const float* src, int w,
                                                    int h, float* mag_ori,
                                                    int device)
{
  if (!src || !mag_ori || w < 2 || h < 2)
    return fail(SARA_HIP_INVALID_PARAMS, "image must be at least 2x2");
  const sara_hip_status st = select_device(device);
  if (st != SARA_HIP_OK)
    return st;
  DeviceScratch sc;
  float *ds = nullptr, *dd = nullptr;
  const size_t n = size_t(w) * h;
  HIP_TRY(sc.get(ds, n));
  HIP_TRY(sc.get(dd, 2 * n));
  HIP_TRY(hipMemcpy(ds, src, n * sizeof(float), hipMemcpyHostToDevice));
  launch_gradient_polar(ds, n, dd, 2 * n, w, h, 1, 1, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(mag_ori, dd, 2 * n * sizeof(float), hipMemcpyDeviceToHost));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_scale_space_dog_extremum_map(
    const float* a, const float* b, const float* c, int w, int h,
    float edge_ratio_thres, float extremum_thres, int img_padding_sz,
    int8_t* out, int device)
{
  if (!a || !b || !c || !out || w < 3 || h < 3)
    return fail(SARA_HIP_INVALID_PARAMS, "layers must be at least 3x3");
  if (img_padding_sz < 0)
    return fail(SARA_HIP_INVALID_PARAMS, "img_padding_sz must be >= 0");
  const sara_hip_status st = select_device(device);
  if (st != SARA_HIP_OK)
    return st;
  DeviceScratch sc;
  float *da = nullptr, *db = nullptr, *dc = nullptr;
  int8_t* dout = nullptr;
  const size_t n = size_t(w) * h;
  HIP_TRY(sc.get(da, n));
  HIP_TRY(sc.get(db, n));
  HIP_TRY(sc.get(dc, n));
  HIP_TRY(sc.get(dout, n));
  HIP_TRY(hipMemcpy(da, a, n * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(db, b, n * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dc, c, n * sizeof(float), hipMemcpyHostToDevice));
  launch_extremum_map(da, db, dc, w, h, edge_ratio_thres, extremum_thres,
                      img_padding_sz, dout, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, dout, n, hipMemcpyDeviceToHost));
  return SARA_HIP_OK;
}

void sara_hip_selfcheck_atan2f(const float* y, const float* x, float* out,
                               size_t count)
{
  for (size_t i = 0; i < count; ++i)
  {
    // both restatements the kernels use must agree; a mismatch is reported
    // as NaN so that the comparison with libm fails
    static const float tab[sara_hip::kAtanTableFloats] = SARA_ATAN_TABLE_INIT;
    static const std::vector<float> lut = [] {
      std::vector<float> l(sara_hip::kAtanLutFloats);
      for (int j = 0; j < sara_hip::kAtanLutRows; ++j)
        for (int q = 0; q < 8; ++q)
          l[size_t(8 * j + q)] = tab[8 * sara_hip::atan_lut_source_row(j) + q];
      return l;
    }();
    const float a = sara_hip::fdlibm_atan2f_fast(y[i], x[i]);
    const float b = sara_hip::fdlibm_atan2f_table(y[i], x[i], tab);
    const float c = sara_hip::fdlibm_atan2f_lut(y[i], x[i], lut.data());
    const bool same = std::memcmp(&a, &b, sizeof(float)) == 0 &&
                      std::memcmp(&a, &c, sizeof(float)) == 0;
    out[i] = same ? a : std::nanf("");
  }
}

void sara_hip_selfcheck_sincos(const float* theta, float* out_sin, float* out_cos,
                               size_t count)
{
  for (size_t i = 0; i < count; ++i)
  {
    double s, c;
    sara_hip::sincos_reduced_f64_host(double(theta[i]), s, c);
    out_sin[i] = float(s);
    out_cos[i] = float(c);
  }
}

sara_hip_status sara_hip_selfcheck_device_math(unsigned long long* mismatches,
                                               int device)
{
  if (!mismatches)
    return fail(SARA_HIP_INVALID_PARAMS, "null pointer");
  const sara_hip_status st = select_device(device);
  if (st != SARA_HIP_OK)
    return st;
  DeviceScratch sc;
  unsigned long long* d = nullptr;
  HIP_TRY(sc.get(d, 2));
  HIP_TRY(hipMemset(d, 0, 2 * sizeof(unsigned long long)));
  launch_device_math_selfcheck(d, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(mismatches, d, 2 * sizeof(unsigned long long),
                    hipMemcpyDeviceToHost));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_selfcheck_orientation_bins(unsigned long long* mismatches,
                                                   int device)
{
  if (!mismatches)
    return fail(SARA_HIP_INVALID_PARAMS, "null pointer");
  const sara_hip_status st = select_device(device);
  if (st != SARA_HIP_OK)
    return st;
  float thr[40];
  orientation_bin_thresholds(thr);
  DeviceScratch sc;
  float* d_thr = nullptr;
  unsigned long long* d_bad = nullptr;
  HIP_TRY(sc.get(d_thr, 40));
  HIP_TRY(sc.get(d_bad, 1));
  HIP_TRY(hipMemcpy(d_thr, thr, sizeof(thr), hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(d_bad, 0, sizeof(unsigned long long)));
  launch_orientation_bin_selfcheck(d_thr, d_bad, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(mismatches, d_bad, sizeof(unsigned long long),
                    hipMemcpyDeviceToHost));
  return SARA_HIP_OK;
}

sara_hip_status sara_hip_selfcheck_definiteness(const float* hessians,
                                                const int* types, size_t count,
                                                unsigned char* out, int device)
{
  if (!hessians || !types || !out)
    return fail(SARA_HIP_INVALID_PARAMS, "null pointer");
  if (count == 0)
    return SARA_HIP_OK;
  if (count > (size_t(1) << 28))
    return fail(SARA_HIP_CAPACITY_EXCEEDED, "too many matrices");
  const sara_hip_status st = select_device(device);
  if (st != SARA_HIP_OK)
    return st;
  DeviceScratch sc;
  float* dH = nullptr;
  int* dT = nullptr;
  unsigned char* dO = nullptr;
  HIP_TRY(sc.get(dH, 9 * count));
  HIP_TRY(sc.get(dT, count));
  HIP_TRY(sc.get(dO, count));
  HIP_TRY(hipMemcpy(dH, hessians, 9 * count * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dT, types, count * sizeof(int), hipMemcpyHostToDevice));
  launch_definiteness_selfcheck(dH, dT, int(count), dO, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, dO, count, hipMemcpyDeviceToHost));
  return SARA_HIP_OK;
}

}  // extern "C"
