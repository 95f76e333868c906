// Micro-benchmark: where does the marching Gaussian blur spend its time?
// A copy of gaussian_blur_march_kernel (sara_amd/csrc/pyramid_kernels.hip) with
// switches that remove one resource at a time:
//   bit0 NOLOAD   global loads only in the prologue
//   bit1 NOSTORE  no global stores
//   bit2 NOLDS    row window taken from registers, no LDS traffic
//   bit3 SYMCOL   column pass shares the product of symmetric taps
// Usage: blur_limits [frames=64] [w=1920] [h=1080]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct Taps
{
  int size;
  float k[33];
};

template <int N> struct VecOf;
template <> struct VecOf<4> { using type = float4; };
template <> struct VecOf<2> { using type = float2; };

template <int R, int PF, int MODE, int CPL>
__global__ __launch_bounds__(64) void march(const float* __restrict__ src,
                                            size_t src_stride,
                                            float* __restrict__ dst,
                                            size_t dst_stride, int w, int h,
                                            int seg_rows, int nstrips, Taps taps,
                                            int never)
{
  constexpr bool NOLOAD = MODE & 1, NOSTORE = MODE & 2, NOLDS = MODE & 4,
                 SYMCOL = MODE & 8;
  using vec = typename VecOf<CPL>::type;
  constexpr int K = 2 * R + 1;
  constexpr int W = 64 * CPL;
  constexpr int RP = ((R + CPL - 1) / CPL) * CPL;
  constexpr int D = RP - R;
  constexpr int ROWF = RP + W + RP;
  constexpr int NQ = (D + CPL + 2 * R + CPL - 1) / CPL;
  __shared__ __attribute__((aligned(16))) float s_row[2 * ROWF];

  const int lane = threadIdx.x;
  const int strip = blockIdx.x % nstrips;
  const int seg = blockIdx.x / nstrips;
  const size_t b = blockIdx.y;
  src += b * src_stride;
  dst += b * dst_stride;
  const int x0 = strip * W;
  const int y0 = seg * seg_rows;
  const int y1 = min(h, y0 + seg_rows);
  const int col = x0 + CPL * lane;
  const bool col_ok = col < w;
  int hcol = lane < R ? x0 - R + lane : x0 + W + (lane - R);
  hcol = hcol < 0 ? 0 : (hcol > w - 1 ? w - 1 : hcol);
  const int hslot = lane < R ? RP - R + lane : RP + W + (lane - R);
  const int mcol = col_ok ? col : w - CPL;

  auto load_row = [&](int yy, vec& m, float& hv) {
    const int gy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
    const float* rowp = src + size_t(gy) * w;
    m = *reinterpret_cast<const vec*>(rowp + mcol);
    if (!col_ok)
    {
      const float last = (&m.x)[CPL - 1];
#pragma unroll
      for (int c = 0; c < CPL; ++c)
        (&m.x)[c] = last;
    }
    hv = 0.f;
    if (lane < 2 * R)
      hv = rowp[hcol];
  };

  float A[K][CPL];
  vec pm[PF];
  float phv[PF];
  const int T = (y1 - y0) + 2 * R;
#pragma unroll
  for (int q = 0; q < PF; ++q)
    load_row(y0 - R + q, pm[q], phv[q]);

  for (int n0 = 0; n0 < T; n0 += K)
  {
#pragma unroll
    for (int i = 0; i < K; ++i)
    {
      const int n = n0 + i;
      const int yy = y0 - R + n;
      float* rowbuf = s_row + (n & 1) * ROWF;
      float v[NQ * CPL];
      if (!NOLDS)
      {
        *reinterpret_cast<vec*>(rowbuf + RP + CPL * lane) = pm[i % PF];
        if (lane < 2 * R)
          rowbuf[hslot] = phv[i % PF];
      }
      else
      {
#pragma unroll
        for (int q = 0; q < NQ * CPL; ++q)
          v[q] = (&pm[i % PF].x)[q % CPL] + float(q) * phv[i % PF];
      }
      if (!NOLOAD)
        load_row(yy + PF, pm[i % PF], phv[i % PF]);
      else
      {
        pm[i % PF].x += 1.f;
      }
      float t[CPL];
      if (!NOLDS)
      {
        const vec* p = reinterpret_cast<const vec*>(rowbuf + CPL * lane);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const vec x = p[q];
#pragma unroll
          for (int c = 0; c < CPL; ++c)
            v[CPL * q + c] = (&x.x)[c];
        }
      }
#pragma unroll
      for (int c = 0; c < CPL; ++c)
      {
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j)
          sum += v[D + c + j] * taps.k[j];
        t[c] = sum;
      }
      if (!SYMCOL)
      {
#pragma unroll
        for (int j = 0; j < K; ++j)
        {
          const int sl = (i + K - 1 - j) % K;
#pragma unroll
          for (int c = 0; c < CPL; ++c)
          {
            if (j == 0)
              A[sl][c] = 0.f + t[c] * taps.k[0];
            else
              A[sl][c] += t[c] * taps.k[j];
          }
        }
      }
      else
      {
#pragma unroll
        for (int j = 0; j <= R; ++j)
        {
          const int sl = (i + K - 1 - j) % K;
          const int sl2 = (i + K - 1 - (K - 1 - j)) % K;
#pragma unroll
          for (int c = 0; c < CPL; ++c)
          {
            if (MODE & 32)
            {
              float p;
              const float kj = taps.k[j];
              if (j == 0)
                asm volatile("v_mul_f32 %0, %2, %3\n\tv_add_f32 %1, %1, %0\n\t"
                             "v_add_f32 %0, 0, %0"
                             : "=&v"(p), "+v"(A[sl2][c])
                             : "s"(kj), "v"(t[c]));
              else if (j != R)
                asm volatile("v_mul_f32 %0, %3, %4\n\tv_add_f32 %1, %1, %0\n\t"
                             "v_add_f32 %2, %2, %0"
                             : "=&v"(p), "+v"(A[sl][c]), "+v"(A[sl2][c])
                             : "s"(kj), "v"(t[c]));
              else
                asm volatile("v_mul_f32 %0, %2, %3\n\tv_add_f32 %1, %1, %0"
                             : "=&v"(p), "+v"(A[sl][c])
                             : "s"(kj), "v"(t[c]));
              if (j == 0)
                A[sl][c] = p;
              continue;
            }
            const float p = t[c] * taps.k[j];
            if (j == 0)
              A[sl][c] = 0.f + p;
            else
              A[sl][c] += p;
            if (j != R)
              A[sl2][c] += p;
          }
          if (MODE & 16)
            __builtin_amdgcn_sched_barrier(0);
        }
      }
      const int o = yy - R;
      if ((o >= y0) && (o < y1) && col_ok && (!NOSTORE || never))
      {
        vec ov;
#pragma unroll
        for (int c = 0; c < CPL; ++c)
          (&ov.x)[c] = A[i][c];
        *reinterpret_cast<vec*>(dst + size_t(o) * w + col) = ov;
      }
    }
    if (K % PF != 0)
    {
      vec tm[PF];
      float th[PF];
#pragma unroll
      for (int q = 0; q < PF; ++q)
      {
        tm[q] = pm[(K + q) % PF];
        th[q] = phv[(K + q) % PF];
      }
#pragma unroll
      for (int q = 0; q < PF; ++q)
      {
        pm[q] = tm[q];
        phv[q] = th[q];
      }
    }
  }
}

// ---- second generation: asm symmetric column pass + LDS window read one step
// ahead (issued before the column pass of the current row, consumed by the row
// pass of the next), CPL columns per lane.
template <int R, int PF, int CPL, bool EARLY, int NOMEM>
__global__ __launch_bounds__(64) void march2(const float* __restrict__ src,
                                             size_t src_stride,
                                             float* __restrict__ dst,
                                             size_t dst_stride, int w, int h,
                                             int seg_rows, int nstrips,
                                             Taps taps, int never,
                                             unsigned long long* trace)
{
  using vec = typename VecOf<CPL>::type;
  constexpr int K = 2 * R + 1;
  constexpr int W = 64 * CPL;
  constexpr int RP = ((R + CPL - 1) / CPL) * CPL;
  constexpr int D = RP - R;
  constexpr int ROWF = RP + W + RP;
  constexpr int NQ = (D + CPL + 2 * R + CPL - 1) / CPL;
  __shared__ __attribute__((aligned(16))) float s_row[2 * ROWF];
  unsigned long long t_start = 0;
  if (trace)
    t_start = __builtin_readcyclecounter();

  const int lane = threadIdx.x;
  const int strip = blockIdx.x % nstrips;
  const int seg = blockIdx.x / nstrips;
  const size_t b = blockIdx.y;
  src += b * src_stride;
  dst += b * dst_stride;
  const int x0 = strip * W;
  const int y0 = seg * seg_rows;
  const int y1 = min(h, y0 + seg_rows);
  const int col = x0 + CPL * lane;
  const bool col_ok = col < w;
  int hcol = lane < R ? x0 - R + lane : x0 + W + (lane - R);
  hcol = hcol < 0 ? 0 : (hcol > w - 1 ? w - 1 : hcol);
  const int hslot = lane < R ? RP - R + lane : RP + W + (lane - R);
  const int mcol = col_ok ? col : w - CPL;

  auto load_row = [&](int yy, vec& m, float& hv) {
    if (NOMEM)
    {
      (&m.x)[0] += 1.f;
      return;
    }
    const int gy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
    const float* rowp = src + size_t(gy) * w;
    m = *reinterpret_cast<const vec*>(rowp + mcol);
    if (!col_ok)
    {
      const float last = (&m.x)[CPL - 1];
#pragma unroll
      for (int c = 0; c < CPL; ++c)
        (&m.x)[c] = last;
    }
    hv = 0.f;
    if (lane < 2 * R)
      hv = rowp[hcol];
  };
  auto stage = [&](int n, const vec& m, float hv) {
    float* rowbuf = s_row + (n & 1) * ROWF;
    *reinterpret_cast<vec*>(rowbuf + RP + CPL * lane) = m;
    if (lane < 2 * R)
      rowbuf[hslot] = hv;
  };
  float v[NQ * CPL];
  auto read_window = [&](int n) {
    const vec* p =
        reinterpret_cast<const vec*>(s_row + (n & 1) * ROWF + CPL * lane);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const vec x = p[q];
#pragma unroll
      for (int c = 0; c < CPL; ++c)
        v[CPL * q + c] = (&x.x)[c];
    }
  };

  float A[K][CPL];
  vec pm[PF];
  float phv[PF];
  const int T = (y1 - y0) + 2 * R;
#pragma unroll
  for (int q = 0; q < PF; ++q)
  {
    pm[q] = vec{};
    phv[q] = 0.f;
    load_row(y0 - R + q, pm[q], phv[q]);
  }
  if (EARLY)
  {
    stage(0, pm[0], phv[0]);
    load_row(y0 - R + PF, pm[0], phv[0]);
    read_window(0);
  }

  for (int n0 = 0; n0 < T; n0 += K)
  {
#pragma unroll
    for (int i = 0; i < K; ++i)
    {
      const int n = n0 + i;
      const int yy = y0 - R + n;
      if (!EARLY)
      {
        stage(n, pm[i % PF], phv[i % PF]);
        load_row(yy + PF, pm[i % PF], phv[i % PF]);
        read_window(n);
      }
      float t[CPL];
#pragma unroll
      for (int c = 0; c < CPL; ++c)
      {
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j)
          sum += v[D + c + j] * taps.k[j];
        t[c] = sum;
      }
      if (EARLY)
      {
        stage(n + 1, pm[(i + 1) % PF], phv[(i + 1) % PF]);
        load_row(yy + 1 + PF, pm[(i + 1) % PF], phv[(i + 1) % PF]);
        read_window(n + 1);
      }
#pragma unroll
      for (int j = 0; j <= R; ++j)
      {
        const int sl = (i + K - 1 - j) % K;
        const int sl2 = (i + j) % K;
        const float kj = taps.k[j];
#pragma unroll
        for (int c = 0; c < CPL; ++c)
        {
          float p;
          if (j == 0)
          {
            asm volatile("v_mul_f32 %0, %2, %3\n\tv_add_f32 %1, %1, %0\n\t"
                         "v_add_f32 %0, 0, %0"
                         : "=&v"(p), "+v"(A[sl2][c])
                         : "s"(kj), "v"(t[c]));
            A[sl][c] = p;
          }
          else if (j != R)
            asm volatile("v_mul_f32 %0, %3, %4\n\tv_add_f32 %1, %1, %0\n\t"
                         "v_add_f32 %2, %2, %0"
                         : "=&v"(p), "+v"(A[sl][c]), "+v"(A[sl2][c])
                         : "s"(kj), "v"(t[c]));
          else
            asm volatile("v_mul_f32 %0, %2, %3\n\tv_add_f32 %1, %1, %0"
                         : "=&v"(p), "+v"(A[sl][c])
                         : "s"(kj), "v"(t[c]));
        }
      }
      const int o = yy - R;
      if ((o >= y0) && (o < y1) && col_ok && (!NOMEM || never))
      {
        vec ov;
#pragma unroll
        for (int c = 0; c < CPL; ++c)
          (&ov.x)[c] = A[i][c];
        *reinterpret_cast<vec*>(dst + size_t(o) * w + col) = ov;
      }
    }
    if (K % PF != 0)
    {
      vec tm[PF];
      float th[PF];
#pragma unroll
      for (int q = 0; q < PF; ++q)
      {
        tm[q] = pm[(K + q) % PF];
        th[q] = phv[(K + q) % PF];
      }
#pragma unroll
      for (int q = 0; q < PF; ++q)
      {
        pm[q] = tm[q];
        phv[q] = th[q];
      }
    }
  }
  if (trace && lane == 0)
  {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const size_t id = size_t(blockIdx.y) * gridDim.x + blockIdx.x;
    trace[3 * id] = (unsigned long long) (xcc) << 32 | hw;
    trace[3 * id + 1] = t_start;
    trace[3 * id + 2] = __builtin_readcyclecounter();
  }
}

// ---- third generation: row pass and column pass hand-scheduled (asm volatile
// blocks; producer -> consumer distance >= 4 instructions), symmetric column
// taps, CPL in {2, 4}.
#define ROW4_ASM                                                                 \
  "v_mul_f32 %2, %11, %6\n\tv_mul_f32 %3, %11, %7\n\t"                           \
  "v_mul_f32 %4, %12, %7\n\tv_mul_f32 %5, %12, %8\n\t"                           \
  "v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\t"                             \
  "v_mul_f32 %2, %13, %8\n\tv_mul_f32 %3, %13, %9\n\t"                           \
  "v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5\n\t"                             \
  "v_mul_f32 %4, %14, %9\n\tv_mul_f32 %5, %14, %10\n\t"                          \
  "v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\t"                             \
  "v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5"

//! s0 += sum_{q<4} v[q] k[q],  s1 += sum_{q<4} v[q+1] k[q]  (ascending q)
__device__ __forceinline__ void row4(float& s0, float& s1, float va, float vb,
                                     float vc, float vd, float ve, float k0,
                                     float k1, float k2, float k3)
{
  float p0, p1, p2, p3;
  asm volatile(ROW4_ASM
               : "+v"(s0), "+v"(s1), "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3)
               : "v"(va), "v"(vb), "v"(vc), "v"(vd), "v"(ve), "s"(k0), "s"(k1),
                 "s"(k2), "s"(k3));
}
__device__ __forceinline__ void row1(float& s0, float& s1, float va, float vb,
                                     float k0)
{
  float p0, p1;
  asm volatile("v_mul_f32 %2, %6, %4\n\tv_mul_f32 %3, %6, %5\n\t"
               "v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3"
               : "+v"(s0), "+v"(s1), "=&v"(p0), "=&v"(p1)
               : "v"(va), "v"(vb), "s"(k0));
}
//! taps j and j+1 (neither first nor centre) of two columns
__device__ __forceinline__ void col2(float& a0, float& a1, float& b0, float& b1,
                                     float& c0, float& c1, float& d0, float& d1,
                                     float t0, float t1, float kj, float kj1)
{
  float p0, p1, p2, p3;
  asm volatile("v_mul_f32 %8, %14, %12\n\tv_mul_f32 %9, %14, %13\n\t"
               "v_mul_f32 %10, %15, %12\n\tv_mul_f32 %11, %15, %13\n\t"
               "v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %9\n\t"
               "v_add_f32 %2, %2, %8\n\tv_add_f32 %3, %3, %9\n\t"
               "v_add_f32 %4, %4, %10\n\tv_add_f32 %5, %5, %11\n\t"
               "v_add_f32 %6, %6, %10\n\tv_add_f32 %7, %7, %11"
               : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(c0), "+v"(c1),
                 "+v"(d0), "+v"(d1), "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3)
               : "v"(t0), "v"(t1), "s"(kj), "s"(kj1));
}
__device__ __forceinline__ void col1(float& a0, float& a1, float& b0, float& b1,
                                     float t0, float t1, float kj)
{
  float p0, p1;
  asm volatile("v_mul_f32 %4, %8, %6\n\tv_mul_f32 %5, %8, %7\n\t"
               "v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5\n\t"
               "v_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %5"
               : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "=&v"(p0), "=&v"(p1)
               : "v"(t0), "v"(t1), "s"(kj));
}
//! first tap (new = 0 + t k0, old += t k0) and centre tap (mid += t kR)
__device__ __forceinline__ void col_ends(float& new0, float& new1, float& old0,
                                         float& old1, float& mid0, float& mid1,
                                         float t0, float t1, float k0, float kr)
{
  float p2, p3;
  asm volatile("v_mul_f32 %0, %10, %8\n\tv_mul_f32 %1, %10, %9\n\t"
               "v_mul_f32 %6, %11, %8\n\tv_mul_f32 %7, %11, %9\n\t"
               "v_add_f32 %2, %2, %0\n\tv_add_f32 %3, %3, %1\n\t"
               "v_add_f32 %0, 0, %0\n\tv_add_f32 %1, 0, %1\n\t"
               "v_add_f32 %4, %4, %6\n\tv_add_f32 %5, %5, %7"
               : "=&v"(new0), "=&v"(new1), "+v"(old0), "+v"(old1), "+v"(mid0),
                 "+v"(mid1), "=&v"(p2), "=&v"(p3)
               : "v"(t0), "v"(t1), "s"(k0), "s"(kr));
}

#define ROW4V_ASM                                                                 \
  "v_mul_f32 %2, %11, %6\n\tv_mul_f32 %3, %11, %7\n\t"                           \
  "v_mul_f32 %4, %12, %7\n\tv_mul_f32 %5, %12, %8\n\t"                           \
  "v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\t"                             \
  "v_mul_f32 %2, %13, %8\n\tv_mul_f32 %3, %13, %9\n\t"                           \
  "v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5\n\t"                             \
  "v_mul_f32 %4, %14, %9\n\tv_mul_f32 %5, %14, %10\n\t"                          \
  "v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\t"                             \
  "v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5"

//! s0 += sum_{q<4} v[q] k[q],  s1 += sum_{q<4} v[q+1] k[q]  (ascending q)
__device__ __forceinline__ void row4v(float& s0, float& s1, float va, float vb,
                                     float vc, float vd, float ve, float k0,
                                     float k1, float k2, float k3)
{
  float p0, p1, p2, p3;
  asm volatile(ROW4V_ASM
               : "+v"(s0), "+v"(s1), "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3)
               : "v"(va), "v"(vb), "v"(vc), "v"(vd), "v"(ve), "v"(k0), "v"(k1),
                 "v"(k2), "v"(k3));
}
__device__ __forceinline__ void row1v(float& s0, float& s1, float va, float vb,
                                     float k0)
{
  float p0, p1;
  asm volatile("v_mul_f32 %2, %6, %4\n\tv_mul_f32 %3, %6, %5\n\t"
               "v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3"
               : "+v"(s0), "+v"(s1), "=&v"(p0), "=&v"(p1)
               : "v"(va), "v"(vb), "v"(k0));
}
//! taps j and j+1 (neither first nor centre) of two columns
__device__ __forceinline__ void col2v(float& a0, float& a1, float& b0, float& b1,
                                     float& c0, float& c1, float& d0, float& d1,
                                     float t0, float t1, float kj, float kj1)
{
  float p0, p1, p2, p3;
  asm volatile("v_mul_f32 %8, %14, %12\n\tv_mul_f32 %9, %14, %13\n\t"
               "v_mul_f32 %10, %15, %12\n\tv_mul_f32 %11, %15, %13\n\t"
               "v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %9\n\t"
               "v_add_f32 %2, %2, %8\n\tv_add_f32 %3, %3, %9\n\t"
               "v_add_f32 %4, %4, %10\n\tv_add_f32 %5, %5, %11\n\t"
               "v_add_f32 %6, %6, %10\n\tv_add_f32 %7, %7, %11"
               : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(c0), "+v"(c1),
                 "+v"(d0), "+v"(d1), "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3)
               : "v"(t0), "v"(t1), "v"(kj), "v"(kj1));
}
__device__ __forceinline__ void col1v(float& a0, float& a1, float& b0, float& b1,
                                     float t0, float t1, float kj)
{
  float p0, p1;
  asm volatile("v_mul_f32 %4, %8, %6\n\tv_mul_f32 %5, %8, %7\n\t"
               "v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5\n\t"
               "v_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %5"
               : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "=&v"(p0), "=&v"(p1)
               : "v"(t0), "v"(t1), "v"(kj));
}
//! first tap (new = 0 + t k0, old += t k0) and centre tap (mid += t kR)
__device__ __forceinline__ void col_endsv(float& new0, float& new1, float& old0,
                                         float& old1, float& mid0, float& mid1,
                                         float t0, float t1, float k0, float kr)
{
  float p2, p3;
  asm volatile("v_mul_f32 %0, %10, %8\n\tv_mul_f32 %1, %10, %9\n\t"
               "v_mul_f32 %6, %11, %8\n\tv_mul_f32 %7, %11, %9\n\t"
               "v_add_f32 %2, %2, %0\n\tv_add_f32 %3, %3, %1\n\t"
               "v_add_f32 %0, 0, %0\n\tv_add_f32 %1, 0, %1\n\t"
               "v_add_f32 %4, %4, %6\n\tv_add_f32 %5, %5, %7"
               : "=&v"(new0), "=&v"(new1), "+v"(old0), "+v"(old1), "+v"(mid0),
                 "+v"(mid1), "=&v"(p2), "=&v"(p3)
               : "v"(t0), "v"(t1), "v"(k0), "v"(kr));
}

template <int R, int PF, int CPL, int NOMEM>
__global__ __launch_bounds__(64) void march3(const float* __restrict__ src,
                                             size_t src_stride,
                                             float* __restrict__ dst,
                                             size_t dst_stride, int w, int h,
                                             int seg_rows, int nstrips,
                                             Taps taps, int never)
{
  using vec = typename VecOf<CPL>::type;
  constexpr int K = 2 * R + 1;
  constexpr int W = 64 * CPL;
  constexpr int RP = ((R + CPL - 1) / CPL) * CPL;
  constexpr int D = RP - R;
  constexpr int ROWF = RP + W + RP;
  constexpr int NQ = (D + CPL + 2 * R + CPL - 1) / CPL;
  // NOMEM == 4: real memory traffic, halo lanes handled without exec-mask
  // branches (the other lanes write a dummy LDS slot / re-load their column)
  constexpr bool BRFREE = NOMEM == 4;
  __shared__ __attribute__((aligned(16))) float s_row[2 * ROWF + 64];

  const int lane = threadIdx.x;
  const int strip = blockIdx.x % nstrips;
  const int seg = blockIdx.x / nstrips;
  const size_t b = blockIdx.y;
  src += b * src_stride;
  dst += b * dst_stride;
  const int x0 = strip * W;
  const int y0 = seg * seg_rows;
  const int y1 = min(h, y0 + seg_rows);
  const int col = x0 + CPL * lane;
  const bool col_ok = col < w;
  int hcol = lane < R ? x0 - R + lane : x0 + W + (lane - R);
  hcol = hcol < 0 ? 0 : (hcol > w - 1 ? w - 1 : hcol);
  const int hslot = lane < R ? RP - R + lane : RP + W + (lane - R);
  const int mcol = col_ok ? col : w - CPL;

  auto load_row = [&](int yy, vec& m, float& hv) {
    if (NOMEM == 1 || NOMEM == 2)
    {
      (&m.x)[0] += 1.f;
      return;
    }
    const int gy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
    const float* rowp = src + size_t(gy) * w;
    m = *reinterpret_cast<const vec*>(rowp + mcol);
    if (!col_ok)
    {
      const float last = (&m.x)[CPL - 1];
#pragma unroll
      for (int c = 0; c < CPL; ++c)
        (&m.x)[c] = last;
    }
    if (BRFREE)
      hv = rowp[lane < 2 * R ? hcol : mcol];
    else
    {
      hv = 0.f;
      if (lane < 2 * R)
        hv = rowp[hcol];
    }
  };

  float A[K][CPL];
  vec pm[PF];
  float phv[PF];
  const int T = (y1 - y0) + 2 * R;
#pragma unroll
  for (int q = 0; q < PF; ++q)
  {
    pm[q] = vec{};
    phv[q] = 0.f;
    load_row(y0 - R + q, pm[q], phv[q]);
  }

  for (int n0 = 0; n0 < T; n0 += K)
  {
#pragma unroll
    for (int i = 0; i < K; ++i)
    {
      const int n = n0 + i;
      const int yy = y0 - R + n;
      float* rowbuf = s_row + (n & 1) * ROWF;
      if (NOMEM < 2 || BRFREE)
      {
        *reinterpret_cast<vec*>(rowbuf + RP + CPL * lane) = pm[i % PF];
        if (BRFREE)
        {
          float* hp = lane < 2 * R ? rowbuf + hslot : s_row + 2 * ROWF + lane;
          *hp = phv[i % PF];
        }
        else if (lane < 2 * R)
          rowbuf[hslot] = phv[i % PF];
      }
      load_row(yy + PF, pm[i % PF], phv[i % PF]);
      float v[NQ * CPL];
      if (NOMEM == 2)
      {
#pragma unroll
        for (int q = 0; q < NQ * CPL; ++q)
          asm volatile("v_mov_b32 %0, %1" : "=v"(v[q]) : "v"(phv[q % PF]));
      }
      else
      {
        const vec* p = reinterpret_cast<const vec*>(rowbuf + CPL * lane);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const vec x = p[q];
#pragma unroll
          for (int c = 0; c < CPL; ++c)
            v[CPL * q + c] = (&x.x)[c];
        }
      }
      float t[CPL];
#pragma unroll
      for (int c = 0; c < CPL; c += 2)
      {
        float s0 = 0.f, s1 = 0.f;
        constexpr int NB = K / 4;
#pragma unroll
        for (int q = 0; q < NB; ++q)
        {
          const int j = 4 * q;
          row4(s0, s1, v[D + c + j], v[D + c + j + 1], v[D + c + j + 2],
               v[D + c + j + 3], v[D + c + j + 4], taps.k[j], taps.k[j + 1],
               taps.k[j + 2], taps.k[j + 3]);
        }
#pragma unroll
        for (int j = 4 * NB; j < K; ++j)
          row1(s0, s1, v[D + c + j], v[D + c + j + 1], taps.k[j]);
        t[c] = s0;
        t[c + 1] = s1;
      }
      // column pass: tap j to the output that is j steps old, tap K-1-j (the
      // same product) to the one that is K-1-j steps old
#define SL(j) ((i + K - 1 - (j)) % K)
#pragma unroll
      for (int c = 0; c < CPL; c += 2)
      {
        col_ends(A[SL(0)][c], A[SL(0)][c + 1], A[SL(K - 1)][c],
                 A[SL(K - 1)][c + 1], A[SL(R)][c], A[SL(R)][c + 1], t[c],
                 t[c + 1], taps.k[0], taps.k[R]);
#pragma unroll
        for (int j = 1; j + 1 < R; j += 2)
          col2(A[SL(j)][c], A[SL(j)][c + 1], A[SL(K - 1 - j)][c],
               A[SL(K - 1 - j)][c + 1], A[SL(j + 1)][c], A[SL(j + 1)][c + 1],
               A[SL(K - 2 - j)][c], A[SL(K - 2 - j)][c + 1], t[c], t[c + 1],
               taps.k[j], taps.k[j + 1]);
        if ((R - 1) % 2 == 1)
          col1(A[SL(R - 1)][c], A[SL(R - 1)][c + 1], A[SL(R + 1)][c],
               A[SL(R + 1)][c + 1], t[c], t[c + 1], taps.k[R - 1]);
      }
#undef SL
      const int o = yy - R;
      if ((o >= y0) && (o < y1) && col_ok && (NOMEM == 0 || BRFREE || never))
      {
        vec ov;
#pragma unroll
        for (int c = 0; c < CPL; ++c)
          (&ov.x)[c] = A[i][c];
        *reinterpret_cast<vec*>(dst + size_t(o) * w + col) = ov;
      }
    }
    if (K % PF != 0)
    {
      vec tm[PF];
      float th[PF];
#pragma unroll
      for (int q = 0; q < PF; ++q)
      {
        tm[q] = pm[(K + q) % PF];
        th[q] = phv[(K + q) % PF];
      }
#pragma unroll
      for (int q = 0; q < PF; ++q)
      {
        pm[q] = tm[q];
        phv[q] = th[q];
      }
    }
  }
}

template <int R, int PF, int CPL, int NOMEM>
__global__ __launch_bounds__(64) void march3p(const float* __restrict__ src,
                                             size_t src_stride,
                                             float* __restrict__ dst,
                                             size_t dst_stride, int w, int h,
                                             int seg_rows, int nstrips,
                                             Taps taps, int never,
                                             unsigned long long* prof)
{
  // per-phase cycle sums of this wave: vm wait, LDS stage+read, row pass,
  // column pass, store/loop overhead
  unsigned long long ph[5] = {0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_readcyclecounter();
#define PHASE(x)                                          \
  {                                                        \
    const unsigned long long tn = __builtin_readcyclecounter(); \
    ph[x] += tn - tprev;                                   \
    tprev = tn;                                            \
  }

  using vec = typename VecOf<CPL>::type;
  constexpr int K = 2 * R + 1;
  constexpr int W = 64 * CPL;
  constexpr int RP = ((R + CPL - 1) / CPL) * CPL;
  constexpr int D = RP - R;
  constexpr int ROWF = RP + W + RP;
  constexpr int NQ = (D + CPL + 2 * R + CPL - 1) / CPL;
  __shared__ __attribute__((aligned(16))) float s_row[2 * ROWF];

  const int lane = threadIdx.x;
  const int strip = blockIdx.x % nstrips;
  const int seg = blockIdx.x / nstrips;
  const size_t b = blockIdx.y;
  src += b * src_stride;
  dst += b * dst_stride;
  const int x0 = strip * W;
  const int y0 = seg * seg_rows;
  const int y1 = min(h, y0 + seg_rows);
  const int col = x0 + CPL * lane;
  const bool col_ok = col < w;
  int hcol = lane < R ? x0 - R + lane : x0 + W + (lane - R);
  hcol = hcol < 0 ? 0 : (hcol > w - 1 ? w - 1 : hcol);
  const int hslot = lane < R ? RP - R + lane : RP + W + (lane - R);
  const int mcol = col_ok ? col : w - CPL;

  auto load_row = [&](int yy, vec& m, float& hv) {
    if (NOMEM)
    {
      (&m.x)[0] += 1.f;
      return;
    }
    const int gy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
    const float* rowp = src + size_t(gy) * w;
    m = *reinterpret_cast<const vec*>(rowp + mcol);
    if (!col_ok)
    {
      const float last = (&m.x)[CPL - 1];
#pragma unroll
      for (int c = 0; c < CPL; ++c)
        (&m.x)[c] = last;
    }
    hv = 0.f;
    if (lane < 2 * R)
      hv = rowp[hcol];
  };

  float A[K][CPL];
  vec pm[PF];
  float phv[PF];
  const int T = (y1 - y0) + 2 * R;
#pragma unroll
  for (int q = 0; q < PF; ++q)
  {
    pm[q] = vec{};
    phv[q] = 0.f;
    load_row(y0 - R + q, pm[q], phv[q]);
  }

  for (int n0 = 0; n0 < T; n0 += K)
  {
#pragma unroll
    for (int i = 0; i < K; ++i)
    {
      const int n = n0 + i;
      const int yy = y0 - R + n;
      float* rowbuf = s_row + (n & 1) * ROWF;
      PHASE(4)
      asm volatile("" ::"v"((&pm[i % PF].x)[0]), "v"(phv[i % PF]));
      PHASE(0)
      if (NOMEM < 2)
      {
        *reinterpret_cast<vec*>(rowbuf + RP + CPL * lane) = pm[i % PF];
        if (lane < 2 * R)
          rowbuf[hslot] = phv[i % PF];
      }
      load_row(yy + PF, pm[i % PF], phv[i % PF]);
      float v[NQ * CPL];
      if (NOMEM >= 2)
      {
#pragma unroll
        for (int q = 0; q < NQ * CPL; ++q)
          asm volatile("v_mov_b32 %0, %1" : "=v"(v[q]) : "v"(phv[q % PF]));
      }
      else
      {
        const vec* p = reinterpret_cast<const vec*>(rowbuf + CPL * lane);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const vec x = p[q];
#pragma unroll
          for (int c = 0; c < CPL; ++c)
            v[CPL * q + c] = (&x.x)[c];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PHASE(1)
      float t[CPL];
#pragma unroll
      for (int c = 0; c < CPL; c += 2)
      {
        float s0 = 0.f, s1 = 0.f;
        constexpr int NB = K / 4;
#pragma unroll
        for (int q = 0; q < NB; ++q)
        {
          const int j = 4 * q;
          row4(s0, s1, v[D + c + j], v[D + c + j + 1], v[D + c + j + 2],
               v[D + c + j + 3], v[D + c + j + 4], taps.k[j], taps.k[j + 1],
               taps.k[j + 2], taps.k[j + 3]);
        }
#pragma unroll
        for (int j = 4 * NB; j < K; ++j)
          row1(s0, s1, v[D + c + j], v[D + c + j + 1], taps.k[j]);
        t[c] = s0;
        t[c + 1] = s1;
      }
      asm volatile("" ::"v"(t[0]), "v"(t[1]));
      PHASE(2)
#define SL(j) ((i + K - 1 - (j)) % K)
#pragma unroll
      for (int c = 0; c < CPL; c += 2)
      {
        col_ends(A[SL(0)][c], A[SL(0)][c + 1], A[SL(K - 1)][c],
                 A[SL(K - 1)][c + 1], A[SL(R)][c], A[SL(R)][c + 1], t[c],
                 t[c + 1], taps.k[0], taps.k[R]);
#pragma unroll
        for (int j = 1; j + 1 < R; j += 2)
          col2(A[SL(j)][c], A[SL(j)][c + 1], A[SL(K - 1 - j)][c],
               A[SL(K - 1 - j)][c + 1], A[SL(j + 1)][c], A[SL(j + 1)][c + 1],
               A[SL(K - 2 - j)][c], A[SL(K - 2 - j)][c + 1], t[c], t[c + 1],
               taps.k[j], taps.k[j + 1]);
        if ((R - 1) % 2 == 1)
          col1(A[SL(R - 1)][c], A[SL(R - 1)][c + 1], A[SL(R + 1)][c],
               A[SL(R + 1)][c + 1], t[c], t[c + 1], taps.k[R - 1]);
      }
#undef SL
      asm volatile("" ::"v"(A[i][0]), "v"(A[i][1]));
      PHASE(3)
      const int o = yy - R;
      if ((o >= y0) && (o < y1) && col_ok && (!NOMEM || never))
      {
        vec ov;
#pragma unroll
        for (int c = 0; c < CPL; ++c)
          (&ov.x)[c] = A[i][c];
        *reinterpret_cast<vec*>(dst + size_t(o) * w + col) = ov;
      }
    }
    if (K % PF != 0)
    {
      vec tm[PF];
      float th[PF];
#pragma unroll
      for (int q = 0; q < PF; ++q)
      {
        tm[q] = pm[(K + q) % PF];
        th[q] = phv[(K + q) % PF];
      }
#pragma unroll
      for (int q = 0; q < PF; ++q)
      {
        pm[q] = tm[q];
        phv[q] = th[q];
      }
    }
  }
  if (lane == 0)
  {
    const size_t id = size_t(blockIdx.y) * gridDim.x + blockIdx.x;
#pragma unroll
    for (int q = 0; q < 5; ++q)
      prof[id * 6 + q] = ph[q];
    prof[id * 6 + 5] = (unsigned long long) T;
  }
#undef PHASE
}

template <int R, int PF, int CPL, int NOMEM>
__global__ __launch_bounds__(64) void march3v(const float* __restrict__ src,
                                             size_t src_stride,
                                             float* __restrict__ dst,
                                             size_t dst_stride, int w, int h,
                                             int seg_rows, int nstrips,
                                             Taps taps, int never)
{
  using vec = typename VecOf<CPL>::type;
  constexpr int K = 2 * R + 1;
  constexpr int W = 64 * CPL;
  constexpr int RP = ((R + CPL - 1) / CPL) * CPL;
  constexpr int D = RP - R;
  constexpr int ROWF = RP + W + RP;
  constexpr int NQ = (D + CPL + 2 * R + CPL - 1) / CPL;
  __shared__ __attribute__((aligned(16))) float s_row[2 * ROWF];

  const int lane = threadIdx.x;
  const int strip = blockIdx.x % nstrips;
  const int seg = blockIdx.x / nstrips;
  const size_t b = blockIdx.y;
  src += b * src_stride;
  dst += b * dst_stride;
  const int x0 = strip * W;
  const int y0 = seg * seg_rows;
  const int y1 = min(h, y0 + seg_rows);
  const int col = x0 + CPL * lane;
  const bool col_ok = col < w;
  int hcol = lane < R ? x0 - R + lane : x0 + W + (lane - R);
  hcol = hcol < 0 ? 0 : (hcol > w - 1 ? w - 1 : hcol);
  const int hslot = lane < R ? RP - R + lane : RP + W + (lane - R);
  const int mcol = col_ok ? col : w - CPL;

  auto load_row = [&](int yy, vec& m, float& hv) {
    if (NOMEM)
    {
      (&m.x)[0] += 1.f;
      return;
    }
    const int gy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
    const float* rowp = src + size_t(gy) * w;
    m = *reinterpret_cast<const vec*>(rowp + mcol);
    if (!col_ok)
    {
      const float last = (&m.x)[CPL - 1];
#pragma unroll
      for (int c = 0; c < CPL; ++c)
        (&m.x)[c] = last;
    }
    hv = 0.f;
    if (lane < 2 * R)
      hv = rowp[hcol];
  };

  float tk[R + 1];
#pragma unroll
  for (int j = 0; j <= R; ++j)
    asm volatile("v_mov_b32 %0, %1" : "=v"(tk[j]) : "s"(taps.k[j]));
#define TK(j) tk[(j) <= R ? (j) : 2 * R - (j)]
  float A[K][CPL];
  vec pm[PF];
  float phv[PF];
  const int T = (y1 - y0) + 2 * R;
#pragma unroll
  for (int q = 0; q < PF; ++q)
  {
    pm[q] = vec{};
    phv[q] = 0.f;
    load_row(y0 - R + q, pm[q], phv[q]);
  }

  for (int n0 = 0; n0 < T; n0 += K)
  {
#pragma unroll
    for (int i = 0; i < K; ++i)
    {
      const int n = n0 + i;
      const int yy = y0 - R + n;
      float* rowbuf = s_row + (n & 1) * ROWF;
      if (NOMEM < 2)
      {
        *reinterpret_cast<vec*>(rowbuf + RP + CPL * lane) = pm[i % PF];
        if (lane < 2 * R)
          rowbuf[hslot] = phv[i % PF];
      }
      load_row(yy + PF, pm[i % PF], phv[i % PF]);
      float v[NQ * CPL];
      if (NOMEM >= 2)
      {
#pragma unroll
        for (int q = 0; q < NQ * CPL; ++q)
          asm volatile("v_mov_b32 %0, %1" : "=v"(v[q]) : "v"(phv[q % PF]));
      }
      else
      {
        const vec* p = reinterpret_cast<const vec*>(rowbuf + CPL * lane);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const vec x = p[q];
#pragma unroll
          for (int c = 0; c < CPL; ++c)
            v[CPL * q + c] = (&x.x)[c];
        }
      }
      float t[CPL];
#pragma unroll
      for (int c = 0; c < CPL; c += 2)
      {
        float s0 = 0.f, s1 = 0.f;
        constexpr int NB = K / 4;
#pragma unroll
        for (int q = 0; q < NB; ++q)
        {
          const int j = 4 * q;
          row4v(s0, s1, v[D + c + j], v[D + c + j + 1], v[D + c + j + 2],
               v[D + c + j + 3], v[D + c + j + 4], TK(j), TK(j + 1),
               TK(j + 2), TK(j + 3));
        }
#pragma unroll
        for (int j = 4 * NB; j < K; ++j)
          row1v(s0, s1, v[D + c + j], v[D + c + j + 1], TK(j));
        t[c] = s0;
        t[c + 1] = s1;
      }
      // column pass: tap j to the output that is j steps old, tap K-1-j (the
      // same product) to the one that is K-1-j steps old
#define SL(j) ((i + K - 1 - (j)) % K)
#pragma unroll
      for (int c = 0; c < CPL; c += 2)
      {
        col_endsv(A[SL(0)][c], A[SL(0)][c + 1], A[SL(K - 1)][c],
                 A[SL(K - 1)][c + 1], A[SL(R)][c], A[SL(R)][c + 1], t[c],
                 t[c + 1], TK(0), TK(R));
#pragma unroll
        for (int j = 1; j + 1 < R; j += 2)
          col2v(A[SL(j)][c], A[SL(j)][c + 1], A[SL(K - 1 - j)][c],
               A[SL(K - 1 - j)][c + 1], A[SL(j + 1)][c], A[SL(j + 1)][c + 1],
               A[SL(K - 2 - j)][c], A[SL(K - 2 - j)][c + 1], t[c], t[c + 1],
               TK(j), TK(j + 1));
        if ((R - 1) % 2 == 1)
          col1v(A[SL(R - 1)][c], A[SL(R - 1)][c + 1], A[SL(R + 1)][c],
               A[SL(R + 1)][c + 1], t[c], t[c + 1], TK(R - 1));
      }
#undef SL
      const int o = yy - R;
      if ((o >= y0) && (o < y1) && col_ok && (!NOMEM || never))
      {
        vec ov;
#pragma unroll
        for (int c = 0; c < CPL; ++c)
          (&ov.x)[c] = A[i][c];
        *reinterpret_cast<vec*>(dst + size_t(o) * w + col) = ov;
      }
    }
    if (K % PF != 0)
    {
      vec tm[PF];
      float th[PF];
#pragma unroll
      for (int q = 0; q < PF; ++q)
      {
        tm[q] = pm[(K + q) % PF];
        th[q] = phv[(K + q) % PF];
      }
#pragma unroll
      for (int q = 0; q < PF; ++q)
      {
        pm[q] = tm[q];
        phv[q] = th[q];
      }
    }
  }
}

// ---- fourth generation (CPL = 2): as v3, plus the LDS window is read with
// single ds_read_b64 (256 B/clk; the compiler's ds_read2_b64 runs at 128 B/clk)
// issued from asm one step ahead, before the column pass, and waited for inside
// the first row-pass block.  LDSPAD pads the workgroup's LDS footprint to
// control the number of resident waves per SIMD (keep it even: the two VALU
// pipes of a SIMD are bound to waves).
template <int NQ>
struct Window
{
  float2 q[NQ];
};

template <int R, int PF, int NOMEM, int LDSPAD, int V4MODE = 3>
__global__ __launch_bounds__(64) void march4(const float* __restrict__ src,
                                             size_t src_stride,
                                             float* __restrict__ dst,
                                             size_t dst_stride, int w, int h,
                                             int seg_rows, int nstrips,
                                             Taps taps, int never)
{
  constexpr int CPL = 2;
  using vec = float2;
  constexpr int K = 2 * R + 1;
  constexpr int W = 64 * CPL;
  constexpr int RP = ((R + CPL - 1) / CPL) * CPL;
  constexpr int D = RP - R;
  constexpr int ROWF = RP + W + RP;
  constexpr int NQ = (D + CPL + 2 * R + CPL - 1) / CPL;
  __shared__ __attribute__((aligned(16))) float s_row[LDSPAD ? LDSPAD / 4 : 2 * ROWF];

  const int lane = threadIdx.x;
  const int strip = blockIdx.x % nstrips;
  const int seg = blockIdx.x / nstrips;
  const size_t b = blockIdx.y;
  src += b * src_stride;
  dst += b * dst_stride;
  const int x0 = strip * W;
  const int y0 = seg * seg_rows;
  const int y1 = min(h, y0 + seg_rows);
  const int col = x0 + CPL * lane;
  const bool col_ok = col < w;
  int hcol = lane < R ? x0 - R + lane : x0 + W + (lane - R);
  hcol = hcol < 0 ? 0 : (hcol > w - 1 ? w - 1 : hcol);
  const int hslot = lane < R ? RP - R + lane : RP + W + (lane - R);
  const int mcol = col_ok ? col : w - CPL;
  if (LDSPAD && never)
    s_row[LDSPAD / 4 - 64 + lane] = 0.f;

  auto load_row = [&](int yy, vec& m, float& hv) {
    if (NOMEM)
    {
      m.x += 1.f;
      return;
    }
    const int gy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
    const float* rowp = src + size_t(gy) * w;
    m = *reinterpret_cast<const vec*>(rowp + mcol);
    if (!col_ok)
      m.x = m.y;
    hv = 0.f;
    if (lane < 2 * R)
      hv = rowp[hcol];
  };
  auto stage = [&](int n, const vec& m, float hv) {
    float* rowbuf = s_row + (n & 1) * ROWF;
    *reinterpret_cast<vec*>(rowbuf + RP + CPL * lane) = m;
    if (lane < 2 * R)
      rowbuf[hslot] = hv;
  };
  float2 v[NQ];
  // LDS byte address of this lane's window in slot 0; slot 1 is ROWF*4 further
  const unsigned lds_base =
      unsigned(reinterpret_cast<uintptr_t>(s_row)) + 8u * lane;
  auto read_window = [&](int n) {
    const unsigned a = lds_base + ((n & 1) ? ROWF * 4u : 0u);
    if (V4MODE & 1)
    {
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        asm volatile("ds_read_b64 %0, %1 offset:%2"
                     : "=v"(v[q])
                     : "v"(a), "n"(8 * q)
                     : "memory");
    }
    else
    {
      const float2* p = reinterpret_cast<const float2*>(
          s_row + (n & 1) * ROWF + CPL * lane);
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        v[q] = p[q];
    }
  };
#define VF(i_) (((i_) & 1) ? v[(i_) >> 1].y : v[(i_) >> 1].x)

  float A[K][CPL];
  vec pm[PF];
  float phv[PF];
  const int T = (y1 - y0) + 2 * R;
#pragma unroll
  for (int q = 0; q < PF; ++q)
  {
    pm[q] = vec{};
    phv[q] = 0.f;
    load_row(y0 - R + q, pm[q], phv[q]);
  }
  if (V4MODE & 2)
  {
    stage(0, pm[0], phv[0]);
    load_row(y0 - R + PF, pm[0], phv[0]);
    read_window(0);
  }

  for (int n0 = 0; n0 < T; n0 += K)
  {
#pragma unroll
    for (int i = 0; i < K; ++i)
    {
      const int n = n0 + i;
      const int yy = y0 - R + n;
      float t[CPL];
      if (!(V4MODE & 2))
      {
        stage(n, pm[i % PF], phv[i % PF]);
        load_row(yy + PF, pm[i % PF], phv[i % PF]);
        read_window(n);
      }
      {
        float s0 = 0.f, s1 = 0.f;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        constexpr int NB = K / 4;
#pragma unroll
        for (int q = 0; q < NB; ++q)
        {
          const int j = 4 * q;
          row4(s0, s1, VF(D + j), VF(D + j + 1), VF(D + j + 2), VF(D + j + 3),
               VF(D + j + 4), taps.k[j], taps.k[j + 1], taps.k[j + 2],
               taps.k[j + 3]);
        }
#pragma unroll
        for (int j = 4 * NB; j < K; ++j)
          row1(s0, s1, VF(D + j), VF(D + j + 1), taps.k[j]);
        t[0] = s0;
        t[1] = s1;
      }
      if (V4MODE & 2)
      {
        stage(n + 1, pm[(i + 1) % PF], phv[(i + 1) % PF]);
        load_row(yy + 1 + PF, pm[(i + 1) % PF], phv[(i + 1) % PF]);
        read_window(n + 1);
      }
#define SL(j) ((i + K - 1 - (j)) % K)
      {
        constexpr int c = 0;
        col_ends(A[SL(0)][c], A[SL(0)][c + 1], A[SL(K - 1)][c],
                 A[SL(K - 1)][c + 1], A[SL(R)][c], A[SL(R)][c + 1], t[c],
                 t[c + 1], taps.k[0], taps.k[R]);
#pragma unroll
        for (int j = 1; j + 1 < R; j += 2)
          col2(A[SL(j)][c], A[SL(j)][c + 1], A[SL(K - 1 - j)][c],
               A[SL(K - 1 - j)][c + 1], A[SL(j + 1)][c], A[SL(j + 1)][c + 1],
               A[SL(K - 2 - j)][c], A[SL(K - 2 - j)][c + 1], t[c], t[c + 1],
               taps.k[j], taps.k[j + 1]);
        if ((R - 1) % 2 == 1)
          col1(A[SL(R - 1)][c], A[SL(R - 1)][c + 1], A[SL(R + 1)][c],
               A[SL(R + 1)][c + 1], t[c], t[c + 1], taps.k[R - 1]);
      }
#undef SL
      const int o = yy - R;
      if ((o >= y0) && (o < y1) && col_ok && (!NOMEM || never))
        *reinterpret_cast<vec*>(dst + size_t(o) * w + col) =
            make_float2(A[i][0], A[i][1]);
    }
    if (K % PF != 0)
    {
      vec tm[PF];
      float th[PF];
#pragma unroll
      for (int q = 0; q < PF; ++q)
      {
        tm[q] = pm[(K + q) % PF];
        th[q] = phv[(K + q) % PF];
      }
#pragma unroll
      for (int q = 0; q < PF; ++q)
      {
        pm[q] = tm[q];
        phv[q] = th[q];
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef VF
}

template <int R, int PF, int NOMEM, int LDSPAD, int V4MODE = 3>
__global__ __launch_bounds__(64) void march4p(const float* __restrict__ src,
                                             size_t src_stride,
                                             float* __restrict__ dst,
                                             size_t dst_stride, int w, int h,
                                             int seg_rows, int nstrips,
                                             Taps taps, int never,
                                             unsigned long long* prof)
{
  unsigned long long ph[5] = {0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_readcyclecounter();
#define PHASE(x)                                          \
  {                                                        \
    const unsigned long long tn = __builtin_readcyclecounter(); \
    ph[x] += tn - tprev;                                   \
    tprev = tn;                                            \
  }

  constexpr int CPL = 2;
  using vec = float2;
  constexpr int K = 2 * R + 1;
  constexpr int W = 64 * CPL;
  constexpr int RP = ((R + CPL - 1) / CPL) * CPL;
  constexpr int D = RP - R;
  constexpr int ROWF = RP + W + RP;
  constexpr int NQ = (D + CPL + 2 * R + CPL - 1) / CPL;
  __shared__ __attribute__((aligned(16))) float s_row[LDSPAD ? LDSPAD / 4 : 2 * ROWF];

  const int lane = threadIdx.x;
  const int strip = blockIdx.x % nstrips;
  const int seg = blockIdx.x / nstrips;
  const size_t b = blockIdx.y;
  src += b * src_stride;
  dst += b * dst_stride;
  const int x0 = strip * W;
  const int y0 = seg * seg_rows;
  const int y1 = min(h, y0 + seg_rows);
  const int col = x0 + CPL * lane;
  const bool col_ok = col < w;
  int hcol = lane < R ? x0 - R + lane : x0 + W + (lane - R);
  hcol = hcol < 0 ? 0 : (hcol > w - 1 ? w - 1 : hcol);
  const int hslot = lane < R ? RP - R + lane : RP + W + (lane - R);
  const int mcol = col_ok ? col : w - CPL;
  if (LDSPAD && never)
    s_row[LDSPAD / 4 - 64 + lane] = 0.f;

  auto load_row = [&](int yy, vec& m, float& hv) {
    if (NOMEM)
    {
      m.x += 1.f;
      return;
    }
    const int gy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
    const float* rowp = src + size_t(gy) * w;
    m = *reinterpret_cast<const vec*>(rowp + mcol);
    if (!col_ok)
      m.x = m.y;
    hv = 0.f;
    if (lane < 2 * R)
      hv = rowp[hcol];
  };
  auto stage = [&](int n, const vec& m, float hv) {
    float* rowbuf = s_row + (n & 1) * ROWF;
    *reinterpret_cast<vec*>(rowbuf + RP + CPL * lane) = m;
    if (lane < 2 * R)
      rowbuf[hslot] = hv;
  };
  float2 v[NQ];
  // LDS byte address of this lane's window in slot 0; slot 1 is ROWF*4 further
  const unsigned lds_base =
      unsigned(reinterpret_cast<uintptr_t>(s_row)) + 8u * lane;
  auto read_window = [&](int n) {
    const unsigned a = lds_base + ((n & 1) ? ROWF * 4u : 0u);
    if (V4MODE & 1)
    {
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        asm volatile("ds_read_b64 %0, %1 offset:%2"
                     : "=v"(v[q])
                     : "v"(a), "n"(8 * q)
                     : "memory");
    }
    else
    {
      const float2* p = reinterpret_cast<const float2*>(
          s_row + (n & 1) * ROWF + CPL * lane);
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        v[q] = p[q];
    }
  };
#define VF(i_) (((i_) & 1) ? v[(i_) >> 1].y : v[(i_) >> 1].x)

  float A[K][CPL];
  vec pm[PF];
  float phv[PF];
  const int T = (y1 - y0) + 2 * R;
#pragma unroll
  for (int q = 0; q < PF; ++q)
  {
    pm[q] = vec{};
    phv[q] = 0.f;
    load_row(y0 - R + q, pm[q], phv[q]);
  }
  if (V4MODE & 2)
  {
    stage(0, pm[0], phv[0]);
    load_row(y0 - R + PF, pm[0], phv[0]);
    read_window(0);
  }

  for (int n0 = 0; n0 < T; n0 += K)
  {
#pragma unroll
    for (int i = 0; i < K; ++i)
    {
      const int n = n0 + i;
      const int yy = y0 - R + n;
      float t[CPL];
      if (!(V4MODE & 2))
      {
        stage(n, pm[i % PF], phv[i % PF]);
        load_row(yy + PF, pm[i % PF], phv[i % PF]);
        read_window(n);
      }
      {
        float s0 = 0.f, s1 = 0.f;
        PHASE(4)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PHASE(0)
        constexpr int NB = K / 4;
#pragma unroll
        for (int q = 0; q < NB; ++q)
        {
          const int j = 4 * q;
          row4(s0, s1, VF(D + j), VF(D + j + 1), VF(D + j + 2), VF(D + j + 3),
               VF(D + j + 4), taps.k[j], taps.k[j + 1], taps.k[j + 2],
               taps.k[j + 3]);
        }
#pragma unroll
        for (int j = 4 * NB; j < K; ++j)
          row1(s0, s1, VF(D + j), VF(D + j + 1), taps.k[j]);
        t[0] = s0;
        t[1] = s1;
      }
      asm volatile("" ::"v"(t[0]), "v"(t[1]));
      PHASE(1)
      if (V4MODE & 2)
      {
        stage(n + 1, pm[(i + 1) % PF], phv[(i + 1) % PF]);
        load_row(yy + 1 + PF, pm[(i + 1) % PF], phv[(i + 1) % PF]);
        read_window(n + 1);
      }
      PHASE(2)
#define SL(j) ((i + K - 1 - (j)) % K)
      {
        constexpr int c = 0;
        col_ends(A[SL(0)][c], A[SL(0)][c + 1], A[SL(K - 1)][c],
                 A[SL(K - 1)][c + 1], A[SL(R)][c], A[SL(R)][c + 1], t[c],
                 t[c + 1], taps.k[0], taps.k[R]);
#pragma unroll
        for (int j = 1; j + 1 < R; j += 2)
          col2(A[SL(j)][c], A[SL(j)][c + 1], A[SL(K - 1 - j)][c],
               A[SL(K - 1 - j)][c + 1], A[SL(j + 1)][c], A[SL(j + 1)][c + 1],
               A[SL(K - 2 - j)][c], A[SL(K - 2 - j)][c + 1], t[c], t[c + 1],
               taps.k[j], taps.k[j + 1]);
        if ((R - 1) % 2 == 1)
          col1(A[SL(R - 1)][c], A[SL(R - 1)][c + 1], A[SL(R + 1)][c],
               A[SL(R + 1)][c + 1], t[c], t[c + 1], taps.k[R - 1]);
      }
#undef SL
      asm volatile("" ::"v"(A[i][0]), "v"(A[i][1]));
      PHASE(3)
      const int o = yy - R;
      if ((o >= y0) && (o < y1) && col_ok && (!NOMEM || never))
        *reinterpret_cast<vec*>(dst + size_t(o) * w + col) =
            make_float2(A[i][0], A[i][1]);
    }
    if (K % PF != 0)
    {
      vec tm[PF];
      float th[PF];
#pragma unroll
      for (int q = 0; q < PF; ++q)
      {
        tm[q] = pm[(K + q) % PF];
        th[q] = phv[(K + q) % PF];
      }
#pragma unroll
      for (int q = 0; q < PF; ++q)
      {
        pm[q] = tm[q];
        phv[q] = th[q];
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef VF
  if (lane == 0)
  {
    const size_t id = size_t(blockIdx.y) * gridDim.x + blockIdx.x;
#pragma unroll
    for (int q = 0; q < 5; ++q)
      prof[id * 6 + q] = ph[q];
    prof[id * 6 + 5] = (unsigned long long) T;
  }
#undef PHASE
}

static Taps make_taps(int R)
{
  Taps t;
  t.size = 2 * R + 1;
  const float sigma = R / 4.f;
  float sum = 0.f;
  for (int i = 0; i < t.size; ++i)
  {
    const float x = float(i - R);
    t.k[i] = std::exp(-x * x / (2 * sigma * sigma));
    sum += t.k[i];
  }
  for (int i = 0; i < t.size; ++i)
    t.k[i] /= sum;
  return t;
}

static int g_waves = 3072, g_minrows = 4;

template <int R, int PF, int MODE, int CPL = 4>
void run(const float* src, float* dst, int w, int h, int batch, const char* tag)
{
  {
    char name[128];
    snprintf(name, sizeof name, "R=%d CPL=%d %s", R, CPL, tag);
    if (getenv("ONLY") && strcmp(getenv("ONLY"), name) != 0)
      return;
  }
  const int nstrips = (w + 64 * CPL - 1) / (64 * CPL);
  int nseg = (g_waves + nstrips * batch - 1) / (nstrips * batch);
  const int min_rows = std::max(32, g_minrows * R);
  nseg = std::max(1, std::min(nseg, (h + min_rows - 1) / min_rows));
  const int seg_rows = (h + nseg - 1) / nseg;
  nseg = (h + seg_rows - 1) / seg_rows;
  const dim3 grid(nstrips * nseg, batch);
  const Taps taps = make_taps(R);
  const size_t stride = size_t(w) * h;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep)
  {
    hipEventRecord(a);
    march<R, PF, MODE, CPL><<<grid, 64>>>(src, stride, dst, stride, w, h, seg_rows,
                                     nstrips, taps, 0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    best = std::min(best, ms);
  }
  const double bytes = 8.0 * stride * batch;
  const double rows = double(seg_rows + 2 * R) / seg_rows;
  const double pairs = double(stride) * batch * rows * 2 * (2 * R + 1);
  printf("R=%2d PF=%d CPL=%d %-28s seg_rows=%3d grid=%5d x %d : %7.1f us  %5.2f TB/s  "
         "%5.1f T mul-add/s\n",
         R, PF, CPL, tag, seg_rows, nstrips * nseg, batch, best * 1e3,
         bytes / best / 1e9, pairs / best / 1e9);
}

template <int R, int PF, int CPL, bool EARLY, int NOMEM>
void run2(const float* src, float* dst, int w, int h, int batch, const char* tag)
{
  {
    char name[128];
    snprintf(name, sizeof name, "R=%d CPL=%d %s", R, CPL, tag);
    if (getenv("ONLY") && strcmp(getenv("ONLY"), name) != 0)
      return;
  }
  const int nstrips = (w + 64 * CPL - 1) / (64 * CPL);
  int nseg = (g_waves + nstrips * batch - 1) / (nstrips * batch);
  const int min_rows = std::max(32, g_minrows * R);
  nseg = std::max(1, std::min(nseg, (h + min_rows - 1) / min_rows));
  const int seg_rows = (h + nseg - 1) / nseg;
  nseg = (h + seg_rows - 1) / seg_rows;
  const dim3 grid(nstrips * nseg, batch);
  const Taps taps = make_taps(R);
  const size_t stride = size_t(w) * h;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep)
  {
    hipEventRecord(a);
    march2<R, PF, CPL, EARLY, NOMEM><<<grid, 64>>>(src, stride, dst, stride, w, h,
                                                   seg_rows, nstrips, taps, 0,
                                                   nullptr);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    best = std::min(best, ms);
  }
  if (getenv("TRACE"))
  {
    const size_t nb = size_t(grid.x) * grid.y;
    unsigned long long* tr;
    hipMalloc(&tr, nb * 24);
    march2<R, PF, CPL, EARLY, NOMEM><<<grid, 64>>>(src, stride, dst, stride, w, h,
                                                   seg_rows, nstrips, taps, 0, tr);
    std::vector<unsigned long long> ht(nb * 3);
    hipMemcpy(ht.data(), tr, nb * 24, hipMemcpyDeviceToHost);
    FILE* f = fopen(getenv("TRACE"), "w");
    for (size_t i = 0; i < nb; ++i)
      fprintf(f, "%llx %llu %llu\n", ht[3 * i], ht[3 * i + 1], ht[3 * i + 2]);
    fclose(f);
    hipFree(tr);
  }
  const double bytes = 8.0 * stride * batch;
  printf("R=%2d PF=%d CPL=%d %-28s seg_rows=%3d grid=%5d x %d : %7.1f us  %5.2f TB/s\n",
         R, PF, CPL, tag, seg_rows, nstrips * nseg, batch, best * 1e3,
         bytes / best / 1e9);
}

template <int R, int PF, int CPL, int NOMEM, bool TAPV = false>
void run3(const float* src, float* dst, float* ref, int w, int h, int batch,
          const char* tag)
{
  {
    char name[128];
    snprintf(name, sizeof name, "R=%d CPL=%d %s", R, CPL, tag);
    if (getenv("ONLY") && strcmp(getenv("ONLY"), name) != 0)
      return;
  }
  const int nstrips = (w + 64 * CPL - 1) / (64 * CPL);
  int nseg = (g_waves + nstrips * batch - 1) / (nstrips * batch);
  const int min_rows = std::max(32, g_minrows * R);
  nseg = std::max(1, std::min(nseg, (h + min_rows - 1) / min_rows));
  const int seg_rows = (h + nseg - 1) / nseg;
  nseg = (h + seg_rows - 1) / seg_rows;
  const dim3 grid(nstrips * nseg, batch);
  const Taps taps = make_taps(R);
  const size_t stride = size_t(w) * h;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep)
  {
    hipEventRecord(a);
    if (TAPV)
      march3v<R, PF, CPL, NOMEM><<<grid, 64>>>(src, stride, dst, stride, w, h,
                                               seg_rows, nstrips, taps, 0);
    else
      march3<R, PF, CPL, NOMEM><<<grid, 64>>>(src, stride, dst, stride, w, h,
                                              seg_rows, nstrips, taps, 0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    best = std::min(best, ms);
  }
  long bad = -1;
  if (ref && (NOMEM == 0 || NOMEM == 4))
  {
    // bit-exactness against the first-generation kernel
    const int ns1 = (w + 255) / 256;
    int nsg = (g_waves + ns1 * batch - 1) / (ns1 * batch);
    nsg = std::max(1, std::min(nsg, (h + min_rows - 1) / min_rows));
    const int sr = (h + nsg - 1) / nsg;
    nsg = (h + sr - 1) / sr;
    march<R, PF, 0, 4><<<dim3(ns1 * nsg, 1), 64>>>(src, stride, ref, stride, w, h,
                                                   sr, ns1, taps, 0);
    std::vector<float> x(stride), y(stride);
    hipMemcpy(x.data(), dst, stride * 4, hipMemcpyDeviceToHost);
    hipMemcpy(y.data(), ref, stride * 4, hipMemcpyDeviceToHost);
    bad = 0;
    for (size_t i = 0; i < stride; ++i)
      bad += memcmp(&x[i], &y[i], 4) != 0;
  }
  const double bytes = 8.0 * stride * batch;
  printf("R=%2d PF=%d CPL=%d %-28s seg_rows=%3d grid=%5d x %d : %7.1f us  %5.2f TB/s  mismatches=%ld\n",
         R, PF, CPL, tag, seg_rows, nstrips * nseg, batch, best * 1e3,
         bytes / best / 1e9, bad);
}

template <int R, int PF, int NOMEM, int LDSPAD, int V4MODE = 3>
void run4(const float* src, float* dst, float* ref, int w, int h, int batch,
          const char* tag)
{
  constexpr int CPL = 2;
  {
    char name[128];
    snprintf(name, sizeof name, "R=%d CPL=%d %s", R, CPL, tag);
    if (getenv("ONLY") && strcmp(getenv("ONLY"), name) != 0)
      return;
  }
  const int nstrips = (w + 64 * CPL - 1) / (64 * CPL);
  int nseg = (g_waves + nstrips * batch - 1) / (nstrips * batch);
  const int min_rows = std::max(32, g_minrows * R);
  nseg = std::max(1, std::min(nseg, (h + min_rows - 1) / min_rows));
  const int seg_rows = (h + nseg - 1) / nseg;
  nseg = (h + seg_rows - 1) / seg_rows;
  const dim3 grid(nstrips * nseg, batch);
  const Taps taps = make_taps(R);
  const size_t stride = size_t(w) * h;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep)
  {
    hipEventRecord(a);
    march4<R, PF, NOMEM, LDSPAD, V4MODE><<<grid, 64>>>(src, stride, dst, stride, w, h,
                                               seg_rows, nstrips, taps, 0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    best = std::min(best, ms);
  }
  long bad = -1;
  if (ref && !NOMEM)
  {
    const int ns1 = (w + 255) / 256;
    int nsg = (3072 + ns1 * batch - 1) / (ns1 * batch);
    nsg = std::max(1, std::min(nsg, (h + min_rows - 1) / min_rows));
    const int sr = (h + nsg - 1) / nsg;
    nsg = (h + sr - 1) / sr;
    march<R, PF <= 4 ? PF : 4, 0, 4><<<dim3(ns1 * nsg, 1), 64>>>(
        src, stride, ref, stride, w, h, sr, ns1, taps, 0);
    std::vector<float> x(stride), y(stride);
    hipMemcpy(x.data(), dst, stride * 4, hipMemcpyDeviceToHost);
    hipMemcpy(y.data(), ref, stride * 4, hipMemcpyDeviceToHost);
    bad = 0;
    for (size_t i = 0; i < stride; ++i)
    {
      const bool m = memcmp(&x[i], &y[i], 4) != 0;
      if (m && bad < 12 && getenv("SHOW"))
        printf("  (%zu,%zu) got %.9g want %.9g\n", i % w, i / w, x[i], y[i]);
      bad += m;
    }
  }
  const double bytes = 8.0 * stride * batch;
  printf("R=%2d PF=%d CPL=%d %-28s seg_rows=%3d grid=%5d x %d : %7.1f us  %5.2f TB/s  mismatches=%ld\n",
         R, PF, CPL, tag, seg_rows, nstrips * nseg, batch, best * 1e3,
         bytes / best / 1e9, bad);
}

template <int R, int V4MODE>
void run4p(const float* src, float* dst, int w, int h, int batch)
{
  constexpr int CPL = 2;
  const int nstrips = (w + 64 * CPL - 1) / (64 * CPL);
  int nseg = (g_waves + nstrips * batch - 1) / (nstrips * batch);
  const int min_rows = std::max(32, g_minrows * R);
  nseg = std::max(1, std::min(nseg, (h + min_rows - 1) / min_rows));
  const int seg_rows = (h + nseg - 1) / nseg;
  nseg = (h + seg_rows - 1) / seg_rows;
  const dim3 grid(nstrips * nseg, batch);
  const Taps taps = make_taps(R);
  const size_t stride = size_t(w) * h;
  const size_t nb = size_t(grid.x) * grid.y;
  unsigned long long* prof;
  hipMalloc(&prof, nb * 48);
  for (int rep = 0; rep < 2; ++rep)
    march4p<R, 4, 0, 0, V4MODE><<<grid, 64>>>(src, stride, dst, stride, w, h,
                                              seg_rows, nstrips, taps, 0, prof);
  hipDeviceSynchronize();
  std::vector<unsigned long long> hp(nb * 6);
  hipMemcpy(hp.data(), prof, nb * 48, hipMemcpyDeviceToHost);
  double sum[5] = {0, 0, 0, 0, 0}, steps = 0;
  for (size_t i = 0; i < nb; ++i)
  {
    for (int q = 0; q < 5; ++q)
      sum[q] += double(hp[i * 6 + q]);
    steps += double(hp[i * 6 + 5]);
  }
  printf("R=%2d v4 mode %d waves=%zu: cycles per step and wave: window wait %.0f  row pass %.0f  "
         "stage+issue reads %.0f  column pass %.0f  store+loop %.0f  (total %.0f)\n",
         R, V4MODE, nb, sum[0] / steps, sum[1] / steps, sum[2] / steps, sum[3] / steps,
         sum[4] / steps, (sum[0] + sum[1] + sum[2] + sum[3] + sum[4]) / steps);
  hipFree(prof);
}

template <int R, int PF, int CPL>
void run3p(const float* src, float* dst, int w, int h, int batch)
{
  const int nstrips = (w + 64 * CPL - 1) / (64 * CPL);
  int nseg = (g_waves + nstrips * batch - 1) / (nstrips * batch);
  const int min_rows = std::max(32, g_minrows * R);
  nseg = std::max(1, std::min(nseg, (h + min_rows - 1) / min_rows));
  const int seg_rows = (h + nseg - 1) / nseg;
  nseg = (h + seg_rows - 1) / seg_rows;
  const dim3 grid(nstrips * nseg, batch);
  const Taps taps = make_taps(R);
  const size_t stride = size_t(w) * h;
  const size_t nb = size_t(grid.x) * grid.y;
  unsigned long long* prof;
  hipMalloc(&prof, nb * 48);
  for (int rep = 0; rep < 2; ++rep)
    march3p<R, PF, CPL, 0><<<grid, 64>>>(src, stride, dst, stride, w, h, seg_rows,
                                         nstrips, taps, 0, prof);
  hipDeviceSynchronize();
  std::vector<unsigned long long> hp(nb * 6);
  hipMemcpy(hp.data(), prof, nb * 48, hipMemcpyDeviceToHost);
  double sum[5] = {0, 0, 0, 0, 0}, steps = 0;
  for (size_t i = 0; i < nb; ++i)
  {
    for (int q = 0; q < 5; ++q)
      sum[q] += double(hp[i * 6 + q]);
    steps += double(hp[i * 6 + 5]);
  }
  printf("R=%2d CPL=%d waves=%zu: cycles per step and wave: vmwait %.0f  lds stage+read %.0f  "
         "row pass %.0f  column pass %.0f  store+loop %.0f  (total %.0f)\n",
         R, CPL, nb, sum[0] / steps, sum[1] / steps, sum[2] / steps, sum[3] / steps,
         sum[4] / steps, (sum[0] + sum[1] + sum[2] + sum[3] + sum[4]) / steps);
  hipFree(prof);
}

template <int R, int PF>
void sweep3(const float* src, float* dst, float* ref, int w, int h, int batch)
{
  run<R, PF, 0>(src, dst, w, h, batch, "full");
  run3<R, PF, 4, 0>(src, dst, ref, w, h, batch, "v3");
  run3<R, PF, 4, 1>(src, dst, ref, w, h, batch, "v3 nomem");
  run3<R, 4, 2, 0>(src, dst, ref, w, h, batch, "v3");
  run3<R, 4, 2, 1>(src, dst, ref, w, h, batch, "v3 nomem");
  run3<R, 4, 2, 2>(src, dst, ref, w, h, batch, "v3 nolds");
  run3<R, 4, 2, 4>(src, dst, ref, w, h, batch, "v3 brfree");
  run3<R, 4, 2, 0, true>(src, dst, ref, w, h, batch, "v3 tapv");
  run3<R, 4, 2, 1, true>(src, dst, ref, w, h, batch, "v3 tapv nomem");
  run3<R, 4, 2, 2, true>(src, dst, ref, w, h, batch, "v3 tapv nolds");
  run3<R, 6, 2, 0>(src, dst, ref, w, h, batch, "v3 pf6");
  run4<R, 4, 0, 0>(src, dst, ref, w, h, batch, "v4");
  run4<R, 4, 1, 0>(src, dst, ref, w, h, batch, "v4 nomem");
  run4<R, 4, 0, 0, 0>(src, dst, ref, w, h, batch, "v4 m0");
  run4<R, 4, 0, 0, 1>(src, dst, ref, w, h, batch, "v4 m1");
  run4<R, 4, 0, 0, 2>(src, dst, ref, w, h, batch, "v4 m2");
  run4<R, 4, 0, 10240>(src, dst, ref, w, h, batch, "v4 occ4");
  run4<R, 4, 0, 6800>(src, dst, ref, w, h, batch, "v4 occ6");
  run4<R, 2, 0, 6800>(src, dst, ref, w, h, batch, "v4 occ6 pf2");
}

template <int R, int PF>
void sweep2(const float* src, float* dst, int w, int h, int batch)
{
  run<R, PF, 0>(src, dst, w, h, batch, "full");
  run2<R, PF, 4, false, 0>(src, dst, w, h, batch, "v2");
  run2<R, PF, 4, true, 0>(src, dst, w, h, batch, "v2 early");
  run2<R, PF, 4, true, 1>(src, dst, w, h, batch, "v2 early nomem");
  run2<R, 4, 2, false, 0>(src, dst, w, h, batch, "v2");
  run2<R, 4, 2, true, 0>(src, dst, w, h, batch, "v2 early");
  run2<R, 4, 2, true, 1>(src, dst, w, h, batch, "v2 early nomem");
  run2<R, 6, 2, true, 0>(src, dst, w, h, batch, "v2 early pf6");
}

template <int R, int PF>
void sweep(const float* src, float* dst, int w, int h, int batch)
{
  run<R, PF, 0>(src, dst, w, h, batch, "full");
  run<R, PF, 1>(src, dst, w, h, batch, "no loads");
  run<R, PF, 2>(src, dst, w, h, batch, "no stores");
  run<R, PF, 3>(src, dst, w, h, batch, "no loads/stores");
  run<R, PF, 7>(src, dst, w, h, batch, "no loads/stores/LDS");
  run<R, PF, 24>(src, dst, w, h, batch, "symcol+sched_barrier");
  run<R, PF, 27>(src, dst, w, h, batch, "symcol+sb, no loads/stores");
  run<R, PF, 8 + 32>(src, dst, w, h, batch, "symcol asm");
  run<R, PF, 8 + 32 + 3>(src, dst, w, h, batch, "symcol asm, no loads/stores");
  run<R, 4, 8 + 32, 2>(src, dst, w, h, batch, "symcol asm");
  run<R, 4, 8 + 32 + 3, 2>(src, dst, w, h, batch, "symcol asm, no loads/stores");
  run<R, 4, 0, 2>(src, dst, w, h, batch, "full");
  run<R, 4, 3, 2>(src, dst, w, h, batch, "no loads/stores");
  run<R, 4, 24, 2>(src, dst, w, h, batch, "symcol+sb");
  run<R, 4, 27, 2>(src, dst, w, h, batch, "symcol+sb, no loads/stores");
}

int main(int argc, char** argv)
{
  const int batch = argc > 1 ? atoi(argv[1]) : 64;
  const int w = argc > 2 ? atoi(argv[2]) : 1920;
  const int h = argc > 3 ? atoi(argv[3]) : 1080;
  if (getenv("WAVES"))
    g_waves = atoi(getenv("WAVES"));
  if (getenv("MINROWS"))
    g_minrows = atoi(getenv("MINROWS"));
  const size_t n = size_t(w) * h * batch;
  float *src, *dst;
  hipMalloc(&src, n * 4);
  hipMalloc(&dst, n * 4);
  std::vector<float> hsrc(size_t(w) * h);
  for (size_t i = 0; i < hsrc.size(); ++i)
    hsrc[i] = float((i * 2654435761u) >> 8 & 0xffff) / 65536.f;
  for (int b = 0; b < batch; ++b)
    hipMemcpy(src + size_t(b) * w * h, hsrc.data(), hsrc.size() * 4,
              hipMemcpyHostToDevice);
  if (getenv("SWEEP1"))
  {
    sweep<12, 2>(src, dst, w, h, batch);
    sweep<10, 3>(src, dst, w, h, batch);
    sweep<8, 4>(src, dst, w, h, batch);
    sweep<5, 4>(src, dst, w, h, batch);
  }
  if (getenv("PHASES"))
  {
    run4p<12, 2>(src, dst, w, h, batch);
    run4p<12, 0>(src, dst, w, h, batch);
    run3p<12, 4, 2>(src, dst, w, h, batch);
    run3p<10, 4, 2>(src, dst, w, h, batch);
    run3p<8, 4, 2>(src, dst, w, h, batch);
    run3p<5, 4, 2>(src, dst, w, h, batch);
    return 0;
  }
  if (!getenv("SWEEP2"))
  {
    float* ref;
    hipMalloc(&ref, size_t(w) * h * 4);
    sweep3<12, 2>(src, dst, ref, w, h, batch);
    sweep3<10, 3>(src, dst, ref, w, h, batch);
    sweep3<8, 4>(src, dst, ref, w, h, batch);
    sweep3<6, 4>(src, dst, ref, w, h, batch);
    sweep3<5, 4>(src, dst, ref, w, h, batch);
    return 0;
  }
  sweep2<12, 2>(src, dst, w, h, batch);
  sweep2<10, 3>(src, dst, w, h, batch);
  sweep2<8, 4>(src, dst, w, h, batch);
  sweep2<6, 4>(src, dst, w, h, batch);
  sweep2<5, 4>(src, dst, w, h, batch);
  return 0;
}
