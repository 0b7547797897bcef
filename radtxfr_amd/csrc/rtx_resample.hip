// smooth / reduceResolution (radiative_transfer.py:1266-1350): the post-processing step right after compute_TUD in
// the reference's main caller (Generate_LWIR_TUD.py:82-85,124-126). Two streaming kernels, fp64 like the reference.
#include <string.h>

#include <mutex>
#include <vector>

#include "rtx_common.h"

#define FIR_MAX_TAPS 8192
#define FIR_BLOCK 256
#define FIR_PER_THREAD 4
#define FIR_TILE (FIR_BLOCK * FIR_PER_THREAD)
#define FIR_CHUNK 512  // taps staged per pass: LDS = (FIR_TILE + FIR_CHUNK) * 8 + FIR_CHUNK * 8 = 16 KiB

// out[r][i] = sum_k taps[k] * in[r][R(i + k - centre)]. A workgroup owns FIR_TILE consecutive outputs of one row and
// walks the taps in chunks: the input span of a chunk and the chunk's taps are staged in LDS (the reflection is
// applied while staging), each thread keeps FIR_PER_THREAD running sums and slides a register window over the span:
// per tap one broadcast LDS read (the tap) and one LDS read (the newest sample) feed FIR_PER_THREAD fp64 FMAs.
template <typename T>
__global__ __launch_bounds__(FIR_BLOCK) void fir_reflect_kernel(const T* __restrict__ in, long long ld_in, long long n,
                                                                const double* __restrict__ taps, int n_taps, int centre,
                                                                double* __restrict__ out, long long ld_out) {
  __shared__ double s_x[FIR_TILE + FIR_CHUNK];
  __shared__ double s_t[FIR_CHUNK];
  const int row = blockIdx.y;
  const long long i0 = (long long)blockIdx.x * FIR_TILE;
  const T* __restrict__ src = in + (size_t)row * (size_t)ld_in;
  double acc[FIR_PER_THREAD];
#pragma unroll
  for (int j = 0; j < FIR_PER_THREAD; ++j) acc[j] = 0.0;
  const int t0 = threadIdx.x * FIR_PER_THREAD;
  for (int k0 = 0; k0 < n_taps; k0 += FIR_CHUNK) {
    const int kc = n_taps - k0 < FIR_CHUNK ? n_taps - k0 : FIR_CHUNK;
    __syncthreads();
    // samples i0 + k0 - centre + [0, FIR_TILE + kc - 1)
    for (int s = threadIdx.x; s < FIR_TILE + kc - 1; s += FIR_BLOCK) {
      long long jdx = i0 + k0 - centre + s;
      if (jdx < 0) jdx = -jdx;
      if (jdx >= n) jdx = 2 * (n - 1) - jdx;
      jdx = jdx < 0 ? 0 : (jdx >= n ? n - 1 : jdx);  // only outputs past the end of the row can get here
      s_x[s] = (double)src[jdx];
    }
    for (int s = threadIdx.x; s < kc; s += FIR_BLOCK) s_t[s] = taps[k0 + s];
    __syncthreads();
    // FIR_PER_THREAD taps at a time: the 2 FIR_PER_THREAD - 1 samples they touch are read once into registers and every
    // (tap, output) pair indexes them statically -- no register window to shift per tap
    int k = 0;
    for (; k + FIR_PER_THREAD <= kc; k += FIR_PER_THREAD) {
      double w[2 * FIR_PER_THREAD - 1], tk[FIR_PER_THREAD];
#pragma unroll
      for (int j = 0; j < 2 * FIR_PER_THREAD - 1; ++j) w[j] = s_x[t0 + k + j];
#pragma unroll
      for (int u = 0; u < FIR_PER_THREAD; ++u) tk[u] = s_t[k + u];
#pragma unroll
      for (int u = 0; u < FIR_PER_THREAD; ++u)
#pragma unroll
        for (int j = 0; j < FIR_PER_THREAD; ++j) acc[j] = fma(tk[u], w[u + j], acc[j]);
    }
    for (; k < kc; ++k) {
      const double tk = s_t[k];
#pragma unroll
      for (int j = 0; j < FIR_PER_THREAD; ++j) acc[j] = fma(tk, s_x[t0 + k + j], acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < FIR_PER_THREAD; ++j) {
    const long long i = i0 + t0 + j;
    if (i < n) out[(size_t)row * (size_t)ld_out + (size_t)i] = acc[j];
  }
}

extern "C" int rtx_fir_reflect(const void* in, int in_is_f64, int64_t ld_in, int n_rows, int64_t n, const double* taps_h,
                               int n_taps, int centre, double* out, int64_t ld_out, void* stream) {
  if (!in || !taps_h || !out) RTX_FAIL("a required pointer is NULL");
  if (n_rows < 1 || n < 1) RTX_FAIL("n_rows=%d n=%lld", n_rows, (long long)n);
  if (n_taps < 1 || n_taps > FIR_MAX_TAPS) RTX_FAIL("n_taps=%d outside [1,%d]", n_taps, FIR_MAX_TAPS);
  if (n_taps - 1 > n) RTX_FAIL("window of %d taps is longer than the %lld samples (one reflection only)", n_taps, (long long)n);
  if (centre < 0 || centre >= n_taps) RTX_FAIL("centre=%d outside the %d taps", centre, n_taps);
  if (ld_in < n || ld_out < n) RTX_FAIL("leading dimension smaller than the row");
  if (n_rows > 65535) RTX_FAIL("n_rows=%d too large", n_rows);
  hipStream_t st = (hipStream_t)stream;
  // device copies of the windows seen so far (a caller smooths every spectrum of a run with the same one or two windows):
  // a hit costs a memcmp; only a new window is uploaded (synchronously: taps_h may be a temporary of the caller)
  // The lock is held until the kernel that reads the taps is enqueued: an eviction by another host thread (hipFree, which
  // waits for the device) can then only come after the launch, never between the look-up and it.
  double* d_taps = nullptr;
  struct Win { int dev; std::vector<double> h; double* d; };
  static std::mutex mu;
  static std::vector<Win> cache;
  std::lock_guard<std::mutex> lock(mu);
  {
    int dev = 0;
    RTX_HIP(hipGetDevice(&dev));
    for (const Win& w : cache)
      if (w.dev == dev && (int)w.h.size() == n_taps && memcmp(w.h.data(), taps_h, (size_t)n_taps * sizeof(double)) == 0) { d_taps = w.d; break; }
    if (!d_taps) {
      if (cache.size() >= 64) {  // bounded: forget the oldest
        (void)hipFree(cache.front().d);
        cache.erase(cache.begin());
      }
      Win w;
      w.dev = dev;
      w.h.assign(taps_h, taps_h + n_taps);
      RTX_HIP(hipMalloc((void**)&w.d, (size_t)n_taps * sizeof(double)));
      RTX_HIP(hipMemcpy(w.d, taps_h, (size_t)n_taps * sizeof(double), hipMemcpyHostToDevice));
      d_taps = w.d;
      cache.push_back(std::move(w));
    }
  }
  const dim3 grid((unsigned)((n + FIR_TILE - 1) / FIR_TILE), (unsigned)n_rows);
  if (in_is_f64)
    hipLaunchKernelGGL(fir_reflect_kernel<double>, grid, dim3(FIR_BLOCK), 0, st, (const double*)in, (long long)ld_in, (long long)n, d_taps,
                       n_taps, centre, out, (long long)ld_out);
  else
    hipLaunchKernelGGL(fir_reflect_kernel<float>, grid, dim3(FIR_BLOCK), 0, st, (const float*)in, (long long)ld_in, (long long)n, d_taps,
                       n_taps, centre, out, (long long)ld_out);
  RTX_LAUNCH_CHECK();
  return 0;
}

// Cardinal cubic spline on a uniform axis. With B-spline coefficients c = prefilter(y), prefilter taps
// sqrt(3) z^|k|, z = sqrt(3) - 2 (the inverse of the [1 4 1]/6 sampling of the cubic B-spline), the spline at
// t = i + f is  sum_j b_j(f) c[i - 1 + j]. One thread per (row, output point); 4 x 81 cached loads.
#define SPL_K 40
#define SPL_EDGE 24
__global__ __launch_bounds__(256) void cubic_resample_kernel(const double* __restrict__ y, long long ld, long long n, double x0, double inv_h,
                                                            const double* __restrict__ x_out, long long n_out,
                                                            double* __restrict__ out, long long ld_out, int* __restrict__ bad) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_out) return;
  const int row = blockIdx.y;
  const double t = (x_out[q] - x0) * inv_h;
  long long i = (long long)floor(t);
  if (!(i - 1 - SPL_EDGE >= 0 && i + 2 + SPL_EDGE <= n - 1)) {  // also catches NaN
    if (bad) *bad = 1;
    else out[(size_t)blockIdx.y * (size_t)ld_out + (size_t)q] = __builtin_nan("");  // unchecked form: the point is marked, not reported
    return;
  }
  const double f = t - (double)i;
  const double* __restrict__ src = y + (size_t)row * (size_t)ld;
  const long long a0 = i - 1 - SPL_K;
  // c[j] = sum_{k=-K..K} pre[|k|] y[i-1+j+k], j = 0..3: one sweep over the 2K+4 samples
  const double z = -0.2679491924311227;  // sqrt(3) - 2
  double pre[SPL_K + 1];
  pre[0] = 1.7320508075688772;
#pragma unroll
  for (int k = 1; k <= SPL_K; ++k) pre[k] = pre[k - 1] * z;
  double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
#pragma unroll
  for (int s = 0; s < 2 * SPL_K + 4; ++s) {
    // sample s contributes to c[j] with k = s - SPL_K - j; samples beyond an end of the row are dropped: the
    // output is >= SPL_EDGE knots inside, where their weight is below z^SPL_EDGE = 2e-14
    const long long a = a0 + s;
    const double v = (a >= 0 && a < n) ? src[a] : 0.0;
    const int k0 = s - SPL_K, k1 = k0 - 1, k2 = k0 - 2, k3 = k0 - 3;
    if (k0 >= -SPL_K && k0 <= SPL_K) c0 = fma(pre[k0 < 0 ? -k0 : k0], v, c0);
    if (k1 >= -SPL_K && k1 <= SPL_K) c1 = fma(pre[k1 < 0 ? -k1 : k1], v, c1);
    if (k2 >= -SPL_K && k2 <= SPL_K) c2 = fma(pre[k2 < 0 ? -k2 : k2], v, c2);
    if (k3 >= -SPL_K && k3 <= SPL_K) c3 = fma(pre[k3 < 0 ? -k3 : k3], v, c3);
  }
  const double g = 1.0 - f;
  const double b0 = g * g * g, b3 = f * f * f;
  const double b1 = fma(fma(3.0, f, -6.0) * f, f, 4.0);
  const double b2 = fma(fma(3.0, g, -6.0) * g, g, 4.0);
  out[(size_t)row * (size_t)ld_out + (size_t)q] = (b0 * c0 + b1 * c1 + b2 * c2 + b3 * c3) * (1.0 / 6.0);
}

// The same without the device-to-host read-back of the range flag (which synchronises the stream): for callers that have
// checked on the host that every x_out lies inside the supported range -- a stream of spectra resampled onto one output
// axis (rt.compute_TUD_batch(reduce=...)). An abscissa outside the range gives NaN instead of an error.
extern "C" int rtx_cubic_resample_unchecked(const double* Ysm, int64_t ld, int n_rows, int64_t n, double x0, double h,
                                            const double* x_out, int64_t n_out, double* out, int64_t ld_out, void* stream) {
  if (!Ysm || !x_out || !out) RTX_FAIL("a required pointer is NULL");
  if (n_rows < 1 || n_rows > 65535) RTX_FAIL("n_rows=%d", n_rows);
  if (n < 2 * SPL_EDGE + 8) RTX_FAIL("n=%lld: the spline needs at least %d samples", (long long)n, 2 * SPL_EDGE + 8);
  if (!(h > 0.0)) RTX_FAIL("h=%g", h);
  if (ld < n || ld_out < n_out) RTX_FAIL("leading dimension smaller than the row");
  if (n_out == 0) return 0;
  hipLaunchKernelGGL(cubic_resample_kernel, dim3((unsigned)((n_out + 255) / 256), (unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, Ysm,
                     (long long)ld, (long long)n, x0, 1.0 / h, x_out, (long long)n_out, out, (long long)ld_out, (int*)nullptr);
  RTX_LAUNCH_CHECK();
  return 0;
}

extern "C" int rtx_cubic_resample(const double* Ysm, int64_t ld, int n_rows, int64_t n, double x0, double h, const double* x_out,
                                  int64_t n_out, double* out, int64_t ld_out, void* stream) {
  if (!Ysm || !x_out || !out) RTX_FAIL("a required pointer is NULL");
  if (n_rows < 1 || n_rows > 65535) RTX_FAIL("n_rows=%d", n_rows);
  if (n < 2 * SPL_EDGE + 8) RTX_FAIL("n=%lld: the spline needs at least %d samples", (long long)n, 2 * SPL_EDGE + 8);
  if (!(h > 0.0)) RTX_FAIL("h=%g", h);
  if (ld < n || ld_out < n_out) RTX_FAIL("leading dimension smaller than the row");
  if (n_out == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  int* d_bad = nullptr;
  RTX_HIP(hipMallocAsync((void**)&d_bad, sizeof(int), st));
  RTX_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), st));
  hipLaunchKernelGGL(cubic_resample_kernel, dim3((unsigned)((n_out + 255) / 256), (unsigned)n_rows), dim3(256), 0, st, Ysm, (long long)ld,
                     (long long)n, x0, 1.0 / h, x_out, (long long)n_out, out, (long long)ld_out, d_bad);
  RTX_LAUNCH_CHECK();
  int bad = 0;
  RTX_HIP(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, st));
  RTX_HIP(hipStreamSynchronize(st));
  RTX_HIP(hipFreeAsync(d_bad, st));
  if (bad) RTX_FAIL("an output abscissa lies within %d samples of an end of the input axis (or is NaN)", SPL_EDGE + 2);
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// End regions of reduceResolution's spline (radiative_transfer.py:1327-1350: scipy's interp1d(kind='cubic') = the
// not-a-knot cubic spline through ALL smoothed samples). Away from the ends that spline is the cardinal spline on uniform
// knots (rtx_cubic_resample); within a window length of an end it is not: the reference smooths the AXIS as well, and the
// reflection padding of smooth() bends the first and last ceil(window/2) knots off the uniform grid, and the not-a-knot
// end condition acts there. Both fade as 0.268^k with the distance k in knots, so the spline on the first (last) m knots,
// with the true not-a-knot condition at the end of the axis and a natural condition at the cut -- 24 or more knots beyond
// the last output -- is the reference's spline there to 0.268^24 = 2e-14. One workgroup per row: the local tridiagonal
// system (second-derivative form on the true, non-uniform knots) is solved by one thread in LDS, then every thread
// evaluates outputs. `high_end`: the window is the LAST m samples; it is mirrored on load so that the true end is index 0.
#define SPL_END_MAX 768
__global__ __launch_bounds__(256) void cubic_end_kernel(const double* __restrict__ y, long long ld, long long i_first, int m, int high_end,
                                                        const double* __restrict__ knots, const double* __restrict__ x_out, long long n_out,
                                                        double* __restrict__ out, long long ld_out) {
  __shared__ double s_x[SPL_END_MAX], s_y[SPL_END_MAX], s_m[SPL_END_MAX], s_c[SPL_END_MAX], s_d[SPL_END_MAX];
  const int row = blockIdx.x;
  const double* __restrict__ src = y + (size_t)row * (size_t)ld + (size_t)i_first;
  for (int j = threadIdx.x; j < m; j += blockDim.x) {
    const int jj = high_end ? m - 1 - j : j;
    s_x[j] = high_end ? -knots[jj] : knots[jj];
    s_y[j] = src[jj];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // unknowns M_1 .. M_{m-2}; M_0 from the not-a-knot condition, M_{m-1} = 0 (natural, at the cut)
    const double h0 = s_x[1] - s_x[0], h1 = s_x[2] - s_x[1];
    double dprev = (s_y[1] - s_y[0]) / h0;
    // forward sweep (Thomas): row i: a_i M_{i-1} + b_i M_i + c_i M_{i+1} = r_i
    double cp = 0.0, dp = 0.0;  // modified coefficients of the previous row
    for (int i = 1; i <= m - 2; ++i) {
      const double hl = s_x[i] - s_x[i - 1], hr = s_x[i + 1] - s_x[i];
      const double dcur = (s_y[i + 1] - s_y[i]) / hr;
      double a = hl, b = 2.0 * (hl + hr), c = hr;
      const double r = 6.0 * (dcur - dprev);
      if (i == 1) {  // M_0 = (1 + h0/h1) M_1 - (h0/h1) M_2 folded in
        b += h0 * (1.0 + h0 / h1);
        c -= h0 * h0 / h1;
        a = 0.0;
      }
      if (i == m - 2) c = 0.0;  // M_{m-1} = 0
      const double den = b - a * cp;
      cp = c / den;
      dp = (r - a * dp) / den;
      s_c[i] = cp;
      s_d[i] = dp;
      dprev = dcur;
    }
    s_m[m - 1] = 0.0;
    double mn = 0.0;
    for (int i = m - 2; i >= 1; --i) {
      mn = s_d[i] - s_c[i] * mn;
      s_m[i] = mn;
    }
    s_m[0] = (1.0 + h0 / h1) * s_m[1] - (h0 / h1) * s_m[2];
  }
  __syncthreads();
  for (long long q = threadIdx.x; q < n_out; q += blockDim.x) {
    const double x = high_end ? -x_out[q] : x_out[q];
    // interval [x_i, x_{i+1}] holding x (clamped to the local knots: the host has checked the range)
    int lo = 0, hi = m - 1;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (s_x[mid] <= x) lo = mid; else hi = mid;
    }
    const double h = s_x[lo + 1] - s_x[lo], A = s_x[lo + 1] - x, B = x - s_x[lo];
    const double v = (s_m[lo] * A * A * A + s_m[lo + 1] * B * B * B) / (6.0 * h) + (s_y[lo] / h - s_m[lo] * h / 6.0) * A +
                     (s_y[lo + 1] / h - s_m[lo + 1] * h / 6.0) * B;
    out[(size_t)row * (size_t)ld_out + (size_t)q] = v;
  }
}

extern "C" int rtx_cubic_end(const double* Ysm, int64_t ld, int n_rows, int64_t i_first, int m, int high_end, const double* knots,
                             const double* x_out, int64_t n_out, double* out, int64_t ld_out, void* stream) {
  if (!Ysm || !knots || !x_out || !out) RTX_FAIL("a required pointer is NULL");
  if (n_rows < 1 || n_rows > 65535) RTX_FAIL("n_rows=%d", n_rows);
  if (m < 4 || m > SPL_END_MAX) RTX_FAIL("m=%d local knots outside [4,%d]", m, SPL_END_MAX);
  if (i_first < 0 || i_first + m > ld) RTX_FAIL("local window [%lld, +%d) outside the row", (long long)i_first, m);
  if (ld_out < n_out) RTX_FAIL("leading dimension smaller than the row");
  if (n_out == 0) return 0;
  hipLaunchKernelGGL(cubic_end_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, Ysm, (long long)ld, (long long)i_first, m,
                     high_end, knots, x_out, (long long)n_out, out, (long long)ld_out);
  RTX_LAUNCH_CHECK();
  return 0;
}
