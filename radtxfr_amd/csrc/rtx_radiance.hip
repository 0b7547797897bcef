// At-sensor radiance and the MAKO instrument-line-shape resampling.
//
// rtx_apparent_radiance : compute_LWIR_apparent_radiance(), reference radiative_transfer.py:1017-1069
// rtx_ils               : ILS_MAKO() triangle, radiative_transfer.py:1236-1256 (kind 0) and the
//                         Gaussian ILS_MAKO.py:21-33 (kind 1)
//
// Both are streaming (HBM-bound) stages: the radiance kernel writes nX*nE*nA*nT floats and reads
// almost nothing, so one workgroup owns one spectral channel, stages B(X, Ts+dT) for that channel
// in LDS (fp64 Planck, once per (a,t)) and streams the [nE][nA][nT] slab with coalesced stores.
// The ILS never builds the reference's dense (nS,nX,nB) temporary: each band only visits the grid
// points under its own weight function.
#include "rtx_common.h"

#define RT_C1 1.19104295315e-16
#define RT_C2 1.43877736830e-02

// ---------------------------------------------------------------------------------------------------
struct RadArgs {
  const double* X;
  long long nX, nE, nA, nT;  // nT >= 1 (1 when dT is absent)
  const float* emis;
  const double* Ts;
  const float *tau, *La, *Ld;
  const double* dT;          // NULL -> no dT axis
  float* L;
  float* Ls;
};

__global__ __launch_bounds__(256) void apparent_radiance_kernel(RadArgs a) {
  extern __shared__ float s_mem[];
  const long long nAT = a.nA * a.nT;
  float* s_B = s_mem;             // [nA*nT]
  float* s_tau = s_B + nAT;       // [nA]
  float* s_La = s_tau + a.nA;
  float* s_Ld = s_La + a.nA;
  for (long long ix = blockIdx.x; ix < a.nX; ix += gridDim.x) {
    __syncthreads();
    const double x100 = a.X[ix] * 100.0;
    const double c1x3 = RT_C1 * (x100 * x100 * x100);
    for (long long e = threadIdx.x; e < nAT; e += blockDim.x) {
      const long long ia = e / a.nT, it = e - ia * a.nT;
      const double T = a.Ts[ia] + (a.dT ? a.dT[it] : 0.0);
      s_B[e] = (float)(c1x3 / (exp(RT_C2 * x100 / T) - 1.0) * 1e4);
    }
    for (long long ia = threadIdx.x; ia < a.nA; ia += blockDim.x) {
      s_tau[ia] = a.tau[ix * a.nA + ia];
      s_La[ia] = a.La[ix * a.nA + ia];
      s_Ld[ia] = a.Ld[ix * a.nA + ia];
    }
    __syncthreads();
    const long long slab = a.nE * nAT;
    const float* em = a.emis + ix * a.nE;
    float* L = a.L + ix * slab;
    float* Ls = a.Ls ? a.Ls + ix * slab : nullptr;
    for (long long o = threadIdx.x; o < slab; o += blockDim.x) {
      const long long ie = o / nAT, r = o - ie * nAT;
      const long long ia = r / a.nT;
      const float e = em[ie];
      const float ls = e * s_B[r] + (1.0f - e) * s_Ld[ia];  // (:1064)
      L[o] = s_tau[ia] * ls + s_La[ia];                      // (:1065)
      if (Ls) Ls[o] = ls;
    }
  }
}

extern "C" int rtx_apparent_radiance(const double* X, int64_t nX, const float* emis, int64_t nE, const double* Ts, int64_t nA,
                                     const float* tau, const float* La, const float* Ld, const double* dT, int64_t nT,
                                     float* L, float* Ls, void* stream) {
  if (nX < 0 || nE < 0 || nA < 0 || nT < 0) RTX_FAIL("negative size");
  if (dT == nullptr) nT = 1;
  if (nX == 0 || nE == 0 || nA == 0 || nT == 0) return 0;
  if (!X || !emis || !Ts || !tau || !La || !Ld || !L) RTX_FAIL("a required pointer is NULL");
  const size_t lds = sizeof(float) * (size_t)(nA * nT + 3 * nA);
  if (lds > 150 * 1024) RTX_FAIL("nA*nT=%lld does not fit the per-channel LDS staging", (long long)(nA * nT));
  RadArgs a;
  a.X = X; a.nX = nX; a.nE = nE; a.nA = nA; a.nT = nT; a.emis = emis; a.Ts = Ts; a.tau = tau; a.La = La; a.Ld = Ld;
  a.dT = dT; a.L = L; a.Ls = Ls;
  if (lds > 64 * 1024)
    RTX_HIP(hipFuncSetAttribute((const void*)apparent_radiance_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  long long blocks = nX < 256 * 16 ? nX : 256 * 16;
  hipLaunchKernelGGL(apparent_radiance_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, a);
  RTX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// ILS. Band b covers the open interval |X - centre_b| < R_b (R = sigma for the triangle; 14 sigma for
// the Gaussian, beyond which exp(-x^2/2) < 3e-43 of the peak, far below fp32 resolution of the sums).
struct IlsArgs {
  int kind;
  GridDev g;
  const double* X;
  long long nx, nS, ldY;
  const float* Y;
  int nB;
  const double* centre;  // device copies
  const double* sigma;
  float* Yout;
};

__device__ __forceinline__ double ils_x(const IlsArgs& a, long long i) { return a.X ? a.X[i] : grid_x(a.g, a.g.offset + i); }

// first index with X[i] > v (strict=1) or X[i] >= v (strict=0), X ascending
__device__ long long ils_bound(const IlsArgs& a, double v, int strict) {
  long long lo = 0, hi = a.nx;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    const double x = ils_x(a, mid);
    const bool right = strict ? (x > v) : (x >= v);
    if (right) hi = mid; else lo = mid + 1;
  }
  return lo;
}

__device__ __forceinline__ float ils_weight(int kind, double x, double c, double s) {
  if (kind == 0) {
    const float w = 1.0f - fabsf((float)(x - c)) / (float)s;  // tri(), :1236-1239
    return w < 0.f ? 0.f : w;
  }
  const float z = (float)((x - c) / s);
  return __expf(-0.5f * z * z) / ((float)s * 2.5066282746310002f);  // g(), ILS_MAKO.py:24
}

// nS small: lanes stride over the band's grid points, every lane handles all nS columns of its rows;
// wavefront shuffles + one LDS exchange reduce over the points. One workgroup per band.
template <int NS_MAX>
__global__ __launch_bounds__(256) void ils_points_kernel(IlsArgs a) {
  const int b = blockIdx.x;
  const double c = a.centre[b], s = a.sigma[b];
  const double R = a.kind == 0 ? s : 14.0 * s;
  const long long lo = ils_bound(a, c - R, 1), hi = ils_bound(a, c + R, 0);
  float acc[NS_MAX];
  float wsum = 0.f;
#pragma unroll
  for (int q = 0; q < NS_MAX; ++q) acc[q] = 0.f;
  for (long long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const float w = ils_weight(a.kind, ils_x(a, i), c, s);
    wsum += w;
    const float* y = a.Y + i * a.ldY;
#pragma unroll
    for (int q = 0; q < NS_MAX; ++q)
      if (q < a.nS) acc[q] = fmaf(w, y[q], acc[q]);
  }
  __shared__ float s_red[4][NS_MAX + 1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int off = 32; off > 0; off >>= 1) {
    wsum += __shfl_down(wsum, off);
#pragma unroll
    for (int q = 0; q < NS_MAX; ++q) acc[q] += __shfl_down(acc[q], off);
  }
  if (lane == 0) {
    s_red[wave][NS_MAX] = wsum;
#pragma unroll
    for (int q = 0; q < NS_MAX; ++q) s_red[wave][q] = acc[q];
  }
  __syncthreads();
  if (threadIdx.x < a.nS) {
    const int q = threadIdx.x;
    const float N = (s_red[0][NS_MAX] + s_red[1][NS_MAX]) + (s_red[2][NS_MAX] + s_red[3][NS_MAX]);
    const float v = (s_red[0][q] + s_red[1][q]) + (s_red[2][q] + s_red[3][q]);
    a.Yout[(size_t)b * a.nS + q] = v / N;  // N = 0 -> NaN, as the reference (quirk 13)
  }
}

// nS large: lanes <-> spectra (coalesced along the sample axis of Y[nx][nS]); the 4 waves of a
// workgroup split the band's grid points and combine through LDS. Grid = (band, 64-column block).
__global__ __launch_bounds__(256) void ils_columns_kernel(IlsArgs a) {
  const int b = blockIdx.x;
  const long long col = (long long)blockIdx.y * 64 + (threadIdx.x & 63);
  const int wave = threadIdx.x >> 6;
  const double c = a.centre[b], s = a.sigma[b];
  const double R = a.kind == 0 ? s : 14.0 * s;
  const long long lo = ils_bound(a, c - R, 1), hi = ils_bound(a, c + R, 0);
  float acc = 0.f, wsum = 0.f;
  const bool live = col < a.nS;
  for (long long i = lo + wave; i < hi; i += 4) {
    const float w = ils_weight(a.kind, ils_x(a, i), c, s);
    wsum += w;
    if (live) acc = fmaf(w, a.Y[i * a.ldY + col], acc);
  }
  __shared__ float s_acc[4][64];
  __shared__ float s_w[4];
  s_acc[wave][threadIdx.x & 63] = acc;
  if ((threadIdx.x & 63) == 0) s_w[wave] = wsum;
  __syncthreads();
  if (wave == 0 && live) {
    const int l = threadIdx.x;
    const float N = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
    const float v = (s_acc[0][l] + s_acc[1][l]) + (s_acc[2][l] + s_acc[3][l]);
    a.Yout[(size_t)b * a.nS + col] = v / N;
  }
}

extern "C" int rtx_ils(int kind, const rtx_grid* grid, const double* X, int64_t nx, const float* Y, int64_t nS, int64_t ldY,
                       int nB, const double* centre_d, const double* sigma_d, float* Y_out, void* stream) {
  if (kind != 0 && kind != 1) RTX_FAIL("kind must be 0 (triangle) or 1 (Gaussian)");
  if (!X) {
    if (rtx_check_grid(grid)) return 1;
    if (nx != grid->n) RTX_FAIL("nx=%lld != grid->n=%lld", (long long)nx, (long long)grid->n);
  }
  if (nB < 0 || nS < 0 || nx < 0) RTX_FAIL("negative size");
  if (nB == 0 || nS == 0) return 0;
  if (!Y || !centre_d || !sigma_d || !Y_out) RTX_FAIL("a required pointer is NULL");
  if (ldY < nS) RTX_FAIL("ldY=%lld < nS=%lld", (long long)ldY, (long long)nS);
  IlsArgs a;
  a.kind = kind;
  if (grid) a.g = to_dev(grid); else { a.g.xmin = a.g.xmax = a.g.step = 0; a.g.n_total = a.g.offset = a.g.n = 0; }
  a.X = X; a.nx = nx; a.nS = nS; a.ldY = ldY; a.Y = Y; a.nB = nB; a.centre = centre_d; a.sigma = sigma_d; a.Yout = Y_out;
  hipStream_t st = (hipStream_t)stream;
  if (nS <= 4) hipLaunchKernelGGL(ils_points_kernel<4>, dim3(nB), dim3(256), 0, st, a);
  else if (nS <= 16) hipLaunchKernelGGL(ils_points_kernel<16>, dim3(nB), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(ils_columns_kernel, dim3(nB, (unsigned)((nS + 63) / 64)), dim3(256), 0, st, a);
  RTX_LAUNCH_CHECK();
  return 0;
}
