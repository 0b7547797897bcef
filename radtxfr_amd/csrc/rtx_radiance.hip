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
// points under its own weight function; on a large materialised Y it runs in one pass over the rows
// (ils_rows_kernel: every row is fetched once although ~3 triangles or ~15 Gaussians cover it).
#include <map>
#include <mutex>
#include <stdlib.h>
#include <string.h>

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

// General case (several atmospheres and/or a dT axis). Grid = (spectral channel, emissivity chunk).
// Threads form a TA x TE tile: ta walks the [nA][nT] slab of one emissivity (fastest output axis, so
// stores coalesce), te walks emissivities; no integer division in the streaming loop.
__global__ __launch_bounds__(256) void apparent_radiance_kernel(RadArgs a, int log2TA, long long e_chunk) {
  extern __shared__ float s_mem[];
  const long long nAT = a.nA * a.nT;
  float* s_B = s_mem;             // [nA*nT]
  float* s_tau = s_B + nAT;       // [nA]
  float* s_La = s_tau + a.nA;
  float* s_Ld = s_La + a.nA;
  const int TA = 1 << log2TA, TE = 256 >> log2TA;
  const int ta = threadIdx.x & (TA - 1), te = threadIdx.x >> log2TA;
  const long long e_lo = (long long)blockIdx.y * e_chunk;
  const long long e_hi = e_lo + e_chunk < a.nE ? e_lo + e_chunk : a.nE;
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
    const float* em = a.emis + ix * a.nE;
    float* L = a.L + ix * a.nE * nAT;
    float* Ls = a.Ls ? a.Ls + ix * a.nE * nAT : nullptr;
    for (long long ie = e_lo + te; ie < e_hi; ie += TE) {
      const float e = em[ie];
      long long ia = ta / a.nT;  // once per emissivity, then advanced incrementally
      long long it = ta - ia * a.nT;
      for (long long r = ta; r < nAT; r += TA) {
        const float ls = e * s_B[r] + (1.0f - e) * s_Ld[ia];  // (:1064)
        const long long o = ie * nAT + r;
        L[o] = s_tau[ia] * ls + s_La[ia];                      // (:1065)
        if (Ls) Ls[o] = ls;
        it += TA;
        while (it >= a.nT) { it -= a.nT; ++ia; }
      }
    }
  }
}

// One atmosphere, no dT axis (config C4): pure streaming. One spectral channel per WAVE-iteration, eight 16-B loads per lane
// along the emissivity axis issued before the first is used (8 KB of a 2000-emissivity row in flight per wave), Planck per
// lane in fp32 from an fp64 exponent. (One channel per workgroup-iteration with two dependent load/store pairs per thread:
// 4.7 TB/s read + write; this form: see DESIGN.md 4.4.)
__global__ __launch_bounds__(256) void apparent_radiance_row_kernel(RadArgs a) {
  const long long nE4 = a.nE >> 2;  // nE % 4 == 0 on this path
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long long ix = (long long)blockIdx.x * 4 + wave; ix < a.nX; ix += (long long)gridDim.x * 4) {
    const double x = a.X[ix];
    const double x100 = x * 100.0;
    const float B = planck_f32(RT_C1 * (x100 * x100 * x100) * 1e4, x, 100.0 * RT_C2 * 1.4426950408889634 / a.Ts[0]);
    const float tau = a.tau[ix], La = a.La[ix], Ld = a.Ld[ix];
    const float4* em = reinterpret_cast<const float4*>(a.emis + ix * a.nE);
    float4* L = reinterpret_cast<float4*>(a.L + ix * a.nE);
    float4* Ls = a.Ls ? reinterpret_cast<float4*>(a.Ls + ix * a.nE) : nullptr;
    for (long long q0 = 0; q0 < nE4; q0 += 512) {
      float4 e[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const long long q = q0 + lane + 64 * t;
        e[t] = q < nE4 ? em[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const long long q = q0 + lane + 64 * t;
        if (q < nE4) {
          float4 ls, l;
          ls.x = e[t].x * B + (1.0f - e[t].x) * Ld; ls.y = e[t].y * B + (1.0f - e[t].y) * Ld;
          ls.z = e[t].z * B + (1.0f - e[t].z) * Ld; ls.w = e[t].w * B + (1.0f - e[t].w) * Ld;
          l.x = tau * ls.x + La; l.y = tau * ls.y + La; l.z = tau * ls.z + La; l.w = tau * ls.w + La;
          L[q] = l;
          if (Ls) Ls[q] = ls;
        }
      }
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
  RadArgs a;
  a.X = X; a.nX = nX; a.nE = nE; a.nA = nA; a.nT = nT; a.emis = emis; a.Ts = Ts; a.tau = tau; a.La = La; a.Ld = Ld;
  a.dT = dT; a.L = L; a.Ls = Ls;
  hipStream_t st = (hipStream_t)stream;
  const bool aligned16 = (nE % 4 == 0) && (((uintptr_t)emis | (uintptr_t)L | (uintptr_t)Ls) % 16 == 0);
  if (nA == 1 && nT == 1 && dT == nullptr && aligned16) {
    const long long rows4 = (nX + 3) / 4;
    const long long blocks = rows4 < 256 * 32 ? rows4 : 256 * 32;
    hipLaunchKernelGGL(apparent_radiance_row_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    RTX_LAUNCH_CHECK();
    return 0;
  }
  const long long nAT = nA * nT;
  const size_t lds = sizeof(float) * (size_t)(nAT + 3 * nA);
  if (lds > 150 * 1024) RTX_FAIL("nA*nT=%lld does not fit the per-channel LDS staging", (long long)nAT);
  if (lds > 64 * 1024)
    RTX_HIP(hipFuncSetAttribute((const void*)apparent_radiance_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int log2TA = 0;
  while (log2TA < 8 && (2LL << log2TA) <= nAT) ++log2TA;  // TA = largest power of two <= min(nAT, 256)
  // enough workgroups to fill 256 CUs even for the reference's 128-channel use: split the emissivity axis
  long long bx = nX < 4096 ? nX : 4096;
  long long chunks = (2048 + bx - 1) / bx;
  if (chunks > nE) chunks = nE;
  if (chunks < 1) chunks = 1;
  const long long e_chunk = (nE + chunks - 1) / chunks;
  chunks = (nE + e_chunk - 1) / e_chunk;
  hipLaunchKernelGGL(apparent_radiance_kernel, dim3((unsigned)bx, (unsigned)chunks), dim3(256), lds, st, a, log2TA, e_chunk);
  RTX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// ILS. Band b covers the open interval |X - centre_b| < R_b (R = sigma for the triangle; for the Gaussian the
// points whose weight is within 2e-11 (7 sigma; 14 sigma = 3e-43 for a band centred outside the grid) of the largest one,
// far below fp32 resolution of the sums).
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

// z0sq: (distance from the band centre to the nearest grid point / sigma)^2 for a Gaussian band centred OUTSIDE the
// grid (0 inside). The reference does not clip the Gaussian band list (ILS_MAKO.py:13-19): such a band comes out as the
// average of the grid's edge region under the far Gaussian tail, as long as its fp64 weights do not all underflow
// (~38 sigma); weights are taken relative to the nearest point (the common factor cancels in the normalisation).
__device__ __forceinline__ float ils_weight(int kind, double x, double c, double s, double z0sq = 0.0) {
  if (kind == 0) {
    const float w = 1.0f - fabsf((float)(x - c)) / (float)s;  // tri(), :1236-1239
    return w < 0.f ? 0.f : w;
  }
  const double z = (x - c) / s;
  return __expf((float)(-0.5 * (z * z - z0sq))) / ((float)s * 2.5066282746310002f);  // g(), ILS_MAKO.py:24
}

// Support radius of band (c, s) on this grid; for the Gaussian also z0sq (see ils_weight) and whether the reference's
// weights all underflow to zero (0/0 = NaN there, ILS_MAKO.py:27: an empty support here).
__device__ __forceinline__ double ils_reach(const IlsArgs& a, double c, double s, double& z0sq, bool& dead) {
  z0sq = 0.0;
  dead = false;
  if (a.kind == 0) return s;
  const double x0 = ils_x(a, 0), x1 = ils_x(a, a.nx - 1);
  const double d0 = fmax(fmax(x0 - c, c - x1), 0.0);
  z0sq = (d0 / s) * (d0 / s);
  dead = 0.5 * z0sq + log(s * 2.5066282746310002) > 744.44;  // exp(-744.44) = 2^-1074, the smallest denormal
  // A band centred inside the grid: 7 sigma (relative weight < 2.3e-11 of the largest one: four orders below fp32 resolution
  // of the normalised sum; the reference sums all of them). A band centred outside is the average of the edge region under
  // its far tail: 14 sigma of weight RELATIVE to the nearest point (< 3e-43).
  const double ns2 = d0 > 0.0 ? 196.0 : 49.0;
  return sqrt(d0 * d0 + ns2 * s * s);
}

// nS small: lanes stride over the band's grid points, every lane handles all nS columns of its rows;
// wavefront shuffles + one LDS exchange reduce over the points. One workgroup per band.
template <int NS_MAX>
__global__ __launch_bounds__(256) void ils_points_kernel(IlsArgs a) {
  const int b = blockIdx.x;
  const double c = a.centre[b], s = a.sigma[b];
  double z0sq;
  bool dead;
  const double R = ils_reach(a, c, s, z0sq, dead);
  const long long lo = dead ? 0 : ils_bound(a, c - R, 1), hi = dead ? 0 : ils_bound(a, c + R, 0);
  float acc[NS_MAX];
  float wsum = 0.f;
#pragma unroll
  for (int q = 0; q < NS_MAX; ++q) acc[q] = 0.f;
  for (long long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const float w = ils_weight(a.kind, ils_x(a, i), c, s, z0sq);
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
  double z0sq;
  bool dead;
  const double R = ils_reach(a, c, s, z0sq, dead);
  const long long lo = dead ? 0 : ils_bound(a, c - R, 1), hi = dead ? 0 : ils_bound(a, c + R, 0);
  float acc = 0.f, wsum = 0.f;
  const bool live = col < a.nS;
  for (long long i0 = lo + wave; i0 < hi; i0 += 256) {  // two-level sums (64 rows per block)
    float pa = 0.f, pw = 0.f;
    const long long blk_end = i0 + 256 < hi ? i0 + 256 : hi;
    for (long long i = i0; i < blk_end; i += 4) {
      const float w = ils_weight(a.kind, ils_x(a, i), c, s, z0sq);
      pw += w;
      if (live) pa = fmaf(w, a.Y[i * a.ldY + col], pa);
    }
    acc += pa;
    wsum += pw;
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

// nS large and a multiple of 4: lanes <-> 4 adjacent spectra (16-B loads, 1 KiB contiguous per wave-instruction
// and grid row); the 4 waves split the band's grid points. Grid = (band, 256-column block).
__global__ __launch_bounds__(256) void ils_columns4_kernel(IlsArgs a) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long col4 = (long long)blockIdx.y * 64 + lane;  // index of the float4 column group
  const long long nS4 = a.nS >> 2, ld4 = a.ldY >> 2;
  const double c = a.centre[b], s = a.sigma[b];
  double z0sq;
  bool dead;
  const double R = ils_reach(a, c, s, z0sq, dead);
  const long long lo = dead ? 0 : ils_bound(a, c - R, 1), hi = dead ? 0 : ils_bound(a, c + R, 0);
  const bool live = col4 < nS4;
  const float4* Y4 = reinterpret_cast<const float4*>(a.Y) + (live ? col4 : 0);
  // two-level sums: a wave adds up to ~6000 rows per band; 64-row blocks keep the fp32 error at ~1e-7
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float wsum = 0.f;
  long long i = lo + wave;
  while (i < hi) {
    float4 pa = make_float4(0.f, 0.f, 0.f, 0.f);
    float pw = 0.f;
    const long long blk_end = i + 256 < hi ? i + 256 : hi;
    for (; i + 4 < blk_end; i += 8) {  // two rows in flight per wave
      const float w0 = ils_weight(a.kind, ils_x(a, i), c, s, z0sq), w1 = ils_weight(a.kind, ils_x(a, i + 4), c, s, z0sq);
      const float4 y0 = Y4[i * ld4], y1 = Y4[(i + 4) * ld4];
      pw += w0 + w1;
      pa.x = fmaf(w0, y0.x, pa.x); pa.y = fmaf(w0, y0.y, pa.y); pa.z = fmaf(w0, y0.z, pa.z); pa.w = fmaf(w0, y0.w, pa.w);
      pa.x = fmaf(w1, y1.x, pa.x); pa.y = fmaf(w1, y1.y, pa.y); pa.z = fmaf(w1, y1.z, pa.z); pa.w = fmaf(w1, y1.w, pa.w);
    }
    for (; i < blk_end; i += 4) {
      const float w0 = ils_weight(a.kind, ils_x(a, i), c, s, z0sq);
      const float4 y0 = Y4[i * ld4];
      pw += w0;
      pa.x = fmaf(w0, y0.x, pa.x); pa.y = fmaf(w0, y0.y, pa.y); pa.z = fmaf(w0, y0.z, pa.z); pa.w = fmaf(w0, y0.w, pa.w);
    }
    acc.x += pa.x; acc.y += pa.y; acc.z += pa.z; acc.w += pa.w;
    wsum += pw;
  }
  __shared__ float4 s_acc[4][64];
  __shared__ float s_w[4];
  s_acc[wave][lane] = acc;
  if (lane == 0) s_w[wave] = wsum;
  __syncthreads();
  if (wave == 0 && live) {
    const float N = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
    float4 v;
    v.x = ((s_acc[0][lane].x + s_acc[1][lane].x) + (s_acc[2][lane].x + s_acc[3][lane].x)) / N;
    v.y = ((s_acc[0][lane].y + s_acc[1][lane].y) + (s_acc[2][lane].y + s_acc[3][lane].y)) / N;
    v.z = ((s_acc[0][lane].z + s_acc[1][lane].z) + (s_acc[2][lane].z + s_acc[3][lane].z)) / N;
    v.w = ((s_acc[0][lane].w + s_acc[1][lane].w) + (s_acc[2][lane].w + s_acc[3][lane].w)) / N;
    reinterpret_cast<float4*>(a.Yout + (size_t)b * a.nS)[col4] = v;
  }
}

// ---- ILS over a large materialised Y in ONE pass over its rows ---------------------------------------------------------------
// ils_columns4_kernel streams a band's support per workgroup; neighbouring bands' supports overlap (triangle: sigma = 1.6
// band spacings, every row lies under ~3.2 of them; Gaussian: 14 sigma either side, ~28), so every row of Y comes from HBM
// that many times -- 14.5 GB for C4's 4.56 GB input with the triangle, 128 GB with the Gaussian. Here a workgroup owns
// ILS_CH consecutive rows x 256 columns, reads them once from HBM and keeps the sums of every band that reaches the chunk
// (ILS_CAP / ILS_CAP_GAUSS at a time in registers, further passes over the chunk come from the cache; the weight of a row under a band is
// uniform over the lanes); the chunks' partial sums go to a workspace and ils_rows_reduce_kernel adds a band's chunks in a
// fixed order (deterministic: no atomics) and normalises. A chunk reached by more bands than the workspace has slots for
// raises a flag and the reduce kernel then redoes its band the old way, so the result never depends on the band list being
// well behaved.
#ifndef ILS_CH
#define ILS_CH 1024
#endif
#ifndef ILS_CAP
#define ILS_CAP 4  // bands per register pass (C4 triangle, ms: 6 -> 1.22, 5 -> 1.08, 4 -> 1.06)
#endif
#ifndef ILS_CAP_GAUSS
#define ILS_CAP_GAUSS 16  // the Gaussian's ~15 bands per chunk (7 sigma either side) in one pass (8: 1.43 ms, 16: 1.38)
#endif
#define ILS_SLOTS_TRI 12
#define ILS_SLOTS_GAUSS 24
struct IlsRowsArgs {
  IlsArgs a;
  float* P;    // [n_chunks][slots][nS]
  float* Wt;   // [n_chunks][slots]
  int* b0;     // [n_chunks] first band of the chunk, [n_chunks .. 2 n_chunks) number of bands
  int* overflow;
  int n_chunks, slots;
};

template <int KIND, int CAP>
__global__ __launch_bounds__(256) void ils_rows_kernel(IlsRowsArgs r) {
  const IlsArgs& a = r.a;
  const int chunk = blockIdx.x;
  const long long r0 = (long long)chunk * ILS_CH, r1 = r0 + ILS_CH < a.nx ? r0 + ILS_CH : a.nx;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long col4 = (long long)blockIdx.y * 64 + lane;
  const long long nS4 = a.nS >> 2, ld4 = a.ldY >> 2;
  const bool live = col4 < nS4;
  const float4* Y4 = reinterpret_cast<const float4*>(a.Y) + (live ? col4 : 0);
  __shared__ int s_b0, s_b1;
  __shared__ float4 s_red[4][64];
  __shared__ float s_w[4];
  if (threadIdx.x == 0) { s_b0 = a.nB; s_b1 = -1; }
  __syncthreads();
  const double x_lo = ils_x(a, r0), x_hi = ils_x(a, r1 - 1);
  for (int b = threadIdx.x; b < a.nB; b += 256) {  // bands whose open support (c - R, c + R) meets the chunk
    const double c = a.centre[b], sg = a.sigma[b];
    double z0sq;
    bool dead;
    const double R = ils_reach(a, c, sg, z0sq, dead);
    if (!dead && c + R > x_lo && c - R < x_hi) { atomicMin(&s_b0, b); atomicMax(&s_b1, b); }
  }
  __syncthreads();
  const int b0 = s_b0, n_act = s_b1 - b0 + 1;
  if (blockIdx.y == 0 && threadIdx.x == 0) {
    r.b0[chunk] = b0;
    r.b0[r.n_chunks + chunk] = n_act > 0 ? n_act : 0;
    if (n_act > r.slots) *r.overflow = 1;
  }
  if (n_act <= 0) return;
  const int n_use = n_act < r.slots ? n_act : r.slots;
  for (int g0 = 0; g0 < n_use; g0 += CAP) {
    double cj[CAP], isd[CAP], z0[CAP];
    float isj[CAP], wsum[CAP];
    float4 acc[CAP];
#pragma unroll
    for (int j = 0; j < CAP; ++j) {
      const bool v = g0 + j < n_use;
      const double c = v ? a.centre[b0 + g0 + j] : 0.0, sg = v ? a.sigma[b0 + g0 + j] : 1.0;
      double z0sq;
      bool dead;
      (void)ils_reach(a, c, sg, z0sq, dead);
      cj[j] = c;
      z0[j] = z0sq;
      isd[j] = 1.0 / sg;
      // triangle: 1/sigma; Gaussian: 1/(sigma sqrt(2 pi)). An unused slot or a band the reference gives 0/0: weight 0
      isj[j] = KIND == 0 ? (v ? 1.0f / (float)sg : __builtin_inff()) : ((v && !dead) ? 1.0f / ((float)sg * 2.5066282746310002f) : 0.0f);
      wsum[j] = 0.f;
      acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // A row's weight under a band is the same for every lane: lane l computes the weights of ITS row of a 64-row block
    // (wave w's rows: first + w + 4 l) for the CAP bands once, and the row loop fetches row t's weights with
    // v_readlane -- one weight evaluation per row and band instead of 64 identical ones (the Gaussian's fp64 argument
    // and exponential were 3/4 of that kernel's instructions).
    for (long long blk = r0; blk < r1; blk += 256) {
      const long long my_row = blk + wave + 4 * lane;
      const bool my_in = my_row < r1;
      const double xr = ils_x(a, my_in ? my_row : r1 - 1);
      float wv[CAP];
#pragma unroll
      for (int j = 0; j < CAP; ++j) {
        float w;
        if (KIND == 0) {
          w = 1.0f - fabsf((float)(xr - cj[j])) * isj[j];  // tri(), :1236-1239
        } else {
          const double z = (xr - cj[j]) * isd[j];
          w = __expf((float)(-0.5 * (z * z - z0[j]))) * isj[j];  // g(), ILS_MAKO.py:24, relative to the nearest point for a band outside the grid
        }
        wv[j] = (my_in && w > 0.f) ? w : 0.f;
        wsum[j] += wv[j];  // per lane; summed over the lanes below
      }
      const long long n_rows = (r1 - blk - wave + 3) / 4;  // rows of this wave in the block
      const int nt = n_rows > 64 ? 64 : (n_rows > 0 ? (int)n_rows : 0);
      for (int t = 0; t < nt; t += 4) {  // four rows in flight
        float4 y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long long it = blk + wave + 4 * (t + u);
          y[u] = (t + u < nt) ? Y4[it * ld4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
          for (int j = 0; j < CAP; ++j) {
            const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wv[j]), (t + u) & 63));  // 0 past the last row
            acc[j].x = fmaf(w, y[u].x, acc[j].x); acc[j].y = fmaf(w, y[u].y, acc[j].y);
            acc[j].z = fmaf(w, y[u].z, acc[j].z); acc[j].w = fmaf(w, y[u].w, acc[j].w);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < CAP; ++j) {  // the wave's weight sum: over its lanes, fixed order
      float v = wsum[j];
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
      wsum[j] = __shfl(v, 0);
    }
    // the four waves' sums, slot by slot, in a fixed order
#pragma unroll
    for (int j = 0; j < CAP; ++j) {
      if (g0 + j >= n_use) break;
      s_red[wave][lane] = acc[j];
      if (lane == 0) s_w[wave] = wsum[j];
      __syncthreads();
      if (wave == 0) {
        if (live) {
          float4 v;
          v.x = (s_red[0][lane].x + s_red[1][lane].x) + (s_red[2][lane].x + s_red[3][lane].x);
          v.y = (s_red[0][lane].y + s_red[1][lane].y) + (s_red[2][lane].y + s_red[3][lane].y);
          v.z = (s_red[0][lane].z + s_red[1][lane].z) + (s_red[2][lane].z + s_red[3][lane].z);
          v.w = (s_red[0][lane].w + s_red[1][lane].w) + (s_red[2][lane].w + s_red[3][lane].w);
          reinterpret_cast<float4*>(r.P + ((size_t)chunk * r.slots + (g0 + j)) * (size_t)a.nS)[col4] = v;
        }
        if (blockIdx.y == 0 && lane == 0) r.Wt[(size_t)chunk * r.slots + (g0 + j)] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
      }
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(256) void ils_rows_reduce_kernel(IlsRowsArgs r) {
  const IlsArgs& a = r.a;
  const int b = blockIdx.x;
  const double c = a.centre[b], sg = a.sigma[b];
  double z0sq;
  bool dead;
  const double R = ils_reach(a, c, sg, z0sq, dead);
  const long long lo = dead ? 0 : ils_bound(a, c - R, 1), hi = dead ? 0 : ils_bound(a, c + R, 0);
  if (*r.overflow) {  // some chunk is reached by more bands than the workspace holds: this band the per-band way
    // (ils_columns4_kernel's arithmetic; the launch geometry is the same)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long col4 = (long long)blockIdx.y * 64 + lane;
    const long long nS4 = a.nS >> 2, ld4 = a.ldY >> 2;
    const bool live = col4 < nS4;
    const float4* Y4 = reinterpret_cast<const float4*>(a.Y) + (live ? col4 : 0);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float wsum = 0.f;
    for (long long i = lo + wave; i < hi; i += 4) {
      const float w0 = ils_weight(a.kind, ils_x(a, i), c, sg, z0sq);
      const float4 y0 = Y4[i * ld4];
      wsum += w0;
      acc.x = fmaf(w0, y0.x, acc.x); acc.y = fmaf(w0, y0.y, acc.y); acc.z = fmaf(w0, y0.z, acc.z); acc.w = fmaf(w0, y0.w, acc.w);
    }
    __shared__ float4 s_acc[4][64];
    __shared__ float s_w[4];
    s_acc[wave][lane] = acc;
    if (lane == 0) s_w[wave] = wsum;
    __syncthreads();
    if (wave == 0 && live) {
      const float N = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
      float4 v;
      v.x = ((s_acc[0][lane].x + s_acc[1][lane].x) + (s_acc[2][lane].x + s_acc[3][lane].x)) / N;
      v.y = ((s_acc[0][lane].y + s_acc[1][lane].y) + (s_acc[2][lane].y + s_acc[3][lane].y)) / N;
      v.z = ((s_acc[0][lane].z + s_acc[1][lane].z) + (s_acc[2][lane].z + s_acc[3][lane].z)) / N;
      v.w = ((s_acc[0][lane].w + s_acc[1][lane].w) + (s_acc[2][lane].w + s_acc[3][lane].w)) / N;
      reinterpret_cast<float4*>(a.Yout + (size_t)b * a.nS)[col4] = v;
    }
    return;
  }
  const long long col4 = (long long)blockIdx.y * 256 + threadIdx.x;  // one thread per float4 column group
  const long long nS4 = a.nS >> 2;
  if (col4 >= nS4) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float N = 0.f;
  if (hi > lo) {
    const int c0 = (int)(lo / ILS_CH), c1 = (int)((hi - 1) / ILS_CH);
    for (int ch = c0; ch <= c1; ++ch) {
      const int slot = b - r.b0[ch];
      if (slot < 0 || slot >= r.b0[r.n_chunks + ch]) continue;
      const float4 v = reinterpret_cast<const float4*>(r.P + ((size_t)ch * r.slots + slot) * (size_t)a.nS)[col4];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      N += r.Wt[(size_t)ch * r.slots + slot];
    }
  }
  float4 o;
  o.x = acc.x / N; o.y = acc.y / N; o.z = acc.z / N; o.w = acc.w / N;  // an empty support: 0/0 = NaN, like the reference
  reinterpret_cast<float4*>(a.Yout + (size_t)b * a.nS)[col4] = o;
}

// grow-only workspace of the one-pass form, per (device, stream): two rtx_ils calls in flight on different streams (or
// from different host threads on their own streams) must not share partial sums and the overflow flag; calls on ONE stream
// are ordered by it. A workspace is re-allocated only to grow, and hipFree waits for the device, so a launch already
// enqueued on that stream never loses its buffer.
static int ils_rows_workspace(size_t bytes, hipStream_t st, void** out) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, std::pair<void*, size_t>> ws;
  int dev = 0;
  RTX_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  auto& e = ws[std::make_pair(dev, st)];
  if (e.second < bytes) {
    if (e.first) RTX_HIP(hipFree(e.first));
    e.first = nullptr; e.second = 0;
    RTX_HIP(hipMalloc(&e.first, bytes));
    e.second = bytes;
  }
  *out = e.first;
  return 0;
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
  else if (nS % 4 == 0 && ldY % 4 == 0 && (((uintptr_t)Y | (uintptr_t)Y_out) % 16 == 0)) {
    // the one-pass form where a band's support is long against a chunk (its partial sums then cost little next to Y
    // itself) and Y is big enough to matter; otherwise one workgroup per band and column block
    static int one_pass = -1;
    if (one_pass < 0) { const char* e = getenv("RADTXFR_ILS_KERNEL"); one_pass = (e && !strcmp(e, "bands")) ? 0 : 1; }
    const long long n_chunks = (nx + ILS_CH - 1) / ILS_CH;
    if (one_pass && nx >= (long long)nB * 2 * ILS_CH && (double)nx * (double)nS >= 3.0e7 && n_chunks < (1 << 20)) {
      IlsRowsArgs r;
      r.a = a; r.n_chunks = (int)n_chunks; r.slots = kind == 0 ? ILS_SLOTS_TRI : ILS_SLOTS_GAUSS;
      const size_t nP = (size_t)n_chunks * r.slots * (size_t)nS, nW = (size_t)n_chunks * r.slots;
      void* base = nullptr;
      if (ils_rows_workspace((nP + nW) * sizeof(float) + (2 * (size_t)n_chunks + 1) * sizeof(int), st, &base)) return 1;
      r.P = (float*)base; r.Wt = r.P + nP; r.b0 = (int*)(r.Wt + nW); r.overflow = r.b0 + 2 * n_chunks;
      RTX_HIP(hipMemsetAsync(r.overflow, 0, sizeof(int), st));
      if (kind == 0) hipLaunchKernelGGL((ils_rows_kernel<0, ILS_CAP>), dim3((unsigned)n_chunks, (unsigned)((nS / 4 + 63) / 64)), dim3(256), 0, st, r);
      else hipLaunchKernelGGL((ils_rows_kernel<1, ILS_CAP_GAUSS>), dim3((unsigned)n_chunks, (unsigned)((nS / 4 + 63) / 64)), dim3(256), 0, st, r);
      RTX_LAUNCH_CHECK();
      // (band, column block): 256 float4 column groups per workgroup when reducing, 64 when a band is redone the old way
      hipLaunchKernelGGL(ils_rows_reduce_kernel, dim3(nB, (unsigned)((nS / 4 + 63) / 64)), dim3(256), 0, st, r);
    } else {
      hipLaunchKernelGGL(ils_columns4_kernel, dim3(nB, (unsigned)((nS / 4 + 63) / 64)), dim3(256), 0, st, a);
    }
  }
  else hipLaunchKernelGGL(ils_columns_kernel, dim3(nB, (unsigned)((nS + 63) / 64)), dim3(256), 0, st, a);
  RTX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// Knot spectra -> monochromatic grid: np.interp(X, Xk, F[:, s]) for every column s (C4/C5 of SURVEY 8d:
// emissivities live on the ~1 cm^-1 ASTER-DB knots, Generate_ASTER_emissivity_DB.py:48-52,81, and are
// "linearly interpolated to the hi-res grid"). Lanes <-> columns (coalesced rows of F and of the output),
// one grid point per workgroup-iteration; the knot interval is found once per point (wave-uniform).
struct InterpArgs {
  GridDev g;
  const double* X;
  long long nx;
  const double* Xk;
  long long nk, nS;
  const float* F;   // [nk][nS]
  float* out;       // [nx][nS]
};

__global__ __launch_bounds__(256) void interp_knots_kernel(InterpArgs a) {
  constexpr int R = 64;  // grid points per workgroup-iteration: their knot searches run in parallel
  __shared__ long long s_j[R];
  __shared__ float s_t[R];
  const long long n_groups = (a.nx + R - 1) / R;
  const bool vec = (a.nS % 4 == 0) && ((((uintptr_t)a.F | (uintptr_t)a.out) & 15) == 0);
  for (long long grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const long long i0 = grp * R;
    const int rows = (int)(a.nx - i0 < R ? a.nx - i0 : R);
    __syncthreads();
    if ((int)threadIdx.x < rows) {
      const long long i = i0 + threadIdx.x;
      const double x = a.X ? a.X[i] : grid_x(a.g, a.g.offset + i);
      // j = last knot with Xk[j] <= x, clamped to [0, nk-2]; outside the knots np.interp holds the end values
      long long lo = 0, hi = a.nk - 1;
      while (hi - lo > 1) {
        const long long mid = (lo + hi) >> 1;
        if (a.Xk[mid] <= x) lo = mid; else hi = mid;
      }
      const double x0 = a.Xk[lo], x1 = a.Xk[lo + 1];
      double t = (x - x0) / (x1 - x0);
      t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
      s_j[threadIdx.x] = lo;
      s_t[threadIdx.x] = (float)t;
    }
    __syncthreads();
    if (vec) {
      // a row per wave-iteration, four 16-B knot-row pairs per lane in flight before the first store
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      const long long nS4 = a.nS >> 2;
      for (int r = wave; r < rows; r += 4) {
        const float tf = s_t[r];
        const float4* f04 = reinterpret_cast<const float4*>(a.F + s_j[r] * a.nS);
        const float4* f14 = reinterpret_cast<const float4*>(a.F + (s_j[r] + 1) * a.nS);
        float4* o4 = reinterpret_cast<float4*>(a.out + (i0 + r) * a.nS);
        for (long long q0 = 0; q0 < nS4; q0 += 256) {
          float4 u[4], v[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const long long q = q0 + lane + 64 * t;
            const bool in = q < nS4;
            u[t] = in ? f04[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            v[t] = in ? f14[q] : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const long long q = q0 + lane + 64 * t;
            if (q < nS4) {
              float4 w;
              w.x = fmaf(tf, v[t].x - u[t].x, u[t].x); w.y = fmaf(tf, v[t].y - u[t].y, u[t].y);
              w.z = fmaf(tf, v[t].z - u[t].z, u[t].z); w.w = fmaf(tf, v[t].w - u[t].w, u[t].w);
              o4[q] = w;
            }
          }
        }
      }
    } else {
      for (int r = 0; r < rows; ++r) {
        const float tf = s_t[r];
        const float* f0 = a.F + s_j[r] * a.nS;
        const float* f1 = f0 + a.nS;
        float* o = a.out + (i0 + r) * a.nS;
        for (long long sidx = threadIdx.x; sidx < a.nS; sidx += blockDim.x) o[sidx] = fmaf(tf, f1[sidx] - f0[sidx], f0[sidx]);
      }
    }
  }
}

extern "C" int rtx_interp_knots(const rtx_grid* grid, const double* X, int64_t nx, const double* Xk, int64_t nk,
                                const float* F, int64_t nS, float* out, void* stream) {
  if (!X) {
    if (rtx_check_grid(grid)) return 1;
    if (nx != grid->n) RTX_FAIL("nx=%lld != grid->n=%lld", (long long)nx, (long long)grid->n);
  }
  if (nk < 2) RTX_FAIL("need at least 2 knots");
  if (nx < 0 || nS < 0) RTX_FAIL("negative size");
  if (nx == 0 || nS == 0) return 0;
  if (!Xk || !F || !out) RTX_FAIL("a required pointer is NULL");
  InterpArgs a;
  if (grid) a.g = to_dev(grid); else { a.g.xmin = a.g.xmax = a.g.step = 0; a.g.n_total = a.g.offset = a.g.n = 0; }
  a.X = X; a.nx = nx; a.Xk = Xk; a.nk = nk; a.nS = nS; a.F = F; a.out = out;
  const long long groups = (nx + 63) / 64;
  const long long blocks = groups < 256 * 16 ? groups : 256 * 16;
  hipLaunchKernelGGL(interp_knots_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  RTX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// Fused C4 (SURVEY 8d row "C+D fused"): band radiances of MANY emissivity spectra given on knots, without
// materialising anything at monochromatic resolution.
//   L_b,k = sum_i w_b(i) [ tau_i (eps_k(nu_i) B_i + (1 - eps_k(nu_i)) Ld_i) + La_i ] / sum_i w_b(i)
// with eps_k(nu_i) = (1-f_i) E[j(i)][k] + f_i E[j(i)+1][k] (np.interp) is linear in the knot values, so
//   L_b,k = ( C_b + sum_j M[b][j] E[j][k] ) / N_b,
//   N_b = sum_i w_i,  C_b = sum_i w_i (tau_i Ld_i + La_i),  M[b][j] = sum over the two knot intervals touching j of
//   w_i tau_i (B_i - Ld_i) times the hat function of knot j at nu_i.
// rtx_band_moments does the monochromatic pass once (band_basis_moments_kernel in its PLANCK mode, further down: wavefront
// reductions per knot interval, fixed order: deterministic); rtx_band_mix is the tiny [nB x nk] x [nk x nE] contraction
// restricted to each band's knots.
struct MixArgs {
  const float *N, *C, *M;
  const int2* jrange;
  long long nk, nE;
  const float* E;  // [nk][nE]
  float* out;      // [nB][nE]
};

__global__ __launch_bounds__(256) void band_mix_kernel(MixArgs a) {
  const int b = blockIdx.x;
  const long long k = (long long)blockIdx.y * blockDim.x + threadIdx.x;
  if (k >= a.nE) return;
  const int2 jr = a.jrange[b];
  const float* Mrow = a.M + (size_t)b * a.nk;
  float acc = a.C ? a.C[b] : 0.f;
  for (int j = jr.x; j <= jr.y; ++j) acc = fmaf(Mrow[j], a.E[(size_t)j * a.nE + k], acc);
  a.out[(size_t)b * a.nE + k] = a.N ? acc / a.N[b] : acc;  // N = 0 -> NaN, as the unfused path
}

extern "C" int rtx_band_mix(const float* N, const float* C, const float* M, const int32_t* jrange, int nB, int64_t nk,
                            const float* E, int64_t nE, float* out, void* stream) {
  if (nB < 0 || nE < 0 || nk < 1) RTX_FAIL("bad size");
  if (nB == 0 || nE == 0) return 0;
  if (!M || !jrange || !E || !out) RTX_FAIL("a required pointer is NULL");  // N, C may be NULL: plain contraction
  MixArgs a;
  a.N = N; a.C = C; a.M = M; a.jrange = reinterpret_cast<const int2*>(jrange); a.nk = nk; a.nE = nE; a.E = E; a.out = out;
  hipLaunchKernelGGL(band_mix_kernel, dim3(nB, (unsigned)((nE + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  RTX_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// Fused C5: an HSI cube -- every pixel has its own linear emissivity mixture AND its own surface
// temperature T_p (LWIR_HSI_Generator.py:151-167) -- at monochromatic resolution followed by the ILS.
//   L_b,p = [ C_b + sum_m f_pm ( sum_i w tau B(nu_i,T_p) eps_km(nu_i) - sum_i w tau Ld eps_km(nu_i) ) ] / N_b
// B(nu, T_p) is the only pixel-dependent factor inside the monochromatic sum. Over the support of one band
// (a few cm^-1 to ~25 cm^-1) it is a very smooth function of nu, so it is replaced by its degree-(Q-1)
// interpolant through Q Chebyshev nodes of the band:  B(nu,T) = sum_q l_q(s) B(nu_bq, T),  s = (nu-c_b)/R_b.
// Worst pointwise interpolation error over the MAKO bands, 230-350 K (NumPy): Q = 4 (sensor.hsi_cube's default)
// 4e-10 relative at resFactor 2, 7e-9 without; Q = 5: 2e-12 / 6e-11; Q = 3: 4e-7 / 3e-6 -- fp32 rounding is 6e-8. Then
//   L_b,p = [ C_b + sum_m f_pm ( sum_q B(nu_bq,T_p) AB[q][b][k_m] - ALd[b][k_m] ) ] / N_b
// with pixel-independent tables from ONE monochromatic pass: MB[q][b][j] = sum_i w tau l_q hat_j,
// MLd[b][j] = sum_i w tau Ld hat_j (rtx_band_basis_moments), contracted with the endmember knot spectra
// (rtx_band_mix_stacked), and a per-(band, pixel) kernel with Q Planck evaluations (rtx_pixel_cube).
#define CUBE_QMAX 6
struct BasisArgs {
  int kind, Q;
  GridDev g;
  long long nx;
  const float *tau, *La, *Ld;
  const double* Xk;
  long long nk;
  int nB;
  const double* centre;
  const double* sigma;
  float coef[CUBE_QMAX][CUBE_QMAX];  // l_q(s) = sum_d coef[q][d] s^d
  float node_span;                   // R_b = node_span * sigma_b
  double c2l2e_over_T;               // PLANCK mode (rtx_band_moments): 100*c2*log2(e)/Ts
  float* N;
  float* C;
  float* MLd;  // [nB][nk]  (PLANCK mode: unused)
  float* MB;   // [Q][nB][nk]  (PLANCK mode: M[nB][nk] = sum w tau (B(Ts) - Ld) hat_j)
  int2* jrange;
};

// One workgroup per band; its 16 waves take (knot interval, quarter) tasks. Quarter r of an interval is the points
// first + 64 r + lane + 256 it: counted from the interval's first point, so a band's bits do not depend on where the grid
// shard starts. A wave reduces its sums over the lanes (DPP adds, fixed order) and parks them in LDS; after every round of
// BBM_SEGS intervals one thread per (row, knot) adds the quarters (fixed tree) of the intervals that meet at the knot, in
// interval order, and writes the knot.
// A band is ONE compute unit's work, and the widest MAKO band (14 knot intervals x 1000 points) sets the kernel's time.
// History for C5's 256 bands: 127 us (round 2: the whole workgroup walked the intervals one after the other, 28 barriers per
// interval) -> 48 (a wave per interval) -> 52 (16 waves, quarters, knots counted in one pass instead of two binary searches,
// loads issued ahead: no gain -- stamps inside the kernel: 7.6 us before the first task, 40 us in the tasks of the widest
// band, i.e. ~200 instructions per point, 50 of them fp64, on one CU) -> the per-point quantities (x - c, the interpolation
// weight, the node abscissa) are LINEAR in the point index: one fp64 evaluation per task, an FMA per point; basis polynomials
// of degree Q - 1 (not CUBE_QMAX - 1) with their coefficients in vector registers; knot -> grid index once per round, not per
// task; rows written once by the combine (zeros outside the band's knots stored at the end, off the critical path).
#ifndef CUBE_ABLATE
#define CUBE_ABLATE 0  // timing experiments only (tools/build_variant.sh)
#endif
#define BBM_WAVES 16
#define BBM_SEGS 16
#define BBM_PARTS 4
#define BBM_UNROLL 4
#define BBM_NVMAX (2 * CUBE_QMAX + 4)  // G0[0..Q-1], G1[0..Q-1], G0 and G1 of the Ld moment, N, C

// grid_lower_bound without the search: the index from the grid's arithmetic, then stepped until it is the first with X >= v
__device__ __forceinline__ long long grid_lower_bound_direct(const GridDev& g, long long n, double v) {
  const double r = ceil((v - g.xmin) / g.step) - (double)g.offset;
  long long i = !(r > 0.0) ? 0 : (!(r < (double)n) ? n : (long long)r);  // NaN -> 0 (no conversion of a NaN)
  while (i > 0 && grid_x(g, g.offset + i - 1) >= v) --i;
  while (i < n && grid_x(g, g.offset + i) < v) ++i;
  return i;
}

// sum over the wave, valid in lane 63: shifts within the rows of 16 lanes, then the two row broadcasts (fixed order)
__device__ __forceinline__ float wave_sum63(float v) {
#define RTX_DPP_ADD(ctrl, rows) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rows, 0xf, true))
  RTX_DPP_ADD(0x111, 0xf);  // row_shr:1
  RTX_DPP_ADD(0x112, 0xf);  // row_shr:2
  RTX_DPP_ADD(0x114, 0xf);  // row_shr:4
  RTX_DPP_ADD(0x118, 0xf);  // row_shr:8  -> lane 15 of a row: the row's sum
  RTX_DPP_ADD(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
  RTX_DPP_ADD(0x143, 0xc);  // row_bcast:31 into rows 2 and 3
#undef RTX_DPP_ADD
  return v;
}

// PLANCK (rtx_band_moments, config C4; Q = 1): the one row is M[b][j] = sum w tau (B(nu, Ts) - Ld) hat_j, no Ld row.
template <int KIND, int Q, bool PLANCK>
__global__ __launch_bounds__(64 * BBM_WAVES) void band_basis_moments_kernel(BasisArgs a) {
  static_assert(!PLANCK || Q == 1, "PLANCK mode has one row");
  constexpr int NROW = PLANCK ? 1 : Q + 1;  // rows of knot moments
  constexpr int NV = 2 * NROW + 2, IG0 = 0, IG1 = Q, IL0 = 2 * Q, IL1 = 2 * Q + 1, IN = 2 * NROW, IC = 2 * NROW + 1;
  __shared__ float s_seg[BBM_SEGS][BBM_PARTS][BBM_NVMAX];
  __shared__ int s_j[BBM_SEGS][2];  // j0, j1 of the interval; j0 = -1: no points
  __shared__ double s_xk[BBM_SEGS + 1];
  __shared__ long long s_pk[BBM_SEGS + 1];
  __shared__ int s_cnt[2];
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#if CUBE_ABLATE & 256
  const unsigned long long T0 = wall_clock64();
  unsigned long long T1 = 0, T2 = 0, T3 = 0;
#endif
  // this thread's first knot of the count below, asked for together with the band's centre and width: one trip to memory
  const double xk_mine = (long long)threadIdx.x < a.nk ? a.Xk[threadIdx.x] : 1.0e300;
  const double c = a.centre[b], s = a.sigma[b];
  const double R = KIND == 0 ? s : 14.0 * s;
  const double inv_Rn = 1.0 / ((double)a.node_span * s);
  long long lo = grid_lower_bound_direct(a.g, a.nx, c - R);
  while (lo < a.nx && !(grid_x(a.g, a.g.offset + lo) > c - R)) ++lo;
  const long long hi = grid_lower_bound_direct(a.g, a.nx, c + R);
  if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  // knot intervals jj = -1 (left of the first knot) .. nk-1 (right of the last): np.interp holds the end values.
  // first interval that can contain X[lo]: (#knots <= X[lo]) - 1; last one: (#knots <= X[hi-1]) - 1
  if (lo < hi) {
    const double x_first = grid_x(a.g, a.g.offset + lo), x_last = grid_x(a.g, a.g.offset + hi - 1);
    int c1 = xk_mine <= x_first ? 1 : 0, c2 = xk_mine <= x_last ? 1 : 0;
    for (long long j = threadIdx.x + blockDim.x; j < a.nk; j += blockDim.x) {
      const double xk = a.Xk[j];
      c1 += xk <= x_first ? 1 : 0;
      c2 += xk <= x_last ? 1 : 0;
    }
    for (int off = 32; off > 0; off >>= 1) { c1 += __shfl_down(c1, off); c2 += __shfl_down(c2, off); }
    if (lane == 0) { atomicAdd(&s_cnt[0], c1); atomicAdd(&s_cnt[1], c2); }
  }
  float cv[Q][Q];  // the basis coefficients in vector registers (an FMA with a scalar-register source issues at half rate)
#pragma unroll
  for (int q = 0; q < Q; ++q)
#pragma unroll
    for (int d = 0; d < Q; ++d) asm volatile("v_mov_b32 %0, %1" : "=v"(cv[q][d]) : "s"(a.coef[q][d]));
  const float step_f = (float)a.g.step, inv_s = (float)(1.0 / s), inv_Rn_f = (float)inv_Rn;
  __syncthreads();
  const long long jj_first = lo < hi ? s_cnt[0] - 1 : 0, jj_last = lo < hi ? max(s_cnt[0], s_cnt[1]) - 1 : -2;
#if CUBE_ABLATE & 256
  T1 = wall_clock64();
#endif
  float Nacc = 0.f, Cacc = 0.f;  // the last thread only
  int jfirst = 0x7fffffff, jlast = -1;
  long long w_lo = 0, w_hi = -1;  // knots written by the combine
  for (long long base = jj_first; base <= jj_last; base += BBM_SEGS) {
    if (threadIdx.x <= BBM_SEGS) {
      const double xk = a.Xk[min(max(base + (long long)threadIdx.x, 0ll), a.nk - 1)];
      s_xk[threadIdx.x] = xk;
      s_pk[threadIdx.x] = grid_lower_bound_direct(a.g, a.nx, xk);
    }
    __syncthreads();
    for (int task = wave; task < BBM_SEGS * BBM_PARTS && base + task / BBM_PARTS <= jj_last; task += BBM_WAVES) {
      const int slot = task / BBM_PARTS, part = task % BBM_PARTS;
      const long long jj = base + slot;
      const long long j0 = jj < 0 ? 0 : (jj >= a.nk - 1 ? a.nk - 1 : jj);
      const long long j1 = jj < 0 ? 0 : (jj >= a.nk - 1 ? a.nk - 1 : jj + 1);
      const double x0 = s_xk[slot], dxk = s_xk[slot + 1] - x0;  // Xk[j0], Xk[j1] - Xk[j0]
      const long long p_lo = jj == jj_first ? lo : s_pk[slot];
      const long long p_hi = jj == jj_last ? hi : s_pk[slot + 1];
      const double inv_dxk_d = dxk > 0.0 ? 1.0 / dxk : 0.0;
      float v[NV];
#pragma unroll
      for (int q = 0; q < NV; ++q) v[q] = 0.f;
      // x - c, f = (x - Xk[j0]) / dxk and the node abscissa as linear functions of the offset from the interval's first point
      const double x_ref = grid_x(a.g, a.g.offset + p_lo);
      const float d0 = (float)(x_ref - c), f00 = (float)((x_ref - x0) * inv_dxk_d), fstep = (float)(a.g.step * inv_dxk_d);
      // PLANCK: the exponent nu * kT is linear too; its integer part is split off once per task (fp64), the fraction runs in fp32
      const double t_ref = PLANCK ? x_ref * a.c2l2e_over_T : 0.0, n_ref = rint(t_ref);
      const float tf_ref = (float)(t_ref - n_ref), tstep = (float)(a.g.step * a.c2l2e_over_T), x_ref_f = (float)x_ref;
      const int n_ref_i = (int)n_ref;
      // offsets exact in fp32; PLANCK: exponent >= 1.5 over the interval (else planck_f32's fp64 expm1 branch, point by point)
      const bool linear = KIND == 0 && p_hi - p_lo < (1ll << 24) && (!PLANCK || (t_ref >= 1.6 && t_ref < 120.0));
      if (!(CUBE_ABLATE & 2))
      for (long long ib = p_lo + part * 64; ib < p_hi; ib += 64 * BBM_PARTS * BBM_UNROLL) {
        float tt[BBM_UNROLL], ll[BBM_UNROLL], aa[BBM_UNROLL];
#pragma unroll
        for (int u = 0; u < BBM_UNROLL; ++u) {  // all loads first: one trip to memory per BBM_UNROLL points
          const long long i = ib + u * (64 * BBM_PARTS) + lane;
          const long long ic = i < p_hi ? i : p_lo;
          tt[u] = a.tau[ic]; ll[u] = a.Ld[ic]; aa[u] = a.La[ic];
        }
        const int rel0 = (int)(ib - p_lo) + lane;
#pragma unroll
        for (int u = 0; u < BBM_UNROLL; ++u) {
          const long long i = ib + u * (64 * BBM_PARTS) + lane;
          const bool ok = i < p_hi;
          float w, f, sn, B = 0.f;
          if (linear) {
            const float uf = (float)(rel0 + u * (64 * BBM_PARTS));
            const float d = fmaf(uf, step_f, d0);
            w = fmaxf(fmaf(-fabsf(d), inv_s, 1.0f), 0.f);  // tri(), :1236-1239
            f = fmaf(uf, fstep, f00);
            sn = d * inv_Rn_f;
            if (PLANCK) {
              const float xf = fmaf(uf, step_f, x_ref_f);
              const float e = ldexpf(__builtin_amdgcn_exp2f(fmaf(uf, tstep, tf_ref)), n_ref_i);
              B = ((float)(RT_C1 * 1e10) * xf) * (xf * xf) * __builtin_amdgcn_rcpf(e - 1.0f);  // c1 (100 nu)^3 1e4 / (e - 1)
            }
          } else {
            const double x = grid_x(a.g, a.g.offset + i);
            w = ils_weight(KIND, x, c, s);
            f = (float)((x - x0) * inv_dxk_d);
            sn = (float)((x - c) * inv_Rn);
            if (PLANCK) {
              const double x100 = x * 100.0;
              B = planck_f32(RT_C1 * (x100 * x100 * x100) * 1e4, x, a.c2l2e_over_T);
            }
          }
          w = ok ? w : 0.f;  // past the interval's end: the first point again, with weight 0
          const float t = tt[u], ld = ll[u];
          const float wt = w * t, f1 = 1.0f - f;
          v[IN] += w;
          v[IC] = fmaf(w, fmaf(t, ld, aa[u]), v[IC]);
          if (PLANCK) {
            const float gi = wt * (B - ld);
            v[IG1] = fmaf(gi, f, v[IG1]);
            v[IG0] = fmaf(gi, f1, v[IG0]);
          } else {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
              float l = cv[q][Q - 1];
#pragma unroll
              for (int d = Q - 2; d >= 0; --d) l = fmaf(l, sn, cv[q][d]);
              const float gq = wt * l;
              v[IG1 + q] = fmaf(gq, f, v[IG1 + q]);
              v[IG0 + q] = fmaf(gq, f1, v[IG0 + q]);
            }
            const float gl = wt * ld;
            v[IL1] = fmaf(gl, f, v[IL1]);
            v[IL0] = fmaf(gl, f1, v[IL0]);
          }
        }
      }
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const float r = wave_sum63(v[q]);
        if (lane == 63) s_seg[slot][part][q] = r;
      }
      if (lane == 0 && part == 0) {
        s_j[slot][0] = p_hi > p_lo ? (int)j0 : -1;
        s_j[slot][1] = (int)j1;
      }
    }
    __syncthreads();
#if CUBE_ABLATE & 256
    if (!T2) T2 = wall_clock64();
#endif
    const int n_slots = (int)min((long long)BBM_SEGS, jj_last - base + 1);
    const long long jb = base < 0 ? 0 : base;
    if (base == jj_first) w_lo = jb;
    w_hi = min(a.nk - 1, jb + n_slots);
    // knot j of this round collects, in interval order, the j0-sum of the intervals that start at it and the j1-sum of those
    // that end at it (the order one thread walking the intervals would add them in); one thread per (row, knot)
    if (CUBE_ABLATE & 4) {
    } else if (threadIdx.x < (unsigned)(NROW * (BBM_SEGS + 1))) {
      const int qi = threadIdx.x / (BBM_SEGS + 1), jl = threadIdx.x % (BBM_SEGS + 1);
      const int i0 = qi == Q ? IL0 : IG0 + qi, i1 = qi == Q ? IL1 : IG1 + qi;  // rows 0..Q-1: the basis moments, row Q: Ld
      const long long j = jb + jl;
      float acc = 0.f;
      for (int sl = 0; sl < n_slots; ++sl) {
        const int j0 = s_j[sl][0];
        if (j0 < 0) continue;
        if (j0 == j) acc += (s_seg[sl][0][i0] + s_seg[sl][1][i0]) + (s_seg[sl][2][i0] + s_seg[sl][3][i0]);
        if (s_j[sl][1] == j) acc += (s_seg[sl][0][i1] + s_seg[sl][1][i1]) + (s_seg[sl][2][i1] + s_seg[sl][3][i1]);
      }
      if (jl <= n_slots && j <= w_hi) {
        float* row = (qi == Q) ? a.MLd + (size_t)b * a.nk : a.MB + ((size_t)qi * a.nB + b) * a.nk;
        if (base != jj_first && jl == 0) row[j] += acc; else row[j] = acc;  // the knot shared with the previous round
      }
    } else if (threadIdx.x == 64 * BBM_WAVES - 1) {
      for (int sl = 0; sl < n_slots; ++sl) {
        const int j0 = s_j[sl][0];
        if (j0 < 0) continue;
        Nacc += (s_seg[sl][0][IN] + s_seg[sl][1][IN]) + (s_seg[sl][2][IN] + s_seg[sl][3][IN]);
        Cacc += (s_seg[sl][0][IC] + s_seg[sl][1][IC]) + (s_seg[sl][2][IC] + s_seg[sl][3][IC]);
        jfirst = min(jfirst, j0);
        jlast = max(jlast, s_j[sl][1]);
      }
    }
    __syncthreads();
  }
  // zeros on the knots the band does not reach
  for (long long j = threadIdx.x; j < a.nk; j += blockDim.x) {
    if (j >= w_lo && j <= w_hi) continue;
    if (!PLANCK) a.MLd[(size_t)b * a.nk + j] = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) a.MB[((size_t)q * a.nB + b) * a.nk + j] = 0.f;
  }
  if (threadIdx.x == 64 * BBM_WAVES - 1) {
    a.N[b] = Nacc;
    a.C[b] = Cacc;
    a.jrange[b] = make_int2(jfirst == 0x7fffffff ? 0 : jfirst, jlast);
#if CUBE_ABLATE & 256
    T3 = wall_clock64();  // 100 MHz ticks
    a.N[b] = (float)(T1 - T0); a.C[b] = (float)(T2 - T0);
    a.jrange[b] = make_int2((int)(T3 - T0), (int)(jj_last - jj_first + 1));
#endif
  }
}

template <int KIND>
static void launch_basis_moments(const BasisArgs& a, hipStream_t stream) {
  const dim3 grid(a.nB), block(64 * BBM_WAVES);
  switch (a.Q) {
    case 1: hipLaunchKernelGGL((band_basis_moments_kernel<KIND, 1, false>), grid, block, 0, stream, a); break;
    case 2: hipLaunchKernelGGL((band_basis_moments_kernel<KIND, 2, false>), grid, block, 0, stream, a); break;
    case 3: hipLaunchKernelGGL((band_basis_moments_kernel<KIND, 3, false>), grid, block, 0, stream, a); break;
    case 4: hipLaunchKernelGGL((band_basis_moments_kernel<KIND, 4, false>), grid, block, 0, stream, a); break;
    case 5: hipLaunchKernelGGL((band_basis_moments_kernel<KIND, 5, false>), grid, block, 0, stream, a); break;
    default: hipLaunchKernelGGL((band_basis_moments_kernel<KIND, 6, false>), grid, block, 0, stream, a); break;
  }
}

extern "C" int rtx_band_moments(int kind, const rtx_grid* grid, const float* tau, const float* La, const float* Ld, double Ts,
                                const double* Xk, int64_t nk, int nB, const double* centre, const double* sigma,
                                float* N_out, float* C_out, float* M_out, int32_t* jrange_out, void* stream) {
  if (kind != 0 && kind != 1) RTX_FAIL("kind must be 0 (triangle) or 1 (Gaussian)");
  if (rtx_check_grid(grid)) return 1;
  if (nk < 2) RTX_FAIL("need at least 2 knots");
  if (nB < 0) RTX_FAIL("negative size");
  if (nB == 0) return 0;
  if (!(Ts > 0.0)) RTX_FAIL("surface temperature %g", Ts);
  if (!tau || !La || !Ld || !Xk || !centre || !sigma || !N_out || !C_out || !M_out || !jrange_out) RTX_FAIL("a required pointer is NULL");
  // the C5 kernel in its PLANCK mode: one row, integrand w tau (B(nu, Ts) - Ld)
  BasisArgs a;
  a.kind = kind; a.Q = 1; a.g = to_dev(grid); a.nx = grid->n; a.tau = tau; a.La = La; a.Ld = Ld; a.Xk = Xk; a.nk = nk; a.nB = nB;
  a.centre = centre; a.sigma = sigma; a.node_span = 1.0f;
  for (int q = 0; q < CUBE_QMAX; ++q)
    for (int d = 0; d < CUBE_QMAX; ++d) a.coef[q][d] = 0.f;
  a.c2l2e_over_T = 100.0 * RT_C2 * 1.4426950408889634 / Ts;
  a.N = N_out; a.C = C_out; a.MLd = nullptr; a.MB = M_out; a.jrange = reinterpret_cast<int2*>(jrange_out);
  const dim3 g(nB), blk(64 * BBM_WAVES);
  if (kind == 0) hipLaunchKernelGGL((band_basis_moments_kernel<0, 1, true>), g, blk, 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((band_basis_moments_kernel<1, 1, true>), g, blk, 0, (hipStream_t)stream, a);
  RTX_LAUNCH_CHECK();
  return 0;
}

extern "C" int rtx_band_basis_moments(int kind, const rtx_grid* grid, const float* tau, const float* La, const float* Ld,
                                      const double* Xk, int64_t nk, int nB, const double* centre, const double* sigma, int Q,
                                      const float* basis_coef_h, double node_span, float* N_out, float* C_out, float* MLd_out,
                                      float* MB_out, int32_t* jrange_out, void* stream) {
  if (kind != 0 && kind != 1) RTX_FAIL("kind must be 0 (triangle) or 1 (Gaussian)");
  if (rtx_check_grid(grid)) return 1;
  if (nk < 2) RTX_FAIL("need at least 2 knots");
  if (Q < 1 || Q > CUBE_QMAX) RTX_FAIL("Q=%d outside [1,%d]", Q, CUBE_QMAX);
  if (nB < 0) RTX_FAIL("negative size");
  if (nB == 0) return 0;
  if (!(node_span > 0.0)) RTX_FAIL("node_span must be > 0");
  if (!tau || !La || !Ld || !Xk || !centre || !sigma || !basis_coef_h || !N_out || !C_out || !MLd_out || !MB_out || !jrange_out)
    RTX_FAIL("a required pointer is NULL");
  BasisArgs a;
  a.kind = kind; a.Q = Q; a.g = to_dev(grid); a.nx = grid->n; a.tau = tau; a.La = La; a.Ld = Ld; a.Xk = Xk; a.nk = nk; a.nB = nB;
  a.centre = centre; a.sigma = sigma; a.node_span = (float)node_span; a.c2l2e_over_T = 0.0;
  for (int q = 0; q < CUBE_QMAX; ++q)
    for (int d = 0; d < CUBE_QMAX; ++d) a.coef[q][d] = (q < Q && d < Q) ? basis_coef_h[q * Q + d] : 0.f;
  a.N = N_out; a.C = C_out; a.MLd = MLd_out; a.MB = MB_out; a.jrange = reinterpret_cast<int2*>(jrange_out);
  if (kind == 0) launch_basis_moments<0>(a, (hipStream_t)stream); else launch_basis_moments<1>(a, (hipStream_t)stream);
  RTX_LAUNCH_CHECK();
  return 0;
}

// The (Q + 1) contractions of one cube in ONE launch: M[n_stack][nB][nk] (rows 0..Q-1: MB[q], row Q: MLd) with the endmember
// knot spectra E[nk][nE] -> tab[nE][n_stack][nB], the layout pixel_cube_kernel reads with lane = band. lane = endmember.
struct MixStackArgs {
  const float* M;
  const int2* jrange;
  int nB, n_stack;
  long long nk, nE;
  const float* E;
  float* tab;
};

__global__ __launch_bounds__(64) void band_mix_stacked_kernel(MixStackArgs a) {
  const int b = blockIdx.x;
  const long long k = (long long)blockIdx.y * 64 + threadIdx.x;
  if (k >= a.nE) return;
  const int2 jr = a.jrange[b];
  float acc[CUBE_QMAX + 1];
#pragma unroll
  for (int s = 0; s <= CUBE_QMAX; ++s) acc[s] = 0.f;
  for (int j = jr.x; j <= jr.y; ++j) {
    const float e = a.E[(size_t)j * a.nE + k];
#pragma unroll
    for (int s = 0; s <= CUBE_QMAX; ++s)
      if (s < a.n_stack) acc[s] = fmaf(a.M[((size_t)s * a.nB + b) * a.nk + j], e, acc[s]);
  }
#pragma unroll
  for (int s = 0; s <= CUBE_QMAX; ++s)
    if (s < a.n_stack) a.tab[((size_t)k * a.n_stack + s) * a.nB + b] = acc[s];
}

extern "C" int rtx_band_mix_stacked(const float* M, const int32_t* jrange, int nB, int n_stack, int64_t nk, const float* E,
                                    int64_t nE, float* tab, void* stream) {
  if (nB < 0 || nE < 0 || nk < 1) RTX_FAIL("bad size");
  if (n_stack < 1 || n_stack > CUBE_QMAX + 1) RTX_FAIL("n_stack=%d outside [1,%d]", n_stack, CUBE_QMAX + 1);
  if (nB == 0 || nE == 0) return 0;
  if (!M || !jrange || !E || !tab) RTX_FAIL("a required pointer is NULL");
  MixStackArgs a;
  a.M = M; a.jrange = reinterpret_cast<const int2*>(jrange); a.nB = nB; a.n_stack = n_stack; a.nk = nk; a.nE = nE; a.E = E; a.tab = tab;
  hipLaunchKernelGGL(band_mix_stacked_kernel, dim3(nB, (unsigned)((nE + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a);
  RTX_LAUNCH_CHECK();
  return 0;
}

struct CubeArgs {
  int nB, nEnd, nMix;
  long long nPix;
  const double* centre;
  const double* sigma;
  float node_span;
  float s_node[CUBE_QMAX];
  const float *N, *C;
  const float* tab;    // [nEnd][Q+1][nB]: rows 0..Q-1 the Planck-node tables AB[q], row Q the Ld table ALd
  const int* kidx;     // [nPix][nMix]
  const float* frac;   // [nPix][nMix]
  const double* Tpix;  // [nPix]
  float* cube;         // [nB][nPix]
};

// lane = band, a wave walks pixels. Everything a pixel owns (temperature, mixture) is wave-uniform -- scalar loads, one fp64
// division per pixel and workgroup -- and everything a band owns (node abscissae, c1 nu^3, C_b, 1/N_b) lives in the lane's
// registers for the 256 pixels of the workgroup; the tables are read with consecutive lanes on consecutive words out of LDS
// (or, when 64 bands x nEnd x (Q+1) words do not fit, out of L1/L2 with the same indexing). Round 2 had lane = pixel and one
// band per workgroup: 18 gathers, five fp64 Planck exponents and an fp64 division per (band, pixel) -- 146 us for C5, where
// the cube itself is 8 us of HBM writes.
// Planck at node q: log2 of the exponential = nu_q kT with kT = 100 c2 log2(e) / T_p.  nu_q = c_b + R_b s_q, so it is
// c_b kT (fp64: this is where the digits are) + (R_b s_q) kT (fp32: |R_b kT| is a few hundredths for a MAKO band, its
// rounding error 1e-9 of the exponent); the integer part of c_b kT is split off once per (band, pixel).
// The (band, pixel) values go through a 64 x 64 LDS tile so that the cube is written in 256-byte rows along the pixel axis.
#define CUBE_PB 256  // pixels per workgroup
#define CUBE_STAGE_MIX 4  // mixtures of up to this many endmembers ride in the lanes (16 pixels x nMix <= 64)

// One (band, pixel) value. FAST: every node of every (band, pixel) of the workgroup has exponent >= 1.5 (checked once per
// workgroup), the exponential in fp32 from the two-float product; otherwise per pixel, with planck_f32's fp64 branch.
template <int Q, bool FAST>
__device__ __forceinline__ void cube_planck(float (&Bq)[Q], const float (&c1x3)[Q], const float (&dq)[Q], float c_hi, float c_lo,
                                            float dmax, float2 kf, double kT, double c, double Rn, const float* s_node) {
  // c_b kT in two-float arithmetic: c = c_hi + c_lo, kT = k_hi + k_lo; c_hi k_hi = pr + er exactly (FMA), and its integer
  // part leaves pr exactly
  const float pr = c_hi * kf.x, er = fmaf(c_hi, kf.x, -pr);
  if (FAST || __ballot(pr - dmax * kf.x < 1.5f) == 0) {
    const float n = rintf(pr);
    const float f0 = (pr - n) + fmaf(c_hi, kf.y, fmaf(c_lo, kf.x, er));
    const int ni = (int)n;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const float e = ldexpf(__builtin_amdgcn_exp2f(fmaf(dq[q], kf.x, f0)), ni);  // inf above 2^128: B -> 0
      Bq[q] = c1x3[q] * __builtin_amdgcn_rcpf(e - 1.0f);
    }
  } else {  // far-infrared nodes: expm1 in fp64 (planck_f32's small-argument branch)
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const double x = c + Rn * (double)s_node[q], x100 = x * 100.0;
      Bq[q] = planck_f32(RT_C1 * (x100 * x100 * x100) * 1e4, x, kT);
    }
  }
}

template <int Q, bool TAB_LDS, bool STAGED>
__global__ __launch_bounds__(256) void pixel_cube_kernel(CubeArgs a) {
  extern __shared__ float s_tab[];  // TAB_LDS: [nEnd][Q+1][64]
  __shared__ float s_tile[64][65];
  __shared__ double s_kT[CUBE_PB];
  __shared__ float2 s_kTf[CUBE_PB];  // kT as hi + lo floats
  __shared__ float s_kmin[4];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b0 = blockIdx.y * 64;
  const int b = min(b0 + lane, a.nB - 1);
  const long long p0 = (long long)blockIdx.x * CUBE_PB;
  const int n_pl = (int)min((long long)CUBE_PB, a.nPix - p0);  // pixels of this workgroup (>= 1)
  const int nMix = a.nMix;
  {
    const double kT = 100.0 * RT_C2 * 1.4426950408889634 / a.Tpix[p0 + min((int)threadIdx.x, n_pl - 1)];
    s_kT[threadIdx.x] = kT;
    const float hi = (float)kT;
    s_kTf[threadIdx.x] = make_float2(hi, (float)(kT - (double)hi));
    float m = hi == hi ? hi : -1.0f;  // NaN temperature -> the per-pixel path
    for (int off = 32; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off));
    if (lane == 0) s_kmin[wave] = m;
  }
  // the mixtures of this wave's 16 pixels of a round, one (pixel, endmember slot) per lane, loaded one round ahead: the loop
  // below takes them out of the lanes with v_readlane
  const int st_i = STAGED ? min(lane / nMix, 15) : 0, st_m = STAGED ? lane - (lane / nMix) * nMix : 0;
  int kv = 0, kv_n = 0;
  float fv = 0.f, fv_n = 0.f;
  if (STAGED) {
    const long long p = p0 + min(wave * 16 + st_i, n_pl - 1);
    kv_n = a.kidx[p * nMix + st_m];
    fv_n = a.frac[p * nMix + st_m];
  }
  if (TAB_LDS && !(CUBE_ABLATE & 64)) {
    const int n = a.nEnd * (Q + 1) * 64;
    for (int i = threadIdx.x; i < n; i += 256) s_tab[i] = a.tab[(size_t)(i >> 6) * a.nB + min(b0 + (i & 63), a.nB - 1)];
  }
  const double c = a.centre[b], Rn = (double)a.node_span * a.sigma[b];
  float c1x3[Q], dq[Q];
  float dmax = 0.f;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const double x100 = (c + Rn * (double)a.s_node[q]) * 100.0;
    c1x3[q] = (float)(RT_C1 * (x100 * x100 * x100) * 1e4);
    dq[q] = (float)(Rn * (double)a.s_node[q]);
    dmax = fmaxf(dmax, fabsf(dq[q]));
  }
  const float c_hi = (float)c, c_lo = (float)(c - (double)c_hi);
  const float Cb = a.C[b], invN = 1.0f / a.N[b];  // N = 0 -> 0 * inf = NaN, as the unfused path
  const float* tab = TAB_LDS ? s_tab + lane : a.tab + b;
  const size_t ts = TAB_LDS ? 64 : (size_t)a.nB;  // words between a table's rows
  const int k_max = a.nEnd - 1;
  __syncthreads();
  // the smallest exponent any node of this workgroup can have: (c - dmax) * the smallest kT, with a margin for the roundings
  const float kmin = fminf(fminf(s_kmin[0], s_kmin[1]), fminf(s_kmin[2], s_kmin[3]));
  const bool fast = __syncthreads_and((c_hi - dmax) * kmin >= 1.51f && dmax < c_hi);
  for (int g = 0; g < CUBE_PB / 64; ++g) {
    if (g * 64 < n_pl) {  // uniform over the workgroup
      kv = kv_n; fv = fv_n;
      if (STAGED && (g + 1) * 64 < n_pl) {
        const long long p = p0 + min((g + 1) * 64 + wave * 16 + st_i, n_pl - 1);
        kv_n = a.kidx[p * nMix + st_m];
        fv_n = a.frac[p * nMix + st_m];
      }
      for (int i = 0; i < 16; ++i) {  // (unrolled by 2 or 4: the same 46-47 us)
        const int pl = wave * 16 + i;
        const int px = min(g * 64 + pl, n_pl - 1);  // past the end: the last pixel again, not written
        float Bq[Q];
        if (CUBE_ABLATE & 16) {
#pragma unroll
          for (int q = 0; q < Q; ++q) Bq[q] = c1x3[q] * s_kTf[px].x;
        } else if (fast) {
          cube_planck<Q, true>(Bq, c1x3, dq, c_hi, c_lo, dmax, s_kTf[px], 0.0, c, Rn, a.s_node);
        } else {
          cube_planck<Q, false>(Bq, c1x3, dq, c_hi, c_lo, dmax, s_kTf[px], s_kT[px], c, Rn, a.s_node);
        }
        float acc = Cb;
        if (CUBE_ABLATE & 32) acc += Bq[0] + Bq[Q - 1] + Bq[Q / 2]; else
        for (int m = 0; m < nMix; ++m) {
          int k;
          float fr;
          if (STAGED) {
            k = __builtin_amdgcn_readlane(kv, i * nMix + m);
            fr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fv), i * nMix + m));
          } else {
            k = a.kidx[(p0 + px) * nMix + m];
            fr = a.frac[(p0 + px) * nMix + m];
          }
          k = min(max(k, 0), k_max);  // an index outside the table must not leave it
          const float* row = tab + (size_t)k * (Q + 1) * ts;
          float t = -row[Q * ts];
#pragma unroll
          for (int q = 0; q < Q; ++q) t = fmaf(Bq[q], row[q * ts], t);
          acc = fmaf(fr, t, acc);
        }
        s_tile[lane][pl] = acc * invN;
      }
    }
    __syncthreads();
    if (g * 64 + lane < n_pl) {
      float* dst = a.cube + (size_t)b0 * a.nPix + p0 + g * 64 + lane;
#pragma unroll 4
      for (int r = wave * 16; r < wave * 16 + 16; ++r)
        if (b0 + r < a.nB && (!(CUBE_ABLATE & 128) || s_tile[r][lane] == 123.f)) dst[(size_t)r * a.nPix] = s_tile[r][lane];
    }
    __syncthreads();
  }
}

template <int Q>
static void launch_pixel_cube(const CubeArgs& a, hipStream_t stream) {
  const dim3 grid((unsigned)((a.nPix + CUBE_PB - 1) / CUBE_PB), (unsigned)((a.nB + 63) / 64));
  const size_t tab_bytes = (size_t)a.nEnd * (Q + 1) * 64 * sizeof(float);
  const bool staged = a.nMix <= CUBE_STAGE_MIX;
  if (tab_bytes <= 40 * 1024) {  // + 21 KB of static LDS: within the 64 KB every HIP launch may ask for
    if (staged) hipLaunchKernelGGL((pixel_cube_kernel<Q, true, true>), grid, dim3(256), tab_bytes, stream, a);
    else hipLaunchKernelGGL((pixel_cube_kernel<Q, true, false>), grid, dim3(256), tab_bytes, stream, a);
  } else {
    if (staged) hipLaunchKernelGGL((pixel_cube_kernel<Q, false, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((pixel_cube_kernel<Q, false, false>), grid, dim3(256), 0, stream, a);
  }
}

extern "C" int rtx_pixel_cube(int nB, int Q, const double* centre, const double* sigma, double node_span, const float* s_node_h,
                              const float* N, const float* C, const float* tab, int nEnd, int64_t nPix, int nMix,
                              const int32_t* kidx, const float* frac, const double* Tpix, float* cube, void* stream) {
  if (Q < 1 || Q > CUBE_QMAX) RTX_FAIL("Q=%d outside [1,%d]", Q, CUBE_QMAX);
  if (nB < 0 || nPix < 0 || nEnd < 1 || nMix < 1) RTX_FAIL("bad size");
  if (nB == 0 || nPix == 0) return 0;
  if (!centre || !sigma || !s_node_h || !N || !C || !tab || !kidx || !frac || !Tpix || !cube) RTX_FAIL("a required pointer is NULL");
  CubeArgs a;
  a.nB = nB; a.nEnd = nEnd; a.nMix = nMix; a.nPix = nPix; a.centre = centre; a.sigma = sigma; a.node_span = (float)node_span;
  for (int q = 0; q < CUBE_QMAX; ++q) a.s_node[q] = q < Q ? s_node_h[q] : 0.f;
  a.N = N; a.C = C; a.tab = tab; a.kidx = kidx; a.frac = frac; a.Tpix = Tpix; a.cube = cube;
  switch (Q) {
    case 1: launch_pixel_cube<1>(a, (hipStream_t)stream); break;
    case 2: launch_pixel_cube<2>(a, (hipStream_t)stream); break;
    case 3: launch_pixel_cube<3>(a, (hipStream_t)stream); break;
    case 4: launch_pixel_cube<4>(a, (hipStream_t)stream); break;
    case 5: launch_pixel_cube<5>(a, (hipStream_t)stream); break;
    default: launch_pixel_cube<6>(a, (hipStream_t)stream); break;
  }
  RTX_LAUNCH_CHECK();
  return 0;
}
