// Voigt line-sum: gather formulation of the reference's per-line scatter-add
// (misc/hapi.py:11050, 11135-11138; PROFILE_VOIGT :10131 -> pcqsdhc PART1 :9900-9915 ->
// hum1_wei :9833-9844).
//
// Mapping (CDNA4): one workgroup = 4 waves = one tile of 4*64*P consecutive grid points of one layer.
// Lane <-> grid point (coalesced along the wavenumber axis), each lane owns P points 64 apart, so
// a "row" of 64 consecutive points is one wave-instruction wide and every per-line decision
// (window edge, Weideman zone) is wave-uniform per row -- no divergence inside a row.
// The lines that can reach the tile are culled, order-preserving, into LDS by the whole workgroup
// (ballot + prefix), then every wave walks the LDS list with broadcast reads.
//
// Arithmetic: the far-wing branch of hum1_wei (|x|+y >= 15; >95 % of evaluations in the troposphere)
// is evaluated in fp32 from a grid-relative argument (integer index difference times step*cte plus
// a sub-grid residual), which keeps (nu - nu0) exact to ~1e-7 relative; the region test and the
// Weideman-24 branch use the fp64 record (fp64 polynomial when y<1, where fp32 loses Re w).
#include "rtx_common.h"

#include "w24_coeffs.inc"
#define INV_SQRT_PI 0.56418958354775628

// Re w(x+iy) by Weideman's rational expansion (misc/hapi.py:9812-9827), real arithmetic.
template <typename F>
__device__ __forceinline__ F weideman_re(F x, F y, const F* __restrict__ coef) {
  const F L = (F)W24_L;
  // d = L - i z = (L+y) - i x ;  n = L + i z = (L-y) + i x ;  Z = n/d
  const F dr = L + y, di = -x;
  const F nr = L - y, ni = x;
  const F inv = (F)1 / (dr * dr + di * di);
  const F Zr = (nr * dr + ni * di) * inv;
  const F Zi = (ni * dr - nr * di) * inv;
  F pr = coef[0], pi = (F)0;
#pragma unroll
  for (int k = 1; k < 24; ++k) {
    const F tr = pr * Zr - pi * Zi + coef[k];
    const F ti = pr * Zi + pi * Zr;
    pr = tr;
    pi = ti;
  }
  // 1/d = conj(d)*inv ; w = 2 p /d^2 + (1/sqrt(pi))/d
  const F ir = dr * inv, ii = -di * inv;
  const F i2r = ir * ir - ii * ii, i2i = (F)2 * ir * ii;
  return (F)2 * (pr * i2r - pi * i2i) + (F)INV_SQRT_PI * ir;
}

struct VsArgs {
  const LineRec* rec;      // [n_layers][n_lines]
  const LineRec64* rec64;  // [n_layers][n_lines]
  const int2* ranges;      // [n_layers][n_tiles] candidate line range per tile
  long long n_lines;
  int n_tiles;
  int tiles_per_xcd;
  GridDev g;
  float* out32;
  double* out64;
  long long ld;
  double inv_scale;
};

struct RangeArgs {
  const int* ic;
  const int* maxhw;
  long long n_lines;
  int n_tiles, tile, n_layers;
  long long n;
  int2* ranges;
};

// Candidate lines of tile t in layer k: unshifted-centre index within maxhw[k] of the tile.
__global__ __launch_bounds__(256) void tile_ranges_kernel(RangeArgs a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  if (t >= a.n_tiles) return;
  const long long ia = (long long)t * a.tile;
  long long ib = ia + a.tile;
  if (ib > a.n) ib = a.n;
  const long long hw = a.maxhw[k];
  const long long vlo = ia - hw, vhi = ib - 1 + hw;
  // lower_bound(ic, vlo)
  long long lo = 0, hi = a.n_lines;
  while (lo < hi) {
    long long mid = (lo + hi) >> 1;
    if ((long long)a.ic[mid] < vlo) lo = mid + 1; else hi = mid;
  }
  const long long l0 = lo;
  hi = a.n_lines;  // upper_bound(ic, vhi), starting from l0
  while (lo < hi) {
    long long mid = (lo + hi) >> 1;
    if ((long long)a.ic[mid] <= vhi) lo = mid + 1; else hi = mid;
  }
  a.ranges[(size_t)k * a.n_tiles + t] = make_int2((int)l0, (int)lo);
}

template <int P>
__global__ __launch_bounds__(256) void voigt_sum_kernel(VsArgs a) {
  constexpr int WPTS = 64 * P;    // points per wave
  constexpr int TILE = 4 * WPTS;  // points per workgroup
  constexpr int CHUNK = 256;
  __shared__ LineRec s_rec[CHUNK];
  __shared__ int s_idx[CHUNK];
  __shared__ int s_wcount[4];

  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD one
  // contiguous run of tiles -- neighbouring tiles share most of their line records in that XCD's L2.
  const int b = blockIdx.x;
  const int tile = (b & 7) * a.tiles_per_xcd + (b >> 3);
  if (tile >= a.n_tiles) return;  // whole workgroup exits together
  const int k = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long n = a.g.n;
  const int ia = tile * TILE;
  const int ib = (int)((long long)ia + TILE < n ? (long long)ia + TILE : n);
  const int wa = ia + wave * WPTS;
  const int wb = wa + WPTS < ib ? wa + WPTS : ib;  // may be <= wa for a tail wave: it then skips everything
  const LineRec* __restrict__ rec = a.rec + (size_t)k * (size_t)a.n_lines;
  const LineRec64* __restrict__ rec64 = a.rec64 + (size_t)k * (size_t)a.n_lines;
  const int2 rng = a.ranges[(size_t)k * a.n_tiles + tile];

  float acc[P];
  float lanef[P];
#pragma unroll
  for (int r = 0; r < P; ++r) {
    acc[r] = 0.f;
    lanef[r] = (float)(lane + 64 * r);
  }

  for (int base = rng.x; base < rng.y; base += CHUNK) {
    // ---- stage: every thread fetches one candidate, keeps it if its window meets the tile ----
    const int l = base + (int)threadIdx.x;
    LineRec r;
    bool keep = false;
    if (l < rng.y) {
      r = rec[l];
      keep = (r.lo < ib) && (r.hi > ia);
    }
    const unsigned long long m = __ballot(keep);
    const int pos = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wcount[wave] = __popcll(m);
    __syncthreads();
    int off = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) off += (w < wave) ? s_wcount[w] : 0;
    const int total = s_wcount[0] + s_wcount[1] + s_wcount[2] + s_wcount[3];
    if (keep) {
      s_rec[off + pos] = r;
      s_idx[off + pos] = l;
    }
    __syncthreads();

    // ---- every wave walks the culled list (LDS broadcast reads) --------------------------------
    for (int j = 0; j < total && wb > wa; ++j) {
      const LineRec q = s_rec[j];
      if (q.hi <= wa || q.lo >= wb) continue;  // wave-uniform
      const bool inside = (q.lo <= wa) && (q.hi >= wb);
      const bool zone = (q.zw > 0) && (q.i0 + q.zw >= wa) && (q.i0 - q.zw < wb);
      // per-line constants of the asymptote  Re[(1/sqrt(pi)) t/(1/2+t^2)], t = y - i x  (:9834-9835):
      //   = (1/sqrt(pi)) y (x^2+y^2+1/2) / (x^4 + x^2 (2y^2-1) + (y^2+1/2)^2)
      const float y2 = q.y * q.y;
      const float b1 = 2.f * y2 - 1.f;
      const float yh = y2 + 0.5f;
      const float b0 = yh * yh;
      const float Ay = q.A * q.y * (float)INV_SQRT_PI;
      const float Ay0 = Ay * yh;
      const float kf = (float)((long long)wa - (long long)q.i0);  // exact: |wa - i0| < 2^24 for any sane window
      if (inside && !zone) {
#pragma unroll
        for (int r = 0; r < P; ++r) {
          const float u = kf + lanef[r];  // integer-valued, exact
          const float x = fmaf(u, q.a, q.c);
          const float xx = x * x;
          const float den = fmaf(xx + b1, xx, b0);
          const float num = fmaf(xx, Ay, Ay0);
          acc[r] = fmaf(num, __builtin_amdgcn_rcpf(den), acc[r]);
        }
      } else {
        // edge of the window and/or rows that may enter the Weideman region: row by row
        for (int r = 0; r < P; ++r) {
          const int ra = wa + 64 * r;
          if (q.hi <= ra || q.lo >= ra + 64) continue;  // wave-uniform
          const int i = ra + lane;
          const bool in_win = (i >= q.lo) && (i < q.hi);
          const float u = kf + (float)(lane + 64 * r);
          const float x = fmaf(u, q.a, q.c);
          const float xx = x * x;
          // same num * rcp(den) + acc as the fast path, so a point gets the same bits whichever path
          // (i.e. whichever tiling / wavenumber shard) reaches it
          float num = fmaf(xx, Ay, Ay0);
          float rden = __builtin_amdgcn_rcpf(fmaf(xx + b1, xx, b0));
          const bool zrow = (q.zw > 0) && (q.i0 + q.zw >= ra) && (q.i0 - q.zw < ra + 64);
          if (zrow) {
            // region test exactly as the reference forms it (fp64): x = -Im Z1 = -((sg0 - sg)*cte)
            const LineRec64 Q = rec64[s_idx[j]];
            const double sg = grid_x(a.g, a.g.offset + (long long)i);
            const double x64 = -((Q.sg0 - sg) * Q.cte);
            const bool wz = (fabs(x64) + Q.y < 15.0);
            if (wz) {
              if (Q.y < 1.0)
                num = (float)(Q.A * weideman_re<double>(x64, Q.y, W24D));
              else
                num = q.A * weideman_re<float>((float)x64, q.y, W24F);
              rden = 1.0f;
            }
          }
          num = in_win ? num : 0.f;
#pragma unroll
          for (int rr = 0; rr < P; ++rr) acc[rr] = (rr == r) ? fmaf(num, rden, acc[rr]) : acc[rr];
        }
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int r = 0; r < P; ++r) {
    const long long i = (long long)wa + 64 * r + lane;
    if (i < (long long)wb) {
      const size_t o = (size_t)k * (size_t)a.ld + (size_t)i;
      if (a.out32) a.out32[o] = acc[r];
      if (a.out64) a.out64[o] = (double)acc[r] * a.inv_scale;
    }
  }
}

#ifndef RTX_VOIGT_P
#define RTX_VOIGT_P 8
#endif

extern "C" int rtx_voigt_tile_points(void) { return 4 * 64 * RTX_VOIGT_P; }

extern "C" int rtx_voigt_sum(const rtx_prep* P, const rtx_grid* grid, int n_layers, float* out_f32, double* out_f64,
                             int64_t ld, void* stream) {
  if (!P) RTX_FAIL("prep is NULL");
  if (rtx_check_grid(grid)) return 1;
  if (n_layers < 1 || n_layers != P->n_layers) RTX_FAIL("n_layers=%d does not match the last rtx_line_prep (%d)", n_layers, P->n_layers);
  if (!out_f32 && !out_f64) RTX_FAIL("both outputs are NULL");
  if (ld < grid->n) RTX_FAIL("ld=%lld < n=%lld", (long long)ld, (long long)grid->n);
  if (grid->n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  constexpr int TILE = 4 * 64 * RTX_VOIGT_P;
  const long long n_tiles_ll = (grid->n + TILE - 1) / TILE;
  if (n_tiles_ll > P->max_tiles) RTX_FAIL("grid shard of %lld points exceeds the prep capacity (%lld points)", (long long)grid->n, (long long)P->max_tiles * TILE);
  const int n_tiles = (int)n_tiles_ll;
  if (P->n_lines == 0) {
    if (out_f32) RTX_HIP(hipMemset2DAsync(out_f32, ld * sizeof(float), 0, grid->n * sizeof(float), n_layers, st));
    if (out_f64) RTX_HIP(hipMemset2DAsync(out_f64, ld * sizeof(double), 0, grid->n * sizeof(double), n_layers, st));
    return 0;
  }
  RangeArgs ra;
  ra.ic = P->ic; ra.maxhw = P->maxhw; ra.n_lines = P->n_lines; ra.n_tiles = n_tiles; ra.tile = TILE;
  ra.n_layers = n_layers; ra.n = grid->n; ra.ranges = P->ranges;
  hipLaunchKernelGGL(tile_ranges_kernel, dim3((n_tiles + 255) / 256, n_layers), dim3(256), 0, st, ra);
  RTX_LAUNCH_CHECK();
  VsArgs a;
  a.rec = P->rec; a.rec64 = P->rec64; a.ranges = P->ranges; a.n_lines = P->n_lines;
  a.n_tiles = n_tiles; a.tiles_per_xcd = (n_tiles + 7) / 8;
  a.g = to_dev(grid);
  a.out32 = out_f32; a.out64 = out_f64; a.ld = ld; a.inv_scale = 1.0 / P->scale;
  hipLaunchKernelGGL(voigt_sum_kernel<RTX_VOIGT_P>, dim3(8 * a.tiles_per_xcd, n_layers), dim3(256), 0, st, a);
  RTX_LAUNCH_CHECK();
  return 0;
}
