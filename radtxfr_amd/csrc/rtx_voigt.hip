// Voigt line-sum: gather formulation of the reference's per-line scatter-add. This file holds the dispatcher
// (rtx_voigt_sum), the per-tile line ranges, and the register-accumulator kernel that was the first formulation; the
// default is now the nodal kernel of rtx_voigt_scatter.hip. This one stays selectable (RADTXFR_VOIGT_KERNEL=gather) as
// the formulation that is bit-identical across wavenumber shards, and as a cross-check (tests run all three).
// (misc/hapi.py:11050, 11135-11138; PROFILE_VOIGT :10131 -> pcqsdhc PART1 :9900-9915 ->
// hum1_wei :9833-9844).
//
// Mapping (CDNA4): one workgroup = 4 waves = one tile of 4*64*P consecutive grid points of one layer.
// Lane <-> grid point (coalesced along the wavenumber axis), each lane owns P points 64 apart, so a
// "row" of 64 consecutive points is one wave-instruction wide and every per-line decision (window
// edge, Weideman zone) is wave-uniform per row -- no divergence inside a row.
//
// Staging: the 256 threads fetch 256 candidate line records into LDS and classify each one against
// each of the 4 wave spans IN PARALLEL (vector code): 0 = does not reach the wave, 1 = the wave lies
// wholly inside the line's window and outside its Weideman band ("fast"), 2 = anything else. Ballots
// and prefix counts turn that into one order-preserving index list per wave. The wave loop then
// only visits lines it needs and spends no scalar instructions on classification -- the first
// version of this kernel did those tests per (line, wave) in scalar code and was bound by the CU's
// single scalar ALU (rocprof: 0.75 SALU per VALU instruction), not by the vector pipes.
//
// Arithmetic: the far-wing branch of hum1_wei (|x|+y >= 15; ~99 % of evaluations in the troposphere)
// is evaluated in fp32 from a grid-relative argument (integer index difference times step*cte plus
// a sub-grid residual), which keeps (nu - nu0) exact to ~1e-7 relative; the region test and the
// Weideman-24 branch use the fp64 record (fp64 polynomial when y<1, where fp32 loses Re w).
#include <stdlib.h>
#include <string.h>

#include "rtx_common.h"

#include "rtx_voigt_math.h"

struct VsArgs {
  const LineRec* rec;      // [n_layers][n_lines]
  const LineRec64* rec64;  // [n_layers][n_lines]
  const int2* ranges;      // [n_layers][n_tiles] candidate line range per tile
  const int* smally;       // [n_layers] != 0: some line has a Weideman band with y < 1 (fp64 pass needed)
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
  long long lo = 0, hi = a.n_lines;  // lower_bound(ic, vlo)
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

// CORE64 = false: the main pass. Far wing everywhere, fp32 Weideman inside the bands of lines with y >= 1.
// CORE64 = true : a second, usually empty, pass that ADDS the band points of lines with y < 1
//                 (Doppler-dominated: stratosphere, low pressure), where Re w needs the fp64 polynomial.
//                 Keeping that code out of the main kernel keeps it at ~64 VGPRs (the fp64 Horner
//                 chain costs 140 and drops the occupancy of the whole line loop to 3 waves/SIMD).
template <int P, bool CORE64>
#ifndef RTX_VOIGT_WAVES
#define RTX_VOIGT_WAVES 1  /* min waves per SIMD asked of the register allocator */
#endif
__global__ __launch_bounds__(256, RTX_VOIGT_WAVES) void voigt_sum_kernel(VsArgs a) {
  constexpr int WPTS = 64 * P;    // points per wave
  constexpr int TILE = 4 * WPTS;  // points per workgroup
  constexpr int CHUNK = 256;      // candidates staged per round (one per thread)
  __shared__ LineRec s_rec[CHUNK];
  __shared__ int s_list[4][CHUNK + 4];  // per consuming wave: (class << 8 | slot), order-preserving; +4 prefetch slack
  __shared__ int s_cnt[4][4];           // [staging wave][consuming wave]

  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD one
  // contiguous run of tiles -- neighbouring tiles share most of their line records in that XCD's L2.
  const int b = blockIdx.x;
  const int tile = xcd_tile(b);  // XCD-aware order (rtx_common.h)
  if (tile >= a.n_tiles) return;  // whole workgroup exits together
  const int k = blockIdx.y;
  if (CORE64 && a.smally[k] == 0) return;
  // readfirstlane makes the wave index an SGPR, so the per-line control flow below is scalar code
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const long long n = a.g.n;
  const int ia = tile * TILE;
  const int ib = (int)((long long)ia + TILE < n ? (long long)ia + TILE : n);
  const int wa = ia + wave * WPTS;
  const LineRec* __restrict__ rec = a.rec + (size_t)k * (size_t)a.n_lines;
  const LineRec64* __restrict__ rec64 = a.rec64 + (size_t)k * (size_t)a.n_lines;
  const int2 rng = a.ranges[(size_t)k * a.n_tiles + tile];

  float acc[P];
#pragma unroll
  for (int r = 0; r < P; ++r) acc[r] = 0.f;
  const float lanef = (float)lane;
  bool touched = false;  // CORE64: did this wave add anything (wave-uniform)

  for (int base = rng.x; base < rng.y; base += CHUNK) {
    // ---- stage + classify (vector code, all 256 threads) ---------------------------------------
    const int l = base + (int)threadIdx.x;
    const bool valid = l < rng.y;
    const float4* src = reinterpret_cast<const float4*>(rec + (valid ? l : rng.y - 1));  // 48-B record = 3 x 16 B
    const float4 r0 = src[0], r1 = src[1];
    const int4 r2 = reinterpret_cast<const int4*>(src)[2];  // i0, lo, hi, zw
    {
      float4* dst = reinterpret_cast<float4*>(&s_rec[threadIdx.x]);
      dst[0] = r0;
      dst[1] = r1;
      reinterpret_cast<int4*>(dst)[2] = r2;
    }
    const int lo = r2.y, hi = r2.z;
    const int zlo = r2.x - r2.w, zhi = r2.x + r2.w;  // band that can hold |x|+y<15 (zw = 0: none)
    int cls[4], pos[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int wwa = ia + w * WPTS;
      const int wwb = wwa + WPTS < ib ? wwa + WPTS : ib;
      const bool band = (r2.w > 0) && (zhi >= wwa) && (zlo < wwb);
      bool reach = valid && (hi > wwa) && (lo < wwb);  // empty windows have lo = hi = 0
      if (CORE64) reach = reach && band && (r1.z < 1.0f);  // r1.z = y
      const bool fast = (lo <= wwa) && (hi >= wwb) && !band;
      cls[w] = reach ? (fast ? 1 : 2) : 0;
      const unsigned long long m = __ballot(reach);
      pos[w] = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) s_cnt[wave][w] = __popcll(m);
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      int off = 0;
#pragma unroll
      for (int sw = 0; sw < 4; ++sw) off += (sw < wave) ? s_cnt[sw][w] : 0;
      if (cls[w]) s_list[w][off + pos[w]] = (cls[w] << 8) | (int)threadIdx.x;
    }
    const int total = s_cnt[0][wave] + s_cnt[1][wave] + s_cnt[2][wave] + s_cnt[3][wave];
    __syncthreads();

    // ---- this wave's lines, in table order (LDS broadcast reads, software-prefetched) ----------
    const int* __restrict__ list = s_list[wave];
    auto visit = [&](const LineRec& q, const int e) __attribute__((always_inline)) {
      // u = i - i0 as a float: integer-valued and exact while |i - i0| < 2^24 (any sane window);
      // |i0| is clamped by the prologue, so wa - i0 cannot overflow int32
      const float u0 = (float)(wa - __builtin_amdgcn_readfirstlane(q.i0)) + lanef;
#ifndef RTX_ABLATE
#define RTX_ABLATE 0  /* timing experiments only: 1 = no phase B, 2 = every visit takes the fast path, 3 = no math */
#endif
      if (RTX_ABLATE == 3) return;
      if (!CORE64 && ((e >> 8) == 1 || RTX_ABLATE == 2)) {
#pragma unroll
        for (int r = 0; r < P; ++r) {
          RTX_FARWING(u0 + (float)(64 * r), q, num, rden);
          acc[r] = fmaf(num, rden, acc[r]);
        }
      } else {
        // General case. Phase A: every row the window reaches, far-wing formula, lanes outside [lo,hi)
        // or inside the Weideman band masked to zero (vector compares: no scalar work per row).
        const int qlo = __builtin_amdgcn_readfirstlane(q.lo), qhi = __builtin_amdgcn_readfirstlane(q.hi);
        const int qi0 = __builtin_amdgcn_readfirstlane(q.i0), qzw = __builtin_amdgcn_readfirstlane(q.zw);
        const float ulo = (float)(qlo - qi0), uhi = (float)(qhi - qi0);
        const float zw_f = qzw > 0 ? (float)qzw : -1.0f;  // |u| <= zw  <=>  inside the band
        const int rb = qlo > wa ? (qlo - wa) >> 6 : 0;
        const int re = ((qhi - 1 - wa) >> 6) + 1 < P ? ((qhi - 1 - wa) >> 6) + 1 : P;
        // rows [c0,c1) lie wholly inside the window; rows [z0,z1] touch the band: the others need no mask
        const int c0 = qlo > wa ? (qlo - wa + 63) >> 6 : 0;
        const int c1 = (qhi - wa) >> 6;
        const int z0 = qzw > 0 ? ((qi0 - qzw - wa) >> 6) : P;
        const int z1 = qzw > 0 ? ((qi0 + qzw - wa) >> 6) : -1;
#pragma unroll
        for (int r = 0; r < P; ++r) {
          if (CORE64 || r < rb) continue;
          if (r >= re) break;
          const float u = u0 + (float)(64 * r);
          RTX_FARWING(u, q, num, rden);
          if (r < c0 || r >= c1 || (r >= z0 && r <= z1))
            num = (u >= ulo && u < uhi && !(fabsf(u) <= zw_f)) ? num : 0.f;
          acc[r] = fmaf(num, rden, acc[r]);
        }
        // Phase B: the few rows around the line centre that hold band points. A line with y < 1 is
        // left to the CORE64 pass (same predicate on the same fp32 record in both passes).
        const bool small_y = q.y < 1.0f;
        if (qzw > 0 && RTX_ABLATE != 1 && (CORE64 ? small_y : !small_y)) {
          const int zlo_ = qi0 - qzw, zhi_ = qi0 + qzw;
          const int first = (zlo_ > qlo ? zlo_ : qlo) - wa;
          const int last = (zhi_ < qhi - 1 ? zhi_ : qhi - 1) - wa;
          const int r0_ = first > 0 ? first >> 6 : 0;
          const int r1_ = (last >> 6) < P - 1 ? (last >> 6) : P - 1;
          for (int r = r0_; r <= r1_; ++r) {
            const int i = wa + 64 * r + lane;
            const float u = u0 + (float)(64 * r);
            // same num * rcp(den) + acc as every other row, so a point gets the same bits whichever
            // tiling (wavenumber shard) reaches it
            RTX_FARWING(u, q, num, rden);
            // hum1_wei's switch |x|+y < 15 (:9840). fp32 decides unless a lane sits within 2e-3 of the
            // boundary (fp32 error of |x|+y is < 1e-5 here); those lanes repeat the test exactly as
            // the reference forms it, in fp64: x = -Im Z1 = -((sg0 - sg)*cte).
            const float s32 = fabsf(x_) + q.y;
            bool wz = s32 < 15.0f;
            const bool near = fabsf(s32 - 15.0f) < 2e-3f;
            if (CORE64 || __ballot(near)) {
              const LineRec64 Q = rec64[base + (e & 255)];
              const double sg = grid_x(a.g, a.g.offset + (long long)i);
              const double x64 = -((Q.sg0 - sg) * Q.cte);
              const bool wz64 = fabs(x64) + Q.y < 15.0;
              wz = (CORE64 || near) ? wz64 : wz;
              if (CORE64 && wz) {
                num = (float)(Q.A * weideman_re<double>(x64, Q.y));
                rden = 1.0f;
              }
            }
            if (!CORE64 && wz && RTX_ABLATE != 4) {
              num = q.A * weideman_re<float>(x_, q.y);
              rden = 1.0f;
            }
            num = (u >= ulo && u < uhi && fabsf(u) <= zw_f) ? num : 0.f;
            if (CORE64) touched = true;
            if (RTX_ABLATE == 5) { acc[0] = fmaf(num, rden, acc[0]); continue; }
#pragma unroll
            for (int rr = 0; rr < P; ++rr) acc[rr] = (rr == r) ? fmaf(num, rden, acc[rr]) : acc[rr];
          }
        }
      }
    };
    // two records in flight, no register copies: while one line is evaluated the next one's LDS reads land
    int e0 = __builtin_amdgcn_readfirstlane(list[0]), e1 = __builtin_amdgcn_readfirstlane(list[1]);
    LineRec qa = s_rec[e0 & 255], qb = s_rec[e1 & 255];  // garbage slots past the list end are never visited
    int j = 0;
    for (; j + 1 < total; j += 2) {
      const int ea = e0, eb = e1;
      e0 = __builtin_amdgcn_readfirstlane(list[j + 2]);
      e1 = __builtin_amdgcn_readfirstlane(list[j + 3]);
      visit(qa, ea);
      qa = s_rec[e0 & 255];
      visit(qb, eb);
      qb = s_rec[e1 & 255];
    }
    if (j < total) visit(qa, e0);
    __syncthreads();
  }

  if (CORE64 && !touched) return;
#pragma unroll
  for (int r = 0; r < P; ++r) {
    const long long i = (long long)wa + 64 * r + lane;
    if (i < (long long)ib) {
      const size_t o = (size_t)k * (size_t)a.ld + (size_t)i;
      if (CORE64) {  // read-modify-write by the owning lane only: deterministic
        if (a.out32) a.out32[o] += acc[r];
        if (a.out64) a.out64[o] += (double)acc[r] * a.inv_scale;
      } else {
        if (a.out32) a.out32[o] = acc[r];
        if (a.out64) a.out64[o] = (double)acc[r] * a.inv_scale;
      }
    }
  }
}

#ifndef RTX_VOIGT_P
#define RTX_VOIGT_P 4  // measured on MI355X, C3 workload: P=2 12.3 ms, P=4 6.6 ms, P=8 7.3 ms, P=16 14 ms
#endif

extern "C" int rtx_voigt_scatter_tile_points(void);
// capacity granularity of rtx_prep_create: the smaller of the two kernels' tiles
extern "C" int rtx_voigt_tile_points(void) {
  const int g = 4 * 64 * RTX_VOIGT_P, s = rtx_voigt_scatter_tile_points();
  return g < s ? g : s;
}

// rtx_voigt_scatter.hip
int rtx_voigt_sum_scatter(const rtx_prep* P, const rtx_grid* grid, int n_layers, float* out_f32, double* out_f64, int64_t ld,
                          hipStream_t st, void (*launch_ranges)(const rtx_prep*, const rtx_grid*, int, int, int, hipStream_t),
                          int nodal);

static void launch_tile_ranges(const rtx_prep* P, const rtx_grid* grid, int n_layers, int n_tiles, int tile, hipStream_t st) {
  RangeArgs ra;
  ra.ic = P->ic; ra.maxhw = P->maxhw; ra.n_lines = P->n_lines; ra.n_tiles = n_tiles; ra.tile = tile;
  ra.n_layers = n_layers; ra.n = grid->n; ra.ranges = P->ranges;
  hipLaunchKernelGGL(tile_ranges_kernel, dim3((n_tiles + 255) / 256, n_layers), dim3(256), 0, st, ra);
}

// RADTXFR_VOIGT_KERNEL selects the line-sum formulation: "nodal" (default; rtx_voigt_scatter.hip, far rows at
// Chebyshev nodes), "scatter" (same file, every row point by point) or "gather" (the register-accumulator kernel
// of this file, bit-identical across wavenumber shards). The alternatives stay for A/B timing and cross-checks.
static int voigt_kernel_choice() {  // 0 nodal, 1 scatter, 2 gather
  static int cached = -1;
  if (cached < 0) {
    const char* e = getenv("RADTXFR_VOIGT_KERNEL");
    cached = (e && strcmp(e, "gather") == 0) ? 2 : (e && strcmp(e, "scatter") == 0) ? 1 : 0;
  }
  return cached;
}
static bool use_gather_kernel() { return voigt_kernel_choice() == 2; }

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
  if (!use_gather_kernel()) return rtx_voigt_sum_scatter(P, grid, n_layers, out_f32, out_f64, ld, st, launch_tile_ranges, voigt_kernel_choice() == 0);
  launch_tile_ranges(P, grid, n_layers, n_tiles, TILE, st);
  RTX_LAUNCH_CHECK();
  VsArgs a;
  a.rec = P->rec; a.rec64 = P->rec64; a.ranges = P->ranges; a.smally = P->smally; a.n_lines = P->n_lines;
  a.n_tiles = n_tiles; a.tiles_per_xcd = xcd_slots(n_tiles);
  a.g = to_dev(grid);
  a.out32 = out_f32; a.out64 = out_f64; a.ld = ld; a.inv_scale = 1.0 / P->scale;
  hipLaunchKernelGGL((voigt_sum_kernel<RTX_VOIGT_P, false>), dim3(8 * a.tiles_per_xcd, n_layers), dim3(256), 0, st, a);
  RTX_LAUNCH_CHECK();
  // fp64 Weideman pass for Doppler-dominated lines; every workgroup returns at once when the layer has none
  hipLaunchKernelGGL((voigt_sum_kernel<RTX_VOIGT_P, true>), dim3(8 * a.tiles_per_xcd, n_layers), dim3(256), 0, st, a);
  RTX_LAUNCH_CHECK();
  return 0;
}
