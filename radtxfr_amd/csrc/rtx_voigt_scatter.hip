// Voigt line-sum, "scatter into per-wave LDS tiles" formulation (the default; rtx_voigt.hip holds the
// register-accumulator gather kernel it replaced and shares the record layout and the Weideman code).
//
// Why a second formulation: in the gather kernel a (line, wave) visit costs ~60 cycles of bookkeeping
// (record re-read from LDS, scalar set-up) before any useful work, every line is visited by every wave it
// reaches (6e7 visits on the C3 workload, ~1.8 ms), and window edges / Weideman bands force whole visits
// through slow code because register accumulators need compile-time row indices.
// Here a workgroup owns a tile of SC_TILE consecutive grid points of one layer; EACH LINE IS TAKEN BY
// EXACTLY ONE WAVE, which sweeps the line's window across the whole tile row by row (64 points per
// wave-instruction) and accumulates into ITS OWN copy of the tile in LDS (plain ds_read/add/ds_write: no
// other wave touches that copy, so no atomics and a fixed summation order). Rows are a run-time loop, so
//   - the per-line set-up is paid once per (line, tile) instead of once per (line, wave),
//   - only the first/last row of a window is masked and only the 2-3 band rows run the Weideman code,
//   - interior rows run a 4-row software-pipelined body: 7 VALU + v_rcp_f32 + one LDS read and write per row.
// At the end the 4 copies are added in a fixed order and stored with coalesced 256-B wave stores.
// LDS accumulate rate (tools/ubench_lds.hip): 4.7 cycles per wave-level read+add+write per CU, i.e. ~80 % of
// what four SIMDs demand at 22 VALU cycles per row -- the vector pipes stay the binding resource.
//
// Summation order: lines are dealt round-robin to the 4 waves in table order, so a point's value is a fixed
// function of the tiling; a different tiling (another wavenumber shard) regroups the fp32 sums and may differ
// in the last bits (the gather kernel is bit-identical across shards; tests allow 1e-6 here).
#include "rtx_common.h"

#include "rtx_voigt_math.h"

#ifndef RTX_SC_ABLATE
#define RTX_SC_ABLATE 0  /* timing experiments: 1 = no band rows, 2 = no band rows and no edge rows */
#endif
#ifndef RTX_SC_ASYM
#define RTX_SC_ASYM 1
#endif
#ifndef RTX_SC_ROWS
#define RTX_SC_ROWS 20  // rows of 64 points per tile: 1280 points, 5 KiB of LDS per wave copy (measured: 16 -> 5.63 ms, 20 -> 5.35, 24 -> 5.42, 32 -> 6.1)
#endif

struct ScArgs {
  const LineRec* rec;
  const LineRec64* rec64;
  const int2* ranges;
  const int* smally;
  long long n_lines;
  int n_tiles;
  int tiles_per_xcd;
  GridDev g;
  float* out32;
  double* out64;
  long long ld;
  double inv_scale;
};

template <bool CORE64>
__global__ __launch_bounds__(256) void voigt_scatter_kernel(ScArgs a) {
  constexpr int ROWS = RTX_SC_ROWS;
  constexpr int TILE = 64 * ROWS;
  __shared__ float s_acc[4][TILE];  // one private tile per wave

  const int b = blockIdx.x;
  const int tile = (b & 7) * a.tiles_per_xcd + (b >> 3);  // XCD-aware: one contiguous run of tiles per XCD
  if (tile >= a.n_tiles) return;
  const int k = blockIdx.y;
  if (CORE64 && a.smally[k] == 0) return;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const long long n = a.g.n;
  const int ia = tile * TILE;
  const int ib = (int)((long long)ia + TILE < n ? (long long)ia + TILE : n);
  const int nt = ib - ia;  // points of this tile
  const LineRec* __restrict__ rec = a.rec + (size_t)k * (size_t)a.n_lines;
  const LineRec64* __restrict__ rec64 = a.rec64 + (size_t)k * (size_t)a.n_lines;
  const int2 rng = a.ranges[(size_t)k * a.n_tiles + tile];
  float* __restrict__ acc = s_acc[wave];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) acc[r * 64 + lane] = 0.f;
  const float lanef = (float)lane;
  bool touched = false;

  // No staging: the candidate records are wave-uniform data, so each wave fetches the records of ITS lines (every
  // 4th candidate of the tile's range, table order) with scalar loads straight into SGPRs -- no LDS for records, no
  // barrier in the line loop, and the far-wing constants enter the VALU instructions as scalar operands.
  // Tried and measured slower on C3: staging the records through LDS with a ballot-compacted list (6.1 ms),
  // fetching the whole record before the reject test (6.1), a software prefetch of the next record (6.2), and
  // doing the set-up for 64 lines at once in vector code with v_readlane broadcasts (5.3); this form: 5.0 ms.
  {
    const int base = 0;
    const LineRec* __restrict__ pq = rec + (rng.x + wave);
    for (int slot = rng.x + wave; slot < rng.y; slot += 4, pq += 4) {
      const LineRec q = *pq;  // s_load: the 8 resident waves per SIMD cover its latency (a software prefetch measured slower)
      const int qi0 = __builtin_amdgcn_readfirstlane(q.i0), qlo = __builtin_amdgcn_readfirstlane(q.lo);
      const int qhi = __builtin_amdgcn_readfirstlane(q.hi), qzw = __builtin_amdgcn_readfirstlane(q.zw);
      if (!(qhi > ia && qlo < ib)) continue;  // empty windows have lo = hi = 0
      if (CORE64 && !(qzw > 0 && q.y < 1.0f && qi0 + qzw >= ia && qi0 - qzw < ib)) continue;
      // tile-local window [lo_t, hi_t) and its rows [r_lo, r_hi); a row cut by a window edge is "partial".
      // Plain integer arithmetic (min/max/shift): this set-up runs 2.7e7 times per C3 pass on the CU's one scalar ALU.
      const int dlo = qlo - ia, dhi = qhi - ia;
      const int lo_t = dlo > 0 ? dlo : 0, hi_t = dhi < nt ? dhi : nt;
      const int r_lo = lo_t >> 6, r_hi = (hi_t + 63) >> 6;
      const int c0 = (lo_t + 63) >> 6;               // rows wholly inside the window: [c0, c1)
      const int c1 = dhi < nt ? hi_t >> 6 : r_hi;    // a ragged last row of the GRID is not an edge: its stores are masked
      const bool part_l = c0 != r_lo, part_r = c1 != r_hi;
      // rows touching the Weideman band: [z0, z1] (z0 > z1: none in this tile)
      int z0 = ROWS, z1 = -1;
      if (qzw > 0) {
        const int zl = qi0 - qzw - ia, zh = qi0 + qzw - ia;
        if (zh >= 0 && zl < TILE) {
          z0 = zl > 0 ? zl >> 6 : 0;
          z1 = (zh >> 6) < ROWS - 1 ? zh >> 6 : ROWS - 1;
        }
      }
      // u = i - i0 as a float: integer-valued, exact while |i - i0| < 2^24; |i0| is clamped by the prologue
      const float u0 = (float)(ia - qi0) + lanef;
      const float ulo = (float)(qlo - qi0), uhi = (float)(qhi - qi0);
      const float zw_f = qzw > 0 ? (float)qzw : -1.0f;

      if (!CORE64) {
        // interior rows = [c0, c1) minus [z0, z1]: run 0 = [c0, min(c1, z0)), run 1 = [max(c0, z1 + 1), c1)
        for (int run = 0; run < 2; ++run) {
          if (run == 1 && z0 > z1) break;  // no band: run 0 already covered [c0, c1)
          int r = run == 0 ? c0 : (z1 + 1 > c0 ? z1 + 1 : c0);
          const int re = run == 0 ? (z0 < c1 ? z0 : c1) : c1;
          for (; r + 8 <= re; r += 8) {  // 8 rows in flight: halves the scalar loop overhead (the CU's one scalar ALU is busy)
            float* p = acc + r * 64 + lane;
            float av[8], nv[8], dv[8], xv;
#pragma unroll
            for (int t = 0; t < 8; ++t) av[t] = p[64 * t];
            const float ub = u0 + (float)(64 * r);
#pragma unroll
            for (int t = 0; t < 8; ++t) farwing(ub + (float)(64 * t), q, xv, nv[t], dv[t]);
#pragma unroll
            for (int t = 0; t < 8; ++t) p[64 * t] = fmaf(nv[t], dv[t], av[t]);
          }
          for (; r + 4 <= re; r += 4) {  // 4 rows in flight: LDS reads, 4 evaluations, LDS writes
            float* p = acc + r * 64 + lane;
            const float a0 = p[0], a1 = p[64], a2 = p[128], a3 = p[192];
            const float ub = u0 + (float)(64 * r);
            float x0, n0, d0, x1, n1, d1, x2, n2, d2, x3, n3, d3;
            farwing(ub, q, x0, n0, d0);
            farwing(ub + 64.0f, q, x1, n1, d1);
            farwing(ub + 128.0f, q, x2, n2, d2);
            farwing(ub + 192.0f, q, x3, n3, d3);
            p[0] = fmaf(n0, d0, a0);
            p[64] = fmaf(n1, d1, a1);
            p[128] = fmaf(n2, d2, a2);
            p[192] = fmaf(n3, d3, a3);
          }
          for (; r < re; ++r) {
            float* p = acc + r * 64 + lane;
            float x0, n0, d0;
            farwing(u0 + (float)(64 * r), q, x0, n0, d0);
            p[0] = fmaf(n0, d0, p[0]);
          }
        }
        // partial rows (at most two) outside the band rows: far-wing formula, lanes outside [lo,hi) masked
        const int eL = part_l ? r_lo : -1;
        const int eR = (part_r && r_hi - 1 != eL) ? r_hi - 1 : -1;
        for (int pass = 0; pass < (RTX_SC_ABLATE == 2 ? 0 : 2); ++pass) {
          const int r = pass == 0 ? eL : eR;
          if (r < 0 || (r >= z0 && r <= z1)) continue;
          float* p = acc + r * 64 + lane;
          const float u = u0 + (float)(64 * r);
          float x0, n0, d0;
          farwing(u, q, x0, n0, d0);
          n0 = (u >= ulo && u < uhi) ? n0 : 0.f;
          p[0] = fmaf(n0, d0, p[0]);
        }
      }
      // band rows: region test + Weideman for lanes inside |x|+y<15, far-wing formula elsewhere, window-masked.
      // A line with y < 1 belongs to the CORE64 pass (same predicate on the same fp32 record in both passes).
      const bool small_y = q.y < 1.0f;
      if (z0 <= z1 && RTX_SC_ABLATE == 0) {
        const int zb = z0 > r_lo ? z0 : r_lo, ze = z1 < r_hi - 1 ? z1 : r_hi - 1;
        for (int r = zb; r <= ze; ++r) {
          float* p = acc + r * 64 + lane;
          const int i = ia + 64 * r + lane;
          const float u = u0 + (float)(64 * r);
          float x, num, rden;
          farwing(u, q, x, num, rden);
          const bool in_band = fabsf(u) <= zw_f;
          if (CORE64 ? small_y : !small_y) {
            // hum1_wei's switch |x|+y < 15 (:9840): fp32 decides unless a lane sits within 2e-3 of it; those
            // lanes repeat the test exactly as the reference forms it, in fp64: x = -Im Z1 = -((sg0 - sg)*cte)
            const float s32 = fabsf(x) + q.y;
            bool wz = s32 < 15.0f;
            const bool near = fabsf(s32 - 15.0f) < 2e-3f;
            if (CORE64 || __ballot(near)) {
              const LineRec64 Q = rec64[base + slot];
              const double sg = grid_x(a.g, a.g.offset + (long long)i);
              const double x64 = -((Q.sg0 - sg) * Q.cte);
              const bool wz64 = fabs(x64) + Q.y < 15.0;
              wz = (CORE64 || near) ? wz64 : wz;
              if (CORE64 && wz) {
                num = (float)(Q.A * weideman_re<double>(x64, Q.y));
                rden = 1.0f;
              }
            }
            if (!CORE64 && wz) {
              // pressure-broadened lines (y >= 6: ~70 % of the band lines of C3): the whole band has |z| >= 6 and the
              // 6-term asymptotic series agrees with Weideman-24 to 6.4e-8; Weideman itself only for y < 6
              if (RTX_SC_ASYM && q.y >= 6.0f) num = q.A * asym6_re(x, q.y);
              else num = q.A * weideman_re<float>(x, q.y);
              rden = 1.0f;
            }
            if (CORE64) {  // this pass adds ONLY the band lanes of small-y lines
              num = in_band ? num : 0.f;
              touched = true;
            }
          } else {
            // the other pass owns this line's band lanes; here only the far-wing lanes of the row count
            num = (CORE64 || in_band) ? 0.f : num;
          }
          num = (u >= ulo && u < uhi) ? num : 0.f;
          p[0] = fmaf(num, rden, p[0]);
        }
      }
    }
  }
  __syncthreads();  // every wave's tile is complete

  if (CORE64) {  // did any wave add anything? (uniform per workgroup through LDS)
    __shared__ int s_touched;
    if (threadIdx.x == 0) s_touched = 0;
    __syncthreads();
    if (touched && lane == 0) s_touched = 1;
    __syncthreads();
    if (!s_touched) return;
  }
  // ---- fixed-order sum of the four private tiles, coalesced stores ------------------------------------
#pragma unroll 4
  for (int r = wave; r < ROWS; r += 4) {
    const int t = r * 64 + lane;
    const long long i = (long long)ia + t;
    if (i < (long long)ib) {
      const float v = (s_acc[0][t] + s_acc[1][t]) + (s_acc[2][t] + s_acc[3][t]);
      const size_t o = (size_t)k * (size_t)a.ld + (size_t)i;
      if (CORE64) {
        if (a.out32) a.out32[o] += v;
        if (a.out64) a.out64[o] += (double)v * a.inv_scale;
      } else {
        if (a.out32) a.out32[o] = v;
        if (a.out64) a.out64[o] = (double)v * a.inv_scale;
      }
    }
  }
}

extern "C" int rtx_voigt_scatter_tile_points(void) { return 64 * RTX_SC_ROWS; }

int rtx_voigt_sum_scatter(const rtx_prep* P, const rtx_grid* grid, int n_layers, float* out_f32, double* out_f64, int64_t ld,
                          hipStream_t st, void (*launch_ranges)(const rtx_prep*, const rtx_grid*, int, int, int, hipStream_t)) {
  constexpr int TILE = 64 * RTX_SC_ROWS;
  const long long n_tiles_ll = (grid->n + TILE - 1) / TILE;
  if (n_tiles_ll > P->max_tiles) RTX_FAIL("grid shard of %lld points exceeds the prep capacity", (long long)grid->n);
  const int n_tiles = (int)n_tiles_ll;
  launch_ranges(P, grid, n_layers, n_tiles, TILE, st);
  RTX_LAUNCH_CHECK();
  ScArgs a;
  a.rec = P->rec; a.rec64 = P->rec64; a.ranges = P->ranges; a.smally = P->smally; a.n_lines = P->n_lines;
  a.n_tiles = n_tiles; a.tiles_per_xcd = (n_tiles + 7) / 8;
  a.g = to_dev(grid);
  a.out32 = out_f32; a.out64 = out_f64; a.ld = ld; a.inv_scale = 1.0 / P->scale;
  hipLaunchKernelGGL((voigt_scatter_kernel<false>), dim3(8 * a.tiles_per_xcd, n_layers), dim3(256), 0, st, a);
  RTX_LAUNCH_CHECK();
  hipLaunchKernelGGL((voigt_scatter_kernel<true>), dim3(8 * a.tiles_per_xcd, n_layers), dim3(256), 0, st, a);
  RTX_LAUNCH_CHECK();
  return 0;
}
