// Voigt line-sum, "scatter into per-wave LDS tiles" formulations: the default nodal kernel and the point-by-point
// scatter kernel it grew out of, kept as its cross-check (RADTXFR_VOIGT_KERNEL=scatter; rtx_voigt.hip holds the
// dispatcher; both kernels share the record layout and the Weideman code of rtx_voigt_math.h).
//
// A workgroup owns a tile of SC_TILE consecutive grid points of one layer; EACH LINE IS TAKEN BY EXACTLY ONE
// WAVE, which accumulates the line into ITS OWN copy of the tile in LDS (plain ds_read/add/ds_write: no other
// wave touches that copy, so no atomics and a fixed summation order). Rows (64 points = one wave-instruction)
// are a run-time loop, so only the first/last row of a window is masked and only the 2-3 band rows run the
// Weideman code. At the end the 4 copies are added in a fixed order and stored with coalesced 256-B stores.
//
// Two kernels:
//  voigt_scatter_kernel<CORE64>  every row a window reaches is evaluated point by point (7 VALU + v_rcp_f32 + one
//                                LDS read/write per row and line). CORE64 = true is its fp64 pass for y < 1 lines.
//  voigt_nodal_kernel            (default main pass) the same, but only for the rows NEAR a line: within
//                                SC_NEAR rows of its centre, in its Weideman band, or cut by a window edge.
//                                On every other row the line is a smooth far wing -- the rational far-wing form of
//                                hum1_wei, poles at |nu - nu0| ~ gamma0 -- and is evaluated at the row's 8
//                                Chebyshev nodes only; the nodal sums of all far lines are carried to the 64
//                                grid points once per tile by the 64x8 Lagrange matrix (interpolation is linear,
//                                so interpolating the sum equals summing the interpolants). With the nearest
//                                pole >= 3 rows from the row the degree-7 interpolant is within 1.4e-8 of each
//                                line's own (positive) contribution (tools/gen_cheb.py), far
//                                below the fp32 rounding of the direct evaluation. The far evaluation runs with
//                                lane = (line, node): 8 lines x 8 nodes per wave-instruction, records fetched
//                                by vector loads, no scalar work per line: ~1.6 VALU slots per (line, row)
//                                against ~11 for the point-by-point form, and C3's windows are ~80 rows wide.
//
// Summation order: a point's value is a fixed function of the tiling and of the lines that reach the tile (the candidate
// ranges are trimmed to those, rtx_voigt.hip: tile_trim_kernel). A wavenumber shard that starts on a tile boundary of the
// full axis -- dist.tile_aligned_bounds cuts every shard that way: compute_TUD_sharded, hsi_cube_from_atmosphere, bench.py
// -- is therefore bit-identical to the single-rank result, whatever subset of the table the rank uploaded, provided the
// subset holds every line in reach. A shard cut elsewhere regroups the fp32 sums (last-bit differences, ~1e-6).
#include "rtx_common.h"

#include "rtx_voigt_math.h"

#include "cheb8_64.inc"
#include "cheb16_tile.inc"
#include <vector>

#ifndef RTX_SC_ABLATE
#define RTX_SC_ABLATE 0  /* timing experiments: bit 0 = no band rows, 1 = no far rows, 2 = no point-by-point rows in the drain, 3 = no entries at all, 4 = no window-edge rows (and no entries for lines that have nothing else) */
#endif
#ifndef RTX_SC_ASYM
#define RTX_SC_ASYM 1
#endif
#ifndef RTX_SC_OUTER
#define RTX_SC_OUTER 1  // outer band rows of y < 1 lines in fp32 (0: fp64 on every band row, as in round 2)
#endif
#ifndef RTX_SC_ROWS
#define RTX_SC_ROWS 16  // rows of 64 points per tile (1024 points, 4 KiB of LDS per wave copy). Round 2, after the tile level went: 8 -> 2.38 ms, 10 -> 2.38, 12 -> 2.32, 14 -> 2.29, 16 -> 2.24, 18 -> 2.34, 20 -> 2.32 (C3, prologue + line-sum)
#endif
#ifndef RTX_SC_NEAR
#define RTX_SC_NEAR 2  // rows either side of the centre row that stay point-by-point in the nodal kernel (3: 2.48 ms, interpolation error <= 1.4e-8 of a line's own contribution; 2: 2.34 ms, <= 2.3e-7)
#endif
static_assert(RTX_SC_ROWS <= 31, "row masks are 32-bit");

#ifndef RTX_SC_STAMP
#define RTX_SC_STAMP 0  /* 1: per-phase cycle totals of the nodal kernel, printed by the host after every launch (debug builds) */
#endif
#define SC_NSTAMP 8
#define STAMP(i_)                                 \
  if (RTX_SC_STAMP) {                             \
    const long long t_now_ = (long long)clock64(); \
    t_ph[i_] += t_now_ - t_prev;                  \
    t_prev = t_now_;                              \
  }
struct ScArgs {
  unsigned long long* stamp;
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
  // hot-tile split (rtx_common.h): the work list of extra parts and the workspace of their partial tiles
  const SplitItem* items;
  const int* n_items;
  float* part_ws;
};

// Row geometry of one line in one tile, in plain integer arithmetic so that the scalar per-line code and the
// per-lane far-row masks agree exactly. Rows are tile-local, 64 points each.
struct RowGeom {
  int r_lo, r_hi;  // rows the window [lo, hi) reaches: [r_lo, r_hi)
  int c0, c1;      // rows wholly inside the window: [c0, c1)
  int z0, z1;      // rows touching the Weideman band: [z0, z1] (z0 > z1: none in this tile)
  int n0, n1;      // near zone [n0, n1] (unclamped): centre row +- SC_NEAR, widened to the band rows
  bool part_l, part_r;
};
__device__ __forceinline__ RowGeom row_geom(int qi0, int qlo, int qhi, int qzw, int ia, int nt) {
  constexpr int ROWS = RTX_SC_ROWS;
  RowGeom g;
  const int dlo = qlo - ia, dhi = qhi - ia;
  const int lo_t = dlo > 0 ? dlo : 0, hi_t = dhi < nt ? dhi : nt;
  g.r_lo = lo_t >> 6;
  g.r_hi = (hi_t + 63) >> 6;
  g.c0 = (lo_t + 63) >> 6;
  g.c1 = dhi < nt ? hi_t >> 6 : g.r_hi;  // a ragged last row of the GRID is not an edge: its stores are masked
  g.part_l = g.c0 != g.r_lo;
  g.part_r = g.c1 != g.r_hi;
  const int zw = qzw > 0 ? qzw : 0;
  const int zl = qi0 - zw - ia, zh = qi0 + zw - ia;  // |qi0| <= 1e8 + n (prologue clamp), zw <= 4e7: no overflow
  g.z0 = ROWS;
  g.z1 = -1;
  if (qzw > 0 && zh >= 0 && zl < 64 * ROWS) {
    g.z0 = zl > 0 ? zl >> 6 : 0;
    g.z1 = (zh >> 6) < ROWS - 1 ? zh >> 6 : ROWS - 1;
  }
  const int rc = (qi0 - ia) >> 6;  // arithmetic shift: floor
  const int bl = zl >> 6, bh = zh >> 6;
  g.n0 = (rc - RTX_SC_NEAR) < bl ? rc - RTX_SC_NEAR : bl;
  g.n1 = (rc + RTX_SC_NEAR) > bh ? rc + RTX_SC_NEAR : bh;
  return g;
}

// One band row: region test + Weideman for lanes inside |x|+y<15, far-wing formula elsewhere, window-masked.
// A line with y < 1 belongs to the CORE64 pass (same predicate on the same fp32 record in both passes).
// The record fields may live in SGPRs (scalar visit) or be wave-uniform VGPR values (nodal kernel).
// MODE 0: the fp32 main pass (the band lanes of y < 1 lines are left to the fp64 pass); MODE 1: the fp64 pass, which adds
// ONLY those lanes; MODE 2: both in one (the nodal kernel's instantiation for layers that hold y < 1 lines).
template <int MODE>
__device__ __forceinline__ void band_row(const ScArgs& a, const LineRec64* __restrict__ rec64, int slot, const LineRec& q, float u, int i,
                                         float zw_f, float ulo, float uhi, bool small_y, float& num, float& rden, bool& touched) {
  constexpr bool CORE64 = MODE == 1;
  float x;
  farwing(u, q, x, num, rden);
  const bool in_band = fabsf(u) <= zw_f;
  // MODE 2, Doppler-dominated line: a row none of whose lanes is closer than |x| = 5.5 to the centre -- most rows of the
  // wide bands at high wavenumbers -- leaves fp64: the 12-term asymptotic series in fp32 (rtx_voigt_math.h: asymK_re)
  const bool outer_row = MODE == 2 && small_y && RTX_SC_OUTER && __ballot(fabsf(x) < 5.5f) == 0ull;
  if (MODE == 2 && small_y && !outer_row) {
    // Doppler-dominated line: the reference's switch and its Weideman value in fp64 on every lane of the row; lanes
    // outside |x| + y < 15 keep the far-wing value
    const LineRec64 Q = rec64[slot];
    const double sg = grid_x(a.g, a.g.offset + (long long)i);
    const double x64 = -((Q.sg0 - sg) * Q.cte);
    if (fabs(x64) + Q.y < 15.0) {
      num = (float)(Q.A * weideman_re<double>(x64, Q.y));
      rden = 1.0f;
    }
  } else if (CORE64 ? small_y : (!small_y || outer_row)) {
    // hum1_wei's switch |x|+y < 15 (:9840): fp32 decides unless a lane sits within 2e-3 of it; those
    // lanes repeat the test exactly as the reference forms it, in fp64: x = -Im Z1 = -((sg0 - sg)*cte)
    const float s32 = fabsf(x) + q.y;
    bool wz = s32 < 15.0f;
    const bool near = fabsf(s32 - 15.0f) < 2e-3f;
    if (CORE64 || __ballot(near)) {
      const LineRec64 Q = rec64[slot];
      const double sg = grid_x(a.g, a.g.offset + (long long)i);
      const double x64 = -((Q.sg0 - sg) * Q.cte);
      const bool wz64 = fabs(x64) + Q.y < 15.0;
      wz = (CORE64 || near) ? wz64 : wz;
      if (CORE64 && wz) {
        num = (float)(Q.A * weideman_re<double>(x64, Q.y));
        rden = 1.0f;
      }
    }
    if (!CORE64) {
      // pressure-broadened lines (y >= 6: ~70 % of the band lines of C3): the whole band has |z| >= 6 and the
      // 6-term asymptotic series agrees with Weideman-24 to 6.4e-8. For 1 <= y < 6 the series needs |z| >= 8
      // (2e-8; the error relative to Re w grows as y -> 0): a row none of whose band lanes is closer than that --
      // the outer rows of a wide band -- takes the series too; Weideman itself only for the rows around the centre.
      const bool series = RTX_SC_ASYM && (q.y >= 6.0f || __ballot(wz && fmaf(x, x, q.y * q.y) < 64.0f) == 0ull);
      if (wz) {
        if (MODE == 2 && outer_row) num = q.A * asymK_re<12>(x, q.y);
        else if (series) num = q.A * asym6_re(x, q.y);
        else num = q.A * weideman_re<float>(x, q.y);
        rden = 1.0f;
      }
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
}

// One line, taken by one wave: point-by-point rows into the wave's LDS tile. NODAL: only the rows of the near
// zone (the far rows were added at the Chebyshev nodes); the record arrives through scalar loads.
template <bool CORE64, bool NODAL>
__device__ __forceinline__ void visit_line(const ScArgs& a, const LineRec* __restrict__ rec, const LineRec64* __restrict__ rec64,
                                           int slot, float* __restrict__ acc, int ia, int ib, int nt, int lane, float lanef,
                                           bool& touched) {
  const LineRec q = rec[slot];  // s_load: the resident waves cover its latency (a software prefetch measured slower)
  const int qi0 = __builtin_amdgcn_readfirstlane(q.i0), qlo = __builtin_amdgcn_readfirstlane(q.lo);
  const int qhi = __builtin_amdgcn_readfirstlane(q.hi), qzw = __builtin_amdgcn_readfirstlane(q.zw);
  if (!(qhi > ia && qlo < ib)) return;  // empty windows have lo = hi = 0
  if (CORE64 && !(qzw > 0 && q.y < 1.0f && qi0 + qzw >= ia && qi0 - qzw < ib)) return;
  // Plain integer arithmetic (min/max/shift): this set-up runs on the CU's one scalar ALU.
  const RowGeom g = row_geom(qi0, qlo, qhi, qzw, ia, nt);
  const int r_lo = g.r_lo, r_hi = g.r_hi, z0 = g.z0, z1 = g.z1;
  // point-by-point interior rows: [c0, c1), clipped to the near zone for the nodal kernel
  const int c0 = NODAL ? (g.c0 > g.n0 ? g.c0 : g.n0) : g.c0;
  const int c1 = NODAL ? (g.c1 < g.n1 + 1 ? g.c1 : g.n1 + 1) : g.c1;
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
      if (!NODAL) {
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
    const int eL = g.part_l ? r_lo : -1;
    const int eR = (g.part_r && r_hi - 1 != eL) ? r_hi - 1 : -1;
    for (int pass = 0; pass < ((RTX_SC_ABLATE & 4) ? 0 : 2); ++pass) {
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
  const bool small_y = q.y < 1.0f;
  if (z0 <= z1 && !(RTX_SC_ABLATE & 1)) {
    const int zb = z0 > r_lo ? z0 : r_lo, ze = z1 < r_hi - 1 ? z1 : r_hi - 1;
    for (int r = zb; r <= ze; ++r) {
      float* p = acc + r * 64 + lane;
      float num, rden;
      band_row<CORE64 ? 1 : 0>(a, rec64, slot, q, u0 + (float)(64 * r), ia + 64 * r + lane, zw_f, ulo, uhi, small_y, num, rden, touched);
      p[0] = fmaf(num, rden, p[0]);
    }
  }
}

// Fixed-order sum of the four private tiles (+ the interpolated nodal sums), coalesced stores.
template <bool CORE64, bool NODAL>
__device__ __forceinline__ void store_tile(const ScArgs& a, const float (*s_acc)[64 * RTX_SC_ROWS],
                                           const float (*s_nodal)[RTX_SC_ROWS][CHEB_N], int k, int ia, int ib, int wave, int lane) {
  constexpr int ROWS = RTX_SC_ROWS;
  float wl[CHEB_N];
  if (NODAL) {
#pragma unroll
    for (int j = 0; j < CHEB_N; ++j) wl[j] = CHEB_W[lane][j];
  }
#pragma unroll 4
  for (int r = wave; r < ROWS; r += 4) {
    const int t = r * 64 + lane;
    const long long i = (long long)ia + t;
    float v = (s_acc[0][t] + s_acc[1][t]) + (s_acc[2][t] + s_acc[3][t]);
    if (NODAL) {
      float f = 0.f;
#pragma unroll
      for (int j = 0; j < CHEB_N; ++j)
        f = fmaf(wl[j], (s_nodal[0][r][j] + s_nodal[1][r][j]) + (s_nodal[2][r][j] + s_nodal[3][r][j]), f);
      v += f;
    }
    if (i < (long long)ib) {
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

template <bool CORE64>
__device__ __forceinline__ void scatter_tile(const ScArgs& a, float (*s_acc)[64 * RTX_SC_ROWS], int* s_touched, int b, int k) {
  constexpr int ROWS = RTX_SC_ROWS;
  constexpr int TILE = 64 * ROWS;
  const int tile = xcd_tile(b);  // XCD-aware order (rtx_common.h)
  if (tile >= a.n_tiles) return;
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
  for (int slot = rng.x + wave; slot < rng.y; slot += 4) visit_line<CORE64, false>(a, rec, rec64, slot, acc, ia, ib, nt, lane, lanef, touched);
  __syncthreads();  // every wave's tile is complete

  if (CORE64) {  // did any wave add anything? (uniform per workgroup through LDS)
    if (threadIdx.x == 0) *s_touched = 0;
    __syncthreads();
    if (touched && lane == 0) *s_touched = 1;
    __syncthreads();
    if (!*s_touched) return;
  }
  store_tile<CORE64, false>(a, s_acc, nullptr, k, ia, ib, wave, lane);
}

// CORE64 = false: one workgroup per tile slot. CORE64 = true (the pass that only matters for Doppler-dominated lines): a
// grid-stride loop over the tile slots with 1/16 of the workgroups -- when a layer has no such line (every layer of C3)
// each workgroup returns at once, and launching one per tile just to do that cost 30 us per step. The stride is a
// multiple of 8, so a workgroup's slots stay on its XCD (xcd_tile).
template <bool CORE64>
__global__ __launch_bounds__(256) void voigt_scatter_kernel(ScArgs a) {
  __shared__ float s_acc[4][64 * RTX_SC_ROWS];  // one private tile per wave
  __shared__ int s_touched;
  const int k = blockIdx.y;
  if (CORE64) {
    if (a.smally[k] == 0) return;
    for (int b = blockIdx.x; b < 8 * a.tiles_per_xcd; b += gridDim.x) {
      scatter_tile<true>(a, s_acc, &s_touched, b, k);
      __syncthreads();  // the tile copies are reused by the next slot
    }
  } else {
    scatter_tile<false>(a, s_acc, &s_touched, blockIdx.x, k);
  }
}

// ---- main pass: far rows at Chebyshev nodes, near rows point by point --------------------------------------------------
// A workgroup owns a tile of ROWS rows of 64 grid points of one layer. Every candidate line of the tile's range is
// classified ONCE, with lane = candidate (one record load and one row_geom per line, 64 lines per instruction; the
// candidates are dealt to the 4 waves in table order modulo 4, so each wave sees the same mix of far / edge / centre
// lines), and then served by level:
//   row level        rows wholly inside the line's window and outside its near zone (centre row +- SC_NEAR rows, widened
//                    to the Weideman band): the line is the smooth far-wing rational there and is evaluated at the row's 8
//                    Chebyshev nodes only. The member lanes' ids are compacted with one ds_permute; groups of 8 members
//                    are evaluated with lane = (member, node), the members' record fields fetched lane-to-lane with
//                    ds_bpermute (no LDS storage, no second trip to memory). Members whose window covers the whole tile
//                    ("full": 40 % of them on C3) skip the per-row masks.
//   point by point   near-zone rows, the <= 2 rows cut by a window edge, band rows: the member lanes write 64-byte LDS
//                    entries (record fields, window in u, row masks) and the wave drains them, accumulating into its
//                    own copy of the tile in LDS (plain read/add/write, no atomics, fixed order).
// The nodal sums of all lines are carried to the 64 grid points of each row once per tile by the 64 x 8 Lagrange matrix
// (interpolation is linear), fused with the fixed-order sum of the four point-by-point copies and the coalesced stores.
//
// Round-2 history (C3, prologue + line-sum): first nodal kernel (lane = (line, node) for classification too, a tile
// level of 32 nodes, a ring for row-level lines) 2.84 ms -> lane = candidate classification + dense groups 2.58 -> tile
// level dropped 2.44 (its 160 x 32 tile-nodes -> row-nodes matrix, read uncoalesced by 160 threads per tile, cost
// 0.18 ms; evaluating those lines at row level costs less than that) -> this form. Measured slower: the point-by-point
// members broadcast out of their lanes with v_readlane instead of LDS entries (2.78), two point-by-point rows in flight
// (2.60), entries drained once per round with the records reloaded on overflow (2.69), tiles of 12 / 16 / 24 rows
// (2.71 / 2.63 / 2.82), LDS float adds without return (ds_add_f32) instead of read / fma / write for the point-by-point
// rows (10.6 ms: the LDS float atomic is several times slower than the plain pair), row ownership -- each wave owns rows
// w, w+4, ... with register accumulators, the entries of all four waves shared through LDS behind two barriers per round,
// no private tile copies (2.68: the point-by-point rows are bound by their own arithmetic, ~41 SIMD cycles per row
// against a 34-cycle issue cost, not by the LDS round trip, and the shared lists cost more than the copies did), the tile
// level rebuilt on 16-row tiles with a coalesced (transposed) matrix (2.29 vs 2.28: no gain), and a tree of segment levels --
// lines two segment lengths away evaluated at 8 nodes per 16-, 8- or 4-row segment, a third of all (line, tile) pairs, 1.5x
// fewer node evaluations, same 2.3e-7 accuracy -- with its three extra compaction passes (2.44 vs 2.24: a pass's fixed cost,
// one ds_permute + nine ds_bpermute per group of 8 lines, outweighs the evaluations it saves).
// Later in round 2: two waves per workgroup instead of four, 32-entry lists (2.10 -> 1.93: a wave that sees twice the candidates
// fills its groups of 8 members better; one wave per workgroup: the same), the Weideman halves by the real two-term recurrence
// (2.25 -> 2.17). Measured slower: one compaction for full and partial members together with only the last group mixed (15
// spilled registers, 2.08 vs 1.93), the next entry read ahead in the final drain (2.00 vs 1.93).
// A single segment level on top of that -- full members whose centre is >= 8 rows outside the tile evaluated at 8 nodes per
// 4-row segment (2.7e-7), a quarter of their evaluations, sums carried to the row nodes by a constant 4 x 8 x 8 matrix in the
// final stage: 2.02 vs 1.95 (distance 6 / 10: 2.03 / 2.04). The extra compaction pass and one more spilled register cost
// more than the ~40 % of the members it takes off the row level save, as the three-level tree had shown with four waves.
#ifndef RTX_SC_WAVES
#define RTX_SC_WAVES 6
#endif
#ifndef SC_NW
#define SC_NW 2  // waves per workgroup of the nodal kernel (each takes every SC_NW-th candidate and owns a copy of the tile). 4 -> 2: the row-level groups of 8 member lines fill better when a wave sees twice the candidates (2.10 -> 2.02 ms); 1: the same as 2
#endif
#ifndef SC_EDGE_LIST
#define SC_EDGE_LIST 1  // 1: lines that need nothing but ONE window-edge row in this tile (60 % of the point-by-point entries) go to a list of their own: 32-byte entries (two LDS reads and one v_readfirstlane instead of four each, no row-mask loops), 1.96 -> 1.87 ms. Entries per iteration 1 / 2 / 3 / 4: 1.90 / 1.87 / 1.89 / 1.91; capacities edge / general 48/8, 24/20, 16/24: 1.93 / 1.89 / 1.90; the near-zone-only lines (a run of whole rows, no band) through the same list with a row loop: no change (1.90 / 1.90)
#endif
#ifndef SC_EDGE_CAP
#define SC_EDGE_CAP 32
#endif
#ifndef SC_EDGE_UNROLL
#define SC_EDGE_UNROLL 2
#endif
#ifndef RTX_SC_TILE_LEVEL
#define RTX_SC_TILE_LEVEL 1  // full members >= RTX_SC_TILE_DIST points outside a 16-row tile: 16 tile nodes instead of 16 x 8 row nodes
#endif
#define RTX_SC_TILE_DIST 512
#ifndef SC_ENT_CAP
#define SC_ENT_CAP (SC_EDGE_LIST ? 16 : 32)  // point-by-point entries per wave (64 B each). Without the edge list, two waves per workgroup: 16 -> 2.02 ms, 24 -> 1.99, 32 -> 1.96, 48 -> 2.03
#endif
// SMALLY: the instantiation for layers that hold Doppler-dominated (y < 1) lines: their band lanes take the fp64 Weideman
// value right here (band_row MODE 2) instead of in a second pass over the layer.
// part_out == nullptr: the tile of workgroup slot b, candidates = its (possibly cut) range, result stored to the optical
// depths. part_out != nullptr: an extra part of a hot tile -- tile b_or_tile, candidates part_rng -- stored to part_out[0 .. TILE).
template <bool SMALLY>
__device__ __forceinline__ void nodal_tile(const ScArgs& a, const int b_or_tile, const int k, const int2 part_rng = make_int2(0, 0),
                                           float* __restrict__ part_out = nullptr) {
  constexpr int ROWS = RTX_SC_ROWS;
  constexpr int TILE = 64 * ROWS;
  __shared__ float s_acc[SC_NW][TILE];              // one private tile per wave (near rows)
  __shared__ float4 s_ent[SC_NW][SC_ENT_CAP][4];
  __shared__ float4 s_edge[SC_NW][SC_EDGE_LIST ? SC_EDGE_CAP : 1][2];
  __shared__ float s_nodsum[RTX_SC_ROWS][CHEB_N];
  // after the last drain the entry lists are dead: wave w keeps its row-level sums [ROWS][8] in its list
  static_assert((RTX_SC_ROWS * CHEB_N + TCHEB_N + 1) * 4 <= SC_ENT_CAP * 64, "row sums + tile sums fit in one wave's entry list");

  const int tile = part_out ? b_or_tile : xcd_tile(b_or_tile);  // XCD-aware order (rtx_common.h)
  if (tile >= a.n_tiles) return;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const long long n = a.g.n;
  const int ia = tile * TILE;
  const int ib = (int)((long long)ia + TILE < n ? (long long)ia + TILE : n);
  const int nt = ib - ia;
  const LineRec* __restrict__ rec = a.rec + (size_t)k * (size_t)a.n_lines;
  const LineRec64* __restrict__ rec64 = a.rec64 + (size_t)k * (size_t)a.n_lines;
  const int2 rng = part_out ? part_rng : a.ranges[(size_t)k * a.n_tiles + tile];
  float* __restrict__ acc = s_acc[wave];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) acc[r * 64 + lane] = 0.f;
  const float lanef = (float)lane;
  bool touched = false;
  long long t_ph[SC_NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0};  // debug builds (RTX_SC_STAMP): per-phase cycle totals
  long long t_prev = RTX_SC_STAMP ? (long long)clock64() : 0;
  const long long t_begin = t_prev;

  // node evaluations: lane = (member l of the group, node j)
  const int l = lane >> 3, j = lane & 7;
  float nod[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) nod[r] = 0.f;
  // tile level: lane (l, j) holds the tile nodes j and 15 - j of member slot l
  constexpr bool TILE_LEVEL = RTX_SC_TILE_LEVEL && ROWS == 16;
#ifndef RTX_SC_OFF_LDS
#define RTX_SC_OFF_LDS 0
#endif
#if !RTX_SC_OFF_LDS
  const float off_j = CHEB_OFF[j];
#endif
  // the lanes' node offsets live in LDS, a copy per wave read back by the same wave once per round (s_nodsum is free until the
  // final stage): the kernel sits at its 80-register budget (6 waves per SIMD), and these two would be held for its whole length
  if (lane < 8) {
    s_nodsum[wave][lane] = CHEB_OFF[lane];
    if (TILE_LEVEL) s_nodsum[SC_NW + wave][lane] = TCHEB_OFF[lane];
  }
  float tnod_a = 0.f, tnod_b = 0.f;
  int n_tile_members = 0;  // wave-uniform
  int n_ent = 0;  // wave-uniform
  float4(*__restrict__ ent)[4] = s_ent[wave];
  // number of set bits of a wave mask below this lane: v_mbcnt_lo/hi on the (scalar) mask -- no per-lane mask registers
  auto below = [](unsigned long long m) -> int {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
  };
  auto pull = [](int src_lane, float v) -> float {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
  };

  auto drain = [&]() {
    for (int e = 0; e < n_ent; ++e) {
      const float4 e0 = ent[e][0], e1 = ent[e][1], e2 = ent[e][2], e3 = ent[e][3];
      LineRec q;
      q.a = e0.x; q.c = e0.y; q.b1 = e0.z; q.b0 = e0.w;
      q.Ay = e1.x; q.Ay0 = e1.y; q.y = e1.z; q.A = e1.w;
      const float u0 = e2.x + lanef, ulo = e2.y, uhi = e2.z, zw_f = e2.w;
      unsigned m_pp = (unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(e3.x));
      unsigned m_ed = (unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(e3.w));
      unsigned m_bd = (unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(e3.y));
      const int slot = __builtin_amdgcn_readfirstlane(__float_as_int(e3.z));
      STAMP(3);  // entry read
      if (RTX_SC_ABLATE & 4) m_pp = m_ed = 0u;
      // near-zone rows wholly inside the window: far-wing formula at every point, no lane mask
      while (m_pp) {
        const int r = __builtin_ctz(m_pp);
        m_pp &= m_pp - 1u;
        float* p = acc + r * 64 + lane;
        float x0, n0, d0;
        farwing(u0 + (float)(64 * r), q, x0, n0, d0);
        p[0] = fmaf(n0, d0, p[0]);
      }
      // the <= 2 rows cut by a window edge: lanes outside [lo, hi) masked
      while (m_ed) {
        const int r = __builtin_ctz(m_ed);
        m_ed &= m_ed - 1u;
        float* p = acc + r * 64 + lane;
        const float u = u0 + (float)(64 * r);
        float x0, n0, d0;
        farwing(u, q, x0, n0, d0);
        n0 = (u >= ulo && u < uhi) ? n0 : 0.f;
        p[0] = fmaf(n0, d0, p[0]);
      }
      STAMP(4);  // point-by-point rows
      if (RTX_SC_ABLATE & 1) m_bd = 0u;
      const bool small_y = q.y < 1.0f;
      while (m_bd) {
        const int r = __builtin_ctz(m_bd);
        m_bd &= m_bd - 1u;
        float* p = acc + r * 64 + lane;
        float num, rden;
        band_row<SMALLY ? 2 : 0>(a, rec64, slot, q, u0 + (float)(64 * r), ia + 64 * r + lane, zw_f, ulo, uhi, small_y, num, rden, touched);
        p[0] = fmaf(num, rden, p[0]);
      }
      STAMP(5);  // band rows
    }
    n_ent = 0;
  };

  // Window-edge entries: [a c b1 b0] [Ay Ay0 u(lane 0 of the row) packed], packed = first inside lane (left edge) or first
  // outside lane (right edge) | row << 7 | right << 12. SC_EDGE_UNROLL entries per iteration: their reads and evaluations are
  // independent, only the two read-add-writes of the tile copy are ordered (a wave's LDS operations complete in order, so
  // two entries of the same row add up correctly); the list order, hence the summation order, is fixed.
  int n_edge = 0;  // wave-uniform
  float4(*__restrict__ edg)[2] = s_edge[wave];
  auto edge_one = [&](const float4& E0, const float4& E1, int pk) {
    const int r = (pk >> 7) & 31, t = pk & 127;
    const bool right = (pk >> 12) & 1;
    float* p = acc + r * 64 + lane;
    const float u = E1.z + lanef;
    const float x = fmaf(u, E0.x, E0.y);
    const float xx = x * x;
    float num = fmaf(xx, E1.x, E1.y);
    const float rden = __builtin_amdgcn_rcpf(fmaf(xx + E0.z, xx, E0.w));
    const bool inside = (lane >= t) != right;
    num = inside ? num : 0.f;
    p[0] = fmaf(num, rden, p[0]);
  };
  auto drain_edges = [&]() {
    for (int e = 0; e < n_edge; e += SC_EDGE_UNROLL) {
      float4 E0[SC_EDGE_UNROLL], E1[SC_EDGE_UNROLL];
      int pk[SC_EDGE_UNROLL];
#pragma unroll
      for (int q = 0; q < SC_EDGE_UNROLL; ++q) {
        const int eq = e + q < n_edge ? e + q : e;
        E0[q] = edg[eq][0];
        E1[q] = edg[eq][1];
      }
#pragma unroll
      for (int q = 0; q < SC_EDGE_UNROLL; ++q) pk[q] = __builtin_amdgcn_readfirstlane(__float_as_int(E1[q].w));
#pragma unroll
      for (int q = 0; q < SC_EDGE_UNROLL; ++q)
        if (e + q < n_edge) edge_one(E0[q], E1[q], pk[q]);
    }
    n_edge = 0;
  };

  // 64 SC_NW candidates per round: wave w, lane i takes candidate rng.x + w + SC_NW i (+ 64 SC_NW per round)
  for (int base = rng.x + wave; base < rng.y; base += 64 * SC_NW) {
    const int slot = base + SC_NW * lane;
    const bool valid = slot < rng.y;
    const float4* __restrict__ pr = reinterpret_cast<const float4*>(rec + (valid ? slot : rng.y - 1));
    const float4 f0 = pr[0];  // a c b1 b0
    const float4 f1 = pr[1];  // Ay Ay0 y A
    const float4 f2 = pr[2];  // i0 lo hi zw (int bits)
    const int qi0 = __float_as_int(f2.x), qlo = __float_as_int(f2.y), qhi = __float_as_int(f2.z), qzw = __float_as_int(f2.w);
    const bool reach = valid && qhi > ia && qlo < ib;
    const RowGeom g = row_geom(qi0, qlo, qhi, qzw, ia, nt);
    auto bits = [](int lo, int hi) -> unsigned {  // rows [lo, hi), 0 <= lo, hi <= ROWS
      return hi > lo ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
    };
    const unsigned m_reach = reach ? bits(g.r_lo, g.r_hi) : 0u;
    const unsigned m_in = reach ? bits(g.c0, g.c1) : 0u;
    const unsigned m_near = bits(g.n0 > 0 ? g.n0 : 0, (g.n1 < ROWS - 1 ? g.n1 : ROWS - 1) + 1);
    const unsigned m_band = bits(g.z0, g.z1 + 1);
    const unsigned m_far = (RTX_SC_ABLATE & 2) ? 0u : (m_in & ~m_near);           // smooth: Chebyshev nodes of the rows
    const unsigned m_bd = m_reach & m_band;                                       // band rows
    const unsigned m_pp = m_in & m_near & ~m_band;                                 // near-zone rows wholly inside the window
    const unsigned m_ed = (RTX_SC_ABLATE & 16) ? 0u : (m_reach & ~m_in & ~m_band);  // rows cut by a window edge
    const float ub = (float)(ia - qi0);  // integer-valued
    constexpr unsigned ALL_ROWS = (1u << ROWS) - 1u;
    if (RTX_SC_STAMP && __ballot(m_pp == 0xffffffffu)) t_ph[7] += 1;  // (forces the masks, i.e. the record loads, before the stamp)
    STAMP(0);  // record loads + geometry

    // ---- point-by-point rows: 64-byte entries, drained by the whole wave ---------------------------------------
    bool edge_only = false;
    if (SC_EDGE_LIST) {
      // nothing but one window-edge row, cut on one side only
      const bool single = m_pp == 0u && m_bd == 0u && m_ed != 0u && (m_ed & (m_ed - 1u)) == 0u;
      const int re = single ? __builtin_ctz(m_ed) : 0;
      const bool left = g.part_l && re == g.r_lo, rightc = g.part_r && re == g.r_hi - 1;
      edge_only = single && (left != rightc) && !(RTX_SC_ABLATE & 8);
      const int dlo = qlo - ia, dhi = qhi - ia;
      const int t = left ? dlo - 64 * re : dhi - 64 * re;  // in [1, 63] for a cut row
      const int pk = (t & 127) | (re << 7) | (left ? 0 : 1 << 12);
      bool emit = edge_only;
      unsigned long long eb = __ballot(emit);
      while (eb) {
        const int room = SC_EDGE_CAP - n_edge;
        const int r = below(eb);
        if (emit && r < room) {
          float4* d = edg[n_edge + r];
          d[0] = f0;
          d[1] = make_float4(f1.x, f1.y, ub + (float)(64 * re), __int_as_float(pk));
          emit = false;
        }
        const int cnt = __popcll(eb);
        n_edge += cnt < room ? cnt : room;
        if (n_edge == SC_EDGE_CAP) drain_edges();
        eb = __ballot(emit);
      }
    }
    {
      bool emit = !(RTX_SC_ABLATE & 8) && !edge_only && (m_pp | m_ed | m_bd) != 0u;
      unsigned long long eb = __ballot(emit);
      while (eb) {
        const int room = SC_ENT_CAP - n_ent;
        const int r = below(eb);
        if (emit && r < room) {
          float4* d = ent[n_ent + r];
          d[0] = f0;
          d[1] = f1;
          d[2] = make_float4(ub, (float)(qlo - qi0), (float)(qhi - qi0), qzw > 0 ? (float)qzw : -1.0f);
          d[3] = make_float4(__int_as_float((int)m_pp), __int_as_float((int)m_bd), __int_as_float(slot), __int_as_float((int)m_ed));
          emit = false;
        }
        const int cnt = __popcll(eb);
        n_ent += cnt < room ? cnt : room;
        if (n_ent == SC_ENT_CAP) drain();
        eb = __ballot(emit);
      }
    }
    STAMP(1);  // entry emission
    // ---- row level, 8 member lines per pass; x of node j in row r = x0 + r dx (one FMA per row) ----------------
    // Tile level: a full member whose centre lies >= RTX_SC_TILE_DIST points outside the tile is smooth over the WHOLE tile --
    // its 16 rows x 8 nodes are replaced by the tile's 16 Chebyshev nodes (two per lane), carried to the row nodes once per
    // tile by TCHEB_M in the final stage (interpolation error <= 4.1e-8 of the line's own contribution, tests/test_host.py;
    // the classification is a function of the line and the tile alone: no dependence on shards or table subsets).
    bool is_t = false;
    if (TILE_LEVEL) {
      is_t = m_far == ALL_ROWS && (ia - qi0 >= RTX_SC_TILE_DIST || qi0 - (ia + TILE - 1) >= RTX_SC_TILE_DIST);
      const unsigned long long tb = __ballot(is_t);
      if (tb) {
        const int nT = __popcll(tb);
        n_tile_members += nT;
        const float toff_a = s_nodsum[SC_NW + wave][j];
        const int src = __builtin_amdgcn_ds_permute((is_t ? below(tb) : 63) << 2, lane);
        for (int g0 = 0; g0 < nT; g0 += 8) {
          const int e = g0 + l;
          const int sl = __builtin_amdgcn_ds_bpermute(e << 2, src);
          const float qa = pull(sl, f0.x), qc = pull(sl, f0.y), qb1 = pull(sl, f0.z), qb0 = pull(sl, f0.w);
          float qAy = pull(sl, f1.x), qAy0 = pull(sl, f1.y);
          const float qub = pull(sl, ub);
          if (e >= nT) { qAy = 0.f; qAy0 = 0.f; }  // empty slot of the last group
          const float xa = fmaf(qub, qa, fmaf(toff_a, qa, qc));
          const float xb = fmaf(qub, qa, fmaf(1023.0f - toff_a, qa, qc));
          const float xxa = xa * xa, xxb = xb * xb;
          tnod_a = fmaf(fmaf(xxa, qAy, qAy0), __builtin_amdgcn_rcpf(fmaf(xxa + qb1, xxa, qb0)), tnod_a);
          tnod_b = fmaf(fmaf(xxb, qAy, qAy0), __builtin_amdgcn_rcpf(fmaf(xxb + qb1, xxb, qb0)), tnod_b);
        }
      }
    }
    // full members first (no masks), then the partial ones
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const bool is_m = pass == 0 ? (m_far == ALL_ROWS && !is_t) : (m_far != 0u && m_far != ALL_ROWS);
      const unsigned long long rb = __ballot(is_m);
      if (!rb) continue;
      const int nR = __popcll(rb);
#if RTX_SC_OFF_LDS
      const float off_j = s_nodsum[wave][j];
#endif
      // member of rank r sends its lane id to lane r; the others all write to lane 63, which no member targets
      const int src = __builtin_amdgcn_ds_permute((is_m ? below(rb) : 63) << 2, lane);
      for (int g0 = 0; g0 < nR; g0 += 8) {
        const int e = g0 + l;
        const int sl = __builtin_amdgcn_ds_bpermute(e << 2, src);
        const float qa = pull(sl, f0.x), qc = pull(sl, f0.y), qb1 = pull(sl, f0.z), qb0 = pull(sl, f0.w);
        float qAy = pull(sl, f1.x), qAy0 = pull(sl, f1.y);
        const float qub = pull(sl, ub);
        const float x0 = fmaf(qub, qa, fmaf(off_j, qa, qc));  // x of node j in row 0
        const float dx = 64.0f * qa;
        if (pass == 0) {
          if (e >= nR) { qAy = 0.f; qAy0 = 0.f; }  // empty slot of the last group: numerator 0 on every row
#pragma unroll
          for (int r = 0; r < ROWS; ++r) {
            const float x = fmaf((float)r, dx, x0);
            const float xx = x * x;
            const float num = fmaf(xx, qAy, qAy0);
            const float rden = __builtin_amdgcn_rcpf(fmaf(xx + qb1, xx, qb0));
            nod[r] = fmaf(num, rden, nod[r]);
          }
        } else {
          int mf = __builtin_amdgcn_ds_bpermute(sl << 2, (int)m_far);
          mf = e < nR ? mf : 0;
          // row r counts for this member iff bit r of its mask is set: v_bfe_i32 (bit -> 0 / -1) + v_and_b32 on the numerator,
          // two full-rate instructions (written as `x & sbfe(mf, r, 1)` the compiler turns it into and + compare + select,
          // and a select reads its mask from scalar registers: half rate, tools/ubench_enc.hip)
#define SC_PART_ROW(r_)                                                                  \
          if (r_ < ROWS) {                                                                 \
            const float x = fmaf((float)(r_), dx, x0);                                     \
            const float xx = x * x;                                                        \
            float num = fmaf(xx, qAy, qAy0);                                               \
            const float rden = __builtin_amdgcn_rcpf(fmaf(xx + qb1, xx, qb0));             \
            int mb;                                                                        \
            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(mb) : "v"(mf), "n"(r_));                  \
            num = __int_as_float(__float_as_int(num) & mb);                                \
            nod[r_ < ROWS ? r_ : 0] = fmaf(num, rden, nod[r_ < ROWS ? r_ : 0]);            \
          }
          // (row-level members parked in an LDS list until eight are together -- every group full, two ds_read_b128 instead of
          // a ds_permute and nine ds_bpermute per group -- 1.83 against 1.81 ms: the list's emission loop costs what the fuller
          // groups save)
          // (skipping the blocks of four rows that no member of the group has set -- a partial member has ~9 of its 16 rows set --
          // behind a ballot each: 1.815 against 1.823 ms, not worth the branches)
          SC_PART_ROW(0) SC_PART_ROW(1) SC_PART_ROW(2) SC_PART_ROW(3) SC_PART_ROW(4) SC_PART_ROW(5) SC_PART_ROW(6) SC_PART_ROW(7)
          SC_PART_ROW(8) SC_PART_ROW(9) SC_PART_ROW(10) SC_PART_ROW(11) SC_PART_ROW(12) SC_PART_ROW(13) SC_PART_ROW(14) SC_PART_ROW(15)
          SC_PART_ROW(16) SC_PART_ROW(17) SC_PART_ROW(18) SC_PART_ROW(19) SC_PART_ROW(20) SC_PART_ROW(21) SC_PART_ROW(22) SC_PART_ROW(23)
          SC_PART_ROW(24) SC_PART_ROW(25) SC_PART_ROW(26) SC_PART_ROW(27) SC_PART_ROW(28) SC_PART_ROW(29) SC_PART_ROW(30)
#undef SC_PART_ROW
        }
      }
    }
    STAMP(2);  // row level
  }
  drain();
  if (SC_EDGE_LIST) drain_edges();

  // sum the 8 member slots of each node (lanes l = 0..7 of equal j), one copy per wave
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    float v = nod[r];
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (lane < 8) reinterpret_cast<float*>(&s_ent[wave][0][0])[r * CHEB_N + lane] = v;
  }
  if (TILE_LEVEL) {  // the wave's 16 tile-node sums and its member count, behind its row sums
    float va = tnod_a, vb = tnod_b;
    va += __shfl_xor(va, 8);  vb += __shfl_xor(vb, 8);
    va += __shfl_xor(va, 16); vb += __shfl_xor(vb, 16);
    va += __shfl_xor(va, 32); vb += __shfl_xor(vb, 32);
    float* ts = reinterpret_cast<float*>(&s_ent[wave][0][0]) + ROWS * CHEB_N;
    if (lane < 8) { ts[lane] = va; ts[15 - lane] = vb; }
    if (lane == 0) ts[16] = __int_as_float(n_tile_members);
  }
  STAMP(2);
  __syncthreads();
  STAMP(6);  // barrier
  // stage 1: thread (r, jj) adds the waves' row-level sums in a fixed order
  static_assert(SC_NW == 1 || SC_NW == 2 || SC_NW == 4, "one, two or four waves per workgroup");
  for (int o = wave * 64 + lane; o < ROWS * CHEB_N; o += 64 * SC_NW) {  // o = r * CHEB_N + jj (from the live lane id: no late use of threadIdx.x, which cost a spill)
    float v = reinterpret_cast<const float*>(&s_ent[0][0][0])[o];
    if constexpr (SC_NW >= 2) v += reinterpret_cast<const float*>(&s_ent[1][0][0])[o];
    if constexpr (SC_NW == 4) v += reinterpret_cast<const float*>(&s_ent[2][0][0])[o] + reinterpret_cast<const float*>(&s_ent[3][0][0])[o];
    if (TILE_LEVEL) {
      int any = 0;
#pragma unroll
      for (int w = 0; w < SC_NW; ++w) any |= __float_as_int(reinterpret_cast<const float*>(&s_ent[w][0][0])[ROWS * CHEB_N + 16]);
      if (any) {  // uniform over the workgroup
        float c = 0.f;
#pragma unroll
        for (int t = 0; t < TCHEB_N; ++t) {
          float tsum = reinterpret_cast<const float*>(&s_ent[0][0][0])[ROWS * CHEB_N + t];
          if constexpr (SC_NW >= 2) tsum += reinterpret_cast<const float*>(&s_ent[1][0][0])[ROWS * CHEB_N + t];
          if constexpr (SC_NW == 4)
            tsum += reinterpret_cast<const float*>(&s_ent[2][0][0])[ROWS * CHEB_N + t] + reinterpret_cast<const float*>(&s_ent[3][0][0])[ROWS * CHEB_N + t];
          c = fmaf(TCHEB_M[t][o], tsum, c);
        }
        v += c;
      }
    }
    s_nodsum[o >> 3][o & 7] = v;
  }
  __syncthreads();
  // stage 2: row nodes -> grid points, plus the four point-by-point copies; coalesced stores
  {
    float wl[CHEB_N];
#pragma unroll
    for (int jj = 0; jj < CHEB_N; ++jj) wl[jj] = CHEB_W[lane][jj];
#pragma unroll 4
    for (int r = wave; r < ROWS; r += SC_NW) {
      const int t = r * 64 + lane;
      const long long i = (long long)ia + t;
      float f = 0.f;
#pragma unroll
      for (int jj = 0; jj < CHEB_N; ++jj) f = fmaf(wl[jj], s_nodsum[r][jj], f);
      float pp = s_acc[0][t];
      if constexpr (SC_NW >= 2) pp += s_acc[1][t];
      if constexpr (SC_NW == 4) pp += s_acc[2][t] + s_acc[3][t];
      const float v = pp + f;
      if (part_out) {
        part_out[t] = v;  // a partial tile: added to the optical depths by split_combine_kernel, in part order
      } else if (i < (long long)ib) {
        const size_t o = (size_t)k * (size_t)a.ld + (size_t)i;
        if (a.out32) a.out32[o] = v;
        if (a.out64) a.out64[o] = (double)v * a.inv_scale;
      }
    }
  }
  if (RTX_SC_STAMP && !part_out) {
    t_ph[7] = (long long)clock64() - t_begin;  // lifetime (includes the final stages, which have no bucket of their own)
    if (lane == 0) {
      unsigned long long* o = a.stamp + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * SC_NW + wave) * SC_NSTAMP;
      for (int i = 0; i < SC_NSTAMP; ++i) o[i] = (unsigned long long)t_ph[i];
    }
  }
}

// SMALLY = false: one workgroup per tile slot; layers that hold y < 1 lines (per-layer flag from the prologue) are left to
// the other instantiation. SMALLY = true: a grid-stride loop over the tile slots with 1/16 of the workgroups -- when a layer
// has no such line (every layer of C3) each workgroup returns at once, and launching one per tile just to do that cost
// 30 us per step. The stride is a multiple of 8, so a workgroup's slots stay on its XCD (xcd_tile).
#ifndef RTX_SC_WAVES_SMALLY
#define RTX_SC_WAVES_SMALLY 4
#endif
template <bool SMALLY>
__global__ __launch_bounds__(64 * SC_NW, SMALLY ? RTX_SC_WAVES_SMALLY : RTX_SC_WAVES) void voigt_nodal_kernel(ScArgs a) {
  const int k = blockIdx.y;
  if ((a.smally[k] != 0) != SMALLY) return;
  if (SMALLY) {
    for (int b = blockIdx.x; b < 8 * a.tiles_per_xcd; b += gridDim.x) {
      nodal_tile<true>(a, b, k);
      __syncthreads();  // the tile copies are reused by the next slot
    }
  } else {
    nodal_tile<false>(a, blockIdx.x, k);
  }
}

// Extra parts of hot tiles: workgroup i evaluates item i of the work list into its partial tile. Launched with the
// list's CAPACITY (the host-side bound); workgroups beyond the count written by tile_ranges_kernel return at once.
template <bool SMALLY>
__global__ __launch_bounds__(64 * SC_NW, SMALLY ? RTX_SC_WAVES_SMALLY : RTX_SC_WAVES) void voigt_nodal_parts_kernel(ScArgs a) {
  const int i = blockIdx.x;
  if (i >= *a.n_items) return;
  const SplitItem it = a.items[i];
  if ((a.smally[it.k] != 0) != SMALLY) return;
  nodal_tile<SMALLY>(a, it.tile, it.k, make_int2(it.lo, it.hi), a.part_ws + (size_t)i * (size_t)(64 * RTX_SC_ROWS));
}

// optical depths of a hot tile = part 0 (stored by the tile's own workgroup) + part 1 + part 2 + ..., added in that order by
// the workgroup of the tile's part-1 item (its items are consecutive in the list).
__global__ __launch_bounds__(256) void split_combine_kernel(ScArgs a) {
  constexpr int TILE = 64 * RTX_SC_ROWS;
  const int i = blockIdx.x;
  if (i >= *a.n_items) return;
  const SplitItem it = a.items[i];
  if (it.part != 1) return;
  const long long ia = (long long)it.tile * TILE;
  for (int t = threadIdx.x; t < TILE; t += 256) {
    const long long p = ia + t;
    if (p >= a.g.n) break;
    const size_t o = (size_t)it.k * (size_t)a.ld + (size_t)p;
    float v = a.out32 ? a.out32[o] : 0.f;
    double v64 = a.out64 ? a.out64[o] : 0.0;
    for (int e = 0; e < it.extra; ++e) {
      const float w = a.part_ws[(size_t)(i + e) * TILE + t];
      v += w;
      v64 += (double)w * a.inv_scale;
    }
    if (a.out32) a.out32[o] = v;
    if (a.out64) a.out64[o] = v64;
  }
}

__attribute__((visibility("hidden"))) int rtx_voigt_scatter_tile_points(void) { return 64 * RTX_SC_ROWS; }

int rtx_voigt_sum_scatter(const rtx_prep* P, const rtx_grid* grid, int n_layers, float* out_f32, double* out_f64, int64_t ld,
                          hipStream_t st, void (*launch_ranges)(const rtx_prep*, const rtx_grid*, int, int, int, hipStream_t, int),
                          int nodal) {
  constexpr int TILE = 64 * RTX_SC_ROWS;
  const long long n_tiles_ll = (grid->n + TILE - 1) / TILE;
  if (n_tiles_ll > P->max_tiles) RTX_FAIL("grid shard of %lld points exceeds the prep capacity", (long long)grid->n);
  const int n_tiles = (int)n_tiles_ll;
  // the work list of hot-tile parts is rebuilt by every call (a caller may sum twice after one prologue, e.g. once per
  // output precision): its counter starts from zero each time
  if (nodal && P->split_bound > 0 && P->items) RTX_HIP(hipMemsetAsync(P->n_items, 0, sizeof(int), st));
  launch_ranges(P, grid, n_layers, n_tiles, TILE, st, nodal);  // the point-by-point cross-check takes every tile whole
  RTX_LAUNCH_CHECK();
  ScArgs a;
  a.rec = P->rec; a.rec64 = P->rec64; a.ranges = P->ranges; a.smally = P->smally; a.n_lines = P->n_lines;
  a.n_tiles = n_tiles; a.tiles_per_xcd = xcd_slots(n_tiles);
  a.g = to_dev(grid);
  a.out32 = out_f32; a.out64 = out_f64; a.ld = ld; a.inv_scale = 1.0 / P->scale;
  a.stamp = nullptr;
  a.items = P->items; a.n_items = P->n_items; a.part_ws = P->part_ws;
#if RTX_SC_STAMP
  static unsigned long long* d_stamp = nullptr;
  static size_t stamp_cap = 0;
  const size_t n_stamp = (size_t)8 * a.tiles_per_xcd * n_layers * SC_NW * SC_NSTAMP;
  if (stamp_cap < n_stamp) {
    if (d_stamp) RTX_HIP(hipFree(d_stamp));
    RTX_HIP(hipMalloc(&d_stamp, n_stamp * sizeof(unsigned long long)));
    stamp_cap = n_stamp;
  }
  RTX_HIP(hipMemsetAsync(d_stamp, 0, n_stamp * sizeof(unsigned long long), st));
  a.stamp = d_stamp;
#endif
  // RADTXFR_DEBUG_LDS_PAD=<bytes>: extra dynamic LDS per workgroup, to time the kernel at reduced occupancy
  static int lds_pad = -1;
  if (lds_pad < 0) { const char* e = getenv("RADTXFR_DEBUG_LDS_PAD"); lds_pad = e ? atoi(e) : 0; }
  if (nodal) {
    hipLaunchKernelGGL((voigt_nodal_kernel<false>), dim3(8 * a.tiles_per_xcd, n_layers), dim3(64 * SC_NW), (size_t)lds_pad, st, a);
    RTX_LAUNCH_CHECK();
    hipLaunchKernelGGL((voigt_nodal_kernel<true>), dim3(8 * ((a.tiles_per_xcd + 15) / 16), n_layers), dim3(64 * SC_NW), 0, st, a);
    if (P->split_bound > 0 && P->items) {  // the table can have hot tiles: their extra parts, then the sums in part order
      RTX_LAUNCH_CHECK();
      const unsigned cap = (unsigned)P->split_bound;
      hipLaunchKernelGGL((voigt_nodal_parts_kernel<false>), dim3(cap), dim3(64 * SC_NW), 0, st, a);
      hipLaunchKernelGGL((voigt_nodal_parts_kernel<true>), dim3(cap), dim3(64 * SC_NW), 0, st, a);
      hipLaunchKernelGGL(split_combine_kernel, dim3(cap), dim3(256), 0, st, a);
    }
  } else {
    hipLaunchKernelGGL((voigt_scatter_kernel<false>), dim3(8 * a.tiles_per_xcd, n_layers), dim3(256), 0, st, a);
  }
  RTX_LAUNCH_CHECK();
#if RTX_SC_STAMP
  {
    std::vector<unsigned long long> hbuf(n_stamp);
    RTX_HIP(hipMemcpyAsync(hbuf.data(), d_stamp, n_stamp * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    RTX_HIP(hipStreamSynchronize(st));
    if (const char* fn = getenv("RADTXFR_STAMP_FILE")) {
      FILE* fh = fopen(fn, "wb");
      if (fh) { fwrite(hbuf.data(), sizeof(unsigned long long), n_stamp, fh); fclose(fh); }
    }
    double h[SC_NSTAMP] = {0};
    for (size_t i = 0; i < n_stamp; ++i) h[i % SC_NSTAMP] += (double)hbuf[i];
    const double w = (double)SC_NW * 8 * a.tiles_per_xcd * n_layers;
    fprintf(stderr, "[stamp] per wave (s_memtime ticks): load+geom %.0f emit %.0f rowlevel %.0f entry %.0f pp %.0f band %.0f barrier %.0f life %.0f\n",
            h[0] / w, h[1] / w, h[2] / w, h[3] / w, h[4] / w, h[5] / w, h[6] / w, h[7] / w);
  }
#endif
  if (!nodal) {  // the point-by-point formulation keeps its separate fp64 pass
    hipLaunchKernelGGL((voigt_scatter_kernel<true>), dim3(8 * ((a.tiles_per_xcd + 15) / 16), n_layers), dim3(256), 0, st, a);
    RTX_LAUNCH_CHECK();
  }
  return 0;
}
