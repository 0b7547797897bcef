// Voigt line-sum: the dispatcher (rtx_voigt_sum) and the per-tile candidate line ranges. The kernels live in
// rtx_voigt_scatter.hip: the default nodal kernel and, selectable with RADTXFR_VOIGT_KERNEL=scatter, the point-by-point
// scatter kernel it grew out of, kept as the cross-check of the node interpolation (tests run both). The first
// formulation (register accumulators, one gather per grid point) was removed in round 2: three formulations had to be
// kept parity-green for every change and it was 2x slower than the default.
// (misc/hapi.py:11050, 11135-11138; PROFILE_VOIGT :10131 -> pcqsdhc PART1 :9900-9915 -> hum1_wei :9833-9844).
#include <stdlib.h>
#include <string.h>

#include "rtx_common.h"

#include "rtx_voigt_math.h"

struct RangeArgs {
  const int* ic;
  const int* maxhw;
  const LineRec* rec;
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

// Canonical ranges: [first line that REACHES the tile, last such line + 1). The bracket above depends on maxhw -- a maximum
// over whatever table the caller uploaded -- and the line-sum deals candidates to waves and lanes by their position in the
// range, so the order of its fp32 sums would depend on the bracket. Trimmed to the reaching lines, the range -- hence every
// bit of the result -- is a function of the lines that contribute to the tile and of nothing else: a rank that holds only
// the lines in reach of its wavenumber shard reproduces the full-table, full-grid run exactly (shards cut on tile
// boundaries; dist.py, bench.py). One wave per (tile, layer): 64 candidates' windows per step, from each end.
__global__ __launch_bounds__(256) void tile_trim_kernel(RangeArgs a) {
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  const int k = blockIdx.y;
  if (wave >= a.n_tiles) return;
  const long long ia = (long long)wave * a.tile;
  long long ib = ia + a.tile;
  if (ib > a.n) ib = a.n;
  const size_t o = (size_t)k * a.n_tiles + wave;
  const int2 rng = a.ranges[o];
  const LineRec* __restrict__ rec = a.rec + (size_t)k * (size_t)a.n_lines;
  int first = rng.y, last = rng.y;  // nothing reaches: empty range
  for (int base = rng.x; base < rng.y; base += 64) {
    const int s = base + lane;
    const bool reach = s < rng.y && (long long)rec[s].hi > ia && (long long)rec[s].lo < ib;
    const unsigned long long m = __ballot(reach);
    if (m) { first = base + __builtin_ctzll(m); break; }
  }
  if (first < rng.y) {
    for (int top = rng.y; top > first; top -= 64) {
      const int s = top - 1 - lane;
      const bool reach = s >= first && (long long)rec[s].hi > ia && (long long)rec[s].lo < ib;
      const unsigned long long m = __ballot(reach);
      if (m) { last = top - __builtin_ctzll(m); break; }
    }
  }
  if (lane == 0) a.ranges[o] = make_int2(first, last);
}

int rtx_voigt_scatter_tile_points(void);
// points per line-sum workgroup tile: capacity granularity of rtx_prep_create, and the alignment at which a wavenumber
// shard reproduces the full grid's tiles (hence its bits: dist.hsi_cube_from_atmosphere)
extern "C" int rtx_voigt_tile_points(void) { return rtx_voigt_scatter_tile_points(); }

// rtx_voigt_scatter.hip
int rtx_voigt_sum_scatter(const rtx_prep* P, const rtx_grid* grid, int n_layers, float* out_f32, double* out_f64, int64_t ld,
                          hipStream_t st, void (*launch_ranges)(const rtx_prep*, const rtx_grid*, int, int, int, hipStream_t),
                          int nodal);

static void launch_tile_ranges(const rtx_prep* P, const rtx_grid* grid, int n_layers, int n_tiles, int tile, hipStream_t st) {
  RangeArgs ra;
  ra.ic = P->ic; ra.maxhw = P->maxhw; ra.rec = P->rec; ra.n_lines = P->n_lines; ra.n_tiles = n_tiles; ra.tile = tile;
  ra.n_layers = n_layers; ra.n = grid->n; ra.ranges = P->ranges;
  hipLaunchKernelGGL(tile_ranges_kernel, dim3((n_tiles + 255) / 256, n_layers), dim3(256), 0, st, ra);
  hipLaunchKernelGGL(tile_trim_kernel, dim3((n_tiles + 3) / 4, n_layers), dim3(256), 0, st, ra);
}

// RADTXFR_VOIGT_KERNEL=scatter selects the point-by-point cross-check kernel instead of the default nodal one.
static int voigt_kernel_choice() {  // 0 nodal, 1 scatter
  static int cached = -1;
  if (cached < 0) {
    const char* e = getenv("RADTXFR_VOIGT_KERNEL");
    cached = (e && strcmp(e, "scatter") == 0) ? 1 : 0;
  }
  return cached;
}

extern "C" int rtx_voigt_sum(const rtx_prep* P, const rtx_grid* grid, int n_layers, float* out_f32, double* out_f64,
                             int64_t ld, void* stream) {
  if (!P) RTX_FAIL("prep is NULL");
  if (rtx_check_grid(grid)) return 1;
  if (n_layers < 1 || n_layers != P->n_layers) RTX_FAIL("n_layers=%d does not match the last rtx_line_prep (%d)", n_layers, P->n_layers);
  if (!out_f32 && !out_f64) RTX_FAIL("both outputs are NULL");
  if (ld < grid->n) RTX_FAIL("ld=%lld < n=%lld", (long long)ld, (long long)grid->n);
  if (grid->n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (P->n_lines == 0) {
    if (out_f32) RTX_HIP(hipMemset2DAsync(out_f32, ld * sizeof(float), 0, grid->n * sizeof(float), n_layers, st));
    if (out_f64) RTX_HIP(hipMemset2DAsync(out_f64, ld * sizeof(double), 0, grid->n * sizeof(double), n_layers, st));
    return 0;
  }
  return rtx_voigt_sum_scatter(P, grid, n_layers, out_f32, out_f64, ld, st, launch_tile_ranges, voigt_kernel_choice() == 0);
}
