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

extern "C" int rtx_voigt_scatter_tile_points(void);
// points per line-sum workgroup tile: capacity granularity of rtx_prep_create, and the alignment at which a wavenumber
// shard reproduces the full grid's tiles (hence its bits: dist.hsi_cube_from_atmosphere)
extern "C" int rtx_voigt_tile_points(void) { return rtx_voigt_scatter_tile_points(); }

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
