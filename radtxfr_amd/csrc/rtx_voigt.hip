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
  const int2* win;
  long long n_lines;
  int n_tiles, tile, n_layers, n_steps;
  long long n;
  int2* ranges;
  SplitItem* items;  // hot-tile work list (rtx_common.h); NULL: no tile is cut
  int* n_items;
  long long items_cap;
};

// Candidate lines of tile t in layer k, as a CANONICAL range [first line that reaches the tile, last such line + 1).
// Step 1 brackets the lines whose unshifted-centre index lies within maxhw[k] of the tile (two searches on the sorted
// centre indices); step 2 trims the bracket from both ends to the lines whose window [lo, hi) meets the tile. The bracket
// depends on maxhw -- a maximum over whatever table the caller uploaded -- and the line-sum deals candidates to waves and
// lanes by their position in the range, so the order of its fp32 sums would depend on it. Trimmed, the range -- hence
// every bit of the result -- is a function of the lines that contribute to the tile and of nothing else: a rank that holds
// only the lines in reach of its wavenumber shard reproduces the full-table, full-grid run exactly (shards cut on tile
// boundaries: dist.py, bench.py).
// Sixteen lanes per tile (four tiles per wave; one wave per tile was bound by the wave launch rate: 0.28 ms for 66 layers
// x 10 742 tiles): the two searches are 17-ary -- every step probes 16 positions at once, ~5 steps for 100 000 lines
// instead of 17 dependent loads -- and the trim scans 16 candidates' windows per step from each end.
#define RNG_G 16
__device__ __forceinline__ unsigned group_bits(unsigned long long b, int lane) { return (unsigned)(b >> (lane & 48)) & 0xffffu; }

// One step of the 17-ary search for the number of elements of the sorted ic[lo, hi) that are < v (STRICT) or <= v: the 16
// lanes of a group probe 16 positions at once. A group that has converged repeats a no-op.
template <bool STRICT>
__device__ __forceinline__ void bound_step(const int* __restrict__ ic, int& lo, int& hi, int v, int sub, int lane) {
  // 32-bit arithmetic throughout (n_lines <= 2e9; a 64-bit division by 17 is a hundred-instruction sequence on this chip,
  // the unsigned 32-bit one a multiply-high and a shift)
  const unsigned width = (unsigned)(hi - lo);
  const int stride = (int)((width + RNG_G) / (unsigned)(RNG_G + 1));  // >= 1 while width > 0
  const long long pl = (long long)lo + (long long)(sub + 1) * stride - 1;
  const bool inside = width > 0 && pl < (long long)hi;
  const int p = inside ? (int)pl : lo;
  const bool pred = inside && (STRICT ? ic[p] < v : ic[p] <= v);
  const int c = __popc(group_bits(__ballot(pred), lane));  // the predicate is monotone along the probes
  if (width > 0) {
    // probe c was evaluated and failed when it lies inside the bracket: the answer is <= its position
    const long long cap = (long long)lo + (long long)(c + 1) * stride - 1;
    if (c < RNG_G && cap < (long long)hi) hi = (int)cap;
    lo += c * stride;  // probes 0 .. c-1 passed: the answer is > their positions
  }
}

__global__ __launch_bounds__(256) void tile_ranges_kernel(RangeArgs a) {
  const int lane = threadIdx.x & 63, sub = lane & (RNG_G - 1);
  const int t = (int)((blockIdx.x * blockDim.x + threadIdx.x) / RNG_G);
  const int k = blockIdx.y;
  const bool live = t < a.n_tiles;  // dead groups run along (the ballots are wave-wide) on an empty bracket
  const long long ia = (long long)t * a.tile;
  long long ib = ia + a.tile;
  if (ib > a.n) ib = a.n;
  const long long hw = a.maxhw[k];
  // both searches in one loop: their loads are independent, so the chain of dependent memory round trips -- what this
  // kernel's time is made of -- is n_steps long, not twice that
  int l0 = 0, h0 = live ? (int)a.n_lines : 0, l1 = 0, h1 = h0;
  {
    // the centre indices are clamped to +-(1e8 + n) by the prologue, so clamping the two search keys to int32 changes nothing
    const long long v0 = ia - hw, v1 = ib - 1 + hw;
    const int k0 = (int)(v0 < -2147483647LL ? -2147483647LL : v0), k1 = (int)(v1 > 2147483647LL ? 2147483647LL : v1);
    for (int it = 0; it < a.n_steps; ++it) {
      bound_step<true>(a.ic, l0, h0, k0, sub, lane);
      bound_step<false>(a.ic, l1, h1, k1, sub, lane);
    }
  }
  if (l1 < l0) l1 = l0;
  const int2* __restrict__ win = a.win + (size_t)k * (size_t)a.n_lines;
  long long first = l1, last = l1;  // nothing reaches: empty range
  {
    // the trim from both ends, likewise in one loop: 16 candidates' windows from the front and 16 from the back per step
    int base = l0, top = l1;
    bool done_f = base >= l1, done_b = done_f;
    while (__ballot(!done_f || !done_b)) {
      const int sf = base + sub, sb = top - 1 - sub;
      int2 wf = make_int2(0, 0), wb = wf;  // (lo, hi); an empty window reaches nothing
      if (!done_f && sf < l1) wf = win[sf];
      if (!done_b && sb >= l0) wb = win[sb];
      const bool rf = (long long)wf.y > ia && (long long)wf.x < ib && wf.y > wf.x;
      const bool rb = (long long)wb.y > ia && (long long)wb.x < ib && wb.y > wb.x;
      const unsigned mf = group_bits(__ballot(rf), lane), mb = group_bits(__ballot(rb), lane);
      if (!done_f) {
        if (mf) { first = base + __builtin_ctz(mf); done_f = true; }
        else { base += RNG_G; done_f = base >= l1; }
      }
      if (!done_b) {
        if (mb) { last = top - __builtin_ctz(mb); done_b = true; }
        else { top -= RNG_G; done_b = top <= l0; }
      }
    }
    if (first >= l1) last = first = l1;  // no line of the bracket reaches the tile
  }
  if (live && sub == 0) {
    // a hot tile keeps its first RTX_SPLIT_PART candidates; every further part becomes an item of the work list (consecutive
    // slots, part 1 first). The list is sized from a host-side bound (rtx_split_bound) and cannot overflow; should it ever,
    // the tile is simply left whole.
    const long long cnt = last - first;
    if (a.items && cnt > RTX_SPLIT_MIN) {
      const int extra = (int)((cnt - 1) / RTX_SPLIT_PART);
      const long long at = (long long)atomicAdd(a.n_items, extra);
      if (at + extra <= a.items_cap) {
        for (int p = 1; p <= extra; ++p) {
          SplitItem it;
          it.tile = t; it.k = k; it.part = p; it.extra = extra; it.pad0 = it.pad1 = 0;
          it.lo = (int)(first + (long long)p * RTX_SPLIT_PART);
          const long long e = first + (long long)(p + 1) * RTX_SPLIT_PART;
          it.hi = (int)(e < last ? e : last);
          a.items[at + p - 1] = it;
        }
        last = first + RTX_SPLIT_PART;
      }
    }
    a.ranges[(size_t)k * a.n_tiles + t] = make_int2((int)first, (int)last);
  }
}

int rtx_voigt_scatter_tile_points(void);
// points per line-sum workgroup tile: capacity granularity of rtx_prep_create, and the alignment at which a wavenumber
// shard reproduces the full grid's tiles (hence its bits: dist.hsi_cube_from_atmosphere)
extern "C" int rtx_voigt_tile_points(void) { return rtx_voigt_scatter_tile_points(); }

// rtx_voigt_scatter.hip
int rtx_voigt_sum_scatter(const rtx_prep* P, const rtx_grid* grid, int n_layers, float* out_f32, double* out_f64, int64_t ld,
                          hipStream_t st, void (*launch_ranges)(const rtx_prep*, const rtx_grid*, int, int, int, hipStream_t, int),
                          int nodal);

static void launch_tile_ranges(const rtx_prep* P, const rtx_grid* grid, int n_layers, int n_tiles, int tile, hipStream_t st, int split) {
  RangeArgs ra;
  const bool cut = split && P->split_bound > 0 && P->items;
  ra.items = cut ? P->items : nullptr; ra.n_items = P->n_items; ra.items_cap = P->items_cap;
  ra.ic = P->ic; ra.maxhw = P->maxhw; ra.win = P->win; ra.n_lines = P->n_lines; ra.n_tiles = n_tiles; ra.tile = tile;
  ra.n_layers = n_layers; ra.n = grid->n; ra.ranges = P->ranges;
  int n_steps = 1;  // 17-ary search: each step divides the bracket by 17 (rounded up), one more finishes the last <= 16 elements
                    // (the step count is checked against bisect for every size class in a NumPy restatement: tests/test_host.py)
  for (long long w = P->n_lines; w > 0; w /= (RNG_G + 1)) ++n_steps;
  ra.n_steps = n_steps;
  hipLaunchKernelGGL(tile_ranges_kernel, dim3((n_tiles * RNG_G + 255) / 256, n_layers), dim3(256), 0, st, ra);
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
