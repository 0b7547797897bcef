// Internal definitions shared by the HIP translation units of libradtxfr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/radtxfr_hip.h"

// ---- error plumbing (thread-local text behind rtx_last_error) ---------------------------------
void rtx_set_error(const char* fmt, ...);
#define RTX_FAIL(...)          \
  do {                         \
    rtx_set_error(__VA_ARGS__); \
    return 1;                  \
  } while (0)
#define RTX_HIP(call)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) RTX_FAIL("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
  } while (0)
#define RTX_LAUNCH_CHECK()                                                                   \
  do {                                                                                       \
    hipError_t e_ = hipGetLastError();                                                       \
    if (e_ != hipSuccess) RTX_FAIL("%s:%d kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
  } while (0)

// ---- device-side view of the spectral grid ------------------------------------------------------
struct GridDev {
  double xmin, xmax, step;
  long long n_total, offset, n;
};
static inline GridDev to_dev(const rtx_grid* g) {
  GridDev d;
  d.xmin = g->xmin; d.xmax = g->xmax; d.step = g->step;
  d.n_total = g->n_total; d.offset = g->offset; d.n = g->n;
  return d;
}
int rtx_check_grid(const rtx_grid* g);

// X[ig] for a GLOBAL index exactly as np.linspace builds it: arange*step + start (two roundings,
// never fused), last point pinned to xmax. Indices outside [0,n_total) extrapolate.
__device__ __forceinline__ double grid_x(const GridDev& g, long long ig) {
  double v = __dadd_rn(__dmul_rn((double)ig, g.step), g.xmin);
  return (ig == g.n_total - 1) ? g.xmax : v;
}

// ---- XCD-aware tile order of the line-sum kernels ------------------------------------------------
// Workgroup b runs on XCD b & 7 (round-robin dispatch). An XCD walks chunks of RTX_XCD_CHUNK consecutive tiles, so
// neighbouring tiles share their line records in that XCD's L2; the chunks are dealt round-robin over the XCDs, so
// every XCD sees the whole spectrum. (One contiguous eighth of the spectrum per XCD left the XCD with the highest
// wavenumbers -- widest Doppler cores, most Weideman rows -- as the straggler of every launch.)
#ifndef RTX_XCD_CHUNK
#define RTX_XCD_CHUNK 16
#endif
__device__ __forceinline__ int xcd_tile(int b) {
  const int idx = b >> 3, c = idx / RTX_XCD_CHUNK, w = idx - c * RTX_XCD_CHUNK;
  return (c * 8 + (b & 7)) * RTX_XCD_CHUNK + w;
}
static inline int xcd_slots(int n_tiles) {  // block slots per XCD: grid.x = 8 * xcd_slots(n_tiles)
  const int n_chunks = (n_tiles + RTX_XCD_CHUNK - 1) / RTX_XCD_CHUNK;
  return ((n_chunks + 7) / 8) * RTX_XCD_CHUNK;
}

// ---- Planck radiance in fp32 from an fp64 exponent (shared by the TUD and at-sensor kernels) ----------
// B(nu,T) in uW/(cm^2 sr cm^-1):  c1*(100 nu)^3*1e4 / (exp(c2*100 nu/T) - 1)
__device__ __forceinline__ float planck_f32(double c1x3, double x, double c2l2e_over_T) {
  const double t = x * c2l2e_over_T;  // log2 of the exponential, fp64
  if (t < 1.5) {                      // small arguments (far-IR / microwave): expm1 in fp64
    return (float)(c1x3 / expm1(t * 0.6931471805599453));
  }
  const double n = rint(t);
  const float f = (float)(t - n);     // |f| <= 1/2, exact difference
  const float e = ldexpf(__builtin_amdgcn_exp2f(f), (int)n);  // inf above 2^128: B -> 0, as it should
  return (float)c1x3 * __builtin_amdgcn_rcpf(e - 1.0f);
}

// ---- per-(line,layer) records written by the prologue, read by the line-sum --------------------
// fp32 record (48 B): everything the asymptotic (far-wing) evaluation needs, relative to the grid.
// x(i) = (i - i0)*a + c ;  contribution = (xx*Ay + Ay0) / ((xx + b1)*xx + b0),  xx = x*x.
struct __attribute__((aligned(16))) LineRec {
  float a;    // x per grid index = step*cte,  cte = sqrt(ln2)/GammaD
  float c;    // x at grid index i0            = (X[i0]-nu0')*cte
  float b1;   // 2y^2 - 1,        y = Gamma0*cte
  float b0;   // (y^2 + 1/2)^2
  float Ay;   // A*y/sqrt(pi)
  float Ay0;  // Ay*(y^2 + 1/2)
  float y;    // for the fp32 Weideman branch
  float A;    // weight*S(T)*cte/sqrt(pi)*scale  (strength times the profile's prefactor)
  int i0;     // LOCAL grid index nearest to the shifted centre nu0' (may lie outside [0,n))
  int lo;     // window = local indices [lo,hi): bisect(X,nu0-W), bisect(X,nu0+W) clipped to the shard
  int hi;
  int zw;     // half-width, in grid points, of the band around i0 that can hold |x|+y<15 (0: none)
};
// fp64 companion (32 B), read only where the Weideman region is entered.
struct __attribute__((aligned(16))) LineRec64 {
  double sg0;  // nu0 + Shift0
  double cte;
  double y;
  double A;
};

// fp64 record of the speed-dependent Voigt sum (rtx_sdvoigt_sum): PROFILE_SDVOIGT's arguments (misc/hapi.py:10897)
// and the line's weight * S(T).
struct __attribute__((aligned(16))) LineRecSD {
  double nu, cte, Gam0, Shift0, Gam2, WS;
  double inv_Gam2, csqrtY;  // 1 / Gam2 and 1 / (2 cte Gam2): per-line reciprocals the line-sum would otherwise redo per point
};

struct rtx_lines {
  long long n;
  int n_species;
  double *nu, *sw, *elower, *gamma_air, *gamma_self, *n_air, *n_self, *delta_air, *deltap_air, *delta_self;
  double *sd_air, *sd_self;  // optional speed-dependence columns (rtx_lines_set_sd), NULL = 0
  double* deltap_self;       // optional (rtx_lines_set_deltap_self), NULL = 0
  double* zn;                // per line: exp(-c2 E''/Tref) (1 - exp(-c2 nu/Tref)), the layer-independent half of S(T), formed once at creation
  int* species;
  // host side, for the bound on candidates per tile (hot-tile split, below): the sorted centres and column extremes
  double* nu_host;
  double ga_max, gs_max, n_lo, n_hi;
};

// ---- hot tiles ---------------------------------------------------------------------------------------
// Real line lists cluster (band heads: thousands of lines inside a cm^-1), and one line-sum workgroup per (tile, layer)
// then serialises the launch on its few hot tiles (the clustered synthetic table: slowest workgroup = 2x the ideal
// duration of the whole launch, tools/tile_spread.py). A tile with more than RTX_SPLIT_MIN candidates is therefore cut
// into parts of RTX_SPLIT_PART consecutive candidates: part 0 stays with the tile's own workgroup, every further part is
// an item of a work list (written by tile_ranges_kernel) that extra workgroups evaluate into a workspace of partial
// tiles, and a last kernel adds a tile's parts to its optical depths in part order. The cut is a function of the
// (canonical) candidate range alone, so results stay bit-reproducible and independent of how the axis is sharded.
#ifndef RTX_SPLIT_PART
#define RTX_SPLIT_PART 256  // candidates per part (clustered C3 table, prologue + line-sum: 512 -> 2.83 ms, 256 -> 2.57, 128 -> 2.83; unsplit 5.47)
#endif
#ifndef RTX_SPLIT_MIN
#define RTX_SPLIT_MIN 768   // tiles with more candidates than this are cut: three times a part, so that the ~100-300 candidates of an ordinary tile never are
#endif
struct __attribute__((aligned(16))) SplitItem {
  int tile, k;    // tile of the shard, layer
  int lo, hi;     // candidate slots [lo, hi)
  int part;       // 1 .. extra
  int extra;      // number of extra parts of this (tile, layer); its items are consecutive, part 1 first
  int pad0, pad1;
};

struct rtx_prep {
  long long n_lines;
  int max_layers;
  int n_layers;        // of the last rtx_line_prep
  LineRec* rec;        // [max_layers][n_lines]
  LineRec64* rec64;    // [max_layers][n_lines]
  int* ic;             // [n_lines] local grid index nearest the UNSHIFTED centre (sorted)
  int2* win;           // [max_layers][n_lines] the records' windows (lo, hi) on their own: what tile_ranges_kernel scans (8 of a record's 48 bytes)
  int* maxhw;          // [max_layers] max window half-width in grid points (+margin)
  int2* ranges;        // [max_layers][max_tiles] candidate line range per line-sum tile
  int* smally;         // [max_layers] set when a line of that layer has a Weideman band with y < 1
  LineRecSD* recsd;    // [max_layers][n_lines], allocated by the first speed-dependent prologue
  long long max_tiles;
  double* env;         // device copy of T,p,qratio,weight,mass (packed)
  size_t env_cap;
  double scale;
  // hot-tile split: work list and workspace sized from a HOST-side bound on the candidates per tile (no device read-back)
  int* n_items;        // device counter (lives behind maxhw / smally: one memset per prologue)
  SplitItem* items;    // [items_cap]
  float* part_ws;      // [items_cap][tile points]
  long long items_cap;
  long long split_bound;  // extra parts the current (grid, window) bound allows for; 0 = no tile can be hot
  double split_W;         // window half-width [cm^-1] the bound was computed for (valid for any smaller one)
  double split_xmin, split_step;
  long long split_off, split_n;
  int split_layers;
  const void* split_lines;  // the table the bound was made for
};
