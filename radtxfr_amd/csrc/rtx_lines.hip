// Line table upload + the fp64 per-(line,layer) prologue kernel.
//
// Replaces the per-line environment block of hapi.absorptionCoefficient_Voigt
// (reference misc/hapi.py:11068-11134) -- S(T), GammaD, Gamma0, Shift0, OmegaWingF and the two
// bisect() window bounds -- for all lines x all layers in one launch. fp64 throughout, operations
// in the reference's order, so the hard wing cutoff lands on the same grid points.
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include "rtx_common.h"

// ---- error text ------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void rtx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* rtx_last_error(void) { return g_err; }
extern "C" int rtx_version(void) { return RTX_VERSION; }
extern "C" int rtx_device_info(char* name_h, int len, int* n_cu_h) {
  int dev;
  RTX_HIP(hipGetDevice(&dev));
  hipDeviceProp_t p;
  RTX_HIP(hipGetDeviceProperties(&p, dev));
  if (name_h && len > 0) snprintf(name_h, len, "%s (%s)", p.name, p.gcnArchName);
  if (n_cu_h) *n_cu_h = p.multiProcessorCount;
  return 0;
}

int rtx_check_grid(const rtx_grid* g) {
  if (!g) RTX_FAIL("grid is NULL");
  if (g->n_total < 2) RTX_FAIL("grid needs at least 2 points (n_total=%lld)", (long long)g->n_total);
  if (!(g->step > 0.0)) RTX_FAIL("grid step must be > 0 (ascending axis)");
  if (g->offset < 0 || g->n < 0 || g->offset + g->n > g->n_total)
    RTX_FAIL("grid shard [%lld,+%lld) outside [0,%lld)", (long long)g->offset, (long long)g->n, (long long)g->n_total);
  if (g->n_total > 2000000000LL) RTX_FAIL("grid too long for 32-bit point indices");
  return 0;
}

// ---- line table ------------------------------------------------------------------------------------
static int upload(double** dst, const double* src_h, long long n, bool zero_if_null) {
  *dst = nullptr;
  if (!src_h && !zero_if_null) return 0;
  RTX_HIP(hipMalloc((void**)dst, sizeof(double) * (size_t)(n > 0 ? n : 1)));
  if (src_h)
    RTX_HIP(hipMemcpy(*dst, src_h, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
  else
    RTX_HIP(hipMemset(*dst, 0, sizeof(double) * (size_t)n));
  return 0;
}

extern "C" int rtx_lines_free(rtx_lines* L) {
  if (!L) return 0;
  double* p[] = {L->nu, L->sw, L->elower, L->gamma_air, L->gamma_self, L->n_air, L->n_self, L->delta_air, L->deltap_air, L->delta_self,
                 L->sd_air, L->sd_self, L->deltap_self, L->zn};
  for (double* q : p)
    if (q) (void)hipFree(q);
  if (L->species) (void)hipFree(L->species);
  free(L->nu_host);
  delete L;
  return 0;
}

int rtx_lines_fill_zn(rtx_lines* L);
extern "C" int rtx_lines_create(int64_t n, int n_species, const double* nu_h, const double* sw_h, const double* elower_h,
                                const double* gamma_air_h, const double* gamma_self_h, const double* n_air_h,
                                const double* n_self_h, const double* delta_air_h, const double* deltap_air_h,
                                const double* delta_self_h, const int32_t* species_h, rtx_lines** out) {
  if (!out) RTX_FAIL("out is NULL");
  *out = nullptr;
  if (n < 0 || n > 2000000000LL) RTX_FAIL("bad line count %lld", (long long)n);
  if (n_species < 1 || n_species > 4096) RTX_FAIL("n_species must be in [1,4096], got %d", n_species);
  if (n > 0 && (!nu_h || !sw_h || !elower_h || !gamma_air_h || !gamma_self_h || !n_air_h || !delta_air_h || !species_h))
    RTX_FAIL("a required line-table column is NULL");
  for (int64_t i = 0; i < n; ++i) {
    if (i > 0 && nu_h[i] < nu_h[i - 1]) RTX_FAIL("line table must be sorted by nu (row %lld)", (long long)i);
    if (species_h[i] < 0 || species_h[i] >= n_species) RTX_FAIL("species index out of range at row %lld", (long long)i);
  }
  rtx_lines* L = new rtx_lines();
  memset(L, 0, sizeof(*L));
  L->n = n;
  L->n_species = n_species;
  int rc = 0;
  rc |= upload(&L->nu, nu_h, n, true);
  rc |= upload(&L->sw, sw_h, n, true);
  rc |= upload(&L->elower, elower_h, n, true);
  rc |= upload(&L->gamma_air, gamma_air_h, n, true);
  rc |= upload(&L->gamma_self, gamma_self_h, n, true);
  rc |= upload(&L->n_air, n_air_h, n, true);
  rc |= upload(&L->n_self, n_self_h, n, false);
  rc |= upload(&L->delta_air, delta_air_h, n, true);
  rc |= upload(&L->deltap_air, deltap_air_h, n, false);
  rc |= upload(&L->delta_self, delta_self_h, n, false);
  if (!rc) {
    hipError_t e = hipMalloc((void**)&L->species, sizeof(int) * (size_t)(n > 0 ? n : 1));
    if (e == hipSuccess && n > 0) e = hipMemcpy(L->species, species_h, sizeof(int) * (size_t)n, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      rtx_set_error("species upload failed: %s", hipGetErrorString(e));
      rc = 1;
    }
  }
  if (!rc) rc = rtx_lines_fill_zn(L);
  if (!rc && n > 0) {
    L->nu_host = (double*)malloc(sizeof(double) * (size_t)n);
    if (!L->nu_host) { rtx_set_error("out of host memory"); rc = 1; }
  }
  if (rc) {
    rtx_lines_free(L);
    return 1;
  }
  L->n_lo = 1e300; L->n_hi = -1e300;
  for (int64_t i = 0; i < n; ++i) {
    L->nu_host[i] = nu_h[i];
    L->ga_max = fmax(L->ga_max, fabs(gamma_air_h[i]));
    L->gs_max = fmax(L->gs_max, fabs(gamma_self_h[i]));
    const double na = n_air_h[i], ns = (n_self_h && n_self_h[i] != 0.0) ? n_self_h[i] : na;
    L->n_lo = fmin(L->n_lo, fmin(na, ns));
    L->n_hi = fmax(L->n_hi, fmax(na, ns));
  }
  *out = L;
  return 0;
}
extern "C" int64_t rtx_lines_count(const rtx_lines* L) { return L ? L->n : -1; }

extern "C" int rtx_lines_set_sd(rtx_lines* L, const double* sd_air_h, const double* sd_self_h) {
  if (!L) RTX_FAIL("lines is NULL");
  const double* src[2] = {sd_air_h, sd_self_h};
  double** dst[2] = {&L->sd_air, &L->sd_self};
  for (int c = 0; c < 2; ++c) {
    if (*dst[c]) { (void)hipFree(*dst[c]); *dst[c] = nullptr; }
    if (!src[c] || L->n == 0) continue;
    RTX_HIP(hipMalloc((void**)dst[c], (size_t)L->n * sizeof(double)));
    RTX_HIP(hipMemcpy(*dst[c], src[c], (size_t)L->n * sizeof(double), hipMemcpyHostToDevice));
  }
  return 0;
}

extern "C" int rtx_lines_set_deltap_self(rtx_lines* L, const double* deltap_self_h) {
  if (!L) RTX_FAIL("lines is NULL");
  if (L->deltap_self) { (void)hipFree(L->deltap_self); L->deltap_self = nullptr; }
  if (!deltap_self_h || L->n == 0) return 0;
  RTX_HIP(hipMalloc((void**)&L->deltap_self, (size_t)L->n * sizeof(double)));
  RTX_HIP(hipMemcpy(L->deltap_self, deltap_self_h, (size_t)L->n * sizeof(double), hipMemcpyHostToDevice));
  return 0;
}

// ---- prep object -------------------------------------------------------------------------------------
extern "C" int rtx_prep_free(rtx_prep* P) {
  if (!P) return 0;
  if (P->rec) (void)hipFree(P->rec);
  if (P->rec64) (void)hipFree(P->rec64);
  if (P->ic) (void)hipFree(P->ic);
  if (P->win) (void)hipFree(P->win);
  if (P->maxhw) (void)hipFree(P->maxhw);  // smally lives in the same allocation
  if (P->env) (void)hipFree(P->env);
  if (P->ranges) (void)hipFree(P->ranges);
  if (P->recsd) (void)hipFree(P->recsd);
  if (P->items) (void)hipFree(P->items);
  if (P->part_ws) (void)hipFree(P->part_ws);
  delete P;
  return 0;
}

extern "C" int rtx_voigt_tile_points(void);

extern "C" int rtx_prep_create(const rtx_lines* lines, int max_layers, int64_t max_points, rtx_prep** out) {
  if (!out) RTX_FAIL("out is NULL");
  *out = nullptr;
  if (!lines) RTX_FAIL("lines is NULL");
  if (max_layers < 1 || max_layers > 4096) RTX_FAIL("max_layers must be in [1,4096]");
  if (max_points < 1 || max_points > 2000000000LL) RTX_FAIL("max_points must be in [1,2e9]");
  rtx_prep* P = new rtx_prep();
  memset(P, 0, sizeof(*P));
  P->n_lines = lines->n;
  P->max_layers = max_layers;
  size_t nrec = (size_t)(lines->n > 0 ? lines->n : 1) * (size_t)max_layers;
  P->env_cap = (size_t)max_layers * (2 + 2 * (size_t)lines->n_species) + (size_t)lines->n_species;
  const int tile = rtx_voigt_tile_points();
  P->max_tiles = (max_points + tile - 1) / tile;
  hipError_t e = hipMalloc((void**)&P->rec, nrec * sizeof(LineRec));
  if (e == hipSuccess) e = hipMalloc((void**)&P->rec64, nrec * sizeof(LineRec64));
  if (e == hipSuccess) e = hipMalloc((void**)&P->ic, sizeof(int) * (size_t)(lines->n > 0 ? lines->n : 1));
  if (e == hipSuccess) e = hipMalloc((void**)&P->win, nrec * sizeof(int2));
  if (e == hipSuccess) e = hipMalloc((void**)&P->maxhw, (2 * (size_t)max_layers + 1) * sizeof(int));  // [maxhw | smally | n_items]: one memset per prologue
  if (e == hipSuccess) { P->smally = P->maxhw + max_layers; P->n_items = P->maxhw + 2 * max_layers; }
  if (e == hipSuccess) e = hipMalloc((void**)&P->env, sizeof(double) * P->env_cap);
  if (e == hipSuccess) e = hipMalloc((void**)&P->ranges, sizeof(int2) * (size_t)P->max_tiles * (size_t)max_layers);
  if (e != hipSuccess) {
    rtx_set_error("rtx_prep_create: %s (%zu records)", hipGetErrorString(e), nrec);
    rtx_prep_free(P);
    return 1;
  }
  *out = P;
  return 0;
}

// ---- prologue kernel --------------------------------------------------------------------------------
// hapi constants (misc/hapi.py:84-92, :10171, :11085)
#define H_CBOLTS 1.380648813e-16
#define H_CC 2.99792458e10
#define H_CMASSMOL 1.66053873e-27
#define H_C2 1.4388028496642257
#define H_TREF 296.0
#ifndef RTX_PREP_ABLATE
#define RTX_PREP_ABLATE 0
#endif
#define RTX_ENV_MAX 416  /* doubles of per-layer tables carried in the prologue's kernel arguments (3.3 KB of the 4 KB) */

// The reference-temperature half of S(T) (misc/hapi.py:10171-10172) depends on the line alone: formed once per table, by the
// same expression the prologue used to evaluate per (line, layer) -- two of its four fp64 exponentials and two divisions.
__global__ __launch_bounds__(256) void lines_zn_kernel(const double* __restrict__ nu, const double* __restrict__ elower, double* __restrict__ zn,
                                                       long long n) {
  const long long l = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (l < n) zn[l] = exp(-H_C2 * elower[l] / H_TREF) * (1.0 - exp(-H_C2 * nu[l] / H_TREF));
}
int rtx_lines_fill_zn(rtx_lines* L) {
  if (L->n == 0) return 0;
  RTX_HIP(hipMalloc((void**)&L->zn, sizeof(double) * (size_t)L->n));
  hipLaunchKernelGGL(lines_zn_kernel, dim3((unsigned)((L->n + 255) / 256)), dim3(256), 0, 0, L->nu, L->elower, L->zn, (long long)L->n);
  RTX_LAUNCH_CHECK();
  RTX_HIP(hipDeviceSynchronize());
  return 0;
}

struct PrepArgs {
  const double *nu, *sw, *elower, *gamma_air, *gamma_self, *n_air, *n_self, *delta_air, *deltap_air, *delta_self;
  const double *sd_air, *sd_self, *deltap_self;
  const double* zn;
  LineRecSD* recsd;
  const int* species;
  long long n_lines;
  int n_layers, n_species;
  const double *T, *p, *qratio, *weight, *mass;  // the small per-layer tables: device copies, or offsets into env[] (ENV_ARGS)
  double env[RTX_ENV_MAX];                       // T | p | qratio | weight | mass packed, when they fit (no H2D copy at all)
  double dil_air, dil_self, omega_wing, omega_wing_hw, thresh, scale;
  int profile;  // RTX_PROFILE_VOIGT / _LORENTZ / _DOPPLER
  GridDev g;
  LineRec* rec;
  LineRec64* rec64;
  int* ic;
  int2* win;
  int* maxhw;
  int* smally;
};

// The per-layer tables (T, p, qratio, weight, mass: 324 doubles for 4 species x 32 layers) travel in the kernel arguments
// when they fit: the prologue then needs no host-to-device copy, which was 5 hipMemcpyAsync calls (~10 us of host time
// each) per atmosphere. Larger sets go through the prep object's device buffer as before.
// bisect.bisect (= bisect_right) of value v on the FULL grid: number of grid points <= v.
__device__ long long grid_bisect_right(const GridDev& g, double v) {
  double t = (v - g.xmin) / g.step;
  long long k;
  if (!(t > -1.0)) k = 0;
  else if (t >= (double)g.n_total) k = g.n_total;
  else k = (long long)floor(t) + 1;
  if (k < 0) k = 0;
  if (k > g.n_total) k = g.n_total;
#if !(RTX_PREP_ABLATE & 4)
  while (k < g.n_total && grid_x(g, k) <= v) ++k;
  while (k > 0 && grid_x(g, k - 1) > v) --k;
#endif
  return k;
}

__device__ __forceinline__ int clamp_local(long long ig, const GridDev& g) {
  long long l = ig - g.offset;
  if (l < 0) l = 0;
  if (l > g.n) l = g.n;
  return (int)l;
}

// Local centre index clamped to [-1e8, n+1e8]: with n <= 2e9 and zw <= 4e7 every i - i0, lo - i0 and
// i0 +- zw the line-sum forms stays inside int32. A centre that far outside cannot reach the shard.
__device__ __forceinline__ int sat_local(long long v, long long n) {
  const long long M = 100000000LL;
  return (int)(v < -M ? -M : (v > n + M ? n + M : v));
}

template <bool ENV_ARGS>
__global__ __launch_bounds__(256) void line_prep_kernel(PrepArgs a) {
  const long long l = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  int my_hw = 0;
  // ENV_ARGS: a.T ... a.mass hold element offsets into a.env (kernel-argument segment), else device pointers
  const double* eT = ENV_ARGS ? a.env + (size_t)a.T : a.T;
  const double* ep = ENV_ARGS ? a.env + (size_t)a.p : a.p;
  const double* eq = ENV_ARGS ? a.env + (size_t)a.qratio : a.qratio;
  const double* ew = ENV_ARGS ? a.env + (size_t)a.weight : a.weight;
  const double* em = ENV_ARGS ? a.env + (size_t)a.mass : a.mass;
  if (l < a.n_lines) {
    const GridDev& g = a.g;
    const double T = eT[k], p = ep[k];
    const double nu = a.nu[l];
    const int sp = a.species[l];
    const double w = ew[(size_t)sp * a.n_layers + k];
    // GammaD, misc/hapi.py:11085-11087
    const double m = em[sp] * H_CMASSMOL * 1000.0;
    double GammaD = sqrt(2.0 * H_CBOLTS * T * log(2.0) / m / (H_CC * H_CC)) * nu;
    if (a.profile == RTX_PROFILE_DOPPLER)  // absorptionCoefficient_Doppler's own SI constants, misc/hapi.py:11534-11538
      GammaD = (1.1774100225 / 2.99792458e8) * sqrt(1.3806503e-23 / 1.66053873e-27) * sqrt(T) * nu / sqrt(em[sp]);
    // Gamma0 / Shift0 over the diluent mix, misc/hapi.py:11090-11128
    double Gamma0 = 0.0, Shift0 = 0.0;
    const double tr = H_TREF / T;
    // (Tref/T)^n as exp(n log(Tref/T)): the logarithm is the same for every line of the layer; |n log(Tref/T)| < 1, so the
    // result is within 2 ulp of pow's (whose extended-precision logarithm is 200 fp64 instructions of this kernel's ~1000)
    const double ltr = log(tr);
    if (a.dil_air != 0.0) {
#if RTX_PREP_ABLATE & 1  /* timing experiments only */
      Gamma0 += a.dil_air * (a.gamma_air[l] * p / 1.0 * (tr * a.n_air[l]));
#else
      Gamma0 += a.dil_air * (a.gamma_air[l] * p / 1.0 * exp(a.n_air[l] * ltr));
#endif
      const double dp = a.deltap_air ? a.deltap_air[l] : 0.0;
      Shift0 += a.dil_air * ((a.delta_air[l] + dp * (T - H_TREF)) * p / 1.0);
    }
    if (a.dil_self != 0.0) {
      double ns = a.n_self ? a.n_self[l] : a.n_air[l];
      if (a.n_self && ns == 0.0) ns = a.n_air[l];
      Gamma0 += a.dil_self * (a.gamma_self[l] * p / 1.0 * exp(ns * ltr));
      const double ds = a.delta_self ? a.delta_self[l] : 0.0;
      const double dps = a.deltap_self ? a.deltap_self[l] : 0.0;
      Shift0 += a.dil_self * ((ds + dps * (T - H_TREF)) * p / 1.0);
    }
    if (a.profile == RTX_PROFILE_DOPPLER) {  // no pressure broadening; Shift0 = delta_air * p (misc/hapi.py:11543), set by the host through dil_air = 1 / 0 (LineShift)
      Gamma0 = 0.0;
      Shift0 = a.dil_air * a.delta_air[l] * p;
    }
    // OmegaWingF and the window: Voigt misc/hapi.py:11131-11134, Lorentz :11364, Doppler :11540
    const double W = a.profile == RTX_PROFILE_LORENTZ   ? fmax(a.omega_wing, a.omega_wing_hw * Gamma0)
                     : a.profile == RTX_PROFILE_DOPPLER ? fmax(a.omega_wing, a.omega_wing_hw * GammaD)
                                                        : fmax(a.omega_wing, fmax(a.omega_wing_hw * Gamma0, a.omega_wing_hw * GammaD));
    long long glo = grid_bisect_right(g, nu - W);
    long long ghi = grid_bisect_right(g, nu + W);
    int lo = clamp_local(glo, g), hi = clamp_local(ghi, g);
    const double sg0 = nu + Shift0;
    const long long M = 1000000000LL;
    if (k == 0) {
      long long gic = llrint((nu - g.xmin) / g.step);
      if (gic < -M) gic = -M;
      if (gic > M) gic = M;
      a.ic[l] = sat_local(gic - g.offset, g.n);
    }
    // A wavenumber SHARD (one rank of N) is reached by ~1/N of the table, but every rank runs this prologue over all of it
    // (the candidate ranges, hence the order of the fp32 sums, must not depend on where the shard is cut: maxhw below is a
    // maximum over the whole table). A line whose window ends a grid step or more outside the shard contributes nothing
    // here and its record is read at most as a rejected tile candidate, so the
    // rest of the work (S(T): four fp64 exponentials, the profile constants, 80 bytes of records) is skipped for it. Not
    // with an intensity threshold (whether the line counts for maxhw then depends on S) and not for the SD-Voigt records.
    const bool pre_dropped = !(w != 0.0) || !(GammaD > 0.0) || (a.profile == RTX_PROFILE_LORENTZ && !(Gamma0 > 0.0));
    const bool outside = nu + W < grid_x(g, g.offset) - g.step || nu - W > grid_x(g, g.offset + g.n - 1) + g.step;
    const bool skip = outside && !(a.thresh > 0.0) && a.profile != RTX_PROFILE_SDVOIGT;
    if (skip) {
      long long gi0 = llrint((sg0 - g.xmin) / g.step);
      if (gi0 < -M) gi0 = -M;
      if (gi0 > M) gi0 = M;
      const size_t o = (size_t)k * (size_t)a.n_lines + (size_t)l;
      // lo = hi = 0 rejects it; the other fields are finite and give a zero contribution with a non-zero denominator (the
      // line-sum fills the empty slots of a group of 8 from whatever candidate lane is at hand and zeroes the numerator)
      LineRec r;
      r.a = 0.f; r.c = 0.f; r.b1 = 0.f; r.b0 = 1.f; r.Ay = 0.f; r.Ay0 = 0.f; r.y = 15.f; r.A = 0.f;
      r.i0 = sat_local(gi0 - g.offset, g.n); r.lo = 0; r.hi = 0; r.zw = 0;
      a.rec[o] = r;
      a.win[o] = make_int2(0, 0);
      if (!pre_dropped && ghi > glo) {
        double hw = ceil(W / g.step) + 2.0;
        my_hw = hw > 1.0e9 ? 1000000000 : (int)hw;
      }
    } else {
    // S(T): EnvironmentDependency_Intensity, misc/hapi.py:10169-10175 (SigmaTref/SigmaT = qratio)
    const double el = a.elower[l];
#if RTX_PREP_ABLATE & 2
    const double ch = (-H_C2 * el / T) * (1.0 - (-H_C2 * nu / T));
    const double zn = (-H_C2 * el / H_TREF) * (1.0 - (-H_C2 * nu / H_TREF));
#else
    const double ch = exp(-H_C2 * el / T) * (1.0 - exp(-H_C2 * nu / T));
    const double zn = a.zn[l];  // exp(-H_C2 * el / H_TREF) * (1.0 - exp(-H_C2 * nu / H_TREF)), from table creation (lines_zn_kernel)
#endif
    const double S = a.sw[l] * eq[(size_t)sp * a.n_layers + k] * ch / zn;
    const bool dropped = pre_dropped || (S < a.thresh);
    if (dropped || hi <= lo) { lo = 0; hi = 0; }
    // profile parameters: pcqsdhc PART1, misc/hapi.py:9900-9915
    // Lorentz (PROFILE_LORENTZ, misc/hapi.py:10150): with x = (nu - sg0)/Gamma0 the profile is (1/(pi Gamma0)) / (x^2 + 1)
    // = (x^2 K + K) / ((x^2 + 2) x^2 + 1), i.e. the line-sum's far-wing rational with b1 = 2, b0 = 1, Ay = Ay0 = K:
    // the same kernels evaluate it everywhere (no band: y is set to 15), poles at |nu - sg0| = Gamma0 as for Voigt.
    // Doppler (PROFILE_DOPPLER, :10160) is the Voigt profile at y = 0: Re w(x) = exp(-x^2); the band |x| < 15 goes
    // through the fp64 Weideman pass, beyond it the reference's exp(-225) = 1e-98 is dropped.
    const bool lor = a.profile == RTX_PROFILE_LORENTZ;
    const double cte = lor ? 1.0 / Gamma0 : sqrt(log(2.0)) / GammaD;
    const double y = lor ? 15.0 : Gamma0 * cte;
    const double A = dropped ? 0.0 : (lor ? w * S * cte / M_PI * a.scale : w * S * cte / sqrt(M_PI) * a.scale);
    // nearest grid index to the shifted centre (global), then the residual in fp64
    long long gi0 = llrint((sg0 - g.xmin) / g.step);
    if (gi0 < -M) gi0 = -M;
    if (gi0 > M) gi0 = M;
    const double frac_x = (grid_x(g, gi0) - sg0) * cte;  // x at gi0, |.| <= a/2 when inside the grid
    const double ax = g.step * cte;
    LineRec r;
    r.a = (float)ax;
    r.c = (float)frac_x;
    const double yh = y * y + 0.5;
    r.b1 = lor ? 2.0f : (float)(2.0 * (y * y) - 1.0);
    r.b0 = lor ? 1.0f : (float)(yh * yh);
    r.Ay = lor ? (float)A : (float)(A * y * 0.56418958354775628);
    r.Ay0 = lor ? (float)A : (float)(A * y * 0.56418958354775628 * yh);
    r.y = (float)y;
    r.A = (float)A;
    r.i0 = sat_local(gi0 - g.offset, g.n);
    r.lo = lo;
    r.hi = hi;
    // half-width (grid points) of the band that can satisfy |x|+y<15 (hum1_wei switch, misc/hapi.py:9840)
    int zw = 0;
    if (y < 15.0 && hi > lo) {
      double z = ceil((15.0 - y) / ax) + 2.0;
      zw = z > 4.0e7 ? 40000000 : (int)z;
    }
    r.zw = zw;
    if (zw > 0 && r.y < 1.0f) a.smally[k] = 1;  // benign race: every writer stores 1
    LineRec64 r64;
    r64.sg0 = sg0; r64.cte = cte; r64.y = y; r64.A = A;
    const size_t o = (size_t)k * (size_t)a.n_lines + (size_t)l;
#if !(RTX_PREP_ABLATE & 8)
    a.rec[o] = r;
#ifndef RTX_PREP_ABLATE_REC64
    a.rec64[o] = r64;
#endif
#endif
    a.win[o] = make_int2(lo, hi);
    if (a.profile == RTX_PROFILE_SDVOIGT) {
      // Gamma2 = sum_species abun * SD_species * p/pref * gamma_species(Tref) (misc/hapi.py:10884-10890); Shift2 = 0
      double Gam2 = 0.0;
      if (a.dil_air != 0.0 && a.sd_air) Gam2 += a.dil_air * (a.sd_air[l] * p) * a.gamma_air[l];
      if (a.dil_self != 0.0 && a.sd_self) Gam2 += a.dil_self * (a.sd_self[l] * p) * a.gamma_self[l];
      LineRecSD q;
      q.nu = nu; q.cte = cte; q.Gam0 = Gamma0; q.Shift0 = Shift0; q.Gam2 = Gam2; q.WS = dropped ? 0.0 : w * S;
      q.inv_Gam2 = Gam2 != 0.0 ? 1.0 / Gam2 : 0.0;
      q.csqrtY = Gam2 != 0.0 ? 1.0 / (2.0 * cte * Gam2) : 0.0;
      a.recsd[o] = q;
    }
    if (!dropped && ghi > glo) {
      // window half-width in grid points, measured from the unshifted centre, with margin. Taken over every live line
      // whose window meets the FULL axis, not only this shard: the per-tile candidate ranges -- hence the order of the
      // fp32 sums -- then do not depend on where a shard is cut (tile-aligned shards reproduce the full grid's bits)
      double hw = ceil(W / g.step) + 2.0;
      my_hw = hw > 1.0e9 ? 1000000000 : (int)hw;
    }
    }  // !skip
  }
  // block max -> one atomic per block
  for (int off = 32; off > 0; off >>= 1) my_hw = max(my_hw, __shfl_down(my_hw, off));
  __shared__ int s_hw[4];
  if ((threadIdx.x & 63) == 0) s_hw[threadIdx.x >> 6] = my_hw;
  __syncthreads();
  if (threadIdx.x == 0) {
    int v = max(max(s_hw[0], s_hw[1]), max(s_hw[2], s_hw[3]));
    if (v > 0) atomicMax(&a.maxhw[k], v);
  }
}

// ---- hot tiles: host-side bound on the extra parts (rtx_common.h) ------------------------------------------------------
// No window of any line in any layer is wider than W = max(OmegaWing, HW * Gamma0_max, HW * GammaD_max) (misc/hapi.py:11131
// with the column extremes of the table and the layer extremes of this call), and windows are centred on the UNSHIFTED
// centres, so a tile [x_a, x_b] has at most count(nu in [x_a - W, x_b + W]) candidates: one sweep over the sorted host
// copy of the centres. The work list and the partial-tile workspace are sized from that bound -- they cannot overflow, and
// nothing is read back from the device. The sweep (~0.1 ms for 100 000 lines) is redone only when the grid changes or a call
// needs a wider window than the cached bound was made for (25 % headroom). A table that cannot have a hot tile (every
// uniform table of the benchmarks) has bound 0 and never launches the two extra kernels.
static int rtx_split_bound(rtx_prep* P, const rtx_lines* L, const rtx_grid* g, int n_layers, const double* T_h, const double* p_h,
                           const double* mass_h, double dil_air, double dil_self, double omega_wing, double omega_wing_hw, int profile) {
  if (L->n == 0 || g->n == 0) { P->split_bound = 0; return 0; }
  double g0 = 0.0, t_max = 0.0;
  for (int k = 0; k < n_layers; ++k) {
    const double tr = H_TREF / T_h[k];
    g0 = fmax(g0, p_h[k] * fmax(pow(tr, L->n_lo), pow(tr, L->n_hi)));
    t_max = fmax(t_max, T_h[k]);
  }
  g0 *= fabs(dil_air) * L->ga_max + fabs(dil_self) * L->gs_max;
  double m_min = 1e300;
  for (int s = 0; s < L->n_species; ++s)
    if (mass_h[s] > 0.0) m_min = fmin(m_min, mass_h[s]);
  if (!(m_min < 1e300)) m_min = 1.0;
  const double nu_max = L->nu_host[L->n - 1];
  double gd = sqrt(2.0 * H_CBOLTS * t_max * log(2.0) / (m_min * H_CMASSMOL * 1000.0) / (H_CC * H_CC)) * fabs(nu_max);
  if (profile == RTX_PROFILE_DOPPLER) gd = (1.1774100225 / 2.99792458e8) * sqrt(1.3806503e-23 / 1.66053873e-27) * sqrt(t_max) * fabs(nu_max) / sqrt(m_min);
  const double W = fmax(omega_wing, fmax(omega_wing_hw * g0, omega_wing_hw * gd)) + 2.0 * g->step;
  const bool same_grid = P->split_xmin == g->xmin && P->split_step == g->step && P->split_off == g->offset && P->split_n == g->n;
  if (same_grid && P->split_lines == (const void*)L && P->split_W > 0.0 && W <= P->split_W && n_layers <= P->split_layers) return 0;  // the cached bound covers this call
  const double Wc = 1.25 * W;
  const int tile = rtx_voigt_tile_points();
  const long long n_tiles = (g->n + tile - 1) / tile;
  long long extra = 0, a = 0, b = 0;
  for (long long t = 0; t < n_tiles; ++t) {
    const double xa = g->xmin + (double)(g->offset + t * tile) * g->step - Wc;
    long long i_end = g->offset + (t + 1) * tile;
    if (i_end > g->offset + g->n) i_end = g->offset + g->n;
    const double xb = g->xmin + (double)i_end * g->step + Wc;
    while (a < L->n && L->nu_host[a] < xa) ++a;
    if (b < a) b = a;
    while (b < L->n && L->nu_host[b] <= xb) ++b;
    const long long cnt = b - a;
    if (cnt > RTX_SPLIT_MIN) extra += (cnt - 1) / RTX_SPLIT_PART;
  }
  extra *= n_layers;
  if (extra > P->items_cap) {  // grow-only; rare (a new table / grid / much wider wings): allocates, hence synchronises
    if (extra > 1000000LL)  // 4 GB of partial tiles: callers with many states per launch split the launch (afit_xs.py)
      RTX_FAIL("hot-tile work list of %lld items (over 1000000): fewer layers / states per call, or a shorter grid shard", extra);
    if (P->items) { RTX_HIP(hipFree(P->items)); P->items = nullptr; }
    if (P->part_ws) { RTX_HIP(hipFree(P->part_ws)); P->part_ws = nullptr; }
    P->items_cap = 0;
    RTX_HIP(hipMalloc((void**)&P->items, (size_t)extra * sizeof(SplitItem)));
    RTX_HIP(hipMalloc((void**)&P->part_ws, (size_t)extra * (size_t)tile * sizeof(float)));
    P->items_cap = extra;
  }
  P->split_bound = extra;
  P->split_W = Wc; P->split_xmin = g->xmin; P->split_step = g->step; P->split_off = g->offset; P->split_n = g->n;
  P->split_layers = n_layers;
  P->split_lines = (const void*)L;
  return 0;
}

extern "C" int64_t rtx_prep_split_bound(const rtx_prep* P) { return P ? P->split_bound : -1; }

extern "C" int rtx_line_prep_profile(rtx_prep* P, const rtx_lines* L, const rtx_grid* grid, int n_layers, const double* T_h,
                             const double* p_atm_h, const double* qratio_h, const double* weight_h, const double* mass_h,
                             double dil_air, double dil_self, double omega_wing, double omega_wing_hw,
                             double intensity_threshold, double scale, int profile, void* stream) {
  if (!P || !L) RTX_FAIL("prep/lines is NULL");
  if (profile < RTX_PROFILE_VOIGT || profile > RTX_PROFILE_SDVOIGT) RTX_FAIL("profile=%d", profile);
  if (profile == RTX_PROFILE_SDVOIGT && !P->recsd) {
    const size_t nrec = (size_t)(P->n_lines > 0 ? P->n_lines : 1) * (size_t)P->max_layers;
    RTX_HIP(hipMalloc((void**)&P->recsd, nrec * sizeof(LineRecSD)));
  }
  if (rtx_check_grid(grid)) return 1;
  if (P->n_lines != L->n) RTX_FAIL("prep object was created for %lld lines, table has %lld", P->n_lines, L->n);
  if (n_layers < 1 || n_layers > P->max_layers) RTX_FAIL("n_layers=%d outside [1,%d]", n_layers, P->max_layers);
  if (!T_h || !p_atm_h || !qratio_h || !weight_h || !mass_h) RTX_FAIL("a per-layer input is NULL");
  if (!(scale > 0.0)) RTX_FAIL("scale must be > 0");
  for (int k = 0; k < n_layers; ++k)
    if (!(T_h[k] > 0.0) || !(p_atm_h[k] >= 0.0)) RTX_FAIL("layer %d: T=%g p=%g not physical", k, T_h[k], p_atm_h[k]);
  hipStream_t st = (hipStream_t)stream;
  const int ns = L->n_species;
  const size_t nT = (size_t)n_layers, nQ = (size_t)ns * n_layers;
  if (2 * nT + 2 * nQ + ns > P->env_cap) RTX_FAIL("environment tables exceed prep capacity");
  const bool env_args = 2 * nT + 2 * nQ + ns <= RTX_ENV_MAX;
  double* d = P->env;
  if (!env_args) {
    // pageable-source async copies are staged by the runtime before returning: caller may reuse its arrays
    RTX_HIP(hipMemcpyAsync(d, T_h, nT * sizeof(double), hipMemcpyHostToDevice, st));
    RTX_HIP(hipMemcpyAsync(d + nT, p_atm_h, nT * sizeof(double), hipMemcpyHostToDevice, st));
    RTX_HIP(hipMemcpyAsync(d + 2 * nT, qratio_h, nQ * sizeof(double), hipMemcpyHostToDevice, st));
    RTX_HIP(hipMemcpyAsync(d + 2 * nT + nQ, weight_h, nQ * sizeof(double), hipMemcpyHostToDevice, st));
    RTX_HIP(hipMemcpyAsync(d + 2 * nT + 2 * nQ, mass_h, ns * sizeof(double), hipMemcpyHostToDevice, st));
  }
  RTX_HIP(hipMemsetAsync(P->maxhw, 0, (2 * (size_t)P->max_layers + 1) * sizeof(int), st));  // maxhw, smally and n_items
  if (rtx_split_bound(P, L, grid, n_layers, T_h, p_atm_h, mass_h, dil_air, dil_self, omega_wing, omega_wing_hw, profile)) return 1;
  P->n_layers = n_layers;
  P->scale = scale;
  if (L->n == 0) return 0;
  PrepArgs a;
  a.nu = L->nu; a.sw = L->sw; a.elower = L->elower; a.gamma_air = L->gamma_air; a.gamma_self = L->gamma_self;
  a.n_air = L->n_air; a.n_self = L->n_self; a.delta_air = L->delta_air; a.deltap_air = L->deltap_air;
  a.delta_self = L->delta_self; a.species = L->species;
  a.sd_air = L->sd_air; a.sd_self = L->sd_self; a.deltap_self = L->deltap_self; a.recsd = P->recsd; a.zn = L->zn;
  a.n_lines = L->n; a.n_layers = n_layers; a.n_species = ns;
  if (env_args) {  // offsets (in doubles) into a.env, carried in the pointer fields
    a.T = (const double*)(size_t)0; a.p = (const double*)nT; a.qratio = (const double*)(2 * nT);
    a.weight = (const double*)(2 * nT + nQ); a.mass = (const double*)(2 * nT + 2 * nQ);
    memcpy(a.env, T_h, nT * sizeof(double));
    memcpy(a.env + nT, p_atm_h, nT * sizeof(double));
    memcpy(a.env + 2 * nT, qratio_h, nQ * sizeof(double));
    memcpy(a.env + 2 * nT + nQ, weight_h, nQ * sizeof(double));
    memcpy(a.env + 2 * nT + 2 * nQ, mass_h, ns * sizeof(double));
  } else {
    a.T = d; a.p = d + nT; a.qratio = d + 2 * nT; a.weight = d + 2 * nT + nQ; a.mass = d + 2 * nT + 2 * nQ;
  }
  a.dil_air = dil_air; a.dil_self = dil_self; a.omega_wing = omega_wing; a.omega_wing_hw = omega_wing_hw;
  a.thresh = intensity_threshold; a.scale = scale; a.profile = profile;
  a.g = to_dev(grid);
  a.rec = P->rec; a.rec64 = P->rec64; a.ic = P->ic; a.win = P->win; a.maxhw = P->maxhw; a.smally = P->smally;
  dim3 grd((unsigned)((L->n + 255) / 256), (unsigned)n_layers);
  if (env_args) hipLaunchKernelGGL(line_prep_kernel<true>, grd, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(line_prep_kernel<false>, grd, dim3(256), 0, st, a);
  RTX_LAUNCH_CHECK();
  return 0;
}

extern "C" int rtx_line_prep(rtx_prep* P, const rtx_lines* L, const rtx_grid* grid, int n_layers, const double* T_h,
                             const double* p_atm_h, const double* qratio_h, const double* weight_h, const double* mass_h,
                             double dil_air, double dil_self, double omega_wing, double omega_wing_hw,
                             double intensity_threshold, double scale, void* stream) {
  return rtx_line_prep_profile(P, L, grid, n_layers, T_h, p_atm_h, qratio_h, weight_h, mass_h, dil_air, dil_self, omega_wing,
                               omega_wing_hw, intensity_threshold, scale, RTX_PROFILE_VOIGT, stream);
}
