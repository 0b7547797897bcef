// Speed-dependent Voigt line-sum (rtx_sdvoigt_sum): PROFILE_SDVOIGT, misc/hapi.py:10117-10129 -> pcqsdhc :9850-10024
// with anuVC = eta = 0, for which the common part (:10022) reduces to LS = Re(Aterm)/pi.
//
// The profile is a difference of two complex probability functions of nearby arguments (PART4), so it is evaluated
// in fp64 throughout, one thread per grid point looping over the lines whose window can reach its block (gather; the
// windows are the Voigt ones, written by the prologue). This is the cross-section generator's path, not the TUD hot
// path: no fp32 fast path, no node levels.
#include "rtx_common.h"

#include "rtx_voigt_math.h"

#ifndef RTX_SD_FARWING
#define RTX_SD_FARWING 1  // 0: every point through the two complex probability functions, as in round 2 (timing comparisons)
#endif
struct cd {
  double r, i;
};
__device__ __forceinline__ cd cmul(cd a, cd b) { return {a.r * b.r - a.i * b.i, a.r * b.i + a.i * b.r}; }
__device__ __forceinline__ cd cadd(cd a, cd b) { return {a.r + b.r, a.i + b.i}; }
__device__ __forceinline__ cd csub(cd a, cd b) { return {a.r - b.r, a.i - b.i}; }
__device__ __forceinline__ cd cscale(cd a, double s) { return {a.r * s, a.i * s}; }
__device__ __forceinline__ cd cinv(cd a) {
  const double d = 1.0 / (a.r * a.r + a.i * a.i);
  return {a.r * d, -a.i * d};
}
__device__ __forceinline__ cd cdiv(cd a, cd b) { return cmul(a, cinv(b)); }
// |a| without hypot's scaling (the arguments here are O(1e-6 .. 1e6): no overflow or underflow to guard against; the
// library hypot costs more than the rest of the far-wing evaluation, and the 1-ulp difference is far below the 1e-9 the
// parity tests hold this path to)
__device__ __forceinline__ double cabs(cd a) { return sqrt(a.r * a.r + a.i * a.i); }
// principal square root (numpy.sqrt on complex128)
__device__ __forceinline__ cd csqrt_(cd z) {
  const double m = cabs(z);
  if (m == 0.0) return {0.0, z.i};
  if (z.r >= 0.0) {
    const double t = sqrt(0.5 * (m + z.r));
    return {t, z.i / (2.0 * t)};
  }
  const double t = sqrt(0.5 * (m - z.r));
  return {fabs(z.i) / (2.0 * t), copysign(t, z.i)};
}

// hum1_wei, misc/hapi.py:9833-9844: w(x + iy), Weideman's 24-term expansion where |x| + y < 15, else the one-term
// asymptote (1/sqrt(pi)) t / (1/2 + t^2), t = y - ix.
__device__ cd hum1_wei_c(double x, double y) {
  if (fabs(x) + y < 15.0) {
    const double L = W24_L;
    const cd d = {L + y, -x};  // L - i z, z = x + iy
    const cd n = {L - y, x};   // L + i z
    const cd Z = cdiv(n, d);
    cd p = {W24D[0], 0.0};
#pragma unroll
    for (int k = 1; k < 24; ++k) {
      p = cmul(p, Z);
      p.r += W24D[k];
    }
    const cd id = cinv(d);
    const cd w = cadd(cscale(cmul(p, cmul(id, id)), 2.0), cscale(id, INV_SQRT_PI));
    return w;
  }
  const cd t = {y, -x};
  cd den = cmul(t, t);
  den.r += 0.5;
  return cscale(cdiv(t, den), INV_SQRT_PI);
}

// cpf3, misc/hapi.py:9645-9670: 15-term asymptotic series
__device__ cd cpf3_c(double x, double y) {
  const cd zm1 = cinv(cd{x, y});
  const cd zm2 = cmul(zm1, zm1);
  cd zsum = {1.0, 0.0}, zterm = {1.0, 0.0};
#pragma unroll
  for (int k = 0; k < 15; ++k) {
    zterm = cscale(cmul(zterm, zm2), 0.5 + (double)k);
    zsum = cadd(zsum, zterm);
  }
  const cd izm1 = {-zm1.i, zm1.r};  // i * zm1
  return cscale(cmul(zsum, izm1), 0.564189583547756);
}

// Re(Aterm)/pi of pcqsdhc for one point; sg = grid wavenumber
__device__ double sdvoigt_profile(const LineRecSD& q, double sg) {
  const double rpi = 1.7724538509055159;  // sqrt(pi)
  const double cte = q.cte;
  const cd c0t = {q.Gam0 - 1.5 * q.Gam2, q.Shift0};  // c0 - 1.5 c2, c2 = Gam2 (Shift2 = 0)
  const cd num = {c0t.r, (q.nu - sg) + c0t.i};        // i (sg0 - sg) + c0t
  if (q.Gam2 == 0.0) {  // PART1 (:9908-9915)
    const cd Z1 = cscale(num, cte);
    return rpi * cte * hum1_wei_c(-Z1.i, Z1.r).r * (1.0 / M_PI);
  }
  const double c2t = q.Gam2;
  const cd X = cscale(num, q.inv_Gam2);  // 1 / c2t, from the prologue (the same division, once per line instead of per point)
  const double csqrtY = q.csqrtY;        // 1 / (2 cte c2t) = (Gam2 - i 0) / (2 cte Gam2^2)
  const double Y = csqrtY * csqrtY;               // 1 / (2 cte c2t)^2
  // |X| against the two thresholds by its SQUARE (fp64 sqrt is a 20-instruction sequence; the branches are continuous
  // across their thresholds, so the last-bit difference of the comparison cannot be seen)
  const double aX2 = X.r * X.r + X.i * X.i;
  cd A;
  if (aX2 <= (3.0e-8 * Y) * (3.0e-8 * Y)) {  // PART2 (:9974-9989)
    const cd Z1 = cscale(num, cte);
    cd Z2 = csqrt_(cd{X.r + Y, X.i});
    Z2.r += csqrtY;
    A = cscale(csub(hum1_wei_c(-Z1.i, Z1.r), hum1_wei_c(-Z2.i, Z2.r)), rpi * cte);
  } else if (Y * Y <= 1.0e-30 * aX2) {  // PART3 (:9992-10017): Y <= 1e-15 |X|
    const cd sq = csqrt_(X);
    if (cabs(sq) <= 4.0e3) {
      const cd t = cmul(sq, hum1_wei_c(-sq.i, sq.r));
      A = cscale(cd{1.0 / rpi - t.r, -t.i}, 2.0 * rpi / c2t);
    } else {
      const cd ix = cinv(X);
      A = cscale(csub(ix, cscale(cmul(ix, ix), 1.5)), 1.0 / c2t);
    }
  } else {  // PART4 (:9933-9971)
    cd Z1 = csqrt_(cd{X.r + Y, X.i});
    Z1.r -= csqrtY;
    const cd Z2 = {Z1.r + 2.0 * csqrtY, Z1.i};
    const double x1 = -Z1.i, y1 = Z1.r, x2 = -Z2.i, y2 = Z2.r;
    if (RTX_SD_FARWING && fabs(x1) + y1 >= 15.0 && fabs(x2) + y2 >= 15.0) {
      // Far wing -- with the reference caller's WavenumberWingHW = 350 (misc/RT_gen_AbsXS_files.py:90) all but the innermost
      // ~0.02 cm^-1 of every window: both |Z| >= 15/sqrt(2) > 8, so the cpf3 switch (:9948-9958) is off and hum1_wei takes
      // its one-term asymptote w = f(t)/sqrt(pi), f(t) = t/(1/2 + t^2), t = y - ix, for BOTH arguments (:9834-9840). Their
      // difference is formed in closed form, f(t1) - f(t2) = (t1 - t2)(1/2 - t1 t2) / ((1/2 + t1^2)(1/2 + t2^2)), with
      // t1 - t2 = -2 csqrtY exactly: the same quantity as the reference's W1 - W2 (to the ~1e-14 its cancellation leaves),
      // at a tenth of the cost of two complex evaluations.
      const cd t1 = {y1, -x1}, t2 = {y2, -x2};
      const cd p12 = cmul(t1, t2);
      cd d1 = cmul(t1, t1), d2 = cmul(t2, t2);
      d1.r += 0.5;
      d2.r += 0.5;
      const cd nume = cscale(cd{0.5 - p12.r, -p12.i}, -2.0 * csqrtY);
      A = cscale(cdiv(nume, cmul(d1, d2)), cte);  // sqrt(pi) cte (W1 - W2), W = f/sqrt(pi)
      return A.r * (1.0 / M_PI);
    }
    const double S1 = sqrt(x1 * x1 + y1 * y1), S2 = sqrt(x2 * x2 + y2 * y2);
    const bool use3 = fabs(S1 - S2) <= 1.0 && fmax(S1, S2) > 8.0 && fmin(S1, S2) <= 8.0;
    const cd W1 = use3 ? cpf3_c(x1, y1) : hum1_wei_c(x1, y1);
    const cd W2 = use3 ? cpf3_c(x2, y2) : hum1_wei_c(x2, y2);
    A = cscale(csub(W1, W2), rpi * cte);
  }
  return A.r * (1.0 / M_PI);
}

struct SdArgs {
  const LineRec* rec;
  const LineRecSD* recsd;
  const int* ic;
  const int* maxhw;
  long long n_lines;
  GridDev g;
  float* out32;
  double* out64;
  long long ld;
  double scale;
};

__global__ __launch_bounds__(256) void sdvoigt_kernel(SdArgs a) {
  __shared__ int s_rng[2];
  const int k = blockIdx.y;
  const long long i0 = (long long)blockIdx.x * 256;
  if (threadIdx.x == 0) {
    // lines whose UNSHIFTED centre index lies within maxhw of this block (the windows are centred there)
    const long long hw = a.maxhw[k];
    const long long lo_v = i0 - hw, hi_v = i0 + 255 + hw;
    long long lo = 0, hi = a.n_lines;
    while (lo < hi) {
      const long long mid = (lo + hi) >> 1;
      if ((long long)a.ic[mid] < lo_v) lo = mid + 1; else hi = mid;
    }
    const long long first = lo;
    hi = a.n_lines;
    while (lo < hi) {
      const long long mid = (lo + hi) >> 1;
      if ((long long)a.ic[mid] <= hi_v) lo = mid + 1; else hi = mid;
    }
    s_rng[0] = (int)first;
    s_rng[1] = (int)lo;
  }
  __syncthreads();
  const long long i = i0 + threadIdx.x;
  if (i >= a.g.n) return;
  const double sg = grid_x(a.g, a.g.offset + i);
  const LineRec* __restrict__ rec = a.rec + (size_t)k * (size_t)a.n_lines;
  const LineRecSD* __restrict__ rsd = a.recsd + (size_t)k * (size_t)a.n_lines;
  double acc = 0.0;
  const int ii = (int)i;
  const int b_lo = (int)i0, b_hi = (int)(i0 + 256 < a.g.n ? i0 + 256 : a.g.n);
  for (int l = s_rng[0]; l < s_rng[1]; ++l) {
    const int lo = rec[l].lo, hi = rec[l].hi;  // empty windows have lo = hi = 0
    if (hi <= b_lo || lo >= b_hi) continue;    // the whole block lies outside this line's window (uniform branch)
    if (ii >= lo && ii < hi) {
      const LineRecSD q = rsd[l];
      acc += q.WS * sdvoigt_profile(q, sg);
    }
  }
  const size_t o = (size_t)k * (size_t)a.ld + (size_t)i;
  if (a.out64) a.out64[o] = acc;
  if (a.out32) a.out32[o] = (float)(acc * a.scale);
}

extern "C" int rtx_sdvoigt_sum(const rtx_prep* P, const rtx_grid* grid, int n_layers, float* out_f32, double* out_f64, int64_t ld,
                               void* stream) {
  if (!P) RTX_FAIL("prep is NULL");
  if (rtx_check_grid(grid)) return 1;
  if (!P->recsd) RTX_FAIL("rtx_line_prep_profile(..., RTX_PROFILE_SDVOIGT, ...) has not been run on this prep object");
  if (n_layers < 1 || n_layers > P->n_layers) RTX_FAIL("n_layers=%d, the prologue was run for %d", n_layers, P->n_layers);
  if (!out_f32 && !out_f64) RTX_FAIL("no output given");
  if (ld < grid->n) RTX_FAIL("leading dimension smaller than the shard");
  if (grid->n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  SdArgs a;
  a.rec = P->rec; a.recsd = P->recsd; a.ic = P->ic; a.maxhw = P->maxhw; a.n_lines = P->n_lines;
  a.g = to_dev(grid);
  a.out32 = out_f32; a.out64 = out_f64; a.ld = ld; a.scale = P->scale;
  hipLaunchKernelGGL(sdvoigt_kernel, dim3((unsigned)((grid->n + 255) / 256), (unsigned)n_layers), dim3(256), 0, st, a);
  RTX_LAUNCH_CHECK();
  return 0;
}
