// Speed-dependent Voigt line-sum (rtx_sdvoigt_sum): PROFILE_SDVOIGT, misc/hapi.py:10117-10129 -> pcqsdhc :9850-10024
// with anuVC = eta = 0, for which the common part (:10022) reduces to LS = Re(Aterm)/pi.
//
// The profile is a difference of two complex probability functions of nearby arguments (PART4), so it is evaluated
// in fp64 throughout. Default kernel (sdvoigt_tile_kernel, below): a workgroup per tile of 1024 points, far wings at
// Chebyshev nodes (32 per tile, 12 per 64-point row), the rows around a centre and around every regime switch point by
// point. Cross-check (sdvoigt_kernel, RADTXFR_SD_KERNEL=gather): one thread per grid point looping over the lines whose
// window can reach its block. The windows are the Voigt ones, written by the prologue. This is the cross-section
// generator's path (misc/RT_gen_AbsXS_files.py:90), not the TUD hot path.
#include "rtx_common.h"

#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>

#include "rtx_voigt_math.h"

#ifndef SD_COUNT
#define SD_COUNT 0  /* debug builds: wave-evaluations per level, printed after every launch */
#endif
#if SD_COUNT
__device__ unsigned long long sd_dbg[4];
#define SD_TICK(i_) if (lane == 0) atomicAdd(&sd_dbg[i_], 1ull)
#else
#define SD_TICK(i_)
#endif
#ifndef SD_ABLATE
#define SD_ABLATE 0  /* timing builds: 1 no point-by-point rows, 2 no row level, 4 no tile level */
#endif
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
// Reciprocal and square root without the library's range handling: the arguments here are sums of squares of O(1e-6 .. 1e6)
// quantities -- never subnormal, never near overflow -- so v_rcp_f64 / v_rsq_f64 (2^-26) plus two Newton steps (<= 1 ulp, as in
// rtx_voigt_math.h: weideman_re) replace the IEEE division sequence (div_scale / div_fmas / div_fixup) and the scaled sqrt:
// ~8 instead of ~15 and ~30 instructions, four to six of them per profile evaluation in a kernel that is bound by fp64 issue.
__device__ __forceinline__ double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double fast_sqrt(double s) {  // s >= 0
  const double y = __builtin_amdgcn_rsq(s);
  double g = s * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  g = fma(fma(-g, g, s), h, g);
  return s > 0.0 ? g : 0.0;
}
__device__ __forceinline__ cd cinv(cd a) {
  const double d = fast_rcp(a.r * a.r + a.i * a.i);
  return {a.r * d, -a.i * d};
}
__device__ __forceinline__ cd cdiv(cd a, cd b) { return cmul(a, cinv(b)); }
// |a| without hypot's scaling (the arguments here are O(1e-6 .. 1e6): no overflow or underflow to guard against; the
// library hypot costs more than the rest of the far-wing evaluation, and the 1-ulp difference is far below the 1e-9 the
// parity tests hold this path to)
__device__ __forceinline__ double cabs(cd a) { return fast_sqrt(a.r * a.r + a.i * a.i); }
// principal square root (numpy.sqrt on complex128)
__device__ __forceinline__ cd csqrt_(cd z) {
  const double m = cabs(z);
  if (m == 0.0) return {0.0, z.i};
  if (z.r >= 0.0) {
    const double t = fast_sqrt(0.5 * (m + z.r));
    return {t, z.i * fast_rcp(2.0 * t)};
  }
  const double t = fast_sqrt(0.5 * (m - z.r));
  return {fabs(z.i) * fast_rcp(2.0 * t), copysign(t, z.i)};
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

// Re hum1_wei for the branches that only need real parts (PART1, PART2, PART4: Re(Aterm) = sqrt(pi) cte (Re W1 - Re W2)):
// the real two-term recurrence of the Voigt line-sum's fp64 band (rtx_voigt_math.h: weideman_re, half the operations of
// the complex Horner form and two independent chains; within 1.5e-15 absolute of numpy.polyval). y >= 0 only.
__device__ __forceinline__ double hum1_wei_asym_re(double x, double y) {
  const cd t = {y, -x};
  cd den = cmul(t, t);
  den.r += 0.5;
  return cdiv(t, den).r * INV_SQRT_PI;
}
// Re w(z1) - Re w(z2); both Weideman evaluations in one straight-line block when both arguments are inside |x| + y < 15,
// so that their recurrences interleave
__device__ __forceinline__ double hum1_wei_re_diff(double x1, double y1, double x2, double y2) {
  const bool in1 = fabs(x1) + y1 < 15.0, in2 = fabs(x2) + y2 < 15.0;
  if (in1 && in2) return weideman_re<double>(x1, y1) - weideman_re<double>(x2, y2);
  const double w1 = in1 ? weideman_re<double>(x1, y1) : hum1_wei_asym_re(x1, y1);
  const double w2 = in2 ? weideman_re<double>(x2, y2) : hum1_wei_asym_re(x2, y2);
  return w1 - w2;
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
__device__ __noinline__ double sdvoigt_profile(const LineRecSD q, const double sg) {
  const double rpi = 1.7724538509055159;  // sqrt(pi)
  const double cte = q.cte;
  const cd c0t = {q.Gam0 - 1.5 * q.Gam2, q.Shift0};  // c0 - 1.5 c2, c2 = Gam2 (Shift2 = 0)
  const cd num = {c0t.r, (q.nu - sg) + c0t.i};        // i (sg0 - sg) + c0t
  if (q.Gam2 == 0.0) {  // PART1 (:9908-9915)
    const cd Z1 = cscale(num, cte);
    if (Z1.r >= 0.0) {
      const double x = -Z1.i, y = Z1.r;
      return rpi * cte * (fabs(x) + y < 15.0 ? weideman_re<double>(x, y) : hum1_wei_asym_re(x, y)) * (1.0 / M_PI);
    }
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
    if (Z1.r >= 0.0 && Z2.r >= 0.0) return hum1_wei_re_diff(-Z1.i, Z1.r, -Z2.i, Z2.r) * (rpi * cte) * (1.0 / M_PI);
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
    if ((SD_ABLATE & 8) || (RTX_SD_FARWING && fabs(x1) + y1 >= 15.0 && fabs(x2) + y2 >= 15.0)) {
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
    // cpf3 switch (:9948-9958): |S1 - S2| <= 1, max(S1, S2) > 8, min(S1, S2) <= 8 with S = |Z|. The two bounds are tested on
    // the squares, and the two square roots are only taken in the thin shell where they hold
    const double q1 = x1 * x1 + y1 * y1, q2 = x2 * x2 + y2 * y2;
    bool use3 = fmax(q1, q2) > 64.0 && fmin(q1, q2) <= 64.0;
    if (use3) use3 = fabs(sqrt(q1) - sqrt(q2)) <= 1.0;
    if (!use3 && y1 >= 0.0 && y2 >= 0.0) return hum1_wei_re_diff(x1, y1, x2, y2) * (rpi * cte) * (1.0 / M_PI);
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

// ---- cross-check formulation (RADTXFR_SD_KERNEL=gather): every point of every window through sdvoigt_profile ------------
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

// ---- default formulation: far wings at Chebyshev nodes, in fp64 ----------------------------------------------------------
// With the reference caller's WavenumberWingHW = 350 a window is +-10 000 ... 15 000 grid points wide and a grid point sees
// ~1100 lines, all but a handful of them as a smooth far wing: there pcqsdhc is ONE analytic expression of the wavenumber
// (PART4 with both complex probability functions on hum1_wei's one-term asymptote -- the closed-form difference above --
// or PART1 / PART2 on the same asymptote), whose singularities all lie within ~GammaD of the line centre on the real axis
// (poles of 1/2 + t^2 at +-GammaD / (sqrt(2 ln 2)) + i (Gamma0 - Gamma2), the branch point of the square root above the
// centre). A workgroup owns a tile of 1024 points of one state and serves every line that reaches it at one of three levels:
//   tile level   window covers the tile, centre >= 256 points outside it, whole tile in a smooth regime: the profile at the
//                tile's 32 Chebyshev nodes (lane = (line slot, node): 8 lines x 32 nodes per pass), carried to the 1024 points
//                once per tile by a 1024 x 32 Lagrange matrix. Bernstein-ellipse bound with the nearest singularity 256 points
//                from the tile (rho = 2.4 inside it): <= 6e-11 of the line's own contribution.
//   row level    a row of 64 points wholly inside the window, >= 3 rows from the centre row and in a smooth regime: 12 nodes
//                per row (rho = 9: <= 1e-10), carried to the points by a 64 x 12 matrix.
//   point by point   everything else: the <= 5 rows around the centre, the rows that hold a regime switch (|x| + y = 15 of
//                either argument, the PART2 / PART3 thresholds) and the <= 2 rows cut by a window edge.
// Every contribution taken to a node level is positive (records with Gamma0 <= 1.5 Gamma2, whose profile can change sign, have
// no zones and stay point by point), so the sum keeps the relative bound; the parity tests hold this path to 1e-9 of the
// reference as before. Where the regimes switch is a closed form per line (sd_zones): with sqrt(X + Y) = p + iq the far
// condition |x1| + y1 >= 15 reads p + |q| >= K = 15 + csqrtY, and (p + |q|)^2 = |X + Y| + |Im X| makes it
// |Im X| >= (K^4 - R^2) / (2 K^2), R = Re X + Y. A margin of two grid points around every switch keeps the rounding of the
// pointwise tests (which still decide every point-by-point row) out of the classification.
// The sums are formed in candidate order with a fixed lane layout: results are bit-reproducible run to run.
#define SD_TILE 1024
#define SD_ROWS 16
#define SD_TN 32
#define SD_RN 12
#define SD_TDIST 256
#define SD_NEAR 2
#define SD_CHUNK 256

struct SdTab {
  const double* tmT;  // [SD_TN][SD_TILE]: Lagrange basis of tile node j at point p of the tile
  const double* rmT;  // [SD_RN][64]
  double toff[SD_TN]; // node positions, in points from the tile's / row's first point
  double roff[SD_RN];
};

// |nu0' - sg| ranges in which the line is one smooth expression: zone a = PART2 on the asymptote, zone b = PART4 far wing
// (or PART1's). Empty zones have lo = +inf.
struct SdZone {
  double a_lo, a_hi, b_lo, b_hi;
};
__device__ SdZone sd_zones(const LineRecSD& q, double step) {
  const double INF = __longlong_as_double(0x7ff0000000000000LL);
  SdZone z = {INF, -INF, INF, -INF};
  const double m = 2.0 * step;
  if (q.Gam2 == 0.0) {  // PART1: hum1_wei(x, y) with y = Gam0 cte, asymptote from |x| + y >= 15
    if (!(q.cte > 0.0)) return z;
    z.b_lo = fmax(0.0, (15.0 - q.Gam0 * q.cte) / q.cte) + m;
    z.b_hi = INF;
    return z;
  }
  const double a = q.Gam0 - 1.5 * q.Gam2, c2 = q.Gam2, sY = q.csqrtY;
  if (!(c2 > 0.0) || !(a > 0.0) || !(sY > 0.0) || !(q.cte > 0.0)) return z;  // unphysical records stay point by point
  const double Y = sY * sY;
  const double K = 15.0 + sY, K2 = K * K;
  const double R = a * q.inv_Gam2 + Y;
  const double d = 225.0 + 30.0 * sY - a * q.inv_Gam2;  // K^2 - R without the cancellation of the two Y
  const double far4 = d > 0.0 ? c2 * ((d * (K2 + R)) / (2.0 * K2)) : 0.0;
  const double t2 = 3.0e-8 * Y * c2;  // PART2: |X| <= 3e-8 Y  <=>  sqrt(a^2 + delta^2) <= t2
  const double dP2 = t2 > a ? sqrt(t2 * t2 - a * a) : -1.0;
  const double t3 = 1.0e15 * Y * c2;  // PART3: |X| >= 1e15 Y
  if (!(t3 > a)) return z;
  const double dP3 = sqrt(t3 * t3 - a * a);
  z.b_lo = fmax(far4, dP2) + m;
  z.b_hi = dP3 - m;
  if (dP2 > 0.0 && sY >= 8.0) {  // inside PART2: Z1 = num cte on the asymptote; Z2 has y2 ~ 2 csqrtY >= 16
    z.a_lo = fmax(0.0, (15.0 - a * q.cte) / q.cte) + m;
    z.a_hi = dP2 - m;
  }
  return z;
}
__device__ __forceinline__ bool sd_in_zone(const SdZone& z, double dmin, double dmax) {
  return (dmin >= z.a_lo && dmax <= z.a_hi) || (dmin >= z.b_lo && dmax <= z.b_hi);
}

__global__ __launch_bounds__(256) void sdvoigt_tile_kernel(SdArgs a, SdTab tab) {
  __shared__ int s_rng[2];
  // the chunk's two lists share one array of records: tile-level lines from the front, the others from the back (the
  // classifying lane has the record in registers; the evaluation loops then read it from LDS instead of from memory)
  __shared__ LineRecSD s_q[SD_CHUNK];
  __shared__ int2 s_win[SD_CHUNK];
  __shared__ int s_rm[SD_CHUNK];
  __shared__ int s_cnt[2][4];
  __shared__ double s_tn[8][SD_TN];
  __shared__ double s_rn[SD_ROWS][SD_RN];
  const int k = blockIdx.y;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const long long n = a.g.n;
  const int ia = blockIdx.x * SD_TILE;
  const int ib = (int)((long long)ia + SD_TILE < n ? (long long)ia + SD_TILE : n);
  if (tid == 0) {
    // lines whose UNSHIFTED centre index lies within maxhw of this tile (the windows are centred there)
    const long long hw = a.maxhw[k];
    const long long lo_v = (long long)ia - hw, hi_v = (long long)ib - 1 + hw;
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
  const int first = s_rng[0], last = s_rng[1];
  const LineRec* __restrict__ rec = a.rec + (size_t)k * (size_t)a.n_lines;
  const LineRecSD* __restrict__ rsd = a.recsd + (size_t)k * (size_t)a.n_lines;
  const long long g0 = a.g.offset + (long long)ia;  // global index of the tile's first point
  // tile level: thread = (line slot ts, node tj); row level: thread = (row rr, node rj), 192 of the 256 threads
  const int ts = tid >> 5, tj = tid & 31;
  const int rr = tid / SD_RN, rj = tid - rr * SD_RN;
  // node / point wavenumbers are re-formed where they are used (two operations) rather than held in registers
  const double pos_t = (double)g0 + tab.toff[tj];
  const double pos_r = tid < SD_ROWS * SD_RN ? (double)(g0 + 64 * rr) + tab.roff[rj] : 0.0;
  double acc_t = 0.0, acc_r = 0.0, acc_p[4] = {0.0, 0.0, 0.0, 0.0};
  bool any_tile = false;  // uniform

  for (int base = first; base < last; base += SD_CHUNK) {
    // ---- classification, lane = candidate ---------------------------------------------------------------------------
    const int l = base + tid;
    int cls = 0, mask = 0;
    LineRecSD q;
    int lo = 0, hi = 0;
    if (l < last) {
      const int4 w = *reinterpret_cast<const int4*>(reinterpret_cast<const char*>(rec + l) + 32);  // i0 lo hi zw
      const int i0 = w.x;
      lo = w.y;
      hi = w.z;
      if (hi > ia && lo < ib) {
        q = rsd[l];
        const SdZone z = sd_zones(q, a.g.step);
        auto delta = [&](long long p) -> double { return fabs((q.nu - grid_x(a.g, a.g.offset + p)) + q.Shift0); };
        const bool left = (long long)i0 <= (long long)ia - SD_TDIST, right = (long long)i0 >= (long long)ia + SD_TILE - 1 + SD_TDIST;
        if (lo <= ia && hi >= ib && (left || right)) {
          const double d0 = delta(ia), d1 = delta((long long)ia + SD_TILE - 1);
          if (sd_in_zone(z, fmin(d0, d1), fmax(d0, d1))) cls = 1;
        }
        if (cls == 0) {
          cls = 2;
          const int rc = (i0 - ia) >> 6;  // centre row (floor; may lie outside the tile)
          for (int r = 0; r < SD_ROWS; ++r) {
            const int pa = ia + 64 * r;
            if (pa >= ib) break;
            const int pb = pa + 63 < ib - 1 ? pa + 63 : ib - 1;
            if (!(hi > pa && lo <= pb)) continue;  // not reached
            bool nodal = lo <= pa && hi > pb && (r < rc - SD_NEAR || r > rc + SD_NEAR);
            if (nodal) {
              const double d0 = delta(pa), d1 = delta((long long)pa + 63);
              nodal = sd_in_zone(z, fmin(d0, d1), fmax(d0, d1));
            }
            mask |= nodal ? (1 << r) : (1 << (16 + r));
          }
        }
      }
    }
    // ordered compaction of the two lists (candidate order)
    const unsigned long long bt = __ballot(cls == 1), br = __ballot(cls == 2);
    if (lane == 0) {
      s_cnt[0][wave] = __popcll(bt);
      s_cnt[1][wave] = __popcll(br);
    }
    __syncthreads();  // (also: the previous chunk's lists are no longer read)
    int ot = 0, orr = 0, nT = 0, nR = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int ct = s_cnt[0][w], cr = s_cnt[1][w];
      if (w < wave) { ot += ct; orr += cr; }
      nT += ct; nR += cr;
    }
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    if (cls == 1) s_q[ot + __popcll(bt & below)] = q;
    if (cls == 2) {
      const int o = SD_CHUNK - 1 - (orr + __popcll(br & below));
      s_q[o] = q;
      s_win[o] = make_int2(lo, hi);
      s_rm[o] = mask;
    }
    __syncthreads();
    // ---- tile level: 8 lines x 32 nodes per pass ----------------------------------------------------------------------
    any_tile = any_tile || nT > 0;
    for (int g = 0; g < nT; g += 8) {
      const int e = g + ts;
      if (e < nT && !(SD_ABLATE & 4)) {
        const LineRecSD qe = s_q[e];
        acc_t += qe.WS * sdvoigt_profile(qe, __dadd_rn(__dmul_rn(pos_t, a.g.step), a.g.xmin));
        SD_TICK(0);
      }
    }
    // ---- the other lines: far rows at the row nodes, the rest point by point ------------------------------------------
    // a wave's rows: row level rows tid / 12 of its 64 threads, point by point rows wave, wave + 4, wave + 8, wave + 12
    const int far_rows = wave < 3 ? (((1 << ((64 * wave + 63) / SD_RN + 1)) - 1) & ~((1 << ((64 * wave) / SD_RN)) - 1)) & 0xffff : 0;
    const int my_rows = far_rows | (0x1111 << (16 + wave));
    for (int e = 0; e < nR; ++e) {
      const int o = SD_CHUNK - 1 - e;
      const int me = __builtin_amdgcn_readfirstlane(s_rm[o]);
      if (!(me & my_rows)) continue;  // nothing of this line in this wave's rows
      const LineRecSD qe = s_q[o];
      if (!(SD_ABLATE & 2) && (me & far_rows) && tid < SD_ROWS * SD_RN && ((me >> rr) & 1)) { acc_r += qe.WS * sdvoigt_profile(qe, __dadd_rn(__dmul_rn(pos_r, a.g.step), a.g.xmin)); SD_TICK(1); }
      if (!(SD_ABLATE & 1) && ((me >> 16) & (0x1111 << wave))) {
        const int2 win = s_win[o];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int r = wave + 4 * m;  // the row of point tid + 256 m
          if ((me >> (16 + r)) & 1) {
            const int i = ia + tid + 256 * m;
            if (i >= win.x && i < win.y) acc_p[m] += qe.WS * sdvoigt_profile(qe, grid_x(a.g, g0 + tid + 256 * m));
            SD_TICK(2);
          }
        }
      }
    }
  }
  // ---- nodal sums -> grid points ------------------------------------------------------------------------------------------
  __syncthreads();
  s_tn[ts][tj] = acc_t;
  if (tid < SD_ROWS * SD_RN) s_rn[rr][rj] = acc_r;
  __syncthreads();
  if (tid < SD_TN) {
    double v = s_tn[0][tid];
#pragma unroll
    for (int s = 1; s < 8; ++s) v += s_tn[s][tid];
    s_tn[0][tid] = v;
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int p = tid + 256 * m;
    const long long i = (long long)ia + p;
    if (i >= n) break;
    double v = acc_p[m];
    const int r = p >> 6, c = p & 63;
    double f = 0.0;
#pragma unroll
    for (int j = 0; j < SD_RN; ++j) f = fma(tab.rmT[j * 64 + c], s_rn[r][j], f);
    v += f;
    if (any_tile) {
      double t = 0.0;
#pragma unroll 8
      for (int j = 0; j < SD_TN; ++j) t = fma(tab.tmT[j * SD_TILE + p], s_tn[0][j], t);
      v += t;
    }
    const size_t o = (size_t)k * (size_t)a.ld + (size_t)i;
    if (a.out64) a.out64[o] = v;
    if (a.out32) a.out32[o] = (float)(v * a.scale);
  }
}

// Chebyshev nodes (first kind) on [0, L - 1] and the Lagrange basis of every node at the L integer points, formed once per
// device in long double on the host.
static int sd_tables(SdTab* out) {
  static std::mutex mu;
  static SdTab cache[64];
  static bool have[64] = {false};
  int dev = 0;
  RTX_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) RTX_FAIL("device %d", dev);
  std::lock_guard<std::mutex> lock(mu);
  if (!have[dev]) {
    SdTab t;
    auto build = [](int N, int L, double* off, std::vector<double>& mT) {
      std::vector<long double> x(N);
      const long double h = (long double)(L - 1) / 2.0L, pi = 3.14159265358979323846264338327950288L;
      for (int j = 0; j < N; ++j) x[j] = h + h * cosl((2 * j + 1) * pi / (2.0L * N));
      mT.assign((size_t)N * L, 0.0);
      for (int j = 0; j < N; ++j) {
        off[j] = (double)x[j];
        for (int p = 0; p < L; ++p) {
          long double v = 1.0L;
          for (int m = 0; m < N; ++m)
            if (m != j) v *= ((long double)p - x[m]) / (x[j] - x[m]);
          mT[(size_t)j * L + p] = (double)v;
        }
      }
    };
    std::vector<double> tm, rm;
    build(SD_TN, SD_TILE, t.toff, tm);
    build(SD_RN, 64, t.roff, rm);
    double *d_tm = nullptr, *d_rm = nullptr;
    RTX_HIP(hipMalloc((void**)&d_tm, tm.size() * sizeof(double)));
    RTX_HIP(hipMalloc((void**)&d_rm, rm.size() * sizeof(double)));
    RTX_HIP(hipMemcpy(d_tm, tm.data(), tm.size() * sizeof(double), hipMemcpyHostToDevice));
    RTX_HIP(hipMemcpy(d_rm, rm.data(), rm.size() * sizeof(double), hipMemcpyHostToDevice));
    t.tmT = d_tm;
    t.rmT = d_rm;
    cache[dev] = t;
    have[dev] = true;
  }
  *out = cache[dev];
  return 0;
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
  if (grid->n > 2000000000LL) RTX_FAIL("grid shard of %lld points", (long long)grid->n);
  hipStream_t st = (hipStream_t)stream;
  SdArgs a;
  a.rec = P->rec; a.recsd = P->recsd; a.ic = P->ic; a.maxhw = P->maxhw; a.n_lines = P->n_lines;
  a.g = to_dev(grid);
  a.out32 = out_f32; a.out64 = out_f64; a.ld = ld; a.scale = P->scale;
  const char* e_k = getenv("RADTXFR_SD_KERNEL");  // "gather": the point-by-point cross-check (read per call: tests switch it)
  const bool gather = e_k && !strcmp(e_k, "gather");
  if (gather) {
    hipLaunchKernelGGL(sdvoigt_kernel, dim3((unsigned)((grid->n + 255) / 256), (unsigned)n_layers), dim3(256), 0, st, a);
  } else {
    SdTab tab;
    if (sd_tables(&tab)) return 1;
    hipLaunchKernelGGL(sdvoigt_tile_kernel, dim3((unsigned)((grid->n + SD_TILE - 1) / SD_TILE), (unsigned)n_layers), dim3(256), 0, st, a, tab);
  }
  RTX_LAUNCH_CHECK();
#if SD_COUNT
  {
    unsigned long long h[4];
    RTX_HIP(hipStreamSynchronize(st));
    RTX_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(sd_dbg), sizeof(h)));
    fprintf(stderr, "[sd count] cumulative wave-evaluations: tile %llu row %llu point-by-point %llu\n", h[0], h[1], h[2]);
  }
#endif
  return 0;
}
