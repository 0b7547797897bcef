// Shared arithmetic of the two Voigt line-sum kernels (rtx_voigt.hip: gather, rtx_voigt_scatter.hip: scatter).
#pragma once
#include "rtx_common.h"
#include "w24_coeffs.inc"
#define INV_SQRT_PI 0.56418958354775628

// Re w(x+iy) by Weideman's rational expansion (misc/hapi.py:9812-9827), real arithmetic:
//   Z = (L + i z)/(L - i z),  w = 2 p(Z)/(L - i z)^2 + (1/sqrt(pi))/(L - i z),  p = 24-term polynomial.
// p is evaluated as pe(Z^2) + Z*po(Z^2): two independent 12-step chains instead of one 24-step chain (the serial chain
// stalled the band rows), each half by the real two-term recurrence below.
template <typename F>
__device__ __forceinline__ F weideman_re(F x, F y) {
  const F L = (F)W24_L;
  constexpr const F* coef = []() constexpr -> const F* { if constexpr (sizeof(F) == 4) return (const F*)W24F; else return (const F*)W24D; }();
  // d = L - i z = (L+y) - i x ;  n = L + i z = (L-y) + i x ;  Z = n/d = n*conj(d)/|d|^2
  const F dr = L + y, nr = L - y;
  const F dd = fma(dr, dr, x * x);
  F inv;
  if constexpr (sizeof(F) == 4) {
    inv = __builtin_amdgcn_rcpf(dd);
    inv = fma(fma(-dd, inv, (F)1), inv, inv);  // one Newton step: < 1 ulp
  } else {
    // dd = (L+y)^2 + x^2 lies in [17, 400]: no scaling to guard, so v_rcp_f64 + two Newton steps (<= 1 ulp) instead of the
    // IEEE division sequence (div_scale / div_fmas / div_fixup: ~20 instructions of the ~85 of a band row)
    inv = __builtin_amdgcn_rcp(dd);
    inv = fma(fma(-dd, inv, (F)1), inv, inv);
    inv = fma(fma(-dd, inv, (F)1), inv, inv);
  }
  const F Zr = fma(nr, dr, -(x * x)) * inv;   // Re[(nr + i x)(dr + i x)]
  const F Zi = (x * (nr + dr)) * inv;         // Im[...] = x*dr + nr*x
  const F Wr = fma(Zr, Zr, -(Zi * Zi)), Wi = (F)2 * Zr * Zi;  // W = Z^2
  // coef[] is in polyval order: p = sum_k coef[k] Z^(23-k); odd powers <-> even k
  F or_, oi, er, ei;  // po(W): coefficients of Z^23, Z^21, ... (k = 0, 2, ...); pe(W): Z^22, Z^20, ... (k = 1, 3, ...)
  {
    // The coefficients are REAL: q(W) = sum c_k W^k is the remainder of the division by W^2 - r W + s (r = 2 Re W,
    // s = |W|^2), i.e. the real recurrence b_k = c_k + r b_{k+1} - s b_{k+2} and q = c_0 - s b_2 + b_1 W: two FMAs per
    // coefficient instead of the four of a complex Horner step (47 operations for both halves instead of 88). |W| <= 1
    // (= 1 only on the real axis, y = 0), so rounding errors are not amplified: in fp32, over the band of the main pass
    // (y >= 1, |z| < 8 or y < 6), the result is within 1.8e-6 of the fp64 reference, the same as the complex Horner form
    // (1.9e-6); in fp64, over 1e-6 <= y < 1, within 1.5e-15 (absolute) of numpy.polyval.
    const F r = (F)2 * Wr, ns = -fma(Wr, Wr, Wi * Wi);
    F ob1 = coef[0], ob2 = (F)0, eb1 = coef[1], eb2 = (F)0;
#pragma unroll
    for (int k = 2; k < 22; k += 2) {
      const F ob0 = fma(r, ob1, fma(ns, ob2, coef[k]));
      const F eb0 = fma(r, eb1, fma(ns, eb2, coef[k + 1]));
      ob2 = ob1; ob1 = ob0; eb2 = eb1; eb1 = eb0;
    }
    or_ = fma(ob1, Wr, fma(ns, ob2, coef[22])); oi = ob1 * Wi;
    er = fma(eb1, Wr, fma(ns, eb2, coef[23])); ei = eb1 * Wi;
  }
  const F pr = fma(or_, Zr, fma(-oi, Zi, er));  // p = pe + Z*po
  const F pi = fma(or_, Zi, fma(oi, Zr, ei));
  // 1/d = conj(d)*inv = (dr + i x)*inv ; w = 2 p /d^2 + (1/sqrt(pi))/d
  const F ir = dr * inv, ii = x * inv;
  const F i2r = fma(ir, ir, -(ii * ii)), i2i = (F)2 * ir * ii;
  return fma((F)2, fma(pr, i2r, -(pi * i2i)), (F)INV_SQRT_PI * ir);
}

// Re w(x+iy) for |z| >= 6 by the asymptotic series  w ~ (i/(sqrt(pi) z)) sum_k (2k-1)!! / (2 z^2)^k, k = 0..5.
// Against the Weideman-24 value (what the reference computes inside |x|+y<15) the truncation error is < 6.5e-8
// for y >= 6 (checked in fp64 over the whole band); evaluated in fp32 it is at rounding level (3e-7).
__device__ __forceinline__ float asym6_re(float x, float y) {
  const float r2 = fmaf(x, x, y * y);
  float inv = __builtin_amdgcn_rcpf(r2);
  inv = fmaf(fmaf(-r2, inv, 1.0f), inv, inv);
  const float zr = x * inv, zi = -y * inv;                         // 1/z
  const float ur = fmaf(zr, zr, -(zi * zi)), ui = 2.0f * zr * zi;  // 1/z^2
  float pr = 945.0f / 32.0f, pi = 0.0f;
  constexpr float c[5] = {105.0f / 16.0f, 15.0f / 8.0f, 0.75f, 0.5f, 1.0f};
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const float tr = fmaf(pr, ur, fmaf(-pi, ui, c[k]));
    const float ti = fmaf(pr, ui, pi * ur);
    pr = tr;
    pi = ti;
  }
  return -(float)INV_SQRT_PI * fmaf(zr, pi, zi * pr);  // Re[(i/sqrt(pi)) (1/z) p] = -(1/sqrt(pi)) Im[(1/z) p]
}

// The same series with K terms, for the OUTER band rows of Doppler-dominated (y < 1) lines: where every lane of a row has
// |x| >= 5.5 the 12-term series in fp32 is within 4.3e-7 of the Weideman-24 value in the parity metric (error over
// max(value, 1e-3 of the line's peak); checked in NumPy float32 against the oracle's hum1_wei for 1e-5 <= y < 1,
// 5 <= |x| < 15 -- tests/test_host.py), because the imaginary parts it sums all have one sign: no cancellation, unlike the
// rational expansion, whose real part is the small difference that made those lines need fp64. ~65 fp32 operations
// against ~85 fp64 ones (4 cycles each) per row.
template <int K>
__device__ __forceinline__ float asymK_re(float x, float y) {
  const float r2 = fmaf(x, x, y * y);
  float inv = __builtin_amdgcn_rcpf(r2);
  inv = fmaf(fmaf(-r2, inv, 1.0f), inv, inv);
  const float zr = x * inv, zi = -y * inv;                         // 1/z
  const float ur = fmaf(zr, zr, -(zi * zi)), ui = 2.0f * zr * zi;  // 1/z^2
  float c[K];
  c[0] = 1.0f;
#pragma unroll
  for (int k = 1; k < K; ++k) c[k] = c[k - 1] * (float)(2 * k - 1) * 0.5f;  // (2k-1)!! / 2^k (compile-time constants)
  float pr = c[K - 1], pi = 0.0f;
#pragma unroll
  for (int k = K - 2; k >= 0; --k) {
    const float tr = fmaf(pr, ur, fmaf(-pi, ui, c[k]));
    const float ti = fmaf(pr, ui, pi * ur);
    pr = tr;
    pi = ti;
  }
  return -(float)INV_SQRT_PI * fmaf(zr, pi, zi * pr);
}

// One far-wing evaluation: Re[(1/sqrt(pi)) t/(1/2+t^2)], t = y - i x  (hum1_wei, :9834-9835) times the
// line strength, as  (xx*Ay + Ay0) / ((xx + b1)*xx + b0)  with per-line constants from the fp64 prologue;
// x = u*a + c with u = i - i0 an exact integer-valued float.  7 full-rate VALU ops + 1 v_rcp_f32.
#define RTX_FARWING(u_, q_, num_, rden_)                      \
  const float x_ = fmaf((u_), (q_).a, (q_).c);                \
  const float xx_ = x_ * x_;                                  \
  float num_ = fmaf(xx_, (q_).Ay, (q_).Ay0);                  \
  float rden_ = __builtin_amdgcn_rcpf(fmaf(xx_ + (q_).b1, xx_, (q_).b0))

// Function form of RTX_FARWING (the scatter kernel evaluates several rows in one scope).
__device__ __forceinline__ void farwing(float u, const LineRec& q, float& x, float& num, float& rden) {
  x = fmaf(u, q.a, q.c);
  const float xx = x * x;
  num = fmaf(xx, q.Ay, q.Ay0);
  rden = __builtin_amdgcn_rcpf(fmaf(xx + q.b1, xx, q.b0));
}
