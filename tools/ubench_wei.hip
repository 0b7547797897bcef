// Microbenchmark: cost of one fp32 Weideman-24 evaluation per lane (the band-row body of the line-sum kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#define INV_SQRT_PI 0.56418958354775628
#include "../radtxfr_amd/csrc/w24_coeffs.inc"
template <typename F>
__device__ __forceinline__ F weideman_re(F x, F y) {
  const F L = (F)W24_L;
  constexpr const F* coef = []() constexpr -> const F* { if constexpr (sizeof(F) == 4) return (const F*)W24F; else return (const F*)W24D; }();
  const F dr = L + y, nr = L - y;
  const F dd = fma(dr, dr, x * x);
  F inv;
  if constexpr (sizeof(F) == 4) { inv = __builtin_amdgcn_rcpf(dd); inv = fma(fma(-dd, inv, (F)1), inv, inv); } else { inv = (F)1 / dd; }
  const F Zr = fma(nr, dr, -(x * x)) * inv;
  const F Zi = (x * (nr + dr)) * inv;
  const F Wr = fma(Zr, Zr, -(Zi * Zi)), Wi = (F)2 * Zr * Zi;
  F or_ = coef[0], oi = (F)0, er = coef[1], ei = (F)0;
#pragma unroll
  for (int k = 2; k < 24; k += 2) {
    const F t0 = fma(or_, Wr, fma(-oi, Wi, coef[k]));
    const F t1 = fma(or_, Wi, oi * Wr);
    const F t2 = fma(er, Wr, fma(-ei, Wi, coef[k + 1]));
    const F t3 = fma(er, Wi, ei * Wr);
    or_ = t0; oi = t1; er = t2; ei = t3;
  }
  const F pr = fma(or_, Zr, fma(-oi, Zi, er));
  const F pi = fma(or_, Zi, fma(oi, Zr, ei));
  const F ir = dr * inv, ii = x * inv;
  const F i2r = fma(ir, ir, -(ii * ii)), i2i = (F)2 * ir * ii;
  return fma((F)2, fma(pr, i2r, -(pi * i2i)), (F)INV_SQRT_PI * ir);
}
template <typename F> __global__ __launch_bounds__(256) void k(F* out, int iters) {
  F x = (F)(threadIdx.x & 63) * (F)0.2 - (F)6.0, y = (F)1.5, acc = 0;
  for (int i = 0; i < iters; ++i) { acc += weideman_re<F>(x, y); x += (F)1e-4; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <typename F> void run(const char* nm, int wps) {
  int blocks = 256 * wps, iters = 2000; F* out; hipMalloc(&out, sizeof(F) * blocks * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<F><<<blocks, 256>>>(out, 10); hipDeviceSynchronize();
  hipEventRecord(e0); k<F><<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double evals = (double)iters * blocks * 256;
  // wave-evals per SIMD: blocks*4 waves / 1024 SIMDs
  double simd_cycles_per_wave_eval = (ms * 1e-3 * 1.9e9) / (iters * (blocks * 4.0 / 1024.0));
  printf("%s waves/SIMD=%d: %.3f ms, %.3e lane-evals/s, ~%.0f SIMD-cycles per wave-eval (@1.9GHz)\n", nm, wps, ms, evals / (ms * 1e-3), simd_cycles_per_wave_eval);
  hipFree(out);
}
int main() { for (int w : {1, 2, 4, 8}) { run<float>("wei f32", w); } for (int w : {1, 4}) run<double>("wei f64", w); return 0; }
