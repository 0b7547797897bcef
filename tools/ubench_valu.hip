// Instruction-throughput microbenchmark for the VALU ops the line-sum / TUD kernels lean on.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int OP> __global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
  float a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  const float m = 0.999f, c = 1e-3f;
  if (OP == 0) {  // v_fma_f32 x8 independent
    for (int i = 0; i < iters; ++i) {
      a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c);
      a4 = fmaf(a4, m, c); a5 = fmaf(a5, m, c); a6 = fmaf(a6, m, c); a7 = fmaf(a7, m, c);
    }
  } else if (OP == 1) {  // v_pk_fma_f32 x4 (8 lanes-values)
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}; v2f mm = {m, m}, cc = {c, c};
    for (int i = 0; i < iters; ++i) {
      p0 = __builtin_elementwise_fma(p0, mm, cc); p1 = __builtin_elementwise_fma(p1, mm, cc);
      p2 = __builtin_elementwise_fma(p2, mm, cc); p3 = __builtin_elementwise_fma(p3, mm, cc);
    }
    a0 = p0.x + p0.y; a1 = p1.x + p1.y; a2 = p2.x + p2.y; a3 = p3.x + p3.y;
  } else if (OP == 2) {  // v_rcp_f32 x8
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_rcpf(a0); a1 = __builtin_amdgcn_rcpf(a1); a2 = __builtin_amdgcn_rcpf(a2); a3 = __builtin_amdgcn_rcpf(a3);
      a4 = __builtin_amdgcn_rcpf(a4); a5 = __builtin_amdgcn_rcpf(a5); a6 = __builtin_amdgcn_rcpf(a6); a7 = __builtin_amdgcn_rcpf(a7);
    }
  } else if (OP == 3) {  // v_exp_f32 x8
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_exp2f(a0 * 1e-3f); a1 = __builtin_amdgcn_exp2f(a1 * 1e-3f); a2 = __builtin_amdgcn_exp2f(a2 * 1e-3f); a3 = __builtin_amdgcn_exp2f(a3 * 1e-3f);
      a4 = __builtin_amdgcn_exp2f(a4 * 1e-3f); a5 = __builtin_amdgcn_exp2f(a5 * 1e-3f); a6 = __builtin_amdgcn_exp2f(a6 * 1e-3f); a7 = __builtin_amdgcn_exp2f(a7 * 1e-3f);
    }
  } else if (OP == 4) {  // v_fma_f64 x8
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7; const double dm = 0.999, dc = 1e-3;
    for (int i = 0; i < iters; ++i) {
      d0 = fma(d0, dm, dc); d1 = fma(d1, dm, dc); d2 = fma(d2, dm, dc); d3 = fma(d3, dm, dc);
      d4 = fma(d4, dm, dc); d5 = fma(d5, dm, dc); d6 = fma(d6, dm, dc); d7 = fma(d7, dm, dc);
    }
    a0 = d0 + d1 + d2 + d3; a4 = d4 + d5 + d6 + d7;
  } else if (OP == 5) {  // fp32 exp only (mul folded away): pure v_exp
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
      a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7);
      a0 -= 1.f; a1 -= 1.f; a2 -= 1.f; a3 -= 1.f; a4 -= 1.f; a5 -= 1.f; a6 -= 1.f; a7 -= 1.f;
    }
  } else if (OP == 6) {  // mix: 8 fma + 2 rcp per "pair of evals" (proxy of the asymptote body)
    for (int i = 0; i < iters; ++i) {
      a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c);
      a4 = fmaf(a4, m, c); a5 = fmaf(a5, m, c); a6 = fmaf(a6, m, c); a7 = fmaf(a7, m, c);
      a0 = __builtin_amdgcn_rcpf(a0); a4 = __builtin_amdgcn_rcpf(a4);
    }
  } else if (OP == 7) {  // v_pk_mul + v_pk_add
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}; v2f mm = {m, m}, cc = {c, c};
    for (int i = 0; i < iters; ++i) {
      p0 = p0 * mm; p1 = p1 + cc; p2 = p2 * mm; p3 = p3 + cc;
    }
    a0 = p0.x + p0.y; a1 = p1.x + p1.y; a2 = p2.x + p2.y; a3 = p3.x + p3.y;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int OP> int run(const char* name, double ops_per_iter_per_thread, int wavesPerSimd) {
  int blocks = 256 * wavesPerSimd, threads = 256, iters = 20000;
  float* out; CK(hipMalloc(&out, sizeof(float) * blocks * threads));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<OP><<<blocks, threads>>>(out, 100, 1.0f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k<OP><<<blocks, threads>>>(out, iters, 1.0f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double ops = ops_per_iter_per_thread * iters * (double)blocks * threads;
  printf("%-28s waves/SIMD=%d  %8.3f ms  %8.2f Tlane-op/s  (%.1f lane-ops/clk/CU @2.4GHz)\n", name, wavesPerSimd, ms,
         ops / ms * 1e-9, ops / (ms * 1e-3) / 256 / 2.4e9);
  CK(hipFree(out));
  return 0;
}
int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_fma_f32", 8, w); run<1>("v_pk_fma_f32 (2/inst)", 8, w); run<7>("v_pk_mul/add_f32 (2/inst)", 8, w);
    run<2>("v_rcp_f32", 8, w); run<5>("v_exp_f32+sub", 16, w);
    run<4>("v_fma_f64", 8, w); run<6>("8fma+2rcp mix", 10, w);
  }
  return 0;
}
