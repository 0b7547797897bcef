// What does a wave-instruction of each kind cost a SIMD that is full of waves? Each test is a loop whose body is
// U x (8 independent v_fma_f32 + KS s_add_u32 + KL ds_read_b32), i.e. one taken branch per U groups:
//   - U = 1, 2, 4, 8 with KS = KL = 0: the cost of a taken branch (s_cbranch + instruction refetch)
//   - U = 8 with KS = 0 .. 8: do scalar instructions of other waves issue beside the vector ones?
//   - U = 8 with KL = 1, 2: the same for LDS reads
// Reported: SIMD cycles (at 2.1 GHz) per group of 8 v_fma, per wave, min of 5 launches after a long warm-up.
// Build: hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 tools/ubench_issue.hip -o tools/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

template <int U, int KS, int KL> __global__ __launch_bounds__(256) void k(float* out, int iters, float seed, float m, float c) {
  __shared__ float lds[256 * 4];
  float a0 = seed + threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
  float a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  for (int t = threadIdx.x; t < 1024; t += 256) lds[t] = 0.f;
  __syncthreads();
  unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4;
  float l0 = 0.f, l1 = 0.f;
  const unsigned p = (unsigned)(size_t)(lds + threadIdx.x);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c);
      if (KS >= 1) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s0) : : "scc");
      if (KL >= 1) asm volatile("ds_read_b32 %0, %1" : "=v"(l0) : "v"(p));
      a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c);
      if (KS >= 2) asm volatile("s_add_u32 %0, %0, 5" : "+s"(s1) : : "scc");
      if (KS >= 5) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s0) : : "scc");
      if (KS >= 6) asm volatile("s_add_u32 %0, %0, 5" : "+s"(s1) : : "scc");
      a4 = fmaf(a4, m, c); a5 = fmaf(a5, m, c);
      if (KS >= 3) asm volatile("s_add_u32 %0, %0, 7" : "+s"(s2) : : "scc");
      if (KL >= 2) asm volatile("ds_read_b32 %0, %1 offset:1024" : "=v"(l1) : "v"(p));
      a6 = fmaf(a6, m, c); a7 = fmaf(a7, m, c);
      if (KS >= 4) asm volatile("s_add_u32 %0, %0, 9" : "+s"(s3) : : "scc");
      if (KS >= 7) asm volatile("s_add_u32 %0, %0, 7" : "+s"(s2) : : "scc");
      if (KS >= 8) asm volatile("s_add_u32 %0, %0, 9" : "+s"(s3) : : "scc");
      if (KL >= 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(s0 + s1 + s2 + s3) + l0 + l1;
}

template <int U, int KS, int KL> int run(int wavesPerSimd) {
  int blocks = 256 * wavesPerSimd, threads = 256, iters = 40000 / U;
  float* out; CK(hipMalloc(&out, sizeof(float) * blocks * threads));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<U, KS, KL><<<blocks, threads>>>(out, 10 * iters, 1.0f, 0.999f, 1e-3f);  // warm-up long enough for the clocks to settle
  CK(hipDeviceSynchronize());
  float ms = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    k<U, KS, KL><<<blocks, threads>>>(out, iters, 1.0f, 0.999f, 1e-3f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    ms = t < ms ? t : ms;
  }
  printf("loop body %d x (8 v_fma + %d s_add + %d ds_read)  waves/SIMD=%d  %8.3f ms  %6.2f cycles per group and wave\n", U, KS, KL, wavesPerSimd, ms,
         ms * 1e-3 * 2.1e9 / (iters * U) / wavesPerSimd);
  CK(hipFree(out));
  return 0;
}
int main() {
  for (int w : {4, 6, 8}) {
    run<1, 0, 0>(w); run<2, 0, 0>(w); run<4, 0, 0>(w); run<8, 0, 0>(w);
    run<8, 1, 0>(w); run<8, 2, 0>(w); run<8, 4, 0>(w); run<8, 8, 0>(w);
    run<8, 0, 1>(w); run<8, 0, 2>(w); run<8, 4, 2>(w);
  }
  return 0;
}
