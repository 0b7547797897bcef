// Issue rate of fp32 vector instructions by ENCODING and operand kind, at 1 / 2 / 4 / 8 waves per SIMD:
//   0  v_fmac_f32  v, v, v        VOP2, 4 bytes
//   1  v_fma_f32   v, v, v, v     VOP3, 8 bytes
//   2  v_fma_f32   v, s, v, v     VOP3 with a scalar-register operand
//   3  v_fmamk_f32 v, v, K, v     VOP2 + 32-bit literal, 8 bytes
//   4  v_pk_fma_f32               VOP3P, 8 bytes, two results per lane
//   5  v_mul_f32   v, v, v        VOP2, 4 bytes
//   6  v_fmac_f32 v, v, v again (the "s" constraint on a float is given a VGPR)
//   7  what the compiler makes of fmaf(a, m, c) with m a kernel argument: v_fma_f32 v, s, v, v
//   8-12  forms with a true SGPR operand / an inline constant
// Loop body: 64 instructions on 8 independent accumulators (one taken branch per 64), inline asm so that the
// compiler cannot re-encode or pack them. Reported: cycles (s_memtime/clock64 ticks of wave 0... no: wall time x
// an assumed 2.1 GHz) per instruction and SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_enc.hip -o tools/ubench_enc
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X(a0) X(a1) X(a2) X(a3) X(a4) X(a5) X(a6) X(a7)
#define I0(a) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(m), "v"(c));
#define I1(a) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(c));
#define I2(a) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a) : "s"(ms), "v"(c));
#define I3(a) asm volatile("v_fmamk_f32 %0, %0, 0x3f7fbe77, %1" : "+v"(a) : "v"(c));
#define I5(a) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a) : "v"(m));
#define I6(a) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "s"(ms), "v"(c));
#define I8(a) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a) : "s"(msb), "v"(c));
#define I9(a) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "s"(msb), "v"(c));
#define I10(a) asm volatile("v_fma_f32 %0, 1.0, %0, %1" : "+v"(a) : "v"(c));
#define I11(a) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a) : "s"(msb));
#define I12(a) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a) : "v"(m), "s"(msb));
#define I13(a) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(c));
#define I14(a) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a), "v"(c) : "vcc");
#define I15(a) asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "s"(msb));
#define I16(a) asm volatile("v_rcp_f32 %0, %0" : "+v"(a));
#define I17(a) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
#define I18(a) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "s"(mask));
#define I19(a) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(mask) : "v"(a), "v"(c));
#define P20(p) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "s"(spair), "v"(cc));
#define I21(a) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a) : "s"(msb));
#define I22(a) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(dm), "v"(dc));
#define P4(p) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(mm), "v"(cc));

template <int OP, int LDS = 0> __global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters, float seed, float m, float c, float ms_in) {
  __shared__ float lds[LDS ? 1024 : 1];
  if (LDS) {
    for (int t = threadIdx.x; t < 1024; t += 256) lds[t] = 0.f;
    if (LDS > 1) __syncthreads();
  }
  float a0 = seed + threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
  float a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  const float ms = __builtin_amdgcn_readfirstlane(ms_in);
  const int msb = __builtin_amdgcn_readfirstlane(__float_as_int(ms_in));
  v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
  const v2f mm = {m, m}, cc = {c, c};
  unsigned long long mask = __builtin_amdgcn_read_exec();
  const unsigned long long spair = ((unsigned long long)(unsigned)msb << 32) | (unsigned)msb;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7; const double dm = m, dc = c;
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (OP == 0) { REP8(I0) }
      if (OP == 1) { REP8(I1) }
      if (OP == 2) { REP8(I2) }
      if (OP == 3) { REP8(I3) }
      if (OP == 5) { REP8(I5) }
      if (OP == 6) { REP8(I6) }
      if (OP == 8) { REP8(I8) }
      if (OP == 9) { REP8(I9) }
      if (OP == 10) { REP8(I10) }
      if (OP == 11) { REP8(I11) }
      if (OP == 12) { REP8(I12) }
      if (OP == 13) { REP8(I13) }
      if (OP == 14) { REP8(I14) }
      if (OP == 15) { REP8(I15) }
      if (OP == 16) { REP8(I16) }
      if (OP == 17) { REP8(I17) }
      if (OP == 18) { REP8(I18) }
      if (OP == 19) { REP8(I19) }
      if (OP == 21) { REP8(I21) }
      if (OP == 20) { P20(p0) P20(p1) P20(p2) P20(p3) P20(p4) P20(p5) P20(p6) P20(p7) }
      if (OP == 22) { I22(d0) I22(d1) I22(d2) I22(d3) I22(d4) I22(d5) I22(d6) I22(d7) }
      if (OP == 7) { a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c);
                     a4 = fmaf(a4, m, c); a5 = fmaf(a5, m, c); a6 = fmaf(a6, m, c); a7 = fmaf(a7, m, c); }
      if (OP == 4) { P4(p0) P4(p1) P4(p2) P4(p3) P4(p4) P4(p5) P4(p6) P4(p7) }
    }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y +
                                                p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y + (float)(mask & 1) + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + (LDS ? lds[threadIdx.x] : 0.f);
}

template <int OP, int LDS = 0> int run(const char* name, int wavesPerSimd) {
  int blocks = 256 * wavesPerSimd, threads = 256, iters = 5000;
  float* out; CK(hipMalloc(&out, sizeof(float) * blocks * threads));
  long long* cyc; CK(hipMalloc(&cyc, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<OP, LDS><<<blocks, threads>>>(out, cyc, 10 * iters, 1.0f, 0.999f, 1e-3f, 0.999f);
  CK(hipDeviceSynchronize());
  float ms = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    k<OP, LDS><<<blocks, threads>>>(out, cyc, iters, 1.0f, 0.999f, 1e-3f, 0.999f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    ms = t < ms ? t : ms;
  }
  long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
  const double n_inst = 64.0 * iters;
  printf("%-34s waves/SIMD=%d  %8.3f ms  wall: %5.2f cycles@2.1GHz per instruction and SIMD;  wave 0's own clock: %5.2f ticks per instruction and SIMD\n", name,
         wavesPerSimd, ms, ms * 1e-3 * 2.1e9 / n_inst / wavesPerSimd, (double)h / n_inst / wavesPerSimd);
  CK(hipFree(out)); CK(hipFree(cyc));
  return 0;
}
int main() {
  for (int w : {1, 4, 8}) {
    run<0>("v_fmac_f32 v,v,v (VOP2, 4 B)", w);
    run<6>("v_fmac_f32 v,s,v (VOP2, 4 B)", w);
    run<5>("v_mul_f32 v,v,v (VOP2, 4 B)", w);
    run<1>("v_fma_f32 v,v,v,v (VOP3, 8 B)", w);
    run<2>("v_fma_f32 v,s,v,v (VOP3, 8 B)", w);
    run<3>("v_fmamk_f32 (VOP2 + literal, 8 B)", w);
    run<4>("v_pk_fma_f32 (VOP3P, 8 B, 2/lane)", w);
    run<7>("compiler: fmaf(a, m, c)", w);
    run<8>("asm v_fma_f32 v, SGPR, v, v", w);
    run<12>("asm v_fma_f32 v, v, v, SGPR", w);
    run<9>("asm v_fmac_f32 v, SGPR, v", w);
    run<11>("asm v_add_f32 v, SGPR, v", w);
    run<10>("asm v_fma_f32 v, 1.0, v, v", w);
    run<15>("v_mov_b32 v, SGPR", w);
    run<21>("v_add_u32 v, SGPR, v", w);
    run<13>("v_cndmask_b32 v, v, v, vcc", w);
    run<18>("v_cndmask_b32 v, v, v, s[a:b]", w);
    run<14>("v_cmp_lt_f32 vcc, v, v", w);
    run<19>("v_cmp_lt_f32 s[a:b], v, v", w);
    run<20>("v_pk_fma_f32 v, v, s[a:b], v", w);
    run<16>("v_rcp_f32", w);
    run<17>("v_exp_f32", w);
    run<22>("v_fma_f64 v, v, v, v", w);
  }
  return 0;
}
