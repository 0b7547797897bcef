// Is straight-line VALU code instruction-fetch bound? Same number of v_fma_f32 per wave, issued from
//   (a) a small loop body (64 FMAs, 512 B of code, stays in the instruction buffer / cache line set), and
//   (b) a long unrolled body (BODY FMAs of 8 B each, e.g. 4096 -> 32 KB of code walked linearly by every wave),
// with 8-byte VOP3 encodings (v_fma_f32 d, a, b, c with d != c) or 4-byte VOP2 (v_fmac_f32).
// hipcc -O3 --offload-arch=gfx950 tools/ubench_ifetch.hip -o tools/ubench_ifetch && tools/ubench_ifetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int BODY, bool VOP2>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = (float)threadIdx.x * 1e-3f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < BODY / 16; ++j) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (VOP2) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        else asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(x[(i + 1) & 15]), "v"(a), "v"(b));
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int BODY, bool VOP2>
static void run(const char* name, int waves_per_simd) {
  int n_cu = 256;
  const int blocks = n_cu * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD
  float* d;
  hipMalloc(&d, (size_t)blocks * 256 * sizeof(float));
  const long long total = 1 << 22;  // FMAs per wave
  const int iters = (int)(total / BODY);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<BODY, VOP2><<<blocks, 256>>>(d, iters, 1.0001f, 1e-6f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<BODY, VOP2><<<blocks, 256>>>(d, iters, 1.0001f, 1e-6f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)total * blocks * 4;  // wave-instructions
  printf("%-28s waves/SIMD=%d  %8.3f ms  %.2f cycles/instr/SIMD @2.4GHz\n", name, waves_per_simd, ms,
         ms * 1e-3 * 2.4e9 * (n_cu * 4) / instr);
  hipFree(d);
}

int main() {
  for (int w : {2, 4, 8}) {
    run<64, false>("VOP3 8B, body 64 (0.5 KB)", w);
    run<1024, false>("VOP3 8B, body 1024 (8 KB)", w);
    run<4096, false>("VOP3 8B, body 4096 (32 KB)", w);
    run<8192, false>("VOP3 8B, body 8192 (64 KB)", w);
    run<64, true>("VOP2 4B, body 64", w);
    run<4096, true>("VOP2 4B, body 4096 (16 KB)", w);
    run<8192, true>("VOP2 4B, body 8192 (32 KB)", w);
  }
  return 0;
}
