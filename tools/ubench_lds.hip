// Microbenchmark: LDS float accumulate throughput (ds_add_f32 vs read+add+write), 4 or 8 waves per workgroup, each wave on its own region.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, int iters) {
  extern __shared__ float s[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* mine = s + wave * 2048;
  for (int i = lane; i < 2048; i += 64) mine[i] = 0.f;
  __syncthreads();
  float v = 1.0f + lane * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int idx = ((it * 8 + r) & 31) * 64 + lane;
      if (MODE == 0) { __hip_atomic_fetch_add(&mine[idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
      else if (MODE == 1) { mine[idx] += v; }
      else { asm volatile("" :: "v"(v)); }
      v = fmaf(v, 0.999f, 1e-3f);
    }
  }
  __syncthreads();
  float a = 0; for (int i = lane; i < 2048; i += 64) a += mine[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + v;
}
template <int MODE> void run(const char* nm, int blocksPerCU) {
  int blocks = 256 * blocksPerCU, iters = 4000; float* out; hipMalloc(&out, 4 * blocks * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256, 4 * 2048 * 4>>>(out, 10); hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<blocks, 256, 4 * 2048 * 4>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr_per_cu = (double)iters * 8 * 4 * blocksPerCU;  // wave-instructions per CU
  printf("%-22s blocks/CU=%d: %.3f ms -> %.2f cycles per wave-level LDS accumulate per CU (@1.9 GHz)\n", nm, blocksPerCU, ms, ms * 1e-3 * 1.9e9 / instr_per_cu);
  hipFree(out);
}
int main() { for (int b : {1, 2, 4}) { run<0>("ds_add_f32 (atomic)", b); run<1>("read+add+write", b); run<2>("no LDS (fma only)", b); } return 0; }
