// would sharing the far-wing arithmetic ACROSS LAYERS beat the per-(tile, layer) row-level evaluation of
// the line-sum? Two inner loops, same work unit -- one far line x 8 Chebyshev nodes x 8 rows x 32 layers -- timed at the
// line-sum's occupancy (6 waves per SIMD, every CU busy):
//   A  what voigt_nodal_kernel does today, per layer: lane = (line of a group of 8, node); per row
//      x = fma(r, dx, x0); xx = x*x; num = fma(xx, Ay, Ay0); rden = rcp(fma(xx + b1, xx, b0)); nod[r] = fma(num, rden, nod[r])
//      (7 full-rate VALU + 1 v_rcp_f32), the line's six constants in vector registers (ds_bpermute'd once per group and layer).
//   B  the cross-layer form: 1/(nu - nu0) once per (line, node, row), then per LAYER a polynomial in it whose M coefficients
//      belong to the (line, layer) pair. lane = (row, node), the line is wave-uniform, so the coefficients arrive through
//      scalar loads and enter the FMAs as scalar operands (the cheapest way to get per-(line, layer) data to 64 lanes; from
//      LDS the same loop is bound by the LDS pipe: M/4 ds_read_b128 per 2M FMAs). M = 6: what the reference's profile needs,
//      because every layer has its own pressure shift delta*p_k -- 1/(D - s_k)^2 = D^-2 (1 + 2 s_k/D + 3 s_k^2/D^2 + ...)
//      on top of the (gamma0/D)^2 series of the rational; M = 3: a table with no pressure shifts at all.
// Result (MI355X, profiles/r3_ubench_crosslayer.txt): see DESIGN.md 4.2.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_crosslayer.hip -o tools/ubench_crosslayer
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
constexpr int NL = 32, ROWS = 16, LINES = 96;  // layers, rows per tile, far lines per tile (the surface layer's census: 86)

// A: one workgroup (2 waves) per (tile, layer), as the line-sum is launched; each wave takes every other group of 8 lines
__global__ __launch_bounds__(128, 6) void form_a(const float* __restrict__ par, float* __restrict__ out, int n_tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane >> 3, j = lane & 7;
  const int tile = blockIdx.x, k = blockIdx.y;
  float nod[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) nod[r] = 0.f;
  for (int g = wave; g < LINES / 8; g += 2) {
    const float* p = par + ((size_t)(k * LINES + g * 8 + l)) * 8;  // per-(line, layer) record, one per lane-group member
    const float4 f0 = *reinterpret_cast<const float4*>(p), f1 = *reinterpret_cast<const float4*>(p + 4);
    const float x0 = f0.x + 0.01f * j + 1e-3f * tile, dx = f0.y, b1 = f0.z, b0 = f0.w, Ay = f1.x, Ay0 = f1.y;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const float x = fmaf((float)r, dx, x0);
      const float xx = x * x;
      const float num = fmaf(xx, Ay, Ay0);
      const float rden = __builtin_amdgcn_rcpf(fmaf(xx + b1, xx, b0));
      nod[r] = fmaf(num, rden, nod[r]);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < ROWS; ++r) s += nod[r];
  out[((size_t)k * n_tiles + tile) * 128 + threadIdx.x] = s;
}

// B: one workgroup (2 waves) per tile for ALL layers; wave w owns rows 8w .. 8w+7 (lane = (row, node)); lines are wave-uniform
template <int M>
__global__ __launch_bounds__(128, 6) void form_b(const float* __restrict__ geo, const float* __restrict__ coef, float* __restrict__ out, int n_tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = 8 * wave + (lane >> 3), j = lane & 7;
  const int tile = blockIdx.x;
  float acc[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) acc[k] = 0.f;
  for (int ln = 0; ln < LINES; ++ln) {
    const float x0 = geo[ln * 2], dx = geo[ln * 2 + 1];  // scalar loads
    const float d = fmaf((float)r, dx, x0 + 0.01f * j + 1e-3f * tile);
    const float v = __builtin_amdgcn_rcpf(d);  // 1 / (nu - nu0), once for all layers
    const float* __restrict__ c = coef + (size_t)ln * NL * 8;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      float p = c[k * 8 + M - 1];
#pragma unroll
      for (int m = M - 2; m >= 0; --m) p = fmaf(p, v, c[k * 8 + m]);  // M - 1 FMAs with scalar coefficients ...
      acc[k] = fmaf(p * v, v, acc[k]);                                   // ... times v^2, accumulated: 2 more
    }
  }
#pragma unroll
  for (int k = 0; k < NL; ++k) out[((size_t)k * n_tiles + tile) * 128 + threadIdx.x] = acc[k];
}

int main() {
  const int n_tiles = 5372;
  std::vector<float> par((size_t)NL * LINES * 8), geo(LINES * 2), coef((size_t)LINES * NL * 8);
  for (size_t i = 0; i < par.size(); ++i) par[i] = 1.0f + 1e-3f * (float)(i % 97);
  for (size_t i = 0; i < geo.size(); ++i) geo[i] = 3.0f + 1e-2f * (float)(i % 13);
  for (size_t i = 0; i < coef.size(); ++i) coef[i] = 1e-2f * (float)(i % 31);
  float *d_par, *d_geo, *d_coef, *d_out;
  CK(hipMalloc(&d_par, par.size() * 4)); CK(hipMalloc(&d_geo, geo.size() * 4)); CK(hipMalloc(&d_coef, coef.size() * 4));
  CK(hipMalloc(&d_out, (size_t)NL * n_tiles * 128 * 4));
  CK(hipMemcpy(d_par, par.data(), par.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_geo, geo.data(), geo.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_coef, coef.data(), coef.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch, const char* name) -> int {
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(e0));
      launch();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      if (rep > 0 && t < best) best = t;
    }
    const double units = (double)n_tiles * LINES * NL;  // (line, tile, layer) triples: 8 nodes x 16 rows each
    printf("%-58s %7.3f ms for %d tiles x %d far lines x %d layers  = %6.2f ps per (line, node, row, layer)\n", name, best, n_tiles, LINES, NL,
           best * 1e9 / (units * 8 * ROWS));
    return 0;
  };
  if (time([&] { form_a<<<dim3(n_tiles, NL), 128>>>(d_par, d_out, n_tiles); }, "A  per-layer rational (7 VALU + rcp), today's row level")) return 1;
  if (time([&] { form_b<6><<<dim3(n_tiles), 128>>>(d_geo, d_coef, d_out, n_tiles); }, "B  cross-layer series, M = 6 (with pressure shifts)")) return 1;
  if (time([&] { form_b<4><<<dim3(n_tiles), 128>>>(d_geo, d_coef, d_out, n_tiles); }, "B  cross-layer series, M = 4")) return 1;
  if (time([&] { form_b<3><<<dim3(n_tiles), 128>>>(d_geo, d_coef, d_out, n_tiles); }, "B  cross-layer series, M = 3 (no pressure shift anywhere)")) return 1;
  return 0;
}
