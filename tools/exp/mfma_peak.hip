// How fast does the matrix pipe run when NOTHING else does?  One workgroup per CU, W waves, each issuing independent
// v_mfma_f32_32x32x16_bf16 on four accumulators in a loop; no memory traffic.  Prints TFLOP/s for several CU counts:
// the shader clock under matrix load (power management) sets the ceiling any MFMA-bound kernel can approach.
//   hipcc -O3 --offload-arch=gfx950 tools/exp/mfma_peak.hip -o tools/exp/mfma_peak && tools/exp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(512) void spin(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f - i * 0.01f); }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  if (s == 12345.678f) out[0] = s;
}
int main() {
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;                                  // 24 MFMAs per iteration and wave
  for (int waves : {4, 8}) for (int grid : {64, 128, 192, 224, 256}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0); spin<<<grid, waves * 64, 0, 0>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 32 * 32 * 16 * 24.0 * iters * waves * grid;
    printf("waves/CU %d  workgroups %3d: %8.3f ms  %7.1f TFLOP/s bf16 (%5.1f %% of 2516.8; per CU %5.2f)\n", waves, grid, ms, flop / ms / 1e9,
           100.0 * flop / ms / 1e9 / 2516.8, flop / ms / 1e9 / grid);
  }
  return 0;
}
