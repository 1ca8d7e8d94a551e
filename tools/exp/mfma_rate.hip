// Calibration: cycles per v_mfma_f32_32x32x2_f32 on one SIMD with W waves per SIMD and A independent accumulators
// per wave, operands in registers, nothing else in the loop.  hipcc -O3 --offload-arch=gfx950 mfma_rate.hip -o mfma_rate
// Prints shader cycles per MFMA per SIMD (the guide's constant is 64) and TFLOP/s chip-wide for each (W, A).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int A>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, float seed) {
  f32x16 acc[A];
  for (int i = 0; i < A; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = seed * (threadIdx.x % 7 - 3) + 0.37f, b = seed * (threadIdx.x % 5 - 2) - 0.11f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < A; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < A; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int A>
void run(int wg_per_cu, int iters) {
  const int grid = 256 * wg_per_cu;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, grid * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) k<A><<<grid, 256>>>(out, cyc, iters, 0.5f);
  hipEventRecord(e0);
  k<A><<<grid, 256>>>(out, cyc, iters, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(grid);
  hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += v; mean /= grid;
  const double mf = (double)iters * 8 * A;                    // MFMAs per wave
  printf("waves/SIMD %d  accumulators %d: %.1f cycles per MFMA per wave, %.1f per SIMD-slot; %.1f TFLOP/s\n", wg_per_cu, A,
         mean / mf, mean / mf / wg_per_cu, 4096.0 * mf * 4 * grid / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w : {1, 2, 3, 4, 7}) { run<1>(w, 4000 / w); run<2>(w, 2000 / w); run<4>(w, 1000 / w); }
  return 0;
}
