// Experiment (not part of libembnet_hip.so): fp32 GEMM C = A * B^T with the structure DESIGN.md §3.5 "Next"
// describes — operand tiles brought into LDS by LDS-DMA (global_load_lds_dwordx4), no register staging,
// 8 waves per workgroup on a 256x256 tile, one workgroup per CU, two LDS stages, XOR swizzle applied on the
// SOURCE address of the DMA.  Build: hipcc -O3 --offload-arch=gfx950 glds_gemm.hip -o glds_gemm
// Run:   ./glds_gemm M N K      (M, N multiples of 256; K a multiple of 32)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int BM = 256, BN = 256, BK = 32, NT = 512;
constexpr int ROWB = BK * 4;                    // 128 bytes per tile row
constexpr int OP_BYTES = BM * ROWB;             // 32 KB per operand per stage
constexpr int STAGE_BYTES = 2 * OP_BYTES;       // 64 KB

typedef float f32x4 __attribute__((ext_vector_type(4)));
// Fragment reads are issued as inline asm: a C++ load of the staging array makes hipcc (ROCm 7.2) put
// `s_waitcnt vmcnt(0)` in front of the first ds_read of every K tile (it cannot tell the stage being read from the
// stage the LDS-DMA in flight is writing), which serialises DMA and compute.  The asm reads carry no such wait; the
// matching lgkmcnt wait is the wait_frags() statement, tied to the registers so the MFMAs cannot move above it.
__device__ __forceinline__ void frag_read(unsigned op, int row, int j, int h, f32x4& q) {
  const unsigned addr = op + (unsigned)(row * ROWB + (((2 * j + h) ^ (row & 7)) * 16));
  asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(addr));
}
__device__ __forceinline__ void wait_frags(f32x4 (&a)[4], f32x4 (&b)[2]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]));
}

__global__ __launch_bounds__(NT) void glds_gemm_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                       float* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 stages x (A 32 KB + B 32 KB)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_n = N / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;          // 2 (M) x 4 (N) waves, 128 x 64 each
  const int i = lane & 31, h = lane >> 5;
  // DMA role: each wave moves 4 groups of 8 rows per operand per K tile; lane -> (row in group, physical chunk)
  const int drow = lane >> 3, dphys = lane & 7, dlog = dphys ^ (drow & 7);
  const float* ga[4]; const float* gb[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int grp = wave * 4 + g;                                   // 32 groups of 8 rows
    ga[g] = A + (size_t)(m0 + grp * 8 + drow) * K + dlog * 4;
    gb[g] = B + (size_t)(n0 + grp * 8 + drow) * K + dlog * 4;
  }
  auto stage = [&](int s, int kt) {
    char* base = smem + s * STAGE_BYTES;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int grp = wave * 4 + g;
      __builtin_amdgcn_global_load_lds(ga[g] + kt * BK, LDS_PTR(base + grp * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(gb[g] + kt * BK, LDS_PTR(base + OP_BYTES + grp * 1024), 16, 0, 0);
    }
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nt = K / BK;
  auto mfma_step = [&](f32x4 (&fa)[4], f32x4 (&fb)[2]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][t], fb[b][t], acc[a][b], 0, 0, 0);
  };
  auto read_step = [&](int s, int j, f32x4 (&fa)[4], f32x4 (&fb)[2]) {
    const unsigned sa = (unsigned)(s * STAGE_BYTES), sb = sa + OP_BYTES;     // LDS byte addresses (the array starts at 0)
#pragma unroll
    for (int a = 0; a < 4; ++a) frag_read(sa, wm + 32 * a + i, j, h, fa[a]);
#pragma unroll
    for (int b = 0; b < 2; ++b) frag_read(sb, wn + 32 * b + i, j, h, fb[b]);
  };
  f32x4 fa[2][4], fb[2][2];
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_step(0, 0, fa[0], fb[0]);
  for (int kt = 0; kt < nt; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nt) stage(cur ^ 1, kt + 1);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      wait_frags(fa[j & 1], fb[j & 1]);
      read_step(cur, j + 1, fa[(j + 1) & 1], fb[(j + 1) & 1]);
      mfma_step(fa[j & 1], fb[j & 1]);
    }
    // last k-step of the tile: its fragments are in registers, so the stage can be handed over BEFORE its MFMAs:
    // retire the DMA of the next tile, barrier, read the next tile's first fragments, and only then issue the 32 MFMAs
    // that cover the barrier skew and the read latency
    wait_frags(fa[1], fb[1]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < nt) read_step(cur ^ 1, 0, fa[0], fb[0]);
    mfma_step(fa[1], fb[1]);
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h, col = n0 + wn + 32 * b + i;
        C[(size_t)row * N + col] = acc[a][b][r];
      }
}

__global__ void ref_kernel(const float* A, const float* B, float* C, int M, int N, int K) {
  const int col = blockIdx.x * 16 + threadIdx.x, row = blockIdx.y * 16 + threadIdx.y;
  if (row >= M || col >= N) return;
  double s = 0;
  for (int k = 0; k < K; ++k) s += (double)A[(size_t)row * K + k] * B[(size_t)col * K + k];
  C[(size_t)row * N + col] = (float)s;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
  if (M % BM || N % BN || K % BK) { printf("M,N multiples of 256, K of 32\n"); return 1; }
  std::vector<float> ha((size_t)M * K), hb((size_t)N * K);
  srand(1);
  for (auto& v : ha) v = (rand() / (float)RAND_MAX) * 2 - 1;
  for (auto& v : hb) v = (rand() / (float)RAND_MAX) * 2 - 1;
  float *A, *B, *C, *R;
  hipMalloc(&A, ha.size() * 4); hipMalloc(&B, hb.size() * 4); hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&R, (size_t)M * N * 4);
  hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice); hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)glds_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
  const int grid = (M / BM) * (N / BN);
  glds_gemm_kernel<<<grid, NT, 2 * STAGE_BYTES>>>(A, B, C, M, N, K);
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(e)); return 2; }
  if ((size_t)M * N <= (size_t)2048 * 2048) {
    ref_kernel<<<dim3(N / 16, M / 16), dim3(16, 16)>>>(A, B, R, M, N, K);
    std::vector<float> hc((size_t)M * N), hr((size_t)M * N);
    hipMemcpy(hc.data(), C, hc.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hr.data(), R, hr.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (size_t k = 0; k < hc.size(); ++k) { maxerr = fmax(maxerr, fabs((double)hc[k] - hr[k])); maxref = fmax(maxref, fabs((double)hr[k])); }
    printf("check %dx%dx%d: max err %.3e (max |ref| %.3e)\n", M, N, K, maxerr, maxref);
  }
  hipEvent_t s, t; hipEventCreate(&s); hipEventCreate(&t);
  for (int w = 0; w < 3; ++w) glds_gemm_kernel<<<grid, NT, 2 * STAGE_BYTES>>>(A, B, C, M, N, K);
  hipEventRecord(s);
  const int iters = 20;
  for (int w = 0; w < iters; ++w) glds_gemm_kernel<<<grid, NT, 2 * STAGE_BYTES>>>(A, B, C, M, N, K);
  hipEventRecord(t); hipEventSynchronize(t);
  float ms; hipEventElapsedTime(&ms, s, t);
  const double sec = ms * 1e-3 / iters;
  printf("glds_gemm %dx%dx%d: %.1f us  %.1f TFLOP/s (%.1f %% of 157.3)\n", M, N, K, sec * 1e6, 2.0 * M * N * K / sec / 1e12,
         2.0 * M * N * K / sec / 1e12 / 157.3 * 100);
  return 0;
}
