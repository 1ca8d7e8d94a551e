// Experiment (standalone, not in the library): fp32 GEMM C = A * B^T with each fp32 operand split EXACTLY into three
// bf16 pieces (x = x1 + x2 + x3, 8 significant bits each, by truncation) and six bf16 MFMA terms per product
// (x1y1, x1y2, x2y1, x1y3, x3y1, x2y2; the dropped x2y3 + x3y2 + x3y3 are <= 2^-24 |x||y|), fp32 accumulation in the
// matrix core.  gfx950 has no xf32: v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate, so six bf16 MFMAs cost 6/16
// of one fp32 MFMA's time for the same product.  Question answered here: what does the LOOP reach once the split
// arithmetic (VALU), the 1.5x larger LDS image and the fragment reads are paid for?
//   hipcc -O3 --offload-arch=gfx950 -o tools/exp/bf16x6_gemm tools/exp/bf16x6_gemm.hip && tools/exp/bf16x6_gemm
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
typedef __attribute__((ext_vector_type(16))) float f16_t;

#ifndef NW
#define NW 4                                   // waves per workgroup: 4 -> 128 x 128 tile, 8 -> 256 x 128 (each wave 64 x 64)
#endif
constexpr int BM = 32 * NW, BN = 128, BK = 32, NT = 64 * NW;
constexpr int PITCH = 80;                      // bytes per 32-k bf16 row: 64 + 16 pad, 16-byte fragment reads conflict-free
constexpr int PLANE_A = BM * PITCH, PLANE_B = BN * PITCH;

__device__ __forceinline__ uint32_t hi_pair(uint32_t a, uint32_t b) {      // {bf16 trunc(a), bf16 trunc(b)}
  return __builtin_amdgcn_perm(b, a, 0x07060302u);
}

// x = p1 + p2 + p3 exactly; four consecutive k values -> 8 bytes per plane.  -DTRUNC: truncated pieces (first version:
// a systematic 2^-22 shrink); default: round-to-nearest pieces (v_cvt_pk_bf16_f32), as the library does.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t rne_pair(float a, float b) {
  const f32x2v v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split4(const float4 v, uint2& p1, uint2& p2, uint2& p3) {
  const float x[4] = {v.x, v.y, v.z, v.w};
  uint32_t p[3][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float a0 = x[2 * h], a1 = x[2 * h + 1];
#ifdef TRUNC
    p[0][h] = hi_pair(__float_as_uint(a0), __float_as_uint(a1));
#else
    p[0][h] = rne_pair(a0, a1);
#endif
    const float b0 = a0 - __uint_as_float(p[0][h] << 16), b1 = a1 - __uint_as_float(p[0][h] & 0xffff0000u);
#ifdef TRUNC
    p[1][h] = hi_pair(__float_as_uint(b0), __float_as_uint(b1));
#else
    p[1][h] = rne_pair(b0, b1);
#endif
    const float c0 = b0 - __uint_as_float(p[1][h] << 16), c1 = b1 - __uint_as_float(p[1][h] & 0xffff0000u);
    p[2][h] = hi_pair(__float_as_uint(c0), __float_as_uint(c1));
  }
  p1 = make_uint2(p[0][0], p[0][1]); p2 = make_uint2(p[1][0], p[1][1]); p3 = make_uint2(p[2][0], p[2][1]);
}

template <int TERMS>
__global__ __launch_bounds__(NT, 2) void gemm_bf16_split(const float* __restrict__ A, const float* __restrict__ B,
                                                          float* __restrict__ C, int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * PLANE_A + 3 * PLANE_B];
  unsigned char* ldsB = lds + 3 * PLANE_A;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int r = lane & 31, h = lane >> 5;
  constexpr int PA = BM * 8 / NT, PB = BN * 8 / NT;
  float4 ra[PA], rb[PB];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int f = tid + NT * i, row = f >> 3, kq = f & 7;
      ra[i] = *reinterpret_cast<const float4*>(A + (long)(m0 + row) * K + kt * BK + kq * 4);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int f = tid + NT * i, row = f >> 3, kq = f & 7;
      rb[i] = *reinterpret_cast<const float4*>(B + (long)(n0 + row) * K + kt * BK + kq * 4);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int f = tid + NT * i, row = f >> 3, kq = f & 7;
      uint2 p1, p2, p3;
      split4(ra[i], p1, p2, p3);
      unsigned char* d = lds + row * PITCH + kq * 8;
      *reinterpret_cast<uint2*>(d) = p1; *reinterpret_cast<uint2*>(d + PLANE_A) = p2; *reinterpret_cast<uint2*>(d + 2 * PLANE_A) = p3;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int f = tid + NT * i, row = f >> 3, kq = f & 7;
      uint2 p1, p2, p3;
      split4(rb[i], p1, p2, p3);
      unsigned char* d = ldsB + row * PITCH + kq * 8;
      *reinterpret_cast<uint2*>(d) = p1; *reinterpret_cast<uint2*>(d + PLANE_B) = p2; *reinterpret_cast<uint2*>(d + 2 * PLANE_B) = p3;
    }
  };
  f16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int nk = K / BK;
  gload(0);
  lstore();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf8_t a[2][3], b[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          a[i][p] = *reinterpret_cast<const bf8_t*>(lds + p * PLANE_A + (wm * 64 + i * 32 + r) * PITCH + (2 * s + h) * 16);
          b[i][p] = *reinterpret_cast<const bf8_t*>(ldsB + p * PLANE_B + (wn * 64 + i * 32 + r) * PITCH + (2 * s + h) * 16);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f16_t c = acc[i][j];
          if (TERMS >= 6) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
          }
          if (TERMS >= 3) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
          }
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
          acc[i][j] = c;
        }
    }
    __syncthreads();
    if (kt + 1 < nk) { lstore(); __syncthreads(); }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int col = n0 + wn * 64 + j * 32 + r;
        C[(long)row * N + col] = acc[i][j][e];
      }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int TERMS>
static void run(int M, int N, int K, const float* dA, const float* dB, float* dC, const std::vector<float>& hA,
                const std::vector<float>& hB) {
  dim3 grid(N / BN, M / BM);
  gemm_bf16_split<TERMS><<<grid, NT>>>(dA, dB, dC, M, N, K);
  CK(hipDeviceSynchronize());
  std::vector<float> hC((size_t)M * N);
  CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0, worst32 = 0;
  for (int t = 0; t < 4000; ++t) {
    const int i = (int)(((long)t * 7919) % M), j = (int)(((long)t * 104729) % N);
    double ref = 0, mag = 0; float f32 = 0.f;
    for (int k = 0; k < K; ++k) {
      const double a = hA[(size_t)i * K + k], b = hB[(size_t)j * K + k];
      ref += a * b; mag += fabs(a * b);
      f32 = fmaf(hA[(size_t)i * K + k], hB[(size_t)j * K + k], f32);
    }
    worst = fmax(worst, fabs(hC[(size_t)i * N + j] - ref) / mag);
    worst32 = fmax(worst32, fabs((double)f32 - ref) / mag);
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 20; ++w) gemm_bf16_split<TERMS><<<grid, NT>>>(dA, dB, dC, M, N, K);
  const int it = 30;
  CK(hipEventRecord(e0));
  for (int w = 0; w < it; ++w) gemm_bf16_split<TERMS><<<grid, NT>>>(dA, dB, dC, M, N, K);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = 1e3 * ms / it, tf = 2.0 * M * N * K / us / 1e6;
  printf("M %6d N %6d K %5d  terms %d  %9.1f us  %7.1f TFLOP/s (fp32-equivalent)  %7.1f bf16-MFMA TFLOP/s   "
         "max |err| / sum|a b| = %.2e  (k-ordered fp32 fma chain: %.2e)\n", M, N, K, TERMS, us, tf, tf * TERMS, worst, worst32);
}

int main() {
  const int shapes[][3] = {{8192, 8192, 4096}, {8192, 8192, 1152}, {8192, 8192, 576}, {16384, 4096, 2304}};
  for (auto& s : shapes) {
    const int M = s[0], N = s[1], K = s[2];
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    srand(1);
    for (auto& v : hA) v = (float)rand() / (float)RAND_MAX * 2.f - 1.f;
    for (auto& v : hB) v = ((float)rand() / (float)RAND_MAX * 2.f - 1.f) * ((rand() & 7) == 0 ? 37.f : 1.f);
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    run<6>(M, N, K, dA, dB, dC, hA, hB);
    run<3>(M, N, K, dA, dB, dC, hA, hB);
    run<1>(M, N, K, dA, dB, dC, hA, hB);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
  }
  return 0;
}
