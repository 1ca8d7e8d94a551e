// All-pairs Euclidean distance matrix of an embedding block on the fp32 MFMA engine.
//
// Replaces the sklearn call at /root/reference/embedding_net/datagenerators.py:219
// (pairwise_distances(all_embeddings)):  D = sqrt(max(|x|^2 + |y|^2 - 2 X X^T, 0)),
// diagonal forced to 0.  The -2 X X^T term is the GEMM; norms/clamp/diag/sqrt are its
// epilogue.  Roofline: MFMA f32, 2*N*N*E FLOP per launch (full matrix, no symmetry credit).
#include "gemm_engine.h"
#include "../../include/embnet.h"

namespace embnet {

// one wave per row: nn[row] = sum x^2
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float* __restrict__ x, int n, int e,
                                                         float* __restrict__ nn) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  const float* r = x + (long)row * e;
  float s = 0.f;
  for (int k = lane; k < e; k += 64) s = fmaf(r[k], r[k], s);
  s = wave_sum(s);
  if (lane == 0) nn[row] = s;
}

struct PairwiseParams {
  const float* x; const float* nn; float* d; int n, e, squared;
};

template <class G, bool VEC>
__global__ __launch_bounds__(256) void pairwise_kernel(PairwiseParams p) {
  using TA = TileKC<G::BM>;
  using TB = TileKC<G::BN>;
  __shared__ __attribute__((aligned(16))) float smem[(TA::FLOATS + TB::FLOATS) > EPI_FLOATS<G> ? (TA::FLOATS + TB::FLOATS) : EPI_FLOATS<G>];
  const int tiles_n = (p.n + G::BN - 1) / G::BN;
  const int tiles_m = (p.n + G::BM - 1) / G::BM;
  const int id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (id / tiles_n) * G::BM, n0 = (id % tiles_n) * G::BN;

  LoadRowsKC<G::BM, VEC> la; la.init(p.x, p.e, p.n, p.e, m0, threadIdx.x);
  LoadRowsKC<G::BN, VEC> lb; lb.init(p.x, p.e, p.n, p.e, n0, threadIdx.x);
  f32x16 acc[G::TM][G::TN];
  gemm_mainloop<G, TA, TB>(la, lb, 0, (p.e + BK - 1) / BK, smem, acc);

  if ((p.n & 3) == 0) {                                  // 16-byte row stores of the matrix
    for_each_acc_row4<G>(acc, smem, [&](int r, int c, float4 g) {
      const int row = m0 + r, col = n0 + c;
      if (row < p.n && col < p.n) {
        const float nr = p.nn[row];
        const float4 nc = *reinterpret_cast<const float4*>(p.nn + col);
        float o[4] = {fmaxf(nr + nc.x - 2.f * g.x, 0.f), fmaxf(nr + nc.y - 2.f * g.y, 0.f),
                      fmaxf(nr + nc.z - 2.f * g.z, 0.f), fmaxf(nr + nc.w - 2.f * g.w, 0.f)};
#pragma unroll
        for (int j = 0; j < 4; ++j) { if (row == col + j) o[j] = 0.f; if (!p.squared) o[j] = sqrtf(o[j]); }
        *reinterpret_cast<float4*>(p.d + (long)row * p.n + col) = make_float4(o[0], o[1], o[2], o[3]);
      }
    });
    return;
  }
  for_each_acc<G>(acc, [&](int r, int c, float g) {
    const int row = m0 + r, col = n0 + c;
    if (row < p.n && col < p.n) {
      float v = fmaxf(p.nn[row] + p.nn[col] - 2.f * g, 0.f);
      if (row == col) v = 0.f;
      p.d[(long)row * p.n + col] = p.squared ? v : sqrtf(v);
    }
  });
}

}  // namespace embnet

using namespace embnet;

extern "C" size_t embnet_pairwise_workspace_bytes(int n) { return n > 0 ? (size_t)n * sizeof(float) : 0; }

extern "C" int embnet_pairwise_dist_f32(const float* x, int n, int e, float* dist, int squared,
                                        void* workspace, size_t workspace_bytes, void* stream) {
  EMBNET_CHECK_ARG(x && dist && workspace, "pairwise: null pointer");
  EMBNET_CHECK_ARG(((reinterpret_cast<uintptr_t>(dist) | reinterpret_cast<uintptr_t>(workspace)) & 15) == 0,
                   "pairwise: dist and workspace must be 16-byte aligned");
  EMBNET_CHECK_ARG(n > 0 && e > 0, "pairwise: n=%d e=%d must be positive", n, e);
  if (workspace_bytes < embnet_pairwise_workspace_bytes(n))
    return fail(EMBNET_EWORKSPACE, "pairwise: workspace %zu < %zu bytes", workspace_bytes,
                embnet_pairwise_workspace_bytes(n));
  hipStream_t s = (hipStream_t)stream;
  float* nn = (float*)workspace;
  row_sqnorm_kernel<<<cdiv(n, 4), 256, 0, s>>>(x, n, e, nn);
  PairwiseParams p{x, nn, dist, n, e, squared};
  EMBNET_CHECK_ARG((size_t)n * e * 4 <= MAX_OPERAND_BYTES, "pairwise: embedding block exceeds 2 GiB");
  const bool vec = (e & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  using GL = Geom<128, 128, 2, 2>;
  using GS = Geom<64, 64, 2, 2>;
  if (n >= 1024) {
    const int grid = cdiv(n, 128) * cdiv(n, 128);
    if (vec) pairwise_kernel<GL, true><<<grid, 256, 0, s>>>(p); else pairwise_kernel<GL, false><<<grid, 256, 0, s>>>(p);
  } else {
    const int grid = cdiv(n, 64) * cdiv(n, 64);
    if (vec) pairwise_kernel<GS, true><<<grid, 256, 0, s>>>(p); else pairwise_kernel<GS, false><<<grid, 256, 0, s>>>(p);
  }
  return check_launch("pairwise_dist");
}
