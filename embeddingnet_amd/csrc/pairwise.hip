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
  int kt_per_split, splits; float* slabs;      // splits > 1: blockIdx.y takes a K range and writes a raw Gram slab
};

template <class G, bool VEC>
__global__ __launch_bounds__(256) void pairwise_kernel(PairwiseParams p) {
  using TA = TileKC<G::BM>;
  using TB = TileKC<G::BN>;
  __shared__ __attribute__((aligned(16))) float smem[MAIN_FLOATS<TA, TB> > EPI_FLOATS<G> ? MAIN_FLOATS<TA, TB> : EPI_FLOATS<G>];
  prio_hi();                                             // prologue at raised issue priority (gemm_engine.h)
  const int tiles_n = (p.n + G::BN - 1) / G::BN;
  const int tiles_m = (p.n + G::BM - 1) / G::BM;
  const int id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (id / tiles_n) * G::BM, n0 = (id % tiles_n) * G::BN;

  stamp(0);
  LoadRowsKC<G::BM, VEC> la; la.init(p.x, p.e, p.n, p.e, m0, threadIdx.x);
  LoadRowsKC<G::BN, VEC> lb; lb.init(p.x, p.e, p.n, p.e, n0, threadIdx.x);
  f32x16 acc[G::TM][G::TN];
  stamp(2);
  if (p.splits > 1) {
    // N <= 256 at the reference's default encodings_len = 4096 (backbones.py:13) is 4..16 tiles walking 128 K tiles each
    // (98 us): the K range is cut over blockIdx.y, pairwise_finish_kernel adds the slabs and applies the epilogue
    const int kt_total = (p.e + BK - 1) / BK;
    const int kt0 = blockIdx.y * p.kt_per_split, kt1 = min(kt0 + p.kt_per_split, kt_total);
    gemm_mainloop<G, TA, TB>(la, lb, kt0, kt1, smem, acc);
    float* slab = p.slabs + (long)blockIdx.y * p.n * p.n;
    for_each_acc<G>(acc, [&](int r, int c, float g) {
      const int row = m0 + r, col = n0 + c;
      if (row < p.n && col < p.n) slab[(long)row * p.n + col] = g;
    });
    return;
  }
  gemm_mainloop<G, TA, TB>(la, lb, 0, (p.e + BK - 1) / BK, smem, acc);
  stamp(4);

  if ((p.n & 3) == 0) {                                  // 16-byte row stores of the matrix
    // the squared norms of the lane's rows and of its column quad, fetched before the staging loop (gemm_engine.h, EpiIdx)
    constexpr int NJ = EpiIdx<G>::NJ;
    const int ecol = n0 + epi_col<G>();
    const float4 nc0 = *reinterpret_cast<const float4*>(p.nn + (ecol < p.n ? ecol : 0));
    float nrow[G::TM][NJ];
#pragma unroll
    for (int im = 0; im < G::TM; ++im)
#pragma unroll
      for (int j = 0; j < NJ; ++j) nrow[im][j] = p.nn[min(m0 + epi_row<G>(im, j), p.n - 1)];
    float4 nc = nc0;
    settle(nc);
#pragma unroll
    for (int im = 0; im < G::TM; ++im)
#pragma unroll
      for (int j = 0; j < NJ; ++j) settle(nrow[im][j]);
    auto emit = [&](int im, int j, int row, int col, float4 g) {
      const float nr = nrow[im][j];
      float o[4] = {fmaxf(nr + nc.x - 2.f * g.x, 0.f), fmaxf(nr + nc.y - 2.f * g.y, 0.f),
                    fmaxf(nr + nc.z - 2.f * g.z, 0.f), fmaxf(nr + nc.w - 2.f * g.w, 0.f)};
#pragma unroll
      for (int q = 0; q < 4; ++q) { if (row == col + q) o[q] = 0.f; if (!p.squared) o[q] = sqrtf(o[q]); }
      *reinterpret_cast<float4*>(p.d + (long)row * p.n + col) = make_float4(o[0], o[1], o[2], o[3]);
    };
    if (m0 + G::BM <= p.n && n0 + G::BN <= p.n) {          // interior tile (wave-uniform): straight-line stores, no edge tests
      for_each_acc_row4_idx<G>(acc, smem, [&](int im, int j, int r, int c, float4 g) { emit(im, j, m0 + r, n0 + c, g); });
    } else {
      for_each_acc_row4_idx<G>(acc, smem, [&](int im, int j, int r, int c, float4 g) {
        if (m0 + r < p.n && n0 + c < p.n) emit(im, j, m0 + r, n0 + c, g);
      });
    }
    stamp(5);
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

__global__ __launch_bounds__(256) void pairwise_finish_kernel(const float* __restrict__ slabs, int splits, int n,
                                                              const float* __restrict__ nn, int squared, float* __restrict__ d) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x, nn2 = (long)n * n;
  if (i >= nn2) return;
  const int row = (int)(i / n), col = (int)(i % n);
  float g = slabs[i];
  for (int s = 1; s < splits; ++s) g += slabs[(long)s * nn2 + i];
  float v = fmaxf(nn[row] + nn[col] - 2.f * g, 0.f);
  if (row == col) v = 0.f;
  d[i] = squared ? v : sqrtf(v);
}

}  // namespace embnet

using namespace embnet;

#if EMBNET_STAMPS
extern "C" int embnet_debug_set_stamps_pairwise(void* buf) {      // diagnostic build only (gemm_engine.h: stamp())
  return hipMemcpyToSymbol(HIP_SYMBOL(embnet::g_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

// Tile and K-split choice.  A launch should put >= ~512 workgroups on the 256 CUs (two resident per CU cover each other's
// prologue and epilogue): 128x128 tiles from 384 tiles on (N >= ~2 500); below that 64x64 tiles, and where even those are
// fewer than 512 the reduction is cut over workgroups (partial Gram slabs + pairwise_finish_kernel) — round 4 ran N = 1 024 as
// 64 tiles of 128x128 on a quarter of the chip (0.13-0.17 of the fp32 MFMA peak, profiles/r04_gemm_sweep.json; now 0.29-0.52).
static bool pairwise_big_tiles(int n) {
  static const long gl_min = env_long("EMBNET_PAIRWISE_GL_MIN_TILES", 384);
  return (long)cdiv(n, 128) * cdiv(n, 128) >= gl_min;
}
static void pairwise_plan(int n, int e, int& splits, int& kt_per_split) {
  const int kt_total = cdiv(e, BK);
  splits = 1; kt_per_split = kt_total;
  if (pairwise_big_tiles(n)) return;
  const long tiles = (long)cdiv(n, 64) * cdiv(n, 64);
  // measured (tools/exp/run_r05_gemm_mid.sh, profiles/r05_gemm_mid_sweep.txt): the split pays where the tiles alone leave CUs
  // idle (N = 512: 64 tiles) or the reduction is long (E >= 2 048: a second workgroup per CU hides the first one's latencies);
  // at N = 1 024, E <= 512 the 256 tiles already occupy every CU and the slab pass only adds its ~6 us (17.5 vs 11.7 us)
  static const long target = env_long("EMBNET_PAIRWISE_SPLIT_TARGET", 0);
  static const long min_kt = env_long("EMBNET_PAIRWISE_MIN_KT", 4);
  long want = (target > 0 ? target : (kt_total >= 64 ? 512 : 256)) / tiles;
  if (n < 512) {                                     // the batch sizes of a training step: as tuned in round 3
    if (kt_total < 32) return;
    want = 256 / tiles; if (want > kt_total / 8) want = kt_total / 8;
  }
  else if (want > kt_total / min_kt) want = kt_total / min_kt;
  if (want < 2) return;
  kt_per_split = cdiv(kt_total, want);
  splits = cdiv(kt_total, kt_per_split);
}

// n floats of row norms (+ the Gram slabs of a K-split launch)
extern "C" size_t embnet_pairwise_workspace_bytes(int n, int e) {
  if (n <= 0 || e <= 0) return 0;
  int splits, ktps; pairwise_plan(n, e, splits, ktps);
  const size_t norms = ((size_t)n * sizeof(float) + 15) / 16 * 16;
  return norms + (splits > 1 ? (size_t)splits * n * n * sizeof(float) : 0);
}

extern "C" int embnet_pairwise_dist_f32(const float* x, int n, int e, float* dist, int squared,
                                        void* workspace, size_t workspace_bytes, void* stream) {
  EMBNET_CHECK_ARG(x && dist && workspace, "pairwise: null pointer");
  EMBNET_CHECK_ARG(((reinterpret_cast<uintptr_t>(dist) | reinterpret_cast<uintptr_t>(workspace)) & 15) == 0,
                   "pairwise: dist and workspace must be 16-byte aligned");
  EMBNET_CHECK_ARG(n > 0 && e > 0, "pairwise: n=%d e=%d must be positive", n, e);
  if (workspace_bytes < (size_t)n * sizeof(float))
    return fail(EMBNET_EWORKSPACE, "pairwise: workspace %zu < %zu bytes", workspace_bytes, (size_t)n * sizeof(float));
  hipStream_t s = (hipStream_t)stream;
  float* nn = (float*)workspace;
  row_sqnorm_kernel<<<cdiv(n, 4), 256, 0, s>>>(x, n, e, nn);
  PairwiseParams p{x, nn, dist, n, e, squared, 0, 1, nullptr};
  EMBNET_CHECK_ARG((size_t)n * e * 4 <= MAX_OPERAND_BYTES, "pairwise: embedding block exceeds 2 GiB");
  const bool vec = (e & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  using GL = Geom<128, 128, 2, 2>;
  using GS = Geom<64, 64, 2, 2>;
  int splits, ktps; pairwise_plan(n, e, splits, ktps);
  if (splits > 1 && workspace_bytes >= embnet_pairwise_workspace_bytes(n, e)) {
    p.splits = splits; p.kt_per_split = ktps;
    p.slabs = (float*)((char*)workspace + ((size_t)n * sizeof(float) + 15) / 16 * 16);
    const dim3 grid(cdiv(n, 64) * cdiv(n, 64), splits);
    {
      EMBNET_TRACE_FLOP("embnet::pairwise_kernel", 2.0 * n * n * e, 4.0 * ((double)n * e + (double)n * n * splits), s);
      if (vec) pairwise_kernel<GS, true><<<grid, 256, 0, s>>>(p); else pairwise_kernel<GS, false><<<grid, 256, 0, s>>>(p);
    }
    EMBNET_TRACE("embnet::pairwise_finish_kernel", TRACE_BYTES, 4.0 * n * n * (splits + 1), s);
    pairwise_finish_kernel<<<cdiv((long)n * n, 256), 256, 0, s>>>(p.slabs, splits, n, nn, squared, dist);
    return check_launch("pairwise_dist");
  }
  EMBNET_TRACE_FLOP("embnet::pairwise_kernel", 2.0 * n * n * e, 4.0 * ((double)n * e + (double)n * n), s);
  if (pairwise_big_tiles(n)) {
    const int grid = cdiv(n, 128) * cdiv(n, 128);
    if (vec) pairwise_kernel<GL, true><<<grid, 256, 0, s>>>(p); else pairwise_kernel<GL, false><<<grid, 256, 0, s>>>(p);
  } else {
    const int grid = cdiv(n, 64) * cdiv(n, 64);
    if (vec) pairwise_kernel<GS, true><<<grid, 256, 0, s>>>(p); else pairwise_kernel<GS, false><<<grid, 256, 0, s>>>(p);
  }
  return check_launch("pairwise_dist");
}

// ---------------------------------------------------------------------------------------------
// Query-vs-gallery distances and k-nearest-neighbour selection: the evaluation step right after
// training (/root/reference/embedding_net/models.py:128-161 predict_knn / calculate_prediction_accuracy
// through sklearn's KNeighborsClassifier, brute-force Euclidean).  Same GEMM engine, same epilogue
// minus the diagonal rule.
namespace embnet {

struct CrossParams { const float* q; const float* x; const float* qn; const float* xn; float* d; int nq, n, e, squared; };

template <class G, bool VEC>
__global__ __launch_bounds__(256) void cross_dist_kernel(CrossParams p) {
  using TA = TileKC<G::BM>;
  using TB = TileKC<G::BN>;
  __shared__ __attribute__((aligned(16))) float smem[MAIN_FLOATS<TA, TB> > EPI_FLOATS<G> ? MAIN_FLOATS<TA, TB> : EPI_FLOATS<G>];
  prio_hi();                                             // prologue at raised issue priority (gemm_engine.h)
  const int tiles_n = (p.n + G::BN - 1) / G::BN;
  const int m0 = (blockIdx.x / tiles_n) * G::BM, n0 = (blockIdx.x % tiles_n) * G::BN;
  LoadRowsKC<G::BM, VEC> la; la.init(p.q, p.e, p.nq, p.e, m0, threadIdx.x);
  LoadRowsKC<G::BN, VEC> lb; lb.init(p.x, p.e, p.n, p.e, n0, threadIdx.x);
  f32x16 acc[G::TM][G::TN];
  gemm_mainloop<G, TA, TB>(la, lb, 0, (p.e + BK - 1) / BK, smem, acc);
  for_each_acc<G>(acc, [&](int r, int c, float g) {
    const int row = m0 + r, col = n0 + c;
    if (row < p.nq && col < p.n) {
      const float v = fmaxf(p.qn[row] + p.xn[col] - 2.f * g, 0.f);
      p.d[(long)row * p.n + col] = p.squared ? v : sqrtf(v);
    }
  });
}

// k smallest entries of each row, ascending, ties to the smaller column: one wave per row, k rounds of a
// shuffle arg-min over the columns not yet taken (k <= 64: taken columns live one per lane).
__global__ __launch_bounds__(256) void topk_smallest_kernel(const float* __restrict__ d, int rows, int n, int k,
                                                            int* __restrict__ idx, float* __restrict__ val) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* r = d + (long)row * n;
  int taken = -1;                                         // lane j remembers the column chosen in round j
  for (int round = 0; round < k; ++round) {
    float best = INFINITY; int bi = 0x7fffffff;
    for (int c = lane; c < n; c += 64) {
      bool used = false;
      for (int j = 0; j < round; ++j) used |= (__shfl(taken, j, 64) == c);
      const float v = r[c];
      if (!used && (v < best || (v == best && c < bi))) { best = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
      if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == round) taken = bi;
    if (lane == 0) { idx[(long)row * k + round] = bi; val[(long)row * k + round] = best; }
  }
}

// majority vote over the k neighbour labels, ties to the smallest label (scipy.stats.mode, as sklearn predict)
__global__ __launch_bounds__(256) void knn_vote_kernel(const int* __restrict__ idx, const int* __restrict__ labels,
                                                       int rows, int k, int* __restrict__ pred) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  int best_label = 0x7fffffff, best_count = 0;
  for (int i = 0; i < k; ++i) {
    const int li = labels[idx[(long)row * k + i]];
    int cnt = 0;
    for (int j = 0; j < k; ++j) cnt += labels[idx[(long)row * k + j]] == li;
    if (cnt > best_count || (cnt == best_count && li < best_label)) { best_count = cnt; best_label = li; }
  }
  pred[row] = best_label;
}

}  // namespace embnet

extern "C" size_t embnet_cross_dist_workspace_bytes(int nq, int n) { return nq > 0 && n > 0 ? ((size_t)nq + n) * sizeof(float) : 0; }

extern "C" int embnet_cross_dist_f32(const float* q, int nq, const float* x, int n, int e, float* dist, int squared,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  EMBNET_CHECK_ARG(q && x && dist && workspace, "cross_dist: null pointer");
  EMBNET_CHECK_ARG(nq > 0 && n > 0 && e > 0, "cross_dist: nq=%d n=%d e=%d must be positive", nq, n, e);
  EMBNET_CHECK_ARG((size_t)nq * e * 4 <= MAX_OPERAND_BYTES && (size_t)n * e * 4 <= MAX_OPERAND_BYTES,
                   "cross_dist: an embedding block exceeds 2 GiB");
  if (workspace_bytes < embnet_cross_dist_workspace_bytes(nq, n))
    return fail(EMBNET_EWORKSPACE, "cross_dist: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* qn = (float*)workspace; float* xn = qn + nq;
  row_sqnorm_kernel<<<cdiv(nq, 4), 256, 0, s>>>(q, nq, e, qn);
  row_sqnorm_kernel<<<cdiv(n, 4), 256, 0, s>>>(x, n, e, xn);
  CrossParams p{q, x, qn, xn, dist, nq, n, e, squared};
  const bool vec = (e & 3) == 0 && ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(x)) & 15) == 0;
  using GS = Geom<64, 64, 2, 2>;
  using GL = Geom<128, 128, 2, 2>;
  EMBNET_TRACE_FLOP("embnet::cross_dist_kernel", 2.0 * nq * n * e, 4.0 * ((double)nq * e + (double)n * e + (double)nq * n), s);
  static const long gl_min = env_long("EMBNET_PAIRWISE_GL_MIN_TILES", 384);
  if ((long)cdiv(nq, 128) * cdiv(n, 128) >= gl_min) {           // queries x gallery large enough to fill the chip with 128x128 tiles
    const int grid = cdiv(nq, 128) * cdiv(n, 128);
    if (vec) cross_dist_kernel<GL, true><<<grid, 256, 0, s>>>(p); else cross_dist_kernel<GL, false><<<grid, 256, 0, s>>>(p);
  } else {
    const int grid = cdiv(nq, 64) * cdiv(n, 64);
    if (vec) cross_dist_kernel<GS, true><<<grid, 256, 0, s>>>(p); else cross_dist_kernel<GS, false><<<grid, 256, 0, s>>>(p);
  }
  return check_launch("cross_dist");
}

extern "C" int embnet_topk_smallest(const float* dist, int rows, int n, int k, int32_t* idx, float* val, void* stream) {
  EMBNET_CHECK_ARG(dist && idx && val, "topk_smallest: null pointer");
  EMBNET_CHECK_ARG(rows > 0 && n > 0 && k > 0 && k <= 64 && k <= n, "topk_smallest: need 0 < k <= min(64, n) (k=%d n=%d)", k, n);
  topk_smallest_kernel<<<cdiv(rows, 4), 256, 0, (hipStream_t)stream>>>(dist, rows, n, k, idx, val);
  return check_launch("topk_smallest");
}

extern "C" int embnet_knn_vote(const int32_t* idx, const int32_t* labels, int rows, int k, int32_t* pred, void* stream) {
  EMBNET_CHECK_ARG(idx && labels && pred && rows > 0 && k > 0, "knn_vote: bad argument");
  knn_vote_kernel<<<cdiv(rows, 256), 256, 0, (hipStream_t)stream>>>(idx, labels, rows, k, pred);
  return check_launch("knn_vote");
}
