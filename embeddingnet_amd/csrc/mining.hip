// Online triplet mining on the distance matrix: mine-and-select.
//
// Replaces the Python double loop of TripletsDataGenerator.get_batch_triplets_mining
// (/root/reference/embedding_net/datagenerators.py:225-250) and its three selection
// rules (:188-199).  Rows are class-contiguous: class c owns rows [cK,(c+1)K).
// For every ordered positive pair (i<j) of a class, in the order combinations() yields,
//   loss[q] = (D[i,j] - D[i,neg_q]) + margin      (two f32 roundings, like NumPy)
// over the N-K out-of-class columns neg_q in ascending order, then
//   hardest      first arg-max, kept iff loss > 0
//   random_hard  uniform pick among {loss > 0}
//   semihard     uniform pick among {0 < loss < margin}
// One wavefront scans one pair (ballot/popcount rank-select for the random rules,
// shuffle arg-max for hardest); a single-workgroup pass then compacts the active pairs
// into the (a,p,n) list in pair order and applies the reference's fallback triplet.
// HBM-bound integer/compare work: 4*N*(N-K)*(K-1)/2.. bytes read from an L2-resident matrix.
#include "common.h"
#include "../../include/embnet.h"

namespace embnet {

__device__ __forceinline__ void decode_pair(int q, int k, int& ii, int& jj) {
  // q-th pair of combinations(range(k), 2)
  int i = 0, rem = q;
  while (rem >= k - 1 - i) { rem -= k - 1 - i; ++i; }
  ii = i; jj = i + 1 + rem;
}

__global__ __launch_bounds__(256) void zero_words_kernel(uint32_t* __restrict__ p, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = 0u;
}
__global__ __launch_bounds__(256) void mine_select_kernel(
    const float* __restrict__ D, int n, int p, int k, float margin, int mode, uint64_t seed,
    int* __restrict__ selected, uint32_t* __restrict__ cand_mask, int mask_words) {
  const int ppc = k * (k - 1) / 2;
  const int npairs = p * ppc;
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (pair >= npairs) return;
  const int c = pair / ppc;
  int ii, jj;
  decode_pair(pair % ppc, k, ii, jj);
  const int lo = c * k, nneg = n - k;
  const int i = lo + ii, j = lo + jj;
  const float* row = D + (long)i * n;
  const float dap = row[j];

  auto loss_at = [&](int q) -> float {
    const int col = q < lo ? q : q + k;
    return __fadd_rn(__fsub_rn(dap, row[col]), margin);
  };

  int result = -1;
  if (mode == EMBNET_MINE_HARDEST) {
    float best = -INFINITY; int bq = 0x7fffffff;
    // four loads in flight per lane (a wave with one dword load per trip is a chain of memory round trips: 256 trips x ~2 us
    // at N = 16 384, 2.3 TB/s with every wave slot taken); the visiting order per lane is unchanged: q ascending
    for (int q0 = lane; q0 < nneg; q0 += 4 * 64) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = q0 + 64 * u < nneg ? loss_at(q0 + 64 * u) : -INFINITY;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (v[u] > best) { best = v[u]; bq = q0 + 64 * u; }   // ascending q per lane: first max kept
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oq = __shfl_xor(bq, o, 64);
      if (ov > best || (ov == best && oq < bq)) { best = ov; bq = oq; }
    }
    if (bq != 0x7fffffff && best > 0.f) result = bq;
    if (cand_mask && lane == 0 && result >= 0)
      atomicOr(&cand_mask[(long)pair * mask_words + (result >> 5)], 1u << (result & 31));
  } else {
    const bool semi = mode == EMBNET_MINE_SEMIHARD;
    int total = 0;
    for (int qb = 0; qb < nneg; qb += 4 * 64) {            // four loads in flight per lane, ballots in column order
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int q = qb + 64 * u + lane; v[u] = q < nneg ? loss_at(q) : 0.f; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q0 = qb + 64 * u;
        if (q0 >= nneg) break;
        const bool pred = q0 + lane < nneg && v[u] > 0.f && (!semi || v[u] < margin);
        const unsigned long long m = __ballot(pred);
        total += __popcll(m);
        if (cand_mask && lane < 2 && q0 + 32 * lane < nneg)
          cand_mask[(long)pair * mask_words + (q0 >> 5) + lane] = (uint32_t)(m >> (32 * lane));
      }
    }
    if (total > 0) {
      const uint32_t u = rng_u32(seed, (uint64_t)pair, 0);
      const int want = (int)(((uint64_t)u * (uint64_t)total) >> 32);   // uniform in [0,total)
      int cum = 0;
      for (int qb = 0; qb < nneg && result < 0; qb += 4 * 64) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int q = qb + 64 * u + lane; v[u] = q < nneg ? loss_at(q) : 0.f; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int q0 = qb + 64 * u;
          if (q0 >= nneg || result >= 0) break;
          const bool pred = q0 + lane < nneg && v[u] > 0.f && (!semi || v[u] < margin);
          const unsigned long long m = __ballot(pred);
          const int cnt = __popcll(m);
          if (want < cum + cnt) {
            const int rank = __popcll(m & ((1ull << lane) - 1ull));
            const unsigned long long hit = __ballot(pred && rank == want - cum);
            result = q0 + __ffsll((long long)hit) - 1;
          }
          cum += cnt;
        }
      }
    }
  }
  if (lane == 0) selected[pair] = result < 0 ? -1 : (result < lo ? result : result + k);
}

// `hardest`, one wave per ANCHOR row: the pairs (i, j > i) of an anchor scan the same row of D, so the row is read once and
// every element is tested against the anchor's KP <= 7 positives (the per-pair kernel reads it once per positive: 1.5 x the
// matrix for K = 4).  Same arithmetic, same visiting order per lane, same tie rule per pair: bit-identical selections.
template <int KP>
__global__ __launch_bounds__(256) void mine_hardest_anchor_kernel(const float* __restrict__ D, int n, int k, float margin,
                                                                  int* __restrict__ selected, uint32_t* __restrict__ cand_mask,
                                                                  int mask_words) {
  const int a = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (a >= n) return;
  const int c = a / k, ii = a - c * k, lo = c * k, nneg = n - k, npos = k - 1 - ii;
  if (npos <= 0) return;                                   // the class's last row anchors no pair
  const float* row = D + (long)a * n;
  float dap[KP], best[KP]; int bq[KP];
#pragma unroll
  for (int j = 0; j < KP; ++j) { dap[j] = j < npos ? row[lo + ii + 1 + j] : 0.f; best[j] = -INFINITY; bq[j] = 0x7fffffff; }
  for (int q0 = lane; q0 < nneg; q0 += 4 * 64) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int q = q0 + 64 * u; v[u] = q < nneg ? row[q < lo ? q : q + k] : INFINITY; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = q0 + 64 * u;
      if (q < nneg) {
#pragma unroll
        for (int j = 0; j < KP; ++j) {
          const float l = __fadd_rn(__fsub_rn(dap[j], v[u]), margin);
          if (l > best[j]) { best[j] = l; bq[j] = q; }     // ascending q per lane: first max kept
        }
      }
    }
  }
  const int ppc = k * (k - 1) / 2, pair0 = c * ppc + ii * (2 * k - ii - 1) / 2;
#pragma unroll
  for (int j = 0; j < KP; ++j) {
    if (j >= npos) break;
    float b = best[j]; int q = bq[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(b, o, 64);
      const int oq = __shfl_xor(q, o, 64);
      if (ov > b || (ov == b && oq < q)) { b = ov; q = oq; }
    }
    const int result = (q != 0x7fffffff && b > 0.f) ? q : -1;
    if (lane == 0) {
      const int pair = pair0 + j;
      selected[pair] = result < 0 ? -1 : (result < lo ? result : result + k);
      if (cand_mask && result >= 0) atomicOr(&cand_mask[(long)pair * mask_words + (result >> 5)], 1u << (result & 31));
    }
  }
}

// The random rules (semihard: the reference's default; random_hard), one wave per ANCHOR row: two passes over the row — count the
// candidates of each of the anchor's KP <= 7 pairs, then find every pair's want-th candidate — instead of two passes per pair.
// Same predicates, same column order, same counter RNG (seed, pair): bit-identical selections and candidate masks.
template <int KP>
__global__ __launch_bounds__(256) void mine_random_anchor_kernel(const float* __restrict__ D, int n, int k, float margin, int semi,
                                                                 uint64_t seed, int* __restrict__ selected,
                                                                 uint32_t* __restrict__ cand_mask, int mask_words) {
  const int a = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (a >= n) return;
  const int c = a / k, ii = a - c * k, lo = c * k, nneg = n - k, npos = k - 1 - ii;
  if (npos <= 0) return;
  const float* row = D + (long)a * n;
  const int ppc = k * (k - 1) / 2, pair0 = c * ppc + ii * (2 * k - ii - 1) / 2;
  float dap[KP]; int total[KP];
#pragma unroll
  for (int j = 0; j < KP; ++j) { dap[j] = j < npos ? row[lo + ii + 1 + j] : 0.f; total[j] = 0; }
  auto pred_of = [&](float v, int j, bool in) -> bool {
    const float l = __fadd_rn(__fsub_rn(dap[j], v), margin);
    return in && l > 0.f && (!semi || l < margin);
  };
  for (int qb = 0; qb < nneg; qb += 4 * 64) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int q = qb + 64 * u + lane; v[u] = q < nneg ? row[q < lo ? q : q + k] : 0.f; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q0 = qb + 64 * u;
      if (q0 >= nneg) break;
#pragma unroll
      for (int j = 0; j < KP; ++j) {
        if (j >= npos) break;
        const unsigned long long m = __ballot(pred_of(v[u], j, q0 + lane < nneg));
        total[j] += __popcll(m);
        if (cand_mask && lane < 2 && q0 + 32 * lane < nneg)
          cand_mask[(long)(pair0 + j) * mask_words + (q0 >> 5) + lane] = (uint32_t)(m >> (32 * lane));
      }
    }
  }
  int want[KP], cum[KP], result[KP], open = 0;
#pragma unroll
  for (int j = 0; j < KP; ++j) {
    result[j] = -1; cum[j] = 0; want[j] = 0;
    if (j < npos && total[j] > 0) {
      const uint32_t u32 = rng_u32(seed, (uint64_t)(pair0 + j), 0);
      want[j] = (int)(((uint64_t)u32 * (uint64_t)total[j]) >> 32);      // uniform in [0,total)
      ++open;
    }
  }
  for (int qb = 0; qb < nneg && open > 0; qb += 4 * 64) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int q = qb + 64 * u + lane; v[u] = q < nneg ? row[q < lo ? q : q + k] : 0.f; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q0 = qb + 64 * u;
      if (q0 >= nneg) break;
#pragma unroll
      for (int j = 0; j < KP; ++j) {
        if (j >= npos || total[j] == 0 || result[j] >= 0) continue;
        const bool pred = pred_of(v[u], j, q0 + lane < nneg);
        const unsigned long long m = __ballot(pred);
        const int cnt = __popcll(m);
        if (want[j] < cum[j] + cnt) {
          const int rank = __popcll(m & ((1ull << lane) - 1ull));
          const unsigned long long hit = __ballot(pred && rank == want[j] - cum[j]);
          result[j] = q0 + __ffsll((long long)hit) - 1;
          --open;
        }
        cum[j] += cnt;
      }
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < KP; ++j)
      if (j < npos) selected[pair0 + j] = result[j] < 0 ? -1 : (result[j] < lo ? result[j] : result[j] + k);
  }
}

// Single workgroup: stable compaction of the active pairs into triplets[T][3], T -> *count.
__global__ __launch_bounds__(1024) void mine_compact_kernel(const int* __restrict__ selected, int n, int p,
                                                            int k, int* __restrict__ triplets,
                                                            int* __restrict__ count) {
  __shared__ int wave_tot[16];
  __shared__ int base_s;
  const int ppc = k * (k - 1) / 2, npairs = p * ppc;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int q0 = 0; q0 < npairs; q0 += 1024) {
    const int pair = q0 + tid;
    const int sel = pair < npairs ? selected[pair] : -1;
    const bool act = sel >= 0;
    const unsigned long long m = __ballot(act);
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(m);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wave; ++w) off += wave_tot[w];
    if (act) {
      int ii, jj;
      decode_pair(pair % ppc, k, ii, jj);
      const int lo = (pair / ppc) * k;
      int* t = triplets + 3 * (long)(off + rank);
      t[0] = lo + ii; t[1] = lo + jj; t[2] = sel;
    }
    __syncthreads();
    if (tid == 0) { int s = 0; for (int w = 0; w < 16; ++w) s += wave_tot[w]; base_s += s; }
    __syncthreads();
  }
  if (tid == 0) {
    int total = base_s;
    if (total == 0) {                 // datagenerators.py:246-250 — last pair of the last class, first negative
      triplets[0] = n - 2; triplets[1] = n - 1; triplets[2] = 0;
      total = 1;
    }
    *count = total;
  }
}

// Hermans batch-hard (build-defined, not in the reference): one wave per anchor.
__global__ __launch_bounds__(256) void batch_hard_kernel(const float* __restrict__ D, int n, int k,
                                                         int* __restrict__ triplets, int* __restrict__ count) {
  const int a = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (a == 0 && lane == 0 && count) *count = n;
  if (a >= n) return;
  const int lo = (a / k) * k, hi = lo + k;
  const float* row = D + (long)a * n;
  float bp = -INFINITY, bn = INFINITY; int ip = 0x7fffffff, in_ = 0x7fffffff;
  for (int c0 = lane; c0 < n; c0 += 4 * 64) {              // four loads in flight per lane; columns visited in the same order
    float vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) vv[u] = c0 + 64 * u < n ? row[c0 + 64 * u] : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + 64 * u;
      if (c >= n) break;
      const float v = vv[u];
      if (c >= lo && c < hi) { if (c != a && v > bp) { bp = v; ip = c; } }
      else if (v < bn) { bn = v; in_ = c; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bp, o, 64); const int oi = __shfl_xor(ip, o, 64);
    if (ov > bp || (ov == bp && oi < ip)) { bp = ov; ip = oi; }
    const float nv = __shfl_xor(bn, o, 64); const int ni = __shfl_xor(in_, o, 64);
    if (nv < bn || (nv == bn && ni < in_)) { bn = nv; in_ = ni; }
  }
  if (lane == 0) { int* t = triplets + 3 * (long)a; t[0] = a; t[1] = ip; t[2] = in_; }
}

}  // namespace embnet

using namespace embnet;

extern "C" int embnet_mine_max_triplets(int p, int k) {
  if (p <= 0 || k <= 0) return 0;
  const long t = (long)p * k * (k - 1) / 2;
  return (int)(t > 0 ? t : 1);
}

extern "C" int embnet_mine_triplets(const float* dist, int p, int k, float margin, int mode,
                                    uint64_t seed, int32_t* triplets, int32_t* count,
                                    int32_t* selected, uint32_t* cand_mask, void* stream) {
  EMBNET_CHECK_ARG(dist && triplets && count && selected, "mine_triplets: null pointer");
  EMBNET_CHECK_ARG(p >= 2 && k >= 2, "mine_triplets: need k_classes>=2 and k_samples>=2 (got %d,%d)", p, k);
  EMBNET_CHECK_ARG(mode == EMBNET_MINE_SEMIHARD || mode == EMBNET_MINE_HARDEST || mode == EMBNET_MINE_RANDOM_HARD,
                   "mine_triplets: unknown mode %d", mode);
  hipStream_t s = (hipStream_t)stream;
  const int n = p * k, npairs = p * (k * (k - 1) / 2);
  const int mask_words = (n - k + 31) / 32;
  if (cand_mask) {      // (a kernel, not hipMemsetAsync: a memset node did not zero its target when a captured HIP graph was replayed — DESIGN 3.14)
    const long words = (long)npairs * mask_words;
    zero_words_kernel<<<(int)((words + 255) / 256 < 1024 ? (words + 255) / 256 : 1024), 256, 0, s>>>(cand_mask, words);
  }
  const int by_anchor = (int)env_long("EMBNET_MINE_BY_ANCHOR", 1);      // read per call: tests compare the two forms in one process
  if (mode == EMBNET_MINE_HARDEST && by_anchor && k - 1 <= 7 && n >= 1024) {
    // (below ~1 000 rows the launch is latency-sized either way and the per-pair kernel has 1.5 x the waves)
    EMBNET_TRACE("embnet::mine_hardest_anchor_kernel", TRACE_BYTES, 4.0 * n * n, s);
    const int grid = cdiv(n, 4);
    if (k - 1 <= 1) mine_hardest_anchor_kernel<1><<<grid, 256, 0, s>>>(dist, n, k, margin, selected, cand_mask, mask_words);
    else if (k - 1 <= 3) mine_hardest_anchor_kernel<3><<<grid, 256, 0, s>>>(dist, n, k, margin, selected, cand_mask, mask_words);
    else mine_hardest_anchor_kernel<7><<<grid, 256, 0, s>>>(dist, n, k, margin, selected, cand_mask, mask_words);
  } else if (by_anchor && k - 1 <= 7 && n >= 1024) {
    EMBNET_TRACE("embnet::mine_random_anchor_kernel", TRACE_BYTES, 8.0 * n * n, s);
    const int grid = cdiv(n, 4), semi = mode == EMBNET_MINE_SEMIHARD ? 1 : 0;
    if (k - 1 <= 1) mine_random_anchor_kernel<1><<<grid, 256, 0, s>>>(dist, n, k, margin, semi, seed, selected, cand_mask, mask_words);
    else if (k - 1 <= 3) mine_random_anchor_kernel<3><<<grid, 256, 0, s>>>(dist, n, k, margin, semi, seed, selected, cand_mask, mask_words);
    else mine_random_anchor_kernel<7><<<grid, 256, 0, s>>>(dist, n, k, margin, semi, seed, selected, cand_mask, mask_words);
  } else {
    EMBNET_TRACE("embnet::mine_select_kernel", TRACE_BYTES, 0.0, s);
    mine_select_kernel<<<cdiv(npairs, 4), 256, 0, s>>>(dist, n, p, k, margin, mode, seed, selected, cand_mask, mask_words);
  }
  EMBNET_TRACE("embnet::mine_compact_kernel", TRACE_BYTES, 0.0, s);
  mine_compact_kernel<<<1, 1024, 0, s>>>(selected, n, p, k, triplets, count);
  return check_launch("mine_triplets");
}

extern "C" int embnet_batch_hard(const float* dist, int p, int k, int32_t* triplets, int32_t* count,
                                 void* stream) {
  EMBNET_CHECK_ARG(dist && triplets, "batch_hard: null pointer");
  EMBNET_CHECK_ARG(p >= 2 && k >= 2, "batch_hard: need k_classes>=2 and k_samples>=2 (got %d,%d)", p, k);
  const int n = p * k;
  EMBNET_TRACE("embnet::batch_hard_kernel", TRACE_BYTES, 0.0, (hipStream_t)stream);
  batch_hard_kernel<<<cdiv(n, 4), 256, 0, (hipStream_t)stream>>>(dist, n, k, triplets, count);
  return check_launch("batch_hard");
}
