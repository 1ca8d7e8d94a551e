// The whole loss path of the fused training step in ONE launch, for the batch sizes the reference trains at
// (N = P*K <= 512: configs C1..C5 are 32..256).  What it stands in for, per step, in /root/reference/embedding_net:
//   datagenerators.py:219      pairwise_distances(all_embeddings)           (sklearn arithmetic, f32 result)
//   datagenerators.py:225-250  the mining loop + selection rules :188-199   (+ the fallback triplet :246-250)
//   losses_and_accuracies.py:26-42 triplet_loss on the mined rows, Keras' mean over the T triplets
// and, in this repo, six launches: row norms, distance GEMM, mine-select, compaction, gathered hinge, mean.  At these
// sizes everything is launch latency ("report us and fused-launch count"), so the fusion is about launches:
//
//   grid = P workgroups, one per class (4 waves):
//     A  the class's K anchor rows -> LDS; every wave walks rows of the block (coalesced, L2-resident) and produces the
//        K distances anchor->row: sqrt(max(|x|^2 + |y|^2 - 2 x.y, 0)), 0 on the diagonal  -> D_c[K][N] in LDS
//     B  one wave per ordered positive pair (i<j) of the class: loss_q = (D[i,j] - D[i,neg_q]) + margin over the
//        out-of-class columns, selection rule (first arg-max / ballot rank-select with the same counter RNG as
//        mine_select_kernel) -> selected[pair];  EMBNET_MINE_BATCH_HARD: one wave per anchor instead
//     C  hinge of the selected triplet in the loss's own arithmetic: max(sum (a-p)^2 - sum (a-n)^2 + m, 0)
//   the LAST workgroup to finish (agent-scope ticket) compacts the selections in pair order into triplets[T][3],
//   applies the reference's fallback, and reduces the mean in a fixed order (bitwise reproducible).
// Outputs are exactly those of embnet_mine_triplets + embnet_triplet_gather_fwd, so the backward kernel is shared.
#include "common.h"
#include "../../include/embnet.h"

namespace embnet {

constexpr int FUSED_MAX_N = 512;
constexpr int FUSED_LDS_FLOATS = 16 * 1024;              // 64 KiB: K*(E + N) floats

__device__ __forceinline__ void pair_of(int q, int k, int& ii, int& jj) {      // q-th pair of combinations(range(k), 2)
  int i = 0, rem = q;
  while (rem >= k - 1 - i) { rem -= k - 1 - i; ++i; }
  ii = i; jj = i + 1 + rem;
}

// hinge of (a, p, n): a and p from the class's rows in LDS, n from the block in memory — the arithmetic of
// triplet_gather_fwd_kernel (lanes stride the columns, fmaf chains, butterfly sums)
__device__ __forceinline__ float hinge_term(const float* a, const float* p, const float* n, int e, int lane, float margin) {
  float pos = 0.f, neg = 0.f;
  for (int c = lane; c < e; c += 64) {
    const float av = a[c], dp = av - p[c], dn = av - n[c];
    pos = fmaf(dp, dp, pos); neg = fmaf(dn, dn, neg);
  }
  pos = wave_sum(pos); neg = wave_sum(neg);
  return pos - neg + margin;
}

struct FusedLossParams {
  const float* emb; int n, p, k, e; float margin; int mode; uint64_t seed; const uint64_t* seed_dev;
  int* triplets; int* count; int* selected; float* loss; float* active; float* mean;
  int* ticket; float* pair_term;            // workspace: [1] arrival counter (zero between launches), [slots] hinge terms
  int max_t;
};

constexpr int FUSED_THREADS = 1024;                      // 16 waves: the distance phase is a chain of L2 round trips per wave

__global__ __launch_bounds__(FUSED_THREADS) void fused_triplet_loss_fwd_kernel(FusedLossParams q) {
  __shared__ __attribute__((aligned(16))) float lds[FUSED_LDS_FLOATS];
  __shared__ float norm_a[16];
  __shared__ int s_last;
  const int n = q.n, k = q.k, e = q.e, c = blockIdx.x, lo = c * k;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = FUSED_THREADS / 64;
  float* A = lds;                                        // [k][e] the class's rows
  float* D = lds + k * e;                                // [k][n] distances anchor -> row
  for (int i = tid; i < k * e; i += FUSED_THREADS) A[i] = q.emb[(long)lo * e + i];
  __syncthreads();
  if (wave == 0)
    for (int a = 0; a < k; ++a) {
      float s = 0.f;
      for (int cc = lane; cc < e; cc += 64) s = fmaf(A[a * e + cc], A[a * e + cc], s);
      s = wave_sum(s);
      if (lane == 0) norm_a[a] = s;
    }
  __syncthreads();
  // A: distances.  Row r of the block against the K anchors; the row is read once, 16 anchors per sweep at most.
  // (the first version walked a row 64 columns at a time, one dependent L2 round trip each: 128 of them per wave at
  // N = 128, E = 256 = 124 us for the launch; eight loads in flight per lane and 16 waves leave 8 round trips)
  for (int r = wave; r < n; r += NW) {
    const float* y = q.emb + (long)r * e;
    float dot[16], ny = 0.f;
#pragma unroll
    for (int a = 0; a < 16; ++a) dot[a] = 0.f;
    for (int c0 = 0; c0 < e; c0 += 512) {                // same per-lane column order as before: results unchanged
      float yv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int cc = c0 + lane + 64 * j; yv[j] = cc < e ? y[cc] : 0.f; }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int cc = c0 + lane + 64 * j;
        if (cc < e) {
          ny = fmaf(yv[j], yv[j], ny);
#pragma unroll
          for (int a = 0; a < 16; ++a) if (a < k) dot[a] = fmaf(A[a * e + cc], yv[j], dot[a]);
        }
      }
    }
    ny = wave_sum(ny);
#pragma unroll
    for (int a = 0; a < 16; ++a) {
      if (a < k) {
        const float g = wave_sum(dot[a]);
        if (lane == 0) {
          float d2 = fmaxf(norm_a[a] + ny - 2.f * g, 0.f);
          if (r == lo + a) d2 = 0.f;
          D[a * n + r] = sqrtf(d2);
        }
      }
    }
  }
  __syncthreads();

  const int ppc = k * (k - 1) / 2;
  const int nneg = n - k;
  if (q.mode == EMBNET_MINE_BATCH_HARD) {
    // Hermans batch-hard (build-defined): per anchor the farthest positive and the closest negative; slot = anchor
    for (int a = wave; a < k; a += NW) {
      const float* row = D + a * n;
      float bp = -INFINITY, bn = INFINITY; int ip = 0x7fffffff, in_ = 0x7fffffff;
      for (int col = lane; col < n; col += 64) {
        const float v = row[col];
        if (col >= lo && col < lo + k) { if (col != lo + a && v > bp) { bp = v; ip = col; } }
        else if (v < bn) { bn = v; in_ = col; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bp, o, 64); const int oi = __shfl_xor(ip, o, 64);
        if (ov > bp || (ov == bp && oi < ip)) { bp = ov; ip = oi; }
        const float nv = __shfl_xor(bn, o, 64); const int ni = __shfl_xor(in_, o, 64);
        if (nv < bn || (nv == bn && ni < in_)) { bn = nv; in_ = ni; }
      }
      const float b = hinge_term(A + a * e, A + (ip - lo) * e, q.emb + (long)in_ * e, e, lane, q.margin);
      if (lane == 0) {
        int* t = q.triplets + 3 * (long)(lo + a);
        t[0] = lo + a; t[1] = ip; t[2] = in_;
        q.loss[lo + a] = fmaxf(b, 0.f); q.active[lo + a] = b >= 0.f ? 1.f : 0.f;
      }
    }
  } else {
    for (int pr = wave; pr < ppc; pr += NW) {
      int ii, jj;
      pair_of(pr, k, ii, jj);
      const float* row = D + ii * n;
      const float dap = row[lo + jj];
      auto loss_at = [&](int x) -> float {
        const int col = x < lo ? x : x + k;
        return __fadd_rn(__fsub_rn(dap, row[col]), q.margin);
      };
      const int pair = c * ppc + pr;
      int result = -1;
      if (q.mode == EMBNET_MINE_HARDEST) {
        float best = -INFINITY; int bq = 0x7fffffff;
        for (int x = lane; x < nneg; x += 64) {
          const float v = loss_at(x);
          if (v > best) { best = v; bq = x; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const float ov = __shfl_xor(best, o, 64);
          const int oq = __shfl_xor(bq, o, 64);
          if (ov > best || (ov == best && oq < bq)) { best = ov; bq = oq; }
        }
        if (bq != 0x7fffffff && best > 0.f) result = bq;
      } else {
        const bool semi = q.mode == EMBNET_MINE_SEMIHARD;
        int total = 0;
        for (int x0 = 0; x0 < nneg; x0 += 64) {
          const int x = x0 + lane;
          bool pred = false;
          if (x < nneg) { const float v = loss_at(x); pred = v > 0.f && (!semi || v < q.margin); }
          total += __popcll(__ballot(pred));
        }
        if (total > 0) {
          const uint32_t u = rng_u32(q.seed_dev ? *q.seed_dev : q.seed, (uint64_t)pair, 0);
          const int want = (int)(((uint64_t)u * (uint64_t)total) >> 32);
          int cum = 0;
          for (int x0 = 0; x0 < nneg; x0 += 64) {
            const int x = x0 + lane;
            bool pred = false;
            if (x < nneg) { const float v = loss_at(x); pred = v > 0.f && (!semi || v < q.margin); }
            const unsigned long long m = __ballot(pred);
            const int cnt = __popcll(m);
            if (want < cum + cnt) {
              const int rank = __popcll(m & ((1ull << lane) - 1ull));
              const unsigned long long hit = __ballot(pred && rank == want - cum);
              result = x0 + __ffsll((long long)hit) - 1;
              break;
            }
            cum += cnt;
          }
        }
      }
      const int sel = result < 0 ? -1 : (result < lo ? result : result + k);
      float b = 0.f;
      if (sel >= 0) b = hinge_term(A + ii * e, A + jj * e, q.emb + (long)sel * e, e, lane, q.margin);
      if (lane == 0) { q.selected[pair] = sel; q.pair_term[pair] = b; }
    }
  }

  // ---- last workgroup: compaction, fallback, mean -------------------------------------------------------------
  // hand-off as MI355X_MICROARCH.md prescribes: every storing wave drains its stores, barrier, one lane releases at
  // agent scope, drains, then takes the ticket; the last arriver acquires, drains, and a barrier lets its waves read.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int t = atomicAdd(q.ticket, 1);
    s_last = t == (int)gridDim.x - 1;
    if (s_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      *q.ticket = 0;                                     // re-arm the counter for the next launch
    }
  }
  __syncthreads();
  if (!s_last) return;
  // the reductions keep the 256-thread order of the kernels they stand in for (mine_compact_kernel, mean_first_kernel):
  // waves 4..15 only take part in the barriers
  __shared__ float part[NW];
  __shared__ int wave_tot[NW];
  __shared__ int base_s;
  const bool low = tid < 256;
  if (q.mode == EMBNET_MINE_BATCH_HARD) {
    float s = 0.f;
    if (low) for (int i = tid; i < n; i += 256) s += __hip_atomic_load(&q.loss[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s = wave_sum(s);
    if (lane == 0) part[wave] = s;
    __syncthreads();
    if (tid == 0) { *q.count = n; *q.mean = (part[0] + part[1] + part[2] + part[3]) / (float)n; }
    return;
  }
  const int npairs = q.p * ppc;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int p0 = 0; p0 < npairs; p0 += 256) {             // stable compaction in pair order (mine_compact_kernel)
    const int pair = p0 + tid;
    const int sel = (low && pair < npairs) ? __hip_atomic_load(&q.selected[pair], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
    const bool act = sel >= 0;
    const unsigned long long m = __ballot(act);
    if (lane == 0) wave_tot[wave] = __popcll(m);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wave; ++w) off += wave_tot[w];
    if (act) {
      int ii, jj;
      pair_of(pair % ppc, k, ii, jj);
      const int slot = off + __popcll(m & ((1ull << lane) - 1ull));
      const float b = __hip_atomic_load(&q.pair_term[pair], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int* t = q.triplets + 3 * (long)slot;
      t[0] = (pair / ppc) * k + ii; t[1] = (pair / ppc) * k + jj; t[2] = sel;
      q.loss[slot] = fmaxf(b, 0.f); q.active[slot] = b >= 0.f ? 1.f : 0.f;
    }
    __syncthreads();
    if (tid == 0) base_s += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
  }
  int total = base_s;
  if (total == 0) {                                      // datagenerators.py:246-250: last pair of the last class, first negative
    if (wave == 0) {
      const float b = hinge_term(q.emb + (long)(n - 2) * e, q.emb + (long)(n - 1) * e, q.emb, e, lane, q.margin);
      if (lane == 0) {
        q.triplets[0] = n - 2; q.triplets[1] = n - 1; q.triplets[2] = 0;
        q.loss[0] = fmaxf(b, 0.f); q.active[0] = b >= 0.f ? 1.f : 0.f;
      }
    }
    total = 1;
  }
  __syncthreads();
  for (int t = total + tid; t < q.max_t; t += FUSED_THREADS) { q.loss[t] = 0.f; q.active[t] = 0.f; }
  float s = 0.f;                                         // mean over the live triplets, mean_first_kernel's order
  if (low) for (int t = tid; t < total; t += 256) s += q.loss[t];
  s = wave_sum(s);
  if (lane == 0) part[wave] = s;
  __syncthreads();
  if (tid == 0) { *q.count = total; *q.mean = (part[0] + part[1] + part[2] + part[3]) / (float)total; }
}

}  // namespace embnet

using namespace embnet;

extern "C" int embnet_fused_loss_supported(int p, int k, int e) {
  if (p < 2 || k < 2 || k > 16 || e <= 0) return 0;
  const long n = (long)p * k;
  return n <= FUSED_MAX_N && (long)k * (e + n) <= FUSED_LDS_FLOATS;
}

extern "C" size_t embnet_fused_loss_workspace_bytes(int p, int k) {
  if (p <= 0 || k <= 0) return 0;
  return (size_t)(4 + embnet_mine_max_triplets(p, k)) * sizeof(float);
}

extern "C" int embnet_fused_triplet_loss_fwd(const float* emb, int p, int k, int e, float margin, int mode,
                                             uint64_t seed, const uint64_t* seed_dev, int32_t* triplets, int32_t* count,
                                             int32_t* selected,
                                             float* loss, float* active, float* mean_loss, void* workspace,
                                             size_t workspace_bytes, void* stream) {
  EMBNET_CHECK_ARG(emb && triplets && count && selected && loss && active && mean_loss && workspace,
                   "fused_triplet_loss_fwd: null pointer");
  EMBNET_CHECK_ARG(mode >= EMBNET_MINE_SEMIHARD && mode <= EMBNET_MINE_BATCH_HARD, "fused_triplet_loss_fwd: unknown mode %d", mode);
  EMBNET_CHECK_ARG(embnet_fused_loss_supported(p, k, e),
                   "fused_triplet_loss_fwd: p=%d k=%d e=%d outside the fused path (see embnet_fused_loss_supported)", p, k, e);
  if (workspace_bytes < embnet_fused_loss_workspace_bytes(p, k))
    return fail(EMBNET_EWORKSPACE, "fused_triplet_loss_fwd: workspace %zu < %zu bytes", workspace_bytes,
                embnet_fused_loss_workspace_bytes(p, k));
  const int n = p * k;
  FusedLossParams q{emb, n, p, k, e, margin, mode, seed, seed_dev, triplets, count, selected, loss, active, mean_loss,
                    (int*)workspace, (float*)workspace + 4,
                    mode == EMBNET_MINE_BATCH_HARD ? n : embnet_mine_max_triplets(p, k)};
  EMBNET_TRACE("embnet::fused_triplet_loss_fwd_kernel", TRACE_BYTES, 4.0 * n * e * (p + 1.0), stream);
  fused_triplet_loss_fwd_kernel<<<p, FUSED_THREADS, 0, (hipStream_t)stream>>>(q);
  return check_launch("fused_triplet_loss_fwd");
}
