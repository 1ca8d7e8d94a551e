// Margin losses, their gradients, and the embedding-space elementwise heads.
//
// Reference functions replaced (/root/reference/embedding_net/):
//   losses_and_accuracies.py:14-44  triplet_loss      -> triplet_hinge_{fwd,bwd} ([T,3E] API form)
//                                                        triplet_gather_{fwd,bwd}   (fused-step form:
//                                                        rows gathered from the [N,E] block by (a,p,n))
//   losses_and_accuracies.py:4-11   contrastive_loss  -> contrastive_{fwd,bwd}
//   losses_and_accuracies.py:47-50  accuracy          -> accuracy
//   backbones.py:38,77,118          K.l2_normalize    -> l2norm_{fwd,bwd}
//   models.py:225                   siamese L2 head   -> pair_distance_{fwd,bwd}
// All HBM/L2-bound: one wavefront per row, coalesced reads, shuffle reductions.
#include "common.h"
#include "../../include/embnet.h"

namespace embnet {

// ---- triplet hinge, [T,3E] form -------------------------------------------------
__global__ __launch_bounds__(256) void triplet_hinge_fwd_kernel(const float* __restrict__ y, int t, int e,
                                                                float margin, float* __restrict__ loss) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= t) return;
  const float* a = y + (long)row * 3 * e; const float* p = a + e; const float* n = p + e;
  float pos = 0.f, neg = 0.f;
  for (int c = lane; c < e; c += 64) {
    const float av = a[c], dp = av - p[c], dn = av - n[c];
    pos = fmaf(dp, dp, pos); neg = fmaf(dn, dn, neg);
  }
  pos = wave_sum(pos); neg = wave_sum(neg);
  if (lane == 0) loss[row] = fmaxf(pos - neg + margin, 0.f);
}

// dy[T,3E] = dloss[t] * [2(n-p), 2(p-a), 2(a-n)] on rows with pos-neg+margin >= 0 (TF maximum tie rule)
__global__ __launch_bounds__(256) void triplet_hinge_bwd_kernel(const float* __restrict__ y,
                                                                const float* __restrict__ dloss, int t, int e,
                                                                float margin, float* __restrict__ dy) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= t) return;
  const float* a = y + (long)row * 3 * e; const float* p = a + e; const float* n = p + e;
  float* da = dy + (long)row * 3 * e; float* dp_ = da + e; float* dn_ = dp_ + e;
  float pos = 0.f, neg = 0.f;
  for (int c = lane; c < e; c += 64) {
    const float av = a[c], dp = av - p[c], dn = av - n[c];
    pos = fmaf(dp, dp, pos); neg = fmaf(dn, dn, neg);
  }
  pos = wave_sum(pos); neg = wave_sum(neg);
  const float g = (pos - neg + margin >= 0.f) ? 2.f * dloss[row] : 0.f;
  for (int c = lane; c < e; c += 64) {
    const float av = a[c], pv = p[c], nv = n[c];
    da[c] = g * (nv - pv); dp_[c] = g * (pv - av); dn_[c] = g * (av - nv);
  }
}

// ---- triplet hinge, gather form (fused train step) ---------------------------------
// loss[t] for t < *count, 0 beyond; act[t] = 1 where the hinge passes gradient.
__global__ __launch_bounds__(256) void triplet_gather_fwd_kernel(const float* __restrict__ emb, int e,
                                                                 const int* __restrict__ trip,
                                                                 const int* __restrict__ count, int max_t,
                                                                 float margin, float* __restrict__ loss,
                                                                 float* __restrict__ act) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (t >= max_t) return;
  if (t >= *count) { if (lane == 0) { loss[t] = 0.f; act[t] = 0.f; } return; }
  const float* a = emb + (long)trip[3 * t] * e;
  const float* p = emb + (long)trip[3 * t + 1] * e;
  const float* n = emb + (long)trip[3 * t + 2] * e;
  float pos = 0.f, neg = 0.f;
  for (int c = lane; c < e; c += 64) {
    const float av = a[c], dp = av - p[c], dn = av - n[c];
    pos = fmaf(dp, dp, pos); neg = fmaf(dn, dn, neg);
  }
  pos = wave_sum(pos); neg = wave_sum(neg);
  if (lane == 0) {
    const float b = pos - neg + margin;
    loss[t] = fmaxf(b, 0.f); act[t] = b >= 0.f ? 1.f : 0.f;
  }
}

// mean over the first *count entries (Keras averages the per-triplet losses); single workgroup,
// fixed summation order -> bitwise reproducible.
__global__ __launch_bounds__(256) void mean_first_kernel(const float* __restrict__ v,
                                                         const int* __restrict__ count, int max_t,
                                                         float* __restrict__ out) {
  __shared__ float part[4];
  const int cnt = count ? min(*count, max_t) : max_t;
  float s = 0.f;
  for (int i = threadIdx.x; i < cnt; i += 256) s += v[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = (part[0] + part[1] + part[2] + part[3]) / (float)max(cnt, 1);
}

// demb[r] = (upstream / T) * sum over triplets containing r of the hinge gradient.  One workgroup
// per embedding row: the 256 threads scan the triplet list 256 at a time, the (few) triplets that
// touch row r are compacted IN ORDER into LDS by ballot/popcount, then every thread accumulates its
// columns over those matches.  No atomics; the summation order is the triplet order -> reproducible.
// Work per workgroup: T/256 index checks per thread + matches * E/256 FMAs (was T per thread).
__global__ __launch_bounds__(256) void triplet_gather_bwd_kernel(const float* __restrict__ emb, int n_rows, int e,
                                                                 const int* __restrict__ trip,
                                                                 const int* __restrict__ count, int max_t,
                                                                 const float* __restrict__ act,
                                                                 const float* __restrict__ upstream,
                                                                 float* __restrict__ demb) {
  __shared__ int s_match[256 * 3];
  __shared__ int s_wave[4];
  const int r = blockIdx.x;
  const int cnt = min(*count, max_t);
  const float g = 2.f * (upstream ? *upstream : 1.f) / (float)max(cnt, 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // each thread owns columns c = c0 + tid, + 256, ... of this workgroup's block of 4096 columns (16 live accumulators;
  // blockIdx.y walks the blocks when E > 4096: the reference's default encodings_len is 4096, backbones.py:13)
  const int c0 = blockIdx.y * 4096;
  float accv[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) accv[i] = 0.f;
  for (int t0 = 0; t0 < cnt; t0 += 256) {
    const int t = t0 + threadIdx.x;
    int a = -1, p = -1, n = -1;
    bool hit = false;
    if (t < cnt) {
      a = trip[3 * t]; p = trip[3 * t + 1]; n = trip[3 * t + 2];
      hit = (a == r || p == r || n == r) && act[t] != 0.f;
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += s_wave[w];
    const int total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    if (hit) {
      const int slot = off + __popcll(m & ((1ull << lane) - 1ull));
      s_match[3 * slot] = a; s_match[3 * slot + 1] = p; s_match[3 * slot + 2] = n;
    }
    __syncthreads();
    for (int k = 0; k < total; ++k) {
      const int ma = s_match[3 * k], mp = s_match[3 * k + 1], mn = s_match[3 * k + 2];
      const float* ea = emb + (long)ma * e; const float* ep = emb + (long)mp * e; const float* en = emb + (long)mn * e;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c = c0 + threadIdx.x + 256 * i;
        if (c < e) {
          float d = 0.f;
          if (ma == r) d += en[c] - ep[c];
          if (mp == r) d += ep[c] - ea[c];
          if (mn == r) d += ea[c] - en[c];
          accv[i] += d;
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + threadIdx.x + 256 * i;
    if (c < e) demb[(long)r * e + c] = g * accv[i];
  }
  (void)n_rows;
}

// ---- contrastive / accuracy ----------------------------------------------------------
__global__ __launch_bounds__(256) void contrastive_fwd_kernel(const float* __restrict__ y,
                                                              const float* __restrict__ d, int b,
                                                              float* __restrict__ out) {
  __shared__ float part[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < b; i += 256) {
    const float dv = d[i], yv = y[i], mg = fmaxf(1.f - dv, 0.f);
    s += yv * (dv * dv) + (1.f - yv) * (mg * mg);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = (part[0] + part[1] + part[2] + part[3]) / (float)b;
}

__global__ __launch_bounds__(256) void contrastive_bwd_kernel(const float* __restrict__ y,
                                                              const float* __restrict__ d, int b,
                                                              const float* __restrict__ upstream,
                                                              float* __restrict__ dd) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= b) return;
  const float g = (upstream ? *upstream : 1.f) / (float)b;
  const float dv = d[i], yv = y[i];
  dd[i] = g * (2.f * yv * dv - 2.f * (1.f - yv) * fmaxf(1.f - dv, 0.f));
}

__global__ __launch_bounds__(256) void accuracy_kernel(const float* __restrict__ y, const float* __restrict__ d,
                                                       int b, float* __restrict__ out) {
  __shared__ int part[4];
  int s = 0;
  for (int i = threadIdx.x; i < b; i += 256) s += (y[i] == (d[i] < 0.5f ? 1.f : 0.f)) ? 1 : 0;
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = (float)(part[0] + part[1] + part[2] + part[3]) / (float)b;
}

// ---- row L2 normalisation ---------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, int n, int e,
                                                         float* __restrict__ y, float* __restrict__ rnorm) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  const float* r = x + (long)row * e;
  float s = 0.f;
  for (int c = lane; c < e; c += 64) s = fmaf(r[c], r[c], s);
  s = wave_sum(s);
  const float inv = rsqrtf(fmaxf(s, 1e-12f));
  for (int c = lane; c < e; c += 64) y[(long)row * e + c] = r[c] * inv;
  if (lane == 0) rnorm[row] = s >= 1e-12f ? inv : -inv;     // sign flags the clamped branch for backward
}

__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ rnorm,
                                                         const float* __restrict__ dy, int n, int e,
                                                         float* __restrict__ dx) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  const float* yr = y + (long)row * e; const float* gr = dy + (long)row * e;
  const float inv = rnorm[row];
  if (inv < 0.f) {                        // sum x^2 below epsilon: y = x * 1e6, plain scale
    for (int c = lane; c < e; c += 64) dx[(long)row * e + c] = gr[c] * (-inv);
    return;
  }
  float dot = 0.f;
  for (int c = lane; c < e; c += 64) dot = fmaf(gr[c], yr[c], dot);
  dot = wave_sum(dot);
  for (int c = lane; c < e; c += 64) dx[(long)row * e + c] = (gr[c] - yr[c] * dot) * inv;
}

// ---- siamese L2 head: d = sqrt(max(sum (e1-e2)^2, 1e-7)) ----------------------------
__global__ __launch_bounds__(256) void pair_distance_fwd_kernel(const float* __restrict__ e1,
                                                                const float* __restrict__ e2, int b, int e,
                                                                float* __restrict__ d) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= b) return;
  float s = 0.f;
  for (int c = lane; c < e; c += 64) { const float v = e1[(long)row * e + c] - e2[(long)row * e + c]; s = fmaf(v, v, s); }
  s = wave_sum(s);
  if (lane == 0) d[row] = sqrtf(fmaxf(s, 1e-7f));
}

__global__ __launch_bounds__(256) void pair_distance_bwd_kernel(const float* __restrict__ e1,
                                                                const float* __restrict__ e2,
                                                                const float* __restrict__ d,
                                                                const float* __restrict__ dd, int b, int e,
                                                                float* __restrict__ de1, float* __restrict__ de2) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= b) return;
  const float dv = d[row];
  // below the epsilon clamp the maximum() passes no gradient to the sum
  const float g = (dv * dv > 1e-7f) ? dd[row] / dv : 0.f;
  for (int c = lane; c < e; c += 64) {
    const float v = g * (e1[(long)row * e + c] - e2[(long)row * e + c]);
    de1[(long)row * e + c] = v; de2[(long)row * e + c] = -v;
  }
}

}  // namespace embnet

using namespace embnet;
#define S(stream) ((hipStream_t)(stream))

extern "C" int embnet_triplet_hinge_fwd(const float* y_pred, int t, int e, float margin, float* loss, void* stream) {
  EMBNET_CHECK_ARG(y_pred && loss, "triplet_hinge_fwd: null pointer");
  EMBNET_CHECK_ARG(t > 0 && e > 0, "triplet_hinge_fwd: t=%d e=%d", t, e);
  EMBNET_TRACE("embnet::triplet_hinge_fwd_kernel", TRACE_BYTES, 0.0, S(stream));
  triplet_hinge_fwd_kernel<<<cdiv(t, 4), 256, 0, S(stream)>>>(y_pred, t, e, margin, loss);
  return check_launch("triplet_hinge_fwd");
}

extern "C" int embnet_triplet_hinge_bwd(const float* y_pred, const float* dloss, int t, int e, float margin,
                                        float* dy, void* stream) {
  EMBNET_CHECK_ARG(y_pred && dloss && dy, "triplet_hinge_bwd: null pointer");
  EMBNET_CHECK_ARG(t > 0 && e > 0, "triplet_hinge_bwd: t=%d e=%d", t, e);
  EMBNET_TRACE("embnet::triplet_hinge_bwd_kernel", TRACE_BYTES, 0.0, S(stream));
  triplet_hinge_bwd_kernel<<<cdiv(t, 4), 256, 0, S(stream)>>>(y_pred, dloss, t, e, margin, dy);
  return check_launch("triplet_hinge_bwd");
}

extern "C" int embnet_triplet_gather_fwd(const float* emb, int n, int e, const int32_t* triplets,
                                         const int32_t* count, int max_t, float margin, float* loss,
                                         float* active, float* mean_loss, void* stream) {
  EMBNET_CHECK_ARG(emb && triplets && count && loss && active && mean_loss, "triplet_gather_fwd: null pointer");
  EMBNET_CHECK_ARG(n > 0 && e > 0 && max_t > 0, "triplet_gather_fwd: n=%d e=%d max_t=%d", n, e, max_t);
  {
    EMBNET_TRACE("embnet::triplet_gather_fwd_kernel", TRACE_BYTES, 0.0, S(stream));
    triplet_gather_fwd_kernel<<<cdiv(max_t, 4), 256, 0, S(stream)>>>(emb, e, triplets, count, max_t, margin, loss, active);
  }
  EMBNET_TRACE("embnet::mean_first_kernel", TRACE_BYTES, 0.0, S(stream));
  mean_first_kernel<<<1, 256, 0, S(stream)>>>(loss, count, max_t, mean_loss);
  return check_launch("triplet_gather_fwd");
}

extern "C" int embnet_triplet_gather_bwd(const float* emb, int n, int e, const int32_t* triplets,
                                         const int32_t* count, int max_t, const float* active,
                                         const float* upstream, float* demb, void* stream) {
  EMBNET_CHECK_ARG(emb && triplets && count && active && demb, "triplet_gather_bwd: null pointer");
  EMBNET_CHECK_ARG(n > 0 && e > 0 && max_t > 0, "triplet_gather_bwd: n=%d e=%d max_t=%d", n, e, max_t);
  EMBNET_TRACE("embnet::triplet_gather_bwd_kernel", TRACE_BYTES, 8.0 * n * e, S(stream));
  triplet_gather_bwd_kernel<<<dim3(n, cdiv(e, 4096)), 256, 0, S(stream)>>>(emb, n, e, triplets, count, max_t, active, upstream, demb);
  return check_launch("triplet_gather_bwd");
}

extern "C" int embnet_contrastive_fwd(const float* y_true, const float* dist, int b, float* loss, void* stream) {
  EMBNET_CHECK_ARG(y_true && dist && loss, "contrastive_fwd: null pointer");
  EMBNET_CHECK_ARG(b > 0, "contrastive_fwd: b=%d", b);
  EMBNET_TRACE("embnet::contrastive_fwd_kernel", TRACE_BYTES, 0.0, S(stream));
  contrastive_fwd_kernel<<<1, 256, 0, S(stream)>>>(y_true, dist, b, loss);
  return check_launch("contrastive_fwd");
}

extern "C" int embnet_contrastive_bwd(const float* y_true, const float* dist, int b, const float* upstream,
                                      float* ddist, void* stream) {
  EMBNET_CHECK_ARG(y_true && dist && ddist, "contrastive_bwd: null pointer");
  EMBNET_CHECK_ARG(b > 0, "contrastive_bwd: b=%d", b);
  EMBNET_TRACE("embnet::contrastive_bwd_kernel", TRACE_BYTES, 0.0, S(stream));
  contrastive_bwd_kernel<<<cdiv(b, 256), 256, 0, S(stream)>>>(y_true, dist, b, upstream, ddist);
  return check_launch("contrastive_bwd");
}

extern "C" int embnet_accuracy(const float* y_true, const float* dist, int b, float* acc, void* stream) {
  EMBNET_CHECK_ARG(y_true && dist && acc, "accuracy: null pointer");
  EMBNET_CHECK_ARG(b > 0, "accuracy: b=%d", b);
  EMBNET_TRACE("embnet::accuracy_kernel", TRACE_BYTES, 0.0, S(stream));
  accuracy_kernel<<<1, 256, 0, S(stream)>>>(y_true, dist, b, acc);
  return check_launch("accuracy");
}

extern "C" int embnet_l2norm_fwd(const float* x, int n, int e, float* y, float* rnorm, void* stream) {
  EMBNET_CHECK_ARG(x && y && rnorm, "l2norm_fwd: null pointer");
  EMBNET_CHECK_ARG(n > 0 && e > 0, "l2norm_fwd: n=%d e=%d", n, e);
  EMBNET_TRACE("embnet::l2norm_fwd_kernel", TRACE_BYTES, 0.0, S(stream));
  l2norm_fwd_kernel<<<cdiv(n, 4), 256, 0, S(stream)>>>(x, n, e, y, rnorm);
  return check_launch("l2norm_fwd");
}

extern "C" int embnet_l2norm_bwd(const float* y, const float* rnorm, const float* dy, int n, int e, float* dx,
                                 void* stream) {
  EMBNET_CHECK_ARG(y && rnorm && dy && dx, "l2norm_bwd: null pointer");
  EMBNET_CHECK_ARG(n > 0 && e > 0, "l2norm_bwd: n=%d e=%d", n, e);
  EMBNET_TRACE("embnet::l2norm_bwd_kernel", TRACE_BYTES, 0.0, S(stream));
  l2norm_bwd_kernel<<<cdiv(n, 4), 256, 0, S(stream)>>>(y, rnorm, dy, n, e, dx);
  return check_launch("l2norm_bwd");
}

extern "C" int embnet_pair_distance_fwd(const float* e1, const float* e2, int b, int e, float* dist, void* stream) {
  EMBNET_CHECK_ARG(e1 && e2 && dist, "pair_distance_fwd: null pointer");
  EMBNET_CHECK_ARG(b > 0 && e > 0, "pair_distance_fwd: b=%d e=%d", b, e);
  EMBNET_TRACE("embnet::pair_distance_fwd_kernel", TRACE_BYTES, 0.0, S(stream));
  pair_distance_fwd_kernel<<<cdiv(b, 4), 256, 0, S(stream)>>>(e1, e2, b, e, dist);
  return check_launch("pair_distance_fwd");
}

extern "C" int embnet_pair_distance_bwd(const float* e1, const float* e2, const float* dist, const float* ddist,
                                        int b, int e, float* de1, float* de2, void* stream) {
  EMBNET_CHECK_ARG(e1 && e2 && dist && ddist && de1 && de2, "pair_distance_bwd: null pointer");
  EMBNET_CHECK_ARG(b > 0 && e > 0, "pair_distance_bwd: b=%d e=%d", b, e);
  EMBNET_TRACE("embnet::pair_distance_bwd_kernel", TRACE_BYTES, 0.0, S(stream));
  pair_distance_bwd_kernel<<<cdiv(b, 4), 256, 0, S(stream)>>>(e1, e2, dist, ddist, b, e, de1, de2);
  return check_launch("pair_distance_bwd");
}

// ---- softmax + categorical cross-entropy (backbones.py:146-151: Dense(n_classes, softmax) compiled with
// loss='categorical_crossentropy', metrics=['accuracy']; TF evaluates it from the logits of the softmax op).
// One wave per row: max, sum-exp, loss_row = -sum_c t_c * log_softmax_c; probabilities are kept for backward.
namespace embnet {

__global__ __launch_bounds__(256) void softmax_xent_fwd_kernel(const float* __restrict__ z, const float* __restrict__ t,
                                                               int b, int c, float* __restrict__ prob,
                                                               float* __restrict__ loss, float* __restrict__ correct) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= b) return;
  const float* zr = z + (long)row * c; const float* tr = t + (long)row * c;
  float mx = -INFINITY, tmx = -INFINITY; int zi = 0x7fffffff, ti = 0x7fffffff;
  for (int j = lane; j < c; j += 64) {
    if (zr[j] > mx) { mx = zr[j]; zi = j; }
    if (tr[j] > tmx) { tmx = tr[j]; ti = j; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(mx, o, 64); int oi = __shfl_xor(zi, o, 64);
    if (ov > mx || (ov == mx && oi < zi)) { mx = ov; zi = oi; }
    ov = __shfl_xor(tmx, o, 64); oi = __shfl_xor(ti, o, 64);
    if (ov > tmx || (ov == tmx && oi < ti)) { tmx = ov; ti = oi; }
  }
  float se = 0.f;
  for (int j = lane; j < c; j += 64) se += __expf(zr[j] - mx);
  se = wave_sum(se);
  const float lse = mx + __logf(se);
  float l = 0.f;
  for (int j = lane; j < c; j += 64) {
    prob[(long)row * c + j] = __expf(zr[j] - lse);
    l += tr[j] * (lse - zr[j]);
  }
  l = wave_sum(l);
  if (lane == 0) { loss[row] = l; correct[row] = zi == ti ? 1.f : 0.f; }
}

// dz = (*upstream / b) * (prob * sum_c t_c - t)
__global__ __launch_bounds__(256) void softmax_xent_bwd_kernel(const float* __restrict__ prob, const float* __restrict__ t,
                                                               int b, int c, const float* __restrict__ upstream,
                                                               float* __restrict__ dz) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= b) return;
  float ts = 0.f;
  for (int j = lane; j < c; j += 64) ts += t[(long)row * c + j];
  ts = wave_sum(ts);
  const float g = (upstream ? *upstream : 1.f) / (float)b;
  for (int j = lane; j < c; j += 64)
    dz[(long)row * c + j] = g * (prob[(long)row * c + j] * ts - t[(long)row * c + j]);
}

}  // namespace embnet

extern "C" int embnet_softmax_xent_fwd(const float* logits, const float* targets, int b, int c, float* prob,
                                       float* row_loss, float* row_correct, float* mean_loss, float* accuracy,
                                       void* stream) {
  EMBNET_CHECK_ARG(logits && targets && prob && row_loss && row_correct && mean_loss && accuracy,
                   "softmax_xent_fwd: null pointer");
  EMBNET_CHECK_ARG(b > 0 && c > 0, "softmax_xent_fwd: b=%d c=%d", b, c);
  {
    EMBNET_TRACE("embnet::softmax_xent_fwd_kernel", TRACE_BYTES, 0.0, S(stream));
    softmax_xent_fwd_kernel<<<cdiv(b, 4), 256, 0, S(stream)>>>(logits, targets, b, c, prob, row_loss, row_correct);
  }
  {
    EMBNET_TRACE("embnet::mean_first_kernel", TRACE_BYTES, 0.0, S(stream));
    mean_first_kernel<<<1, 256, 0, S(stream)>>>(row_loss, nullptr, b, mean_loss);
  }
  EMBNET_TRACE("embnet::mean_first_kernel", TRACE_BYTES, 0.0, S(stream));
  mean_first_kernel<<<1, 256, 0, S(stream)>>>(row_correct, nullptr, b, accuracy);
  return check_launch("softmax_xent_fwd");
}

extern "C" int embnet_softmax_xent_bwd(const float* prob, const float* targets, int b, int c, const float* upstream,
                                       float* dlogits, void* stream) {
  EMBNET_CHECK_ARG(prob && targets && dlogits, "softmax_xent_bwd: null pointer");
  EMBNET_CHECK_ARG(b > 0 && c > 0, "softmax_xent_bwd: b=%d c=%d", b, c);
  EMBNET_TRACE("embnet::softmax_xent_bwd_kernel", TRACE_BYTES, 0.0, S(stream));
  softmax_xent_bwd_kernel<<<cdiv(b, 4), 256, 0, S(stream)>>>(prob, targets, b, c, upstream, dlogits);
  return check_launch("softmax_xent_bwd");
}
