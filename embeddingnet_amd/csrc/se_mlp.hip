// The squeeze-and-excite gate of an MBConv block as ONE forward and TWO backward launches:
//     gate = sigmoid(swish(pooled W1 + b1) W2 + b2),   pooled [n, C], W1 [C, S], W2 [S, C], S = max(1, block input filters / 4)
// (the `efficientnet` zoo package's SE branch: two 1x1 convs with bias on the pooled [n,1,1,C] tensor, instantiated at
// /root/reference/embedding_net/backbones.py:84-98).  Composed from dense.hip's GEMM launches and the activation kernels it is
// twelve launches per block — dense forward / data gradient / weight gradient x 2, bias column sums x 2, activation forward /
// backward x 2 — of 6-16 us each for 2.5-28 MFLOP of work: 127 us per block, 2.0 ms of a 24.8 ms EfficientNet-B0 step at
// 64 x 4 images, all of it launch latency on one stream.  The arithmetic is fp32 FMA chains (no matrix instruction: the
// reductions are 4-160 or C long and the matrices 256 rows).
//  * forward (1024 threads per two samples, pooled rows in LDS): z1 = pooled W1 + b1 with W1 [C][S] streamed through LDS in chunks
//    of whole rows — copied with consecutive lanes on consecutive addresses and re-laid at an odd row pitch — and threads =
//    (unit lane, one of 16 channel lanes) walking the chunk's rows, the 16 lanes' sums added in lane order;
//    gate = sigmoid(swish(z1) W2 + b2) with thread = (channel quad, unit group): W2 [S][C] is read coalesced along the channels,
//    the groups' sums meet in LDS in group order.  z1 (pre-activation) and gate are what backward needs.
//  * backward A (same geometry): dz2 = dgate gate (1 - gate); dh = dz2 W2^T by (channel quad, unit group) threads whose products
//    are summed over the quads by 16 row slices in order; dz1 = dh swish'(z1) (stored); dpooled = dz1 W1^T from LDS chunks of W1,
//    one row per thread.
//  * backward B (per 16 channels): dW2[:, chunk] = h^T dz2, db2, dW1[chunk, :] = pooled^T dz1, db1 — sums over the samples in
//    sample order from LDS tiles of 64 samples held transposed (four samples per ds_read_b128): bitwise reproducible.
#include "common.h"
#include "../../include/embnet.h"

namespace embnet {
namespace semlp {

constexpr int SB = 2;            // samples per workgroup (forward, backward A)
constexpr int TH = 1024;         // threads of a forward / backward-A workgroup
constexpr int JG = 16;           // units per thread and pass where the channel quads AND unit groups are spread over the threads
constexpr int CH = 16;           // channels per workgroup (backward B)
constexpr int NT = 64;           // samples per LDS tile (backward B)
constexpr int NTP = NT + 4;      // its row pitch in floats: 16-byte aligned rows, 16 channel lanes x ds_read_b128 cover all 64 banks

__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + __expf(-v)); }

// Everything here is bound by memory LATENCY and by the access pattern, not by bytes or FLOP (2.5-28 MFLOP, < 1 MB per launch).
// Measured forms (profiles/r05_exp_se_mlp.txt, forward / backward A in the C5 step's trace): one (unit, channel-lane) pair per thread
// walking C / 4 channels with four loads in flight 34 / 66 us; 256 threads each owning C / 256 channels (a thread per W1 row: 64
// different cache lines per load instruction) 21 / 62 us; 1024 such threads 28 / 51 us; this form 17 / 48 us (8-21 / 10-27 us back
// to back, tools/exp/se_mlp_bench.py).  Requesting more per round trip (first chunk and first W2 batch at kernel start, chunk
// prefetch, z1 through LDS) and rotating the chunk order per workgroup measured the same in the step and are not kept.

// rc rows of S floats (contiguous in memory) -> LDS rows of pitch SP; consecutive threads on consecutive addresses, four
// requests per thread and round in flight
constexpr int WL_FLOATS = 16384;
__device__ __forceinline__ void stage_rows(const float* __restrict__ src, int rc, int S, int SP, float* wl) {
  const int total = rc * S;
  for (int i0 = 0; i0 < total; i0 += 4 * TH) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = i0 + threadIdx.x + u * TH; v[u] = i < total ? src[i] : 0.f; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + threadIdx.x + u * TH;
      if (i < total) { const int r = i / S; wl[r * SP + (i - r * S)] = v[u]; }
    }
  }
}

// how the second-layer phases spread over the threads: thread t -> channel quad t % c4, unit group t / c4 of G groups of jpg units
struct QuadGroups { int G, jpg; };
__device__ __forceinline__ QuadGroups quad_groups(int c4, int S) {
  int G = TH / c4; if (G > S) G = S; if (G < 1) G = 1;
  return QuadGroups{G, (S + G - 1) / G};
}

// LDS (floats): rows[SB][c] | act[SB][JP] | wl[WL_FLOATS] (a chunk of W1; then the partial sums of both layers)      c % 4 == 0, c / 4 <= TH
__global__ __launch_bounds__(TH) void se_mlp_fwd_kernel(const float* __restrict__ pooled, const float* __restrict__ w1,
                                                        const float* __restrict__ b1, const float* __restrict__ w2,
                                                        const float* __restrict__ b2, int n, int c, int S, int JP,
                                                        float* __restrict__ z1, float* __restrict__ gate) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* rows = sm; float* act = rows + SB * c; float* wl = act + SB * JP; float* red = wl;
  const int tid = threadIdx.x, s0 = blockIdx.x * SB;
  for (int i = tid; i < c * SB; i += TH) {
    const int s = i / c, cc = i - s * c;
    rows[i] = s0 + s < n ? pooled[(long)(s0 + s) * c + cc] : 0.f;
  }
  __syncthreads();
  // z1 = pooled W1 + b1.  W1 [c][S] streams through LDS in chunks of whole rows, copied with consecutive lanes on consecutive
  // addresses (a thread per row requests 64 different cache lines per load instruction: 6.8 us per pass at 16 waves per CU) into
  // rows of odd pitch SP; thread t = (unit lane t % 64, channel lane t / 64) then walks the chunk's rows, lanes along the units
  const int SP = S | 1, rc_max = WL_FLOATS / SP;
  const int jl = tid & 63, cl = tid >> 6;                      // 16 channel lanes
  float acc[SB][3];                                            // units jl, jl + 64, jl + 128 (S <= 160)
#pragma unroll
  for (int s = 0; s < SB; ++s) { acc[s][0] = 0.f; acc[s][1] = 0.f; acc[s][2] = 0.f; }
  for (int r0 = 0; r0 < c; r0 += rc_max) {
    const int rc = min(rc_max, c - r0);
    stage_rows(w1 + (long)r0 * S, rc, S, SP, wl);
    __syncthreads();
    for (int r = cl; r < rc; r += TH / 64) {
      float pv[SB];
#pragma unroll
      for (int s = 0; s < SB; ++s) pv[s] = rows[s * c + r0 + r];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int j = jl + 64 * k;
        if (j < S) {
          const float w = wl[r * SP + j];
#pragma unroll
          for (int s = 0; s < SB; ++s) acc[s][k] = fmaf(pv[s], w, acc[s][k]);
        }
      }
    }
    __syncthreads();
  }
  {                                                            // the 16 channel lanes of a unit, in lane order
    float* part = wl;                                          // [16][SB][JP]
#pragma unroll
    for (int s = 0; s < SB; ++s)
#pragma unroll
      for (int k = 0; k < 3; ++k) if (jl + 64 * k < S) part[(cl * SB + s) * JP + jl + 64 * k] = acc[s][k];
    __syncthreads();
    for (int i = tid; i < SB * S; i += TH) {
      const int s = i / S, j = i - s * S;
      float z = b1[j];
#pragma unroll
      for (int l = 0; l < TH / 64; ++l) z += part[(l * SB + s) * JP + j];
      if (s0 + s < n) z1[(long)(s0 + s) * S + j] = z;
      act[s * JP + j] = z * sigm(z);
    }
  }
  __syncthreads();
  // gate = sigmoid(h W2 + b2): thread t owns channel quad t % c4 and unit group t / c4; the groups' sums meet in LDS
  const int c4 = c >> 2;
  const QuadGroups qg = quad_groups(c4, S);
  const int cq = tid % c4, g = tid / c4;
  float4* part = reinterpret_cast<float4*>(red);               // [G][c4][SB]
  if (g < qg.G) {
    float4 a[SB];
#pragma unroll
    for (int s = 0; s < SB; ++s) a[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* wq = reinterpret_cast<const float4*>(w2) + cq;
    const int j0 = g * qg.jpg, j1 = min(S, j0 + qg.jpg);
    constexpr int JF = 8;                                      // requests per batch (16: scratch at 128 registers)
    for (int jo = j0; jo < j1; jo += JF) {
      float4 w[JF];
#pragma unroll
      for (int j = 0; j < JF; ++j) w[j] = jo + j < j1 ? wq[(long)(jo + j) * c4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < JF; ++j)
#pragma unroll
        for (int s = 0; s < SB; ++s) {
          const float h = jo + j < j1 ? act[s * JP + jo + j] : 0.f;
          a[s].x = fmaf(h, w[j].x, a[s].x); a[s].y = fmaf(h, w[j].y, a[s].y); a[s].z = fmaf(h, w[j].z, a[s].z); a[s].w = fmaf(h, w[j].w, a[s].w);
        }
    }
#pragma unroll
    for (int s = 0; s < SB; ++s) part[(g * c4 + cq) * SB + s] = a[s];
  }
  __syncthreads();
  if (tid < c4) {
    const float4 bq = reinterpret_cast<const float4*>(b2)[tid];
#pragma unroll
    for (int s = 0; s < SB; ++s) {
      float4 a = bq;
      for (int k = 0; k < qg.G; ++k) { const float4 o = part[(k * c4 + tid) * SB + s]; a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w; }
      if (s0 + s < n)
        reinterpret_cast<float4*>(gate + (long)(s0 + s) * c)[tid] = make_float4(sigm(a.x), sigm(a.y), sigm(a.z), sigm(a.w));
    }
  }
}

// LDS (floats): d2[SB][c] | dzs[SB][JP] | sl2[16][64] | red[TH][JG + 1], then wl[WL_FLOATS] (a chunk of W1) in its place
__global__ __launch_bounds__(TH) void se_mlp_bwd_a_kernel(const float* __restrict__ dgate, const float* __restrict__ gate,
                                                          const float* __restrict__ z1, const float* __restrict__ w1,
                                                          const float* __restrict__ w2, int n, int c, int S, int JP,
                                                          float* __restrict__ dz1, float* __restrict__ dpooled) {
  constexpr int RQ = JG + 1;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* d2 = sm; float* dzs = d2 + SB * c; float* sl2 = dzs + SB * JP; float* red = sl2 + 16 * 64; float* wl = red;
  const int tid = threadIdx.x, s0 = blockIdx.x * SB;
  for (int i = tid; i < c * SB; i += TH) {
    const int s = i / c, cc = i - s * c;
    float v = 0.f;
    if (s0 + s < n) { const long o = (long)(s0 + s) * c + cc; const float g = gate[o]; v = dgate[o] * g * (1.f - g); }
    d2[i] = v;
  }
  __syncthreads();
  // dh = dz2 W2^T: thread t owns channel quad t % c4 and unit group t / c4 (JG units per pass: ONE batch of requests);
  // the quads' products of a (group, unit) are summed by 16 slices of rows in row order, the slices in order.  dz1 = dh swish'(z1)
  const int c4 = c >> 2;
  const QuadGroups qg = quad_groups(c4, S);
  const int cq = tid % c4, g = tid / c4;
  const int rows_per_slice = (c4 + 15) / 16;
  for (int jo = 0; jo < qg.jpg; jo += JG) {                      // unit jo + j of every group
    float4 w[JG];
    const int jbase = g * qg.jpg + jo, jend = min(S, (g + 1) * qg.jpg);
    const float4* wq = reinterpret_cast<const float4*>(w2) + cq;
#pragma unroll
    for (int j = 0; j < JG; ++j) w[j] = (g < qg.G && jbase + j < jend) ? wq[(long)(jbase + j) * c4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int s = 0; s < SB; ++s) {
      const float4 dv = g < qg.G ? reinterpret_cast<const float4*>(d2 + s * c)[cq] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < JG; ++j)
        red[tid * RQ + j] = fmaf(dv.x, w[j].x, fmaf(dv.y, w[j].y, fmaf(dv.z, w[j].z, dv.w * w[j].w)));
      __syncthreads();
      // reducer thread: output o = tid % 64 = (group o / JG, unit o % JG) (64 / JG = 4 groups per round), slice tid / 64 of its rows
      for (int g0 = 0; g0 < qg.G; g0 += 64 / JG) {
        const int o = tid & 63, sl = tid >> 6, go = g0 + o / JG, jj = o % JG;
        float r = 0.f;
        if (go < qg.G) {
          const int r0 = sl * rows_per_slice, r1 = min(c4, r0 + rows_per_slice);
          for (int q = r0; q < r1; ++q) r += red[(go * c4 + q) * RQ + jj];
        }
        sl2[sl * 64 + o] = r;
        __syncthreads();
        if (tid < 64 && go < qg.G) {
          const int j = go * qg.jpg + jo + jj;
          if (j < min(S, (go + 1) * qg.jpg)) {
            float dh = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) dh += sl2[k * 64 + tid];
            float v = 0.f;
            if (s0 + s < n) {
              const long oz = (long)(s0 + s) * S + j;
              const float z = z1[oz], sg = sigm(z);
              v = dh * (sg + z * sg * (1.f - sg));
              dz1[oz] = v;
            }
            dzs[s * JP + j] = v;
          }
        }
        __syncthreads();
      }
    }
  }
  // dpooled = dz1 W1^T: chunks of W1 rows through LDS (stage_rows), one row per thread (odd pitch: conflict-free)
  const int SP = S | 1, rc_max = min(WL_FLOATS / SP, TH);
  for (int r0 = 0; r0 < c; r0 += rc_max) {
    const int rc = min(rc_max, c - r0);
    __syncthreads();
    stage_rows(w1 + (long)r0 * S, rc, S, SP, wl);
    __syncthreads();
    if (tid < rc) {
      const float* wr = wl + tid * SP;
      float a[SB];
#pragma unroll
      for (int s = 0; s < SB; ++s) a[s] = 0.f;
#pragma unroll 4
      for (int j = 0; j < S; ++j) {
        const float w = wr[j];
#pragma unroll
        for (int s = 0; s < SB; ++s) a[s] = fmaf(dzs[s * JP + j], w, a[s]);
      }
#pragma unroll
      for (int s = 0; s < SB; ++s) if (s0 + s < n) dpooled[(long)(s0 + s) * c + r0 + tid] = a[s];
    }
  }
}

// LDS (floats), all transposed [feature][NT samples], row pitch NTP: D2[CH] | P[CH] | H[S] | Z[S]
// thread t: channel cc = t % CH, units j = t / CH + 16 k  (256 / CH = 16 unit lanes), up to UMAX units per thread
template <int UMAX>
__global__ __launch_bounds__(256) void se_mlp_bwd_b_kernel(const float* __restrict__ dgate, const float* __restrict__ gate,
                                                           const float* __restrict__ z1, const float* __restrict__ dz1,
                                                           const float* __restrict__ pooled, int n, int c, int S,
                                                           float* __restrict__ dw1, float* __restrict__ db1,
                                                           float* __restrict__ dw2, float* __restrict__ db2) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* D2 = sm; float* P = D2 + CH * NTP; float* H = P + CH * NTP; float* Z = H + (long)S * NTP;
  const int tid = threadIdx.x, c0 = blockIdx.x * CH;
  const int ccl = tid % CH, jl = tid / CH;
  const bool cok = c0 + ccl < c;
  float a2[UMAX], a1[UMAX];                     // dW2[j][cc], dW1[cc][j]
#pragma unroll
  for (int u = 0; u < UMAX; ++u) { a2[u] = 0.f; a1[u] = 0.f; }
  float bsum2 = 0.f, bsum1 = 0.f;               // db2[cc] (threads < CH), db1[j] (workgroup 0, threads 64 .. 64 + S)
  for (int n0 = 0; n0 < n; n0 += NT) {
    __syncthreads();                            // the previous tile is consumed
    // (sample, feature) -> [feature][sample]; every request of a batch goes out before the first value is used
    {
      float g[NT * CH / 256], dg[NT * CH / 256], pv[NT * CH / 256];
#pragma unroll
      for (int u = 0; u < NT * CH / 256; ++u) {
        const int i = tid + 256 * u, nn = i / CH, k = i - nn * CH;
        const bool ok = n0 + nn < n && c0 + k < c;
        const long o = ok ? (long)(n0 + nn) * c + c0 + k : 0;
        g[u] = gate[o]; dg[u] = dgate[o]; pv[u] = pooled[o];
        if (!ok) { dg[u] = 0.f; pv[u] = 0.f; }
      }
#pragma unroll
      for (int u = 0; u < NT * CH / 256; ++u) {
        const int i = tid + 256 * u, nn = i / CH, k = i - nn * CH;
        D2[k * NTP + nn] = dg[u] * g[u] * (1.f - g[u]); P[k * NTP + nn] = pv[u];
      }
    }
    for (int i0 = 0; i0 < NT * S; i0 += 256 * 8) {
      float zv[8], dv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + tid + 256 * u, nn = i / S;
        const bool ok = i < NT * S && n0 + nn < n;
        const long o = ok ? (long)n0 * S + i : 0;
        zv[u] = z1[o]; dv[u] = dz1[o];
        if (!ok) { zv[u] = 0.f; dv[u] = 0.f; }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + tid + 256 * u, nn = i / S, j = i - nn * S;
        if (i < NT * S) { H[j * NTP + nn] = zv[u] * sigm(zv[u]); Z[j * NTP + nn] = dv[u]; }
      }
    }
    __syncthreads();
    const float4* d4 = reinterpret_cast<const float4*>(D2 + ccl * NTP);
    const float4* p4 = reinterpret_cast<const float4*>(P + ccl * NTP);
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
      const int j = jl + 16 * u;
      if (j < S) {
        const float4* h4 = reinterpret_cast<const float4*>(H + (long)j * NTP);
        const float4* z4 = reinterpret_cast<const float4*>(Z + (long)j * NTP);
        float s2 = a2[u], s1 = a1[u];
#pragma unroll 4
        for (int q = 0; q < NT / 4; ++q) {
          const float4 dv = d4[q], hv = h4[q], pv = p4[q], zv = z4[q];
          s2 = fmaf(hv.x, dv.x, s2); s2 = fmaf(hv.y, dv.y, s2); s2 = fmaf(hv.z, dv.z, s2); s2 = fmaf(hv.w, dv.w, s2);
          s1 = fmaf(pv.x, zv.x, s1); s1 = fmaf(pv.y, zv.y, s1); s1 = fmaf(pv.z, zv.z, s1); s1 = fmaf(pv.w, zv.w, s1);
        }
        a2[u] = s2; a1[u] = s1;
      }
    }
    if (tid < CH) for (int nn = 0; nn < NT; ++nn) bsum2 += D2[tid * NTP + nn];
    if (blockIdx.x == 0 && tid >= 64 && tid - 64 < S) for (int nn = 0; nn < NT; ++nn) bsum1 += Z[(tid - 64) * NTP + nn];
  }
#pragma unroll
  for (int u = 0; u < UMAX; ++u) {
    const int j = jl + 16 * u;
    if (j < S && cok) { dw2[(long)j * c + c0 + ccl] = a2[u]; dw1[(long)(c0 + ccl) * S + j] = a1[u]; }
  }
  if (tid < CH && c0 + tid < c) db2[c0 + tid] = bsum2;
  if (blockIdx.x == 0 && tid >= 64 && tid - 64 < S) db1[tid - 64] = bsum1;
}

}  // namespace semlp
}  // namespace embnet

using namespace embnet;
using namespace embnet::semlp;

static inline int jp_of(int S) { return (S + 63) / 64 * 64; }
// forward / backward A: rows[SB][c] | act[SB][JP] | 1024 floats | max(wl[WL_FLOATS], red[TH][JG + 1]) (the forward kernel has no sl2: slack)
static inline size_t lds_rows(int c, int s) { return ((size_t)SB * c + (size_t)SB * jp_of(s) + 16 * 64 + (size_t)TH * (JG + 1)) * 4; }

// 1: the fused launches apply (S <= 160: backward B's per-thread sums; LDS fits)
extern "C" int embnet_se_mlp_supported(int n, int c, int s) {
  if (n <= 0 || c <= 0 || s <= 0 || s > 160 || (c & 3) || c / 4 > TH) return 0;
  const size_t lds_a = lds_rows(c, s), lds_b = ((size_t)2 * CH + 2 * (size_t)s) * NTP * 4;
  return lds_a <= 150 * 1024 && lds_b <= 150 * 1024;
}

extern "C" int embnet_se_mlp_fwd(const float* pooled, const float* w1, const float* b1, const float* w2, const float* b2, int n,
                                 int c, int s, float* z1, float* gate, void* stream) {
  EMBNET_CHECK_ARG(pooled && w1 && b1 && w2 && b2 && z1 && gate, "se_mlp_fwd: null pointer");
  EMBNET_CHECK_ARG(embnet_se_mlp_supported(n, c, s), "se_mlp_fwd: unsupported sizes (see embnet_se_mlp_supported)");
  const int JP = jp_of(s);
  const size_t lds = lds_rows(c, s);
  static bool once = false;
  if (!once) { (void)hipFuncSetAttribute((const void*)se_mlp_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
  EMBNET_TRACE_FLOP("embnet::semlp::se_mlp_fwd_kernel", 4.0 * n * c * s, 4.0 * (2.0 * n * c + 2.0 * c * s), stream);
  se_mlp_fwd_kernel<<<cdiv(n, SB), TH, lds, (hipStream_t)stream>>>(pooled, w1, b1, w2, b2, n, c, s, JP, z1, gate);
  return check_launch("se_mlp_fwd");
}

// dz1 [n, s]: scratch the two launches share.  Writes dpooled [n, c], dw1 [c, s], db1 [s], dw2 [s, c], db2 [c].
extern "C" int embnet_se_mlp_bwd(const float* dgate, const float* gate, const float* z1, const float* pooled, const float* w1,
                                 const float* w2, int n, int c, int s, float* dz1, float* dpooled, float* dw1, float* db1,
                                 float* dw2, float* db2, void* stream) {
  EMBNET_CHECK_ARG(dgate && gate && z1 && pooled && w1 && w2 && dz1 && dpooled && dw1 && db1 && dw2 && db2, "se_mlp_bwd: null pointer");
  EMBNET_CHECK_ARG(embnet_se_mlp_supported(n, c, s), "se_mlp_bwd: unsupported sizes (see embnet_se_mlp_supported)");
  const int JP = jp_of(s);
  const size_t lds_a = lds_rows(c, s), lds_b = ((size_t)2 * CH + 2 * (size_t)s) * NTP * 4;
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void*)se_mlp_bwd_a_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)se_mlp_bwd_b_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)se_mlp_bwd_b_kernel<10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    once = true;
  }
  hipStream_t st = (hipStream_t)stream;
  EMBNET_TRACE_FLOP("embnet::semlp::se_mlp_bwd_a_kernel", 4.0 * n * c * s, 4.0 * (3.0 * n * c + 2.0 * c * s), stream);
  se_mlp_bwd_a_kernel<<<cdiv(n, SB), TH, lds_a, st>>>(dgate, gate, z1, w1, w2, n, c, s, JP, dz1, dpooled);
  EMBNET_TRACE_FLOP("embnet::semlp::se_mlp_bwd_b_kernel", 4.0 * n * c * s, 4.0 * (3.0 * n * c + 2.0 * c * s), stream);
  if (s <= 48) se_mlp_bwd_b_kernel<3><<<cdiv(c, CH), 256, lds_b, st>>>(dgate, gate, z1, dz1, pooled, n, c, s, dw1, db1, dw2, db2);
  else se_mlp_bwd_b_kernel<10><<<cdiv(c, CH), 256, lds_b, st>>>(dgate, gate, z1, dz1, pooled, n, c, s, dw1, db1, dw2, db2);
  return check_launch("se_mlp_bwd");
}
