// HBM-bound pieces of the EfficientNet MBConv block (the `efficientnet` zoo package instantiated at
// /root/reference/embedding_net/backbones.py:84-98) and of the siamese 'l1' head (models.py:217-221):
// depthwise convolution fwd / dgrad / wgrad, swish, sigmoid, squeeze-excite channel scaling,
// per-sample drop-connect, |a-b|.  NHWC fp32.  The 1x1 expand/project convolutions and the SE
// dense layers run on the MFMA engine (conv.hip / dense.hip); nothing here is GEMM-shaped:
// a depthwise tap is one multiply per loaded element, so these are priced in bytes.
#include "dw_geom.h"
#include "../../include/embnet.h"

namespace embnet {

// y[n,oh,ow,c] = sum_{r,s} x[n, oh*st+r-pt, ow*st+s-pl, c] * w[r,s,c]      bytes: 4*(in + out)
template <int V>   // V = 4: float4 over channels (C % 4 == 0), V = 1: scalar
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         DwGeom g, float* __restrict__ y) {
  const int cv = g.C / V;
  const long total = (long)g.N * g.OH * g.OW * cv;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % cv) * V;
  long t = i / cv;
  const int ow = (int)(t % g.OW); t /= g.OW;
  const int oh = (int)(t % g.OH);
  const int n = (int)(t / g.OH);
  float acc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = 0.f;
  for (int r = 0; r < g.R; ++r) {
    const int ih = oh * g.stride + r - g.pad_t;
    if ((unsigned)ih >= (unsigned)g.H) continue;
    for (int s = 0; s < g.S; ++s) {
      const int iw = ow * g.stride + s - g.pad_l;
      if ((unsigned)iw >= (unsigned)g.W) continue;
      const float* xp = x + (((long)n * g.H + ih) * g.W + iw) * g.C + c;
      const float* wp = w + ((long)r * g.S + s) * g.C + c;
      if (V == 4) {
        const float4 xv = *reinterpret_cast<const float4*>(xp), wv = *reinterpret_cast<const float4*>(wp);
        acc[0] = fmaf(xv.x, wv.x, acc[0]); acc[1 % V] = fmaf(xv.y, wv.y, acc[1 % V]);
        acc[2 % V] = fmaf(xv.z, wv.z, acc[2 % V]); acc[3 % V] = fmaf(xv.w, wv.w, acc[3 % V]);
      } else {
        acc[0] = fmaf(xp[0], wp[0], acc[0]);
      }
    }
  }
  float* yp = y + i * V;
  if (V == 4) *reinterpret_cast<float4*>(yp) = make_float4(acc[0], acc[1 % V], acc[2 % V], acc[3 % V]);
  else yp[0] = acc[0];
}

// dx[n,h,w,c] = sum_{r,s} dy[n,(h+pt-r)/st,(w+pl-s)/st,c] * w[r,s,c]  over taps that divide
template <int V>
__global__ __launch_bounds__(256) void dwconv_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                           DwGeom g, float* __restrict__ dx) {
  const int cv = g.C / V;
  const long total = (long)g.N * g.H * g.W * cv;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % cv) * V;
  long t = i / cv;
  const int iw = (int)(t % g.W); t /= g.W;
  const int ih = (int)(t % g.H);
  const int n = (int)(t / g.H);
  float acc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = 0.f;
  for (int r = 0; r < g.R; ++r) {
    const int th = ih + g.pad_t - r;
    if (th < 0 || th % g.stride) continue;
    const int oh = th / g.stride;
    if (oh >= g.OH) continue;
    for (int s = 0; s < g.S; ++s) {
      const int tw = iw + g.pad_l - s;
      if (tw < 0 || tw % g.stride) continue;
      const int ow = tw / g.stride;
      if (ow >= g.OW) continue;
      const float* dp = dy + (((long)n * g.OH + oh) * g.OW + ow) * g.C + c;
      const float* wp = w + ((long)r * g.S + s) * g.C + c;
      if (V == 4) {
        const float4 dv = *reinterpret_cast<const float4*>(dp), wv = *reinterpret_cast<const float4*>(wp);
        acc[0] = fmaf(dv.x, wv.x, acc[0]); acc[1 % V] = fmaf(dv.y, wv.y, acc[1 % V]);
        acc[2 % V] = fmaf(dv.z, wv.z, acc[2 % V]); acc[3 % V] = fmaf(dv.w, wv.w, acc[3 % V]);
      } else {
        acc[0] = fmaf(dp[0], wp[0], acc[0]);
      }
    }
  }
  float* xp = dx + i * V;
  if (V == 4) *reinterpret_cast<float4*>(xp) = make_float4(acc[0], acc[1 % V], acc[2 % V], acc[3 % V]);
  else xp[0] = acc[0];
}

// Square KS x KS kernels, 4 channels per thread, branch-free: every tap issues its 16-byte load unconditionally
// (taps outside the image read element 0 and are multiplied by 0), so the KS*KS loads of a pixel are all in flight
// together.  The generic kernels above branch around each tap, which makes hipcc wait for one load before it
// issues the next (EfficientNet-B0's 5x5 layers: 3-5x slower, measured).
template <int KS>
__global__ __launch_bounds__(256) void dwconv_fwd4_sq_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             DwGeom g, float* __restrict__ y) {
  const int c4 = g.C >> 2;
  const long total = (long)g.N * g.OH * g.OW * c4;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int cq = (int)(i % c4);
  long t = i / c4;
  const int ow = (int)(t % g.OW); t /= g.OW;
  const int oh = (int)(t % g.OH);
  const int n = (int)(t / g.OH);
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* w4 = reinterpret_cast<const float4*>(w);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int r = 0; r < KS; ++r)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int ih = oh * g.stride + r - g.pad_t, iw = ow * g.stride + s - g.pad_l;
      const bool ok = (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
      const float4 xv = x4[ok ? (((long)n * g.H + ih) * g.W + iw) * c4 + cq : 0];
      float4 wv = w4[(r * KS + s) * c4 + cq];
      if (!ok) wv = make_float4(0.f, 0.f, 0.f, 0.f);
      acc.x = fmaf(xv.x, wv.x, acc.x); acc.y = fmaf(xv.y, wv.y, acc.y);
      acc.z = fmaf(xv.z, wv.z, acc.z); acc.w = fmaf(xv.w, wv.w, acc.w);
    }
  reinterpret_cast<float4*>(y)[i] = acc;
}

template <int KS, int ST>   // ST: stride (1 or 2), compile-time so the tap -> output-pixel map needs no division
__global__ __launch_bounds__(256) void dwconv_dgrad4_sq_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                               DwGeom g, float* __restrict__ dx) {
  const int c4 = g.C >> 2;
  const long total = (long)g.N * g.H * g.W * c4;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int cq = (int)(i % c4);
  long t = i / c4;
  const int iw = (int)(t % g.W); t /= g.W;
  const int ih = (int)(t % g.H);
  const int n = (int)(t / g.H);
  const float4* d4 = reinterpret_cast<const float4*>(dy);
  const float4* w4 = reinterpret_cast<const float4*>(w);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int r = 0; r < KS; ++r)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int th = ih + g.pad_t - r, tw = iw + g.pad_l - s;
      const int oh = th >> (ST - 1), ow = tw >> (ST - 1);
      const bool ok = th >= 0 && tw >= 0 && (ST == 1 || (((th | tw) & 1) == 0)) && oh < g.OH && ow < g.OW;
      const float4 dv = d4[ok ? (((long)n * g.OH + oh) * g.OW + ow) * c4 + cq : 0];
      float4 wv = w4[(r * KS + s) * c4 + cq];
      if (!ok) wv = make_float4(0.f, 0.f, 0.f, 0.f);
      acc.x = fmaf(dv.x, wv.x, acc.x); acc.y = fmaf(dv.y, wv.y, acc.y);
      acc.z = fmaf(dv.z, wv.z, acc.z); acc.w = fmaf(dv.w, wv.w, acc.w);
    }
  reinterpret_cast<float4*>(dx)[i] = acc;
}

// ---- register-blocked square kernels (the EfficientNet layers: 3x3 / 5x5, stride 1 / 2, C % 4 == 0) -------------
// One thread per (channel quad, output row, block of TW = 4 output columns).  The per-pixel kernels above issue KS*KS
// 16-byte image loads + KS*KS weight loads per output quad and run at 1.3-2.2 TB/s of algorithmic traffic on
// EfficientNet-B0 (bound by L1 request rate, not HBM); a column block shares its input columns between its outputs and
// its weight loads between TW outputs: 5x5 stride 1 needs 8 image loads per kernel row for 4 outputs (40 + 25 per 4
// outputs instead of 200).  Loads are unconditional (out-of-image taps read element 0 and are zeroed in registers).
constexpr int DW_TW = 4;

// FLIP = false: forward correlation  y[oh,ow] = sum x[oh*ST+r-pt, ow*ST+s-pl] * w[r,s]
// FLIP = true (ST = 1 only): the same loop as the stride-1 DATA GRADIENT: dx = correlate(dy, flipped w) with pads
// KS-1-pt / KS-1-pl — the launcher swaps the roles (g.H/W = size of the tensor read, g.OH/OW = size written).
// TW output columns per thread: 8 where the row is wide enough (5x5 stride 1: 60 + 25 loads per 8 outputs instead of 40 + 25
// per 4; the kernels run at the rate the L1 takes requests, not at the HBM rate), 4 otherwise.
template <int KS, int ST, bool FLIP, int TW>
__global__ __launch_bounds__(256) void dwconv_row4_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          DwGeom g, float* __restrict__ y) {
  constexpr int NX = (TW - 1) * ST + KS;
  const int c4 = g.C >> 2, wb_n = (g.OW + TW - 1) / TW;
  const long total = (long)g.N * g.OH * wb_n * c4;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int cq = (int)(i % c4);
  long t = i / c4;
  const int ow0 = (int)(t % wb_n) * TW; t /= wb_n;
  const int oh = (int)(t % g.OH);
  const int n = (int)(t / g.OH);
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* w4 = reinterpret_cast<const float4*>(w);
  float4 acc[TW];
#pragma unroll
  for (int q = 0; q < TW; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int iw0 = ow0 * ST - g.pad_l;
  // one kernel row at a time for 5x5 (unrolled, hipcc hoists all 40-55 image loads of a block: 256+ registers, one wave per SIMD)
#pragma unroll KS == 3 ? 3 : 1
  for (int r = 0; r < KS; ++r) {
    const int ih = oh * ST + r - g.pad_t;
    const bool rok = (unsigned)ih < (unsigned)g.H;
    const long rbase = ((long)n * g.H + (rok ? ih : 0)) * g.W;
    float4 xr[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int iw = iw0 + j;
      const bool ok = rok && (unsigned)iw < (unsigned)g.W;
      xr[j] = x4[ok ? (rbase + iw) * c4 + cq : 0];
      if (!ok) xr[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_) {
      const float4 wv = w4[(FLIP ? (KS - 1 - r) * KS + (KS - 1 - s_) : r * KS + s_) * c4 + cq];
#pragma unroll
      for (int q = 0; q < TW; ++q) {
        const float4 v = xr[q * ST + s_];
        acc[q].x = fmaf(v.x, wv.x, acc[q].x); acc[q].y = fmaf(v.y, wv.y, acc[q].y);
        acc[q].z = fmaf(v.z, wv.z, acc[q].z); acc[q].w = fmaf(v.w, wv.w, acc[q].w);
      }
    }
  }
  float4* yo = reinterpret_cast<float4*>(y) + (((long)n * g.OH + oh) * g.OW + ow0) * c4 + cq;
#pragma unroll
  for (int q = 0; q < TW; ++q)
    if (ow0 + q < g.OW) yo[(long)q * c4] = acc[q];
}

// The same loop for TWO output rows per thread (oh0, oh0 + 1): the (KS + ST) input rows the pair needs are each fetched once
// and feed both outputs — the kernel runs at the rate the L1 takes requests (header above), and this form issues
// (6*12 + 50) image + weight loads per 16 outputs of a 5x5 stride-1 layer against (5*12 + 25) per 8, and (4*10 + 18) per 16
// against (3*10 + 9) per 8 for 3x3.  A real loop over the input rows (one row's loads in flight; unrolled, hipcc hoists
// every row's loads: 330-490 VGPRs), so the kernel row of an (input row, output row) pair is a run-time, wave-uniform
// index and the weights are loaded where they are used (L1 hits), as in the one-row kernel.
// Row blockIdx.x of stats[2][C][gridDim.x] from the per-thread float4 sums of a workgroup that walks L chunks of 256 threads:
// in chunk k thread t handles channel quad (256 k + t) % c4; threads t, t + c4, .. share a quad and are added in order, the
// chunk's sum per quad is added to the quad's LDS accumulator by ONE thread (fixed order over the chunks), and the row is
// written once at the end.  One row per 256 threads (L = 1) cost a 14x14x672 stride-2 data gradient 37 % extra traffic in
// 4-byte writes 37 KB apart (9 408 rows): 229 us at 1.3 TB/s.  accq: 2 * c4 float4 of dynamic LDS.
__device__ __forceinline__ void dw_block_begin(int c4, float4* accq) {
  for (int q = threadIdx.x; q < 2 * c4; q += 256) accq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
}
__device__ __forceinline__ void dw_block_accumulate(float4 s1, float4 s2, int c4, long chunk, float4* accq) {
  __shared__ float4 red[2][256];
  red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
  __syncthreads();
  if ((int)threadIdx.x < c4) {
    float4 a1 = red[0][threadIdx.x], a2 = red[1][threadIdx.x];
    for (int u = threadIdx.x + c4; u < 256; u += c4) {
      const float4 o1 = red[0][u], o2 = red[1][u];
      a1.x += o1.x; a1.y += o1.y; a1.z += o1.z; a1.w += o1.w;
      a2.x += o2.x; a2.y += o2.y; a2.z += o2.z; a2.w += o2.w;
    }
    const int q = (int)((chunk * 256 + threadIdx.x) % c4);
    float4 b1 = accq[q], b2 = accq[c4 + q];
    b1.x += a1.x; b1.y += a1.y; b1.z += a1.z; b1.w += a1.w;
    b2.x += a2.x; b2.y += a2.y; b2.z += a2.z; b2.w += a2.w;
    accq[q] = b1; accq[c4 + q] = b2;
  }
  __syncthreads();
}
__device__ __forceinline__ void dw_block_flush(int c4, int C, const float4* accq, float* __restrict__ stats) {
  const long P = gridDim.x;
  for (int q = threadIdx.x; q < c4; q += 256) {
    const float4 a1 = accq[q], a2 = accq[c4 + q];
    float* d1 = stats + (long)(4 * q) * P + blockIdx.x;
    float* d2 = d1 + (long)C * P;
    d1[0] = a1.x; d1[P] = a1.y; d1[2 * P] = a1.z; d1[3 * P] = a1.w;
    d2[0] = a2.x; d2[P] = a2.y; d2[2 * P] = a2.z; d2[3 * P] = a2.w;
  }
}
// STATS (forward only): the workgroup also writes the per-channel sum and sum of squares of the outputs it produced as row
// blockIdx.x of stats[2][C][gridDim.x] — the statistics partials of the BatchNormalization that follows (the layout the conv
// epilogues write, embnet_bn_train_fwd's `partials`), so that layer does not read the tensor for them.  Threads i, i + c4, ..
// of a workgroup hold the same channel quad; every (channel, row) is written (dw_block_flush: quads a workgroup never touched as zeros).
// STATS = 2 (stride-1 data gradient, FLIP): y is the gradient of the depthwise layer's INPUT a = act(BN(e)); the workgroup writes
// the BatchNorm-backward sums of that layer instead — sum dz and sum dz * ehat with dz = y * act'(BN(e)) — reading e once per
// output (conv.hip's BnSums for the depthwise data gradient): the BatchNormalization backward skips its reduction pass.
template <int KS, int ST, bool FLIP, int TW, int STATS = 0>
__global__ __launch_bounds__(256) void dwconv_row4x2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            DwGeom g, float* __restrict__ y, float* __restrict__ stats = nullptr,
                                                            const DwBn bn = DwBn{nullptr, nullptr, nullptr, nullptr, nullptr, 0}) {
  constexpr int NX = (TW - 1) * ST + KS, TH = 2, NR = (TH - 1) * ST + KS;
  extern __shared__ float4 dw_accq[];                     // STATS: 2 * c4 float4
  const int c4 = g.C >> 2, wb_n = (g.OW + TW - 1) / TW, hb_n = (g.OH + TH - 1) / TH;
  const long total = (long)g.N * hb_n * wb_n * c4;
  const int L = STATS ? (g.L > 1 ? g.L : 1) : 1;
  if (STATS) dw_block_begin(c4, dw_accq);
  for (int lc = 0; lc < L; ++lc) {
  const long i0 = ((long)blockIdx.x * L + lc) * 256 + threadIdx.x;
  const bool live = i0 < total;
  if (!STATS && !live) return;
  const long i = STATS ? (live ? i0 : total - 1) : i0;
  int cq, ow0, oh0, n;
  if (!STATS && g.img_major) {
    // small maps: consecutive threads = 8 channel quads x every (row pair, column block) of ONE image, so the rows a unit shares
    // with its neighbours are fetched by the same CU close in time (L1 hits) instead of by workgroups on other CUs: the kernels
    // wait on memory (SQ_WAIT_ANY 50-67 % of the wave cycles) with every quad requested (KS + ST) / 2 x 1.5 times from L2
    const int units = hb_n * wb_n, cgs = c4 >> 3;
    long t = i >> 3;
    const int unit = (int)(t % units); t /= units;
    cq = (int)(t % cgs) * 8 + (int)(i & 7);
    n = (int)(t / cgs);
    ow0 = (unit % wb_n) * TW; oh0 = (unit / wb_n) * TH;
  } else {
    cq = (int)(i % c4);
    long t = i / c4;
    ow0 = (int)(t % wb_n) * TW; t /= wb_n;
    oh0 = (int)(t % hb_n) * TH;
    n = (int)(t / hb_n);
  }
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* w4 = reinterpret_cast<const float4*>(w);
  float4 acc[TH][TW];
#pragma unroll
  for (int a = 0; a < TH; ++a)
#pragma unroll
    for (int q = 0; q < TW; ++q) acc[a][q] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int iw0 = ow0 * ST - g.pad_l, ih0 = oh0 * ST - g.pad_t;
#pragma unroll 1
  for (int ir = 0; ir < NR; ++ir) {
    const int ih = ih0 + ir;
    const bool rok = (unsigned)ih < (unsigned)g.H;
    const long rbase = ((long)n * g.H + (rok ? ih : 0)) * g.W;
    float4 xr[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int iw = iw0 + j;
      const bool ok = rok && (unsigned)iw < (unsigned)g.W;
      xr[j] = x4[ok ? (rbase + iw) * c4 + cq : 0];
      if (!ok) xr[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int a = 0; a < TH; ++a) {
      const int r = ir - a * ST;                        // kernel row of this input row for output row oh0 + a (wave-uniform)
      if (r >= 0 && r < KS) {
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
          const float4 k4 = w4[(FLIP ? (KS - 1 - r) * KS + (KS - 1 - s_) : r * KS + s_) * c4 + cq];
#pragma unroll
          for (int q = 0; q < TW; ++q) {
            const float4 v = xr[q * ST + s_];
            acc[a][q].x = fmaf(v.x, k4.x, acc[a][q].x); acc[a][q].y = fmaf(v.y, k4.y, acc[a][q].y);
            acc[a][q].z = fmaf(v.z, k4.z, acc[a][q].z); acc[a][q].w = fmaf(v.w, k4.w, acc[a][q].w);
          }
        }
      }
    }
  }
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  float4 bsc = s1, bsh = s1, bmu = s1, brs = s1;
  if (STATS == 2) {
    bsc = reinterpret_cast<const float4*>(bn.scale)[cq]; bsh = reinterpret_cast<const float4*>(bn.shift)[cq];
    bmu = reinterpret_cast<const float4*>(bn.mean)[cq]; brs = reinterpret_cast<const float4*>(bn.rstd)[cq];
  }
#pragma unroll
  for (int a = 0; a < TH; ++a) {
    if (oh0 + a >= g.OH || (STATS && !live)) break;
    const long o0 = (((long)n * g.OH + oh0 + a) * g.OW + ow0) * c4 + cq;
    float4* yo = reinterpret_cast<float4*>(y) + o0;
    float4 ev[TW];
    if (STATS == 2) {
#pragma unroll
      for (int q = 0; q < TW; ++q) ev[q] = reinterpret_cast<const float4*>(bn.e)[o0 + (long)(ow0 + q < g.OW ? q : 0) * c4];
    }
#pragma unroll
    for (int q = 0; q < TW; ++q)
      if (ow0 + q < g.OW) {
        const float4 v = acc[a][q];
        yo[(long)q * c4] = v;
        if (STATS == 1) {
          s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
          s2.x = fmaf(v.x, v.x, s2.x); s2.y = fmaf(v.y, v.y, s2.y); s2.z = fmaf(v.z, v.z, s2.z); s2.w = fmaf(v.w, v.w, s2.w);
        }
        if (STATS == 2) dw_bn_sums_add(bn, v, ev[q], bsc, bsh, bmu, brs, s1, s2);
      }
  }
  if (STATS) dw_block_accumulate(s1, s2, c4, (long)blockIdx.x * L + lc, dw_accq);
  }
  if (STATS) dw_block_flush(c4, g.C, dw_accq, stats);
}

// Stride-2 data gradient: dx[ih,iw] = sum over (r,s) with (ih+pt-r) and (iw+pl-s) even of dy[(ih+pt-r)/2, (iw+pl-s)/2] * w[r,s].
// Block of 4 dx columns starting at a multiple of 4, so which taps are live in each column only depends on the column's
// offset t and the parity PLP of pad_l: s = s0(t) + 2b with s0 = (t + PLP) & 1, and the dy column is
// base + e(t) - b, e(t) = (t + PLP - s0) / 2, base = w0/2 + (pad_l - PLP)/2.  All register indices are static.
// STATS = 2: also the BatchNorm-backward sums of the layer in front of the depthwise conv (dwconv_row4x2_kernel STATS = 2).
template <int KS, int PLP, int STATS = 0>
__global__ __launch_bounds__(256) void dwconv_dgrad4_s2_row_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                   DwGeom g, float* __restrict__ dx, float* __restrict__ stats = nullptr,
                                                                   const DwBn bn = DwBn{nullptr, nullptr, nullptr, nullptr, nullptr, 0}) {
  constexpr int TW = DW_TW, OFF = (KS - 1) / 2, MAXE = PLP ? 2 : 1, NX = MAXE + OFF + 1;
  extern __shared__ float4 dw_accq[];                     // STATS: 2 * c4 float4
  const int c4 = g.C >> 2, wb_n = (g.W + TW - 1) / TW;
  const long total = (long)g.N * g.H * wb_n * c4;
  const int L = STATS ? (g.L > 1 ? g.L : 1) : 1;
  if (STATS) dw_block_begin(c4, dw_accq);
  for (int lc = 0; lc < L; ++lc) {
  const long i0 = ((long)blockIdx.x * L + lc) * 256 + threadIdx.x;
  const bool live = i0 < total;
  if (!STATS && !live) return;
  const long i = STATS ? (live ? i0 : total - 1) : i0;
  const int cq = (int)(i % c4);
  long t = i / c4;
  const int w0 = (int)(t % wb_n) * TW; t /= wb_n;
  const int ih = (int)(t % g.H);
  const int n = (int)(t / g.H);
  const float4* d4 = reinterpret_cast<const float4*>(dy);
  const float4* w4 = reinterpret_cast<const float4*>(w);
  float4 acc[TW];
#pragma unroll
  for (int q = 0; q < TW; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int r0 = (ih + g.pad_t) & 1;                     // rows r = r0, r0+2, .. reach this dx row
  const int base = (w0 >> 1) + ((g.pad_l - PLP) >> 1) - OFF;
  const long o0 = (((long)n * g.H + ih) * g.W + w0) * c4 + cq;
  // STATS = 2: the BatchNormalization input at this thread's outputs, requested in FRONT of the taps (behind them, under the
  // `live` branch, it was one more exposed memory latency per thread: 14x14x672 5x5 at 1.3 TB/s)
  float4 ev[STATS == 2 ? TW : 1];
  if (STATS == 2) {
#pragma unroll
    for (int q = 0; q < TW; ++q) ev[q] = reinterpret_cast<const float4*>(bn.e)[o0 + (long)(w0 + q < g.W ? q : 0) * c4];
  }
#pragma unroll KS == 3 ? 2 : 1
  for (int a = 0; a < (KS + 1) / 2; ++a) {
    const int r = r0 + 2 * a;
    const int th = ih + g.pad_t - r;
    const int oh = th >> 1;
    const bool rok = r < KS && th >= 0 && oh < g.OH;
    const long rbase = ((long)n * g.OH + (rok ? oh : 0)) * g.OW;
    float4 dr[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int ow = base + j;
      const bool ok = rok && (unsigned)ow < (unsigned)g.OW;
      dr[j] = d4[ok ? (rbase + ow) * c4 + cq : 0];
      if (!ok) dr[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float4* wr = w4 + (long)(rok ? r : 0) * KS * c4 + cq;
#pragma unroll
    for (int q = 0; q < TW; ++q) {
      constexpr int dummy = 0; (void)dummy;
      const int s0 = (q + PLP) & 1, e = (q + PLP - s0) / 2;
#pragma unroll
      for (int b = 0; b < (KS + 1) / 2; ++b) {
        if (s0 + 2 * b < KS) {                            // compile-time after unrolling
          const float4 wv = wr[(long)(s0 + 2 * b) * c4];
          const float4 v = dr[e - b + OFF];
          acc[q].x = fmaf(v.x, wv.x, acc[q].x); acc[q].y = fmaf(v.y, wv.y, acc[q].y);
          acc[q].z = fmaf(v.z, wv.z, acc[q].z); acc[q].w = fmaf(v.w, wv.w, acc[q].w);
        }
      }
    }
  }
  float4* xo = reinterpret_cast<float4*>(dx) + o0;
  if (!STATS) {
#pragma unroll
    for (int q = 0; q < TW; ++q)
      if (w0 + q < g.W) xo[(long)q * c4] = acc[q];
    return;
  }
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  if (live) {
    const float4 bsc = reinterpret_cast<const float4*>(bn.scale)[cq], bsh = reinterpret_cast<const float4*>(bn.shift)[cq];
    const float4 bmu = reinterpret_cast<const float4*>(bn.mean)[cq], brs = reinterpret_cast<const float4*>(bn.rstd)[cq];
#pragma unroll
    for (int q = 0; q < TW; ++q)
      if (w0 + q < g.W) {
        xo[(long)q * c4] = acc[q];
        dw_bn_sums_add(bn, acc[q], ev[STATS == 2 ? q : 0], bsc, bsh, bmu, brs, s1, s2);
      }
  }
  dw_block_accumulate(s1, s2, c4, (long)blockIdx.x * L + lc, dw_accq);
  }
  dw_block_flush(c4, g.C, dw_accq, stats);
}

// Weight gradient, same column blocking, one WAVE per kernel row (workgroup = KS waves): wave r accumulates dw[r, 0..KS)
// for its lanes' channel quads over the workgroup's slab of (row, 4-column block) units, so a thread holds KS accumulator
// quads instead of KS*KS (a 5x5 kernel's 25 plus a unit's image quads took 300+ registers whatever the loop order: one
// wave per SIMD; walking the slab once per kernel row with one accumulator row re-streamed it from HBM five times).
// The KS waves of a workgroup walk the same units at the same time: the 4 dy quads of a unit are fetched once and hit L1
// for the other waves, the image rows overlap between neighbouring units.  Per unit and wave: 4 + (TW-1)*ST+KS loads for
// 4*KS FMAs per channel.  Lanes = (channel-quad lanes) x (unit lanes); the unit lanes are summed with a fixed xor
// butterfly, the slabs by dw_slab_sum_kernel: bitwise reproducible.
template <int KS, int ST>
__global__ __launch_bounds__(64 * KS) void dwconv_wgrad4_wave_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                     DwGeom g, int cq_lanes, int units_per_block,
                                                                     float* __restrict__ partial, int colmajor) {
  constexpr int TW = DW_TW, NX = (TW - 1) * ST + KS, TAPS = KS * KS;
  // the TW dy quads of a unit are the same for the KS waves: each is fetched by ONE wave and handed over through LDS (two
  // buffers, one barrier per trip) — 4 instead of 4*KS of the workgroup's (4 + NX)*KS wave-wide L1 requests per unit
  __shared__ float4 dsh[2][TW][64];
  const int c4 = g.C >> 2, unit_lanes = 64 / cq_lanes, wb_n = (g.OW + TW - 1) / TW;
  const int r = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cl = lane % cq_lanes, ul = lane / cq_lanes;
  const int nunits = g.N * g.OH * wb_n;
  const int u0 = blockIdx.x * units_per_block, u1 = min(u0 + units_per_block, nunits);
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* d4 = reinterpret_cast<const float4*>(dy);
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  {                                                        // grid.y = channel-quad group: late layers have few pixels, many channels
    const int cq = blockIdx.y * cq_lanes + cl;
    const bool cok = cq < c4;
    float4 acc[KS];
#pragma unroll
    for (int j = 0; j < KS; ++j) acc[j] = z4;
    const int trips = (u1 - u0 + unit_lanes - 1) / unit_lanes;          // the same for every wave of the workgroup
    for (int it = 0; it < trips; ++it) {
      const int uu = u0 + it * unit_lanes + ul;
      const bool live = cok && uu < u1;
      const int u = min(uu, u1 - 1);
      // colmajor: consecutive units run DOWN a 4-column block of an image, so the image row wave r reads for unit oh + 1
      // is the one wave r + 1 read for unit oh a moment ago (L1 hit) — row-major, the KS reads of an image row are a whole
      // row of units apart and each comes from L2: (KS + 1) / 2 x the compulsory bytes on the L2 -> L1 path
      int row, ow0, n, oh;
      if (colmajor) {
        const int per_img = g.OH * wb_n;
        n = u / per_img; const int rem = u - n * per_img, wb = rem / g.OH;
        oh = rem - wb * g.OH; ow0 = wb * TW; row = n * g.OH + oh;
      } else {
        row = u / wb_n; ow0 = (u - row * wb_n) * TW;
        n = row / g.OH; oh = row - n * g.OH;
      }
      const int iw0 = ow0 * ST - g.pad_l, ih = oh * ST + r - g.pad_t;
      const bool rok = live && (unsigned)ih < (unsigned)g.H;
      const long rbase = ((long)n * g.H + (rok ? ih : 0)) * g.W;
      float4 xr[NX];
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        const int iw = iw0 + j;
        const bool ok = rok && (unsigned)iw < (unsigned)g.W;
        xr[j] = x4[ok ? (rbase + iw) * c4 + cq : 0];
        if (!ok) xr[j] = z4;
      }
      const long dbase = ((long)row * g.OW + ow0) * c4 + cq;
#pragma unroll
      for (int q = 0; q < TW; ++q)
        if (q % KS == r) {                                 // this wave's share of the unit's dy quads
          const bool ok = live && ow0 + q < g.OW;
          float4 v = d4[ok ? dbase + (long)q * c4 : 0];
          if (!ok) v = z4;
          dsh[it & 1][q][lane] = v;
        }
      __syncthreads();
      float4 d[TW];
#pragma unroll
      for (int q = 0; q < TW; ++q) d[q] = dsh[it & 1][q][lane];
#pragma unroll
      for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
        for (int q = 0; q < TW; ++q) {
          const float4 v = xr[q * ST + s_];
          acc[s_].x = fmaf(v.x, d[q].x, acc[s_].x); acc[s_].y = fmaf(v.y, d[q].y, acc[s_].y);
          acc[s_].z = fmaf(v.z, d[q].z, acc[s_].z); acc[s_].w = fmaf(v.w, d[q].w, acc[s_].w);
        }
    }
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      float4 a = acc[j];
      for (int o = cq_lanes; o < 64; o <<= 1) {           // unit lanes of one channel quad: lanes cl, cl+cq_lanes, ..
        a.x += __shfl_xor(a.x, o, 64); a.y += __shfl_xor(a.y, o, 64);
        a.z += __shfl_xor(a.z, o, 64); a.w += __shfl_xor(a.w, o, 64);
      }
      if (ul == 0 && cok)
        reinterpret_cast<float4*>(partial)[((long)blockIdx.x * TAPS + r * KS + j) * c4 + cq] = a;
    }
  }
}

// dw[r,s,c] = sum_{n,oh,ow} x[n, oh*st+r-pt, ow*st+s-pl, c] * dy[n,oh,ow,c]
// Workgroup = (channel-quad lanes) x (pixel lanes) over a slab of output pixels.  Each thread keeps a
// float4 accumulator per tap for its 4 channels, walks its pixels (one 16-byte dy load + one 16-byte x
// load per tap, neighbours served by L1/L2), then the pixel lanes are combined through LDS and the
// workgroup writes partial[block][tap][c]; a second kernel adds the slabs in fixed order (double).
constexpr int DW_MAX_TAPS = 49;
// KS > 0: square KS x KS kernel, taps known at compile time, loads unconditional (see above); KS = 0: any R x S
// up to MAXT taps, with a branch per tap.
template <int MAXT, int KS>
__global__ __launch_bounds__(256) void dwconv_wgrad4_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            DwGeom g, int cq_lanes, int pixels_per_block,
                                                            float* __restrict__ partial) {
  __shared__ float4 sh[256];
  const int taps = g.R * g.S, c4 = g.C >> 2, pix_lanes = 256 / cq_lanes;
  const int cl = threadIdx.x % cq_lanes, pl = threadIdx.x / cq_lanes;
  const long npix = (long)g.N * g.OH * g.OW;
  const long p0 = (long)blockIdx.x * pixels_per_block, p1 = min(p0 + pixels_per_block, npix);
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* d4 = reinterpret_cast<const float4*>(dy);
  for (int cq0 = 0; cq0 < c4; cq0 += cq_lanes) {
    const int cq = cq0 + cl;
    float4 acc[MAXT];
#pragma unroll
    for (int j = 0; j < MAXT; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cq < c4) {
      for (long p = p0 + pl; p < p1; p += pix_lanes) {
        long t = p;
        const int ow = (int)(t % g.OW); t /= g.OW;
        const int oh = (int)(t % g.OH);
        const int n = (int)(t / g.OH);
        const float4 d = d4[p * c4 + cq];
        if (KS > 0) {
#pragma unroll
          for (int j = 0; j < MAXT; ++j) {
            const int ih = oh * g.stride + j / (KS > 0 ? KS : 1) - g.pad_t, iw = ow * g.stride + j % (KS > 0 ? KS : 1) - g.pad_l;
            const bool ok = (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
            const float4 v = x4[ok ? (((long)n * g.H + ih) * g.W + iw) * c4 + cq : 0];
            const float4 dm = ok ? d : make_float4(0.f, 0.f, 0.f, 0.f);
            acc[j].x = fmaf(v.x, dm.x, acc[j].x); acc[j].y = fmaf(v.y, dm.y, acc[j].y);
            acc[j].z = fmaf(v.z, dm.z, acc[j].z); acc[j].w = fmaf(v.w, dm.w, acc[j].w);
          }
        } else {
          int r = 0, s = 0;
#pragma unroll
          for (int j = 0; j < MAXT; ++j) {
            if (j < taps) {
              const int ih = oh * g.stride + r - g.pad_t, iw = ow * g.stride + s - g.pad_l;
              if ((unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W) {
                const float4 v = x4[(((long)n * g.H + ih) * g.W + iw) * c4 + cq];
                acc[j].x = fmaf(v.x, d.x, acc[j].x); acc[j].y = fmaf(v.y, d.y, acc[j].y);
                acc[j].z = fmaf(v.z, d.z, acc[j].z); acc[j].w = fmaf(v.w, d.w, acc[j].w);
              }
              if (++s == g.S) { s = 0; ++r; }
            }
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      if (j < taps) {                                   // workgroup-uniform
        sh[threadIdx.x] = acc[j];
        __syncthreads();
        if (pl == 0 && cq < c4) {
          float4 a = acc[j];
          for (int k = 1; k < pix_lanes; ++k) {
            const float4 o = sh[k * cq_lanes + cl];
            a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
          }
          reinterpret_cast<float4*>(partial)[((long)blockIdx.x * taps + j) * c4 + cq] = a;
        }
        __syncthreads();
      }
    }
  }
}

// scalar-channel fallback (C % 4 != 0): one thread per channel, serial over the slab
__global__ __launch_bounds__(256) void dwconv_wgrad1_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            DwGeom g, int pixels_per_block, float* __restrict__ partial) {
  const int taps = g.R * g.S;
  const long npix = (long)g.N * g.OH * g.OW;
  const long p0 = (long)blockIdx.x * pixels_per_block, p1 = min(p0 + pixels_per_block, npix);
  for (int c = threadIdx.x; c < g.C; c += 256)
    for (int j = 0; j < taps; ++j) {
      const int r = j / g.S, s = j - r * g.S;
      float acc = 0.f;
      for (long p = p0; p < p1; ++p) {
        long t = p;
        const int ow = (int)(t % g.OW); t /= g.OW;
        const int oh = (int)(t % g.OH);
        const int n = (int)(t / g.OH);
        const int ih = oh * g.stride + r - g.pad_t, iw = ow * g.stride + s - g.pad_l;
        if ((unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W)
          acc = fmaf(x[(((long)n * g.H + ih) * g.W + iw) * g.C + c], dy[p * g.C + c], acc);
      }
      partial[((long)blockIdx.x * taps + j) * g.C + c] = acc;
    }
}

// out[i] = sum_b partial[b][i]: 16 outputs x 16 slab groups per workgroup (slab b goes to group b % 16, two
// independent double accumulators per thread), groups combined through LDS in a fixed tree.  With up to 2048 slabs
// a 4-group walk kept one thread on 512 dependent loads (62 us per layer on EfficientNet-B0); bitwise reproducible.
__global__ __launch_bounds__(256) void dw_slab_sum_kernel(const float* __restrict__ partial, int blocks, long n,
                                                          float* __restrict__ out) {
  __shared__ double sh[256];
  const int ol = threadIdx.x & 15, bl = threadIdx.x >> 4;
  const long i = (long)blockIdx.x * 16 + ol;
  double s0 = 0.0, s1 = 0.0;
  if (i < n) {
    int b = bl;
    for (; b + 16 < blocks; b += 32) { s0 += (double)partial[(long)b * n + i]; s1 += (double)partial[(long)(b + 16) * n + i]; }
    if (b < blocks) s0 += (double)partial[(long)b * n + i];
  }
  sh[threadIdx.x] = s0 + s1;
  __syncthreads();
  for (int g = 8; g >= 1; g >>= 1) {
    if (bl < g) sh[threadIdx.x] += sh[threadIdx.x + g * 16];
    __syncthreads();
  }
  if (bl == 0 && i < n) out[i] = (float)sh[ol];
}

// ---- activations ------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + __expf(-v)); }

// kind 0: sigmoid, 1: swish (x * sigmoid(x))
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, long total, int kind,
                                                      float* __restrict__ y) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const float v = x[i], s = sigmoidf(v);
    y[i] = kind ? v * s : s;
  }
}
// dx from x (recomputed): sigmoid' = s(1-s); swish' = s + x s (1-s)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                      long total, int kind, float* __restrict__ dx) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const float v = x[i], s = sigmoidf(v);
    dx[i] = dy[i] * (kind ? s + v * s * (1.f - s) : s * (1.f - s));
  }
}

// ---- squeeze-excite scaling: y[n,p,c] = x[n,p,c] * s[n,c] ---------------------------------------
__global__ __launch_bounds__(256) void chscale_fwd_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                          long total, int hw, int c, float* __restrict__ y) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int col = (int)(i % c);
    const long n = i / ((long)hw * c);
    y[i] = x[i] * s[n * c + col];
  }
}
// dx = dy * s ; ds[n,c] = sum_p dy*x.  Workgroup = one sample x 64 channels: 16 channel-quad lanes x 16
// pixel lanes, float4 accesses, pixel lanes combined through LDS.
__global__ __launch_bounds__(256) void chscale_bwd4_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                           const float* __restrict__ dy, int hw, int c4,
                                                           float* __restrict__ dx, float* __restrict__ ds) {
  __shared__ float4 sh[256];
  const int n = blockIdx.y, cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int cq = blockIdx.x * 16 + cl;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cq < c4) {
    const float4 sv = reinterpret_cast<const float4*>(s)[(long)n * c4 + cq];
    for (int p = pl; p < hw; p += 16) {
      const long i = ((long)n * hw + p) * c4 + cq;
      const float4 d = reinterpret_cast<const float4*>(dy)[i], xv = reinterpret_cast<const float4*>(x)[i];
      reinterpret_cast<float4*>(dx)[i] = make_float4(d.x * sv.x, d.y * sv.y, d.z * sv.z, d.w * sv.w);
      acc.x = fmaf(d.x, xv.x, acc.x); acc.y = fmaf(d.y, xv.y, acc.y);
      acc.z = fmaf(d.z, xv.z, acc.z); acc.w = fmaf(d.w, xv.w, acc.w);
    }
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  if (pl == 0 && cq < c4) {
    for (int k = 1; k < 16; ++k) { const float4 o = sh[k * 16 + cl]; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
    reinterpret_cast<float4*>(ds)[(long)n * c4 + cq] = acc;
  }
}

// ds only (the scaled tensor's own gradient dy * s is formed by its consumer: embnet_bn_bwd_gap's gate argument): 8 B per element
__global__ __launch_bounds__(256) void chscale_dgate4_kernel(const float* __restrict__ x, const float* __restrict__ dy, int hw, int c4,
                                                             float* __restrict__ ds) {
  __shared__ float4 sh[256];
  const int n = blockIdx.y, cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int cq = blockIdx.x * 16 + cl;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cq < c4) {
    for (int p = pl; p < hw; p += 16) {                  // the accumulation order of chscale_bwd4_kernel
      const long i = ((long)n * hw + p) * c4 + cq;
      const float4 d = reinterpret_cast<const float4*>(dy)[i], xv = reinterpret_cast<const float4*>(x)[i];
      acc.x = fmaf(d.x, xv.x, acc.x); acc.y = fmaf(d.y, xv.y, acc.y);
      acc.z = fmaf(d.z, xv.z, acc.z); acc.w = fmaf(d.w, xv.w, acc.w);
    }
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  if (pl == 0 && cq < c4) {
    for (int k = 1; k < 16; ++k) { const float4 o = sh[k * 16 + cl]; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
    reinterpret_cast<float4*>(ds)[(long)n * c4 + cq] = acc;
  }
}

__global__ __launch_bounds__(256) void chscale_bwd_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                          const float* __restrict__ dy, int hw, int c,
                                                          float* __restrict__ dx, float* __restrict__ ds) {
  const int n = blockIdx.y;
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= c) return;
  const float sv = s[(long)n * c + col];
  float acc = 0.f;
  for (int p = 0; p < hw; ++p) {
    const long i = ((long)n * hw + p) * c + col;
    const float d = dy[i];
    dx[i] = d * sv;
    acc = fmaf(d, x[i], acc);
  }
  ds[(long)n * c + col] = acc;
}

// ---- per-sample drop-connect (Dropout with noise_shape (None,1,1,1)), inverted scaling ---------------
__global__ __launch_bounds__(256) void sample_dropout_kernel(const float* __restrict__ x, long total, long per_sample,
                                                             float rate, uint64_t seed, const uint64_t* __restrict__ seed_add,
                                                             float* __restrict__ y) {
  if (seed_add) seed += *seed_add;
  const float keep_scale = 1.f / (1.f - rate);
  const uint32_t thr = (uint32_t)((double)rate * 4294967296.0);
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride)
    y[i] = rng_u32(seed, (uint64_t)(i / per_sample), 2) >= thr ? x[i] * keep_scale : 0.f;
}

// ---- |a - b| ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void absdiff_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          long total, float* __restrict__ y) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) y[i] = fabsf(a[i] - b[i]);
}
__global__ __launch_bounds__(256) void absdiff_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ dy, long total,
                                                          float* __restrict__ da, float* __restrict__ db) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const float d = a[i] - b[i];
    const float g = d > 0.f ? dy[i] : (d < 0.f ? -dy[i] : 0.f);
    da[i] = g; db[i] = -g;
  }
}

}  // namespace embnet

using namespace embnet;
#define S(stream) ((hipStream_t)(stream))
static inline int ew_blocks(long total) { long b = (total + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

static int make_dw(DwGeom& g, int n, int h, int w, int c, int r, int s, int stride, int pad_t, int pad_l, int oh, int ow,
                   const char* who) {
  EMBNET_CHECK_ARG(n > 0 && h > 0 && w > 0 && c > 0 && r > 0 && s > 0 && stride > 0 && oh > 0 && ow > 0 && pad_t >= 0 &&
                   pad_l >= 0, "%s: bad geometry", who);
  EMBNET_CHECK_ARG(r * s <= DW_MAX_TAPS, "%s: kernel %dx%d larger than 7x7 unsupported", who, r, s);
  EMBNET_CHECK_ARG((oh - 1) * stride + 1 - pad_t <= h && (ow - 1) * stride + 1 - pad_l <= w,
                   "%s: output reaches outside the input", who);
  g = DwGeom{n, h, w, c, r, s, stride, pad_t, pad_l, oh, ow};
  return 0;
}

static bool dw_wide(const DwGeom& g) {
  static const int forced = (int)env_long("EMBNET_DW_TW", 0);            // 4 / 8: A/B
  // eight columns per thread unless that wastes more than an eighth of a row the four-column blocks tile exactly
  return forced ? forced == 8 : (g.OW >= 7 && cdiv(g.OW, 8) * 8 <= cdiv(g.OW, 4) * 4 + g.OW / 8);
}
static bool dw_rows2(const DwGeom& g) { static const int rows2 = (int)env_long("EMBNET_DW_ROWS2", 1); return rows2 && g.OH >= 2; }
// statistics variants: 256-thread chunks per workgroup, so that a launch writes at most ~2048 rows of partials (dw_block_accumulate)
static int dw_stats_chunks(long chunks) {
  static const long target = env_long("EMBNET_DW_STATS_ROWS", 2048);
  return chunks > target && target > 0 ? (int)cdiv(chunks, target) : 1;
}
static long dw_rows2_grid(const DwGeom& g) {
  return cdiv((long)g.N * cdiv(g.OH, 2) * (g.C / 4) * cdiv(g.OW, dw_wide(g) ? 8 : 4), 256);
}

template <int KS, int ST, bool FLIP>
static void launch_dw_rows(const float* x, const float* w, const DwGeom& g_in, float* y, hipStream_t st, float* stats = nullptr,
                           const DwBn* bn = nullptr) {
  DwGeom g = g_in;
  if (ST == 1 && dwt::tile_applies(g, !stats ? 0 : (FLIP ? 2 : 1)) && (!stats || !FLIP || bn)) {   // small maps: LDS-tile kernel (dwconv_tile.hip)
    dwt::launch_tile(x, w, g, FLIP, y, stats, bn, st);
    return;
  }
  static const long img_max = env_long("EMBNET_DW_IMG_MAX", 0);         // experiment: image-major thread order for maps up to this many pixels
  g.img_major = (!stats && img_max > 0 && (long)g.OH * g.OW <= img_max && ((g.C / 4) & 7) == 0) ? 1 : 0;
  const bool wide = dw_wide(g);
  if (dw_rows2(g)) {                                                      // two output rows per thread
    const long grid = dw_rows2_grid(g);
    g.L = dw_stats_chunks(grid);
    const long sgrid = cdiv(grid, g.L);                                  // the statistics variants' workgroups = rows of partials
    const size_t accq = (size_t)g.C * 8;                                  // 2 * C/4 float4
    if (stats && !FLIP) {
      if (wide) dwconv_row4x2_kernel<KS, ST, false, 8, 1><<<sgrid, 256, accq, st>>>(x, w, g, y, stats);
      else dwconv_row4x2_kernel<KS, ST, false, 4, 1><<<sgrid, 256, accq, st>>>(x, w, g, y, stats);
      return;
    }
    if (stats && FLIP && ST == 1 && bn) {                                // data gradient + BatchNorm-backward sums
      if (wide) dwconv_row4x2_kernel<KS, 1, true, 8, 2><<<sgrid, 256, accq, st>>>(x, w, g, y, stats, *bn);
      else dwconv_row4x2_kernel<KS, 1, true, 4, 2><<<sgrid, 256, accq, st>>>(x, w, g, y, stats, *bn);
      return;
    }
    if (wide) dwconv_row4x2_kernel<KS, ST, FLIP, 8><<<grid, 256, 0, st>>>(x, w, g, y);
    else dwconv_row4x2_kernel<KS, ST, FLIP, 4><<<grid, 256, 0, st>>>(x, w, g, y);
    return;
  }
  if (wide) dwconv_row4_kernel<KS, ST, FLIP, 8><<<cdiv((long)g.N * g.OH * cdiv(g.OW, 8) * (g.C / 4), 256), 256, 0, st>>>(x, w, g, y);
  else dwconv_row4_kernel<KS, ST, FLIP, 4><<<cdiv((long)g.N * g.OH * cdiv(g.OW, 4) * (g.C / 4), 256), 256, 0, st>>>(x, w, g, y);
}

// EMBNET_DW_ROWS=0 falls back to the per-pixel kernels (A/B)
static bool dw_rows() { static const bool on = env_long("EMBNET_DW_ROWS", 1) != 0; return on; }

static bool dw_fwd_rows_path(int c, int r, int s, int stride) {
  return (c & 3) == 0 && r == s && (r == 3 || r == 5) && (stride == 1 || stride == 2) && dw_rows();
}

// rows P of the statistics partials [2][c][P] embnet_dwconv2d_fwd_stats_f32 writes for this geometry (0: not available — the
// two-rows-per-thread kernels only).  Every (channel, row) is written; callers that zero wide buffers first (c / 4 > 256: older ABI) still may
extern "C" int embnet_dwconv2d_fwd_stats_rows(int n, int c, int r, int s, int stride, int oh, int ow) {
  if (n <= 0 || c <= 0 || oh <= 0 || ow <= 0 || !dw_fwd_rows_path(c, r, s, stride)) return 0;
  DwGeom g{n, 0, 0, c, r, s, stride, 0, 0, oh, ow};
  if (stride == 1) {                                       // same-size layers on small maps: the LDS-tile kernel's rows
    const DwGeom gs{n, oh, ow, c, r, s, 1, (r - 1) / 2, (s - 1) / 2, oh, ow};
    if (dwt::tile_applies(gs, 1)) return dwt::tile_stats_rows(gs, 1);
  }
  if (!dw_rows2(g)) return 0;
  const long grid = dw_rows2_grid(g);
  return (int)cdiv(grid, dw_stats_chunks(grid));
}

static int dwconv2d_fwd_impl(const float* x, const float* w, float* y, int n, int h, int wd, int c, int r,
                             int s, int stride, int pad_t, int pad_l, int oh, int ow, float* stats, void* stream) {
  EMBNET_CHECK_ARG(x && w && y, "dwconv2d_fwd: null pointer");
  DwGeom g;
  if (int rc = make_dw(g, n, h, wd, c, r, s, stride, pad_t, pad_l, oh, ow, "dwconv2d_fwd")) return rc;
  const long total = (long)n * oh * ow * c;
  const int grid4 = cdiv(total / 4, 256);
  EMBNET_CHECK_ARG(!stats || embnet_dwconv2d_fwd_stats_rows(n, c, r, s, stride, oh, ow) > 0, "dwconv2d_fwd_stats: statistics are not available for this geometry");
  if (stats && stride == 1) {                              // the rows function assumed a same-size layer
    const DwGeom gs{n, oh, ow, c, r, s, 1, (r - 1) / 2, (s - 1) / 2, oh, ow};
    EMBNET_CHECK_ARG(dwt::tile_applies(gs, 1) == dwt::tile_applies(g, 1) && (!dwt::tile_applies(g, 1) || dwt::tile_stats_rows(gs, 1) == dwt::tile_stats_rows(g, 1)),
                     "dwconv2d_fwd_stats: statistics are not available for this geometry");
  }
  if (dw_fwd_rows_path(c, r, s, stride)) {
    EMBNET_TRACE(stride == 1 && dwt::tile_applies(g, stats ? 1 : 0) ? "embnet::dwt::dw_tile_kernel" : "embnet::dwconv_row4_kernel", TRACE_BYTES, 4.0 * total + 4.0 * n * h * wd * c, stream);
    if (r == 3 && stride == 1) launch_dw_rows<3, 1, false>(x, w, g, y, S(stream), stats);
    else if (r == 3) launch_dw_rows<3, 2, false>(x, w, g, y, S(stream), stats);
    else if (stride == 1) launch_dw_rows<5, 1, false>(x, w, g, y, S(stream), stats);
    else launch_dw_rows<5, 2, false>(x, w, g, y, S(stream), stats);
    return check_launch("dwconv2d_fwd");
  }
  if ((c & 3) == 0 && r == s && r == 3) { EMBNET_TRACE("embnet::dwconv_fwd4_sq_kernel", TRACE_BYTES, 4.0 * total + 4.0 * n * h * wd * c, stream); dwconv_fwd4_sq_kernel<3><<<grid4, 256, 0, S(stream)>>>(x, w, g, y); }
  else if ((c & 3) == 0 && r == s && r == 5) { EMBNET_TRACE("embnet::dwconv_fwd4_sq_kernel", TRACE_BYTES, 4.0 * total + 4.0 * n * h * wd * c, stream); dwconv_fwd4_sq_kernel<5><<<grid4, 256, 0, S(stream)>>>(x, w, g, y); }
  else if ((c & 3) == 0) { EMBNET_TRACE("embnet::dwconv_fwd_kernel", TRACE_BYTES, 4.0 * total + 4.0 * n * h * wd * c, stream); dwconv_fwd_kernel<4><<<grid4, 256, 0, S(stream)>>>(x, w, g, y); }
  else { EMBNET_TRACE("embnet::dwconv_fwd_kernel", TRACE_BYTES, 4.0 * total + 4.0 * n * h * wd * c, stream); dwconv_fwd_kernel<1><<<cdiv(total, 256), 256, 0, S(stream)>>>(x, w, g, y); }
  return check_launch("dwconv2d_fwd");
}

extern "C" int embnet_dwconv2d_fwd_f32(const float* x, const float* w, float* y, int n, int h, int wd, int c, int r,
                                       int s, int stride, int pad_t, int pad_l, int oh, int ow, void* stream) {
  return dwconv2d_fwd_impl(x, w, y, n, h, wd, c, r, s, stride, pad_t, pad_l, oh, ow, nullptr, stream);
}

// ... with the statistics partials of the BatchNormalization that follows (see dwconv_row4x2_kernel STATS)
extern "C" int embnet_dwconv2d_fwd_stats_f32(const float* x, const float* w, float* y, int n, int h, int wd, int c, int r,
                                             int s, int stride, int pad_t, int pad_l, int oh, int ow, float* stats, void* stream) {
  EMBNET_CHECK_ARG(stats, "dwconv2d_fwd_stats: null pointer");
  return dwconv2d_fwd_impl(x, w, y, n, h, wd, c, r, s, stride, pad_t, pad_l, oh, ow, stats, stream);
}

// rows P of the [2][c][P] BatchNorm-backward partial sums embnet_dwconv2d_dgrad_bnsums_f32 writes (0: not available — stride-1
// layers on the two-rows-per-thread kernel and stride-2 layers on dwconv_dgrad4_s2_row_kernel only); when c / 4 > 256 the caller
// zeroes the buffer first
extern "C" int embnet_dwconv2d_dgrad_bnsums_rows(int n, int h, int wd, int c, int r, int s, int stride) {
  if (n <= 0 || c <= 0 || h <= 0 || wd <= 0 || (stride != 1 && stride != 2) || !dw_fwd_rows_path(c, r, s, stride)) return 0;
  if (stride == 2) {                                     // dwconv_dgrad4_s2_row_kernel: one thread per (row, 4-column block, quad)
    const long grid = cdiv((long)n * h * cdiv(wd, DW_TW) * (c / 4), 256);
    return (int)cdiv(grid, dw_stats_chunks(grid));
  }
  DwGeom gf{n, 0, 0, c, r, s, 1, 0, 0, h, wd};
  {
    const DwGeom gs{n, h, wd, c, r, s, 1, (r - 1) / 2, (s - 1) / 2, h, wd};
    if (dwt::tile_applies(gs, 2)) return dwt::tile_stats_rows(gs, 2);
  }
  if (!dw_rows2(gf)) return 0;
  const long grid = dw_rows2_grid(gf);
  return (int)cdiv(grid, dw_stats_chunks(grid));
}

// Stride-1 depthwise data gradient that also emits the BatchNorm-backward sums of the layer in front of the depthwise conv (its
// input was act(BN(bn_x)); MBConv: expand_bn -> dwconv): see dwconv_row4x2_kernel STATS = 2 and embnet_conv2d_dgrad_bnsums_f32.
extern "C" int embnet_dwconv2d_dgrad_bnsums_f32(const float* dy, const float* w, float* dx, int n, int h, int wd, int c, int r,
                                                int s, int stride, int pad_t, int pad_l, int oh, int ow, const float* bn_x,
                                                const float* bn_scale, const float* bn_shift, const float* bn_mean,
                                                const float* bn_rstd, int bn_act, float* bn_partial, int bn_rows, void* stream) {
  EMBNET_CHECK_ARG(dy && w && dx && bn_x && bn_scale && bn_shift && bn_mean && bn_rstd && bn_partial, "dwconv2d_dgrad_bnsums: null pointer");
  EMBNET_CHECK_ARG(bn_act >= 0 && bn_act <= 2, "dwconv2d_dgrad_bnsums: activation code %d", bn_act);
  EMBNET_CHECK_ARG(bn_rows > 0 && bn_rows == embnet_dwconv2d_dgrad_bnsums_rows(n, h, wd, c, r, s, stride),
                   "dwconv2d_dgrad_bnsums: rows %d for this geometry (see embnet_dwconv2d_dgrad_bnsums_rows)", bn_rows);
  DwGeom g;
  if (int rc = make_dw(g, n, h, wd, c, r, s, stride, pad_t, pad_l, oh, ow, "dwconv2d_dgrad_bnsums")) return rc;
  const long total = (long)n * h * wd * c;
  const DwBn bn{bn_x, bn_scale, bn_shift, bn_mean, bn_rstd, bn_act};
  if (stride == 2) {
    EMBNET_TRACE("embnet::dwconv_dgrad4_s2_row_kernel", TRACE_BYTES, 8.0 * total + 4.0 * n * oh * ow * c, stream);
    const int gridr = bn_rows;
    g.L = dw_stats_chunks(cdiv((long)n * h * cdiv(wd, DW_TW) * (c / 4), 256));
    const size_t accq = (size_t)c * 8;                                    // 2 * C/4 float4
    if (r == 3) {
      if (pad_l & 1) dwconv_dgrad4_s2_row_kernel<3, 1, 2><<<gridr, 256, accq, S(stream)>>>(dy, w, g, dx, bn_partial, bn);
      else dwconv_dgrad4_s2_row_kernel<3, 0, 2><<<gridr, 256, accq, S(stream)>>>(dy, w, g, dx, bn_partial, bn);
    } else {
      if (pad_l & 1) dwconv_dgrad4_s2_row_kernel<5, 1, 2><<<gridr, 256, accq, S(stream)>>>(dy, w, g, dx, bn_partial, bn);
      else dwconv_dgrad4_s2_row_kernel<5, 0, 2><<<gridr, 256, accq, S(stream)>>>(dy, w, g, dx, bn_partial, bn);
    }
    return check_launch("dwconv2d_dgrad_bnsums");
  }
  const DwGeom gf{n, oh, ow, c, r, s, 1, r - 1 - pad_t, s - 1 - pad_l, h, wd};
  {                                                        // the rows function assumed a same-size layer
    const DwGeom gs{n, h, wd, c, r, s, 1, (r - 1) / 2, (s - 1) / 2, h, wd};
    EMBNET_CHECK_ARG(dwt::tile_applies(gs, 2) == dwt::tile_applies(gf, 2), "dwconv2d_dgrad_bnsums: sums are not available for this geometry");
  }
  EMBNET_TRACE(dwt::tile_applies(gf, 2) ? "embnet::dwt::dw_tile_kernel" : "embnet::dwconv_row4_kernel", TRACE_BYTES, 8.0 * total + 4.0 * n * oh * ow * c, stream);
  if (r == 3) launch_dw_rows<3, 1, true>(dy, w, gf, dx, S(stream), bn_partial, &bn);
  else launch_dw_rows<5, 1, true>(dy, w, gf, dx, S(stream), bn_partial, &bn);
  return check_launch("dwconv2d_dgrad_bnsums");
}

extern "C" int embnet_dwconv2d_dgrad_f32(const float* dy, const float* w, float* dx, int n, int h, int wd, int c, int r,
                                         int s, int stride, int pad_t, int pad_l, int oh, int ow, void* stream) {
  EMBNET_CHECK_ARG(dy && w && dx, "dwconv2d_dgrad: null pointer");
  DwGeom g;
  if (int rc = make_dw(g, n, h, wd, c, r, s, stride, pad_t, pad_l, oh, ow, "dwconv2d_dgrad")) return rc;
  const long total = (long)n * h * wd * c;
  const int grid4 = cdiv(total / 4, 256);
  if ((c & 3) == 0 && r == s && (r == 3 || r == 5) && (stride == 1 || stride == 2) && dw_rows()) {
    const int gridr = cdiv((long)n * h * cdiv(wd, DW_TW) * (c / 4), 256);
    const DwGeom gt{n, oh, ow, c, r, s, 1, r - 1 - pad_t, s - 1 - pad_l, h, wd};
    EMBNET_TRACE(stride != 1 ? "embnet::dwconv_dgrad4_s2_row_kernel" : (dwt::tile_applies(gt, 0) ? "embnet::dwt::dw_tile_kernel" : "embnet::dwconv_row4_kernel"),
                 TRACE_BYTES, 4.0 * total + 4.0 * n * oh * ow * c, stream);
    if (stride == 1) {         // correlation of dy with the flipped kernel: the forward loop with the roles swapped
      const DwGeom gf{n, oh, ow, c, r, s, 1, r - 1 - pad_t, s - 1 - pad_l, h, wd};
      if (r == 3) launch_dw_rows<3, 1, true>(dy, w, gf, dx, S(stream));
      else launch_dw_rows<5, 1, true>(dy, w, gf, dx, S(stream));
    } else if (r == 3) {
      if (pad_l & 1) dwconv_dgrad4_s2_row_kernel<3, 1><<<gridr, 256, 0, S(stream)>>>(dy, w, g, dx);
      else dwconv_dgrad4_s2_row_kernel<3, 0><<<gridr, 256, 0, S(stream)>>>(dy, w, g, dx);
    } else {
      if (pad_l & 1) dwconv_dgrad4_s2_row_kernel<5, 1><<<gridr, 256, 0, S(stream)>>>(dy, w, g, dx);
      else dwconv_dgrad4_s2_row_kernel<5, 0><<<gridr, 256, 0, S(stream)>>>(dy, w, g, dx);
    }
    return check_launch("dwconv2d_dgrad");
  }
  const bool sq = (c & 3) == 0 && r == s && (stride == 1 || stride == 2);
  if (sq && r == 3 && stride == 1) { EMBNET_TRACE("embnet::dwconv_dgrad4_sq_kernel", TRACE_BYTES, 4.0 * total + 4.0 * n * oh * ow * c, stream); dwconv_dgrad4_sq_kernel<3, 1><<<grid4, 256, 0, S(stream)>>>(dy, w, g, dx); }
  else if (sq && r == 3) { EMBNET_TRACE("embnet::dwconv_dgrad4_sq_kernel", TRACE_BYTES, 4.0 * total + 4.0 * n * oh * ow * c, stream); dwconv_dgrad4_sq_kernel<3, 2><<<grid4, 256, 0, S(stream)>>>(dy, w, g, dx); }
  else if (sq && r == 5 && stride == 1) { EMBNET_TRACE("embnet::dwconv_dgrad4_sq_kernel", TRACE_BYTES, 4.0 * total + 4.0 * n * oh * ow * c, stream); dwconv_dgrad4_sq_kernel<5, 1><<<grid4, 256, 0, S(stream)>>>(dy, w, g, dx); }
  else if (sq && r == 5) { EMBNET_TRACE("embnet::dwconv_dgrad4_sq_kernel", TRACE_BYTES, 4.0 * total + 4.0 * n * oh * ow * c, stream); dwconv_dgrad4_sq_kernel<5, 2><<<grid4, 256, 0, S(stream)>>>(dy, w, g, dx); }
  else if ((c & 3) == 0) { EMBNET_TRACE("embnet::dwconv_dgrad_kernel", TRACE_BYTES, 4.0 * total + 4.0 * n * oh * ow * c, stream); dwconv_dgrad_kernel<4><<<grid4, 256, 0, S(stream)>>>(dy, w, g, dx); }
  else { EMBNET_TRACE("embnet::dwconv_dgrad_kernel", TRACE_BYTES, 4.0 * total + 4.0 * n * oh * ow * c, stream); dwconv_dgrad_kernel<1><<<cdiv(total, 256), 256, 0, S(stream)>>>(dy, w, g, dx); }
  return check_launch("dwconv2d_dgrad");
}

static int dw_wgrad_blocks(long npix, int& ppb) {
  long blocks = (npix + 127) / 128;
  if (blocks > 2048) blocks = 2048;
  ppb = (int)((npix + blocks - 1) / blocks);
  return (int)((npix + ppb - 1) / ppb);
}

// slabs of (row, column-block) units for the wave-per-kernel-row kernel: enough workgroups (slabs x channel groups) to
// fill the chip even when a layer has few pixels (7x7x1152: 3584 units), at most 2048 partial slabs for dw_slab_sum
static int dw_wave_slabs(long nunits, int c, int& cq_lanes, int& cgroups, int& upb) {
  // channel-quad lanes per wave: a power of two (the other lane bits walk units).  The smallest power of two >= C/4 leaves
  // lanes idle when C/4 is not one (EfficientNet: C/4 = 24, 36, 168, 288 -> 75, 56, 87, 90 % of the lanes at work, and the
  // kernel runs at the rate the L1 takes wave-wide requests): take the widest group of >= 8 lanes (one 128-byte line per
  // pixel) that wastes the fewest lanes.  Measured per layer (profiles/r04_exp_dw_wgrad_lanes.txt): C = 96, 480, 672, 1152 gain
  // 9 - 29 %; C = 144 (36 quads) loses 30 % in 8-lane groups and more in 4-lane groups, so a C/4 that fits one wave stays whole.
  static const int pack = (int)env_long("EMBNET_DW_WGRAD_PACK", 1);
  const int c4 = c / 4;
  cq_lanes = 1; while (cq_lanes < c4 && cq_lanes < 64) cq_lanes <<= 1;
  if (pack && !(c4 > 32 && c4 <= 64)) {                    // (33..64 quads: one masked 64-lane group is the fastest form measured)
    auto use = [&](int l) { return (double)c4 / ((double)l * cdiv(c4, l)); };
    int best = cq_lanes;
    for (int l = cq_lanes >> 1; l >= 8; l >>= 1) if (use(l) > use(best) + 1e-9) best = l;
    cq_lanes = best;
  }
  cgroups = cdiv(c4, cq_lanes);
  long slabs = (nunits + 15) / 16;
  const long cap = 2048 / cgroups > 0 ? 2048 / cgroups : 1;
  if (slabs > cap) slabs = cap;
  if (slabs < 1) slabs = 1;
  upb = (int)((nunits + slabs - 1) / slabs);
  return (int)((nunits + upb - 1) / upb);
}

extern "C" size_t embnet_dwconv2d_wgrad_workspace_bytes(int n, int c, int r, int s, int oh, int ow) {
  if (n <= 0 || c <= 0 || r <= 0 || s <= 0 || oh <= 0 || ow <= 0) return 0;
  int ppb, cql, cg;            // the larger of the two kernels' slab counts (the query does not know the stride)
  const int b0 = dw_wgrad_blocks((long)n * oh * ow, ppb);
  const int b1 = (c & 3) ? 0 : dw_wave_slabs((long)n * oh * cdiv(ow, DW_TW), c, cql, cg, ppb);
  const DwGeom gs{n, oh, ow, c, r, s, 1, (r - 1) / 2, (s - 1) / 2, oh, ow};       // ... nor the input size: a same-size stride-1 layer
  const int b2 = dwt::tile_wgrad_slabs(gs);
  const int b = b0 > b1 ? b0 : b1;
  return (size_t)(b > b2 ? b : b2) * r * s * c * sizeof(float);
}

extern "C" int embnet_dwconv2d_wgrad_f32(const float* x, const float* dy, float* dw, void* workspace,
                                         size_t workspace_bytes, int n, int h, int wd, int c, int r, int s, int stride,
                                         int pad_t, int pad_l, int oh, int ow, void* stream) {
  EMBNET_CHECK_ARG(x && dy && dw && workspace, "dwconv2d_wgrad: null pointer");
  DwGeom g;
  if (int rc = make_dw(g, n, h, wd, c, r, s, stride, pad_t, pad_l, oh, ow, "dwconv2d_wgrad")) return rc;
  if (workspace_bytes < embnet_dwconv2d_wgrad_workspace_bytes(n, c, r, s, oh, ow))
    return fail(EMBNET_EWORKSPACE, "dwconv2d_wgrad: workspace too small");
  int ppb;
  if (const int slabs = dwt::tile_wgrad_slabs(g)) {                              // small maps, stride 1: LDS-tile kernel (dwconv_tile.hip)
    const long cnt = (long)r * s * c;
    { EMBNET_TRACE("embnet::dwt::dw_tile_wgrad_kernel", TRACE_BYTES, 4.0 * n * c * ((double)h * wd + (double)oh * ow), stream); dwt::launch_tile_wgrad(x, dy, g, (float*)workspace, S(stream)); }
    { EMBNET_TRACE("embnet::dw_slab_sum_kernel", TRACE_BYTES, 4.0 * cnt * (slabs + 1), stream); dw_slab_sum_kernel<<<cdiv(cnt, 16), 256, 0, S(stream)>>>((const float*)workspace, slabs, cnt, dw); }
    return check_launch("dwconv2d_wgrad");
  }
  if ((c & 3) == 0 && r == s && (r == 3 || r == 5) && (stride == 1 || stride == 2) && dw_rows()) {
    int upb, cql, cgroups;
    const int blocks = dw_wave_slabs((long)n * oh * cdiv(ow, DW_TW), c, cql, cgroups, upb);
    const dim3 grid(blocks, cgroups);
    static const int colmajor = (int)env_long("EMBNET_DW_WGRAD_COLMAJOR", 1);
    {
      EMBNET_TRACE("embnet::dwconv_wgrad4_wave_kernel", TRACE_BYTES, 4.0 * n * c * ((double)h * wd + (double)oh * ow), stream);
      if (r == 3 && stride == 1) dwconv_wgrad4_wave_kernel<3, 1><<<grid, 192, 0, S(stream)>>>(x, dy, g, cql, upb, (float*)workspace, colmajor);
      else if (r == 3) dwconv_wgrad4_wave_kernel<3, 2><<<grid, 192, 0, S(stream)>>>(x, dy, g, cql, upb, (float*)workspace, colmajor);
      else if (stride == 1) dwconv_wgrad4_wave_kernel<5, 1><<<grid, 320, 0, S(stream)>>>(x, dy, g, cql, upb, (float*)workspace, colmajor);
      else dwconv_wgrad4_wave_kernel<5, 2><<<grid, 320, 0, S(stream)>>>(x, dy, g, cql, upb, (float*)workspace, colmajor);
    }
    const long cnt = (long)r * s * c;
    { EMBNET_TRACE("embnet::dw_slab_sum_kernel", TRACE_BYTES, 4.0 * cnt * (blocks + 1), stream); dw_slab_sum_kernel<<<cdiv(cnt, 16), 256, 0, S(stream)>>>((const float*)workspace, blocks, cnt, dw); }
    return check_launch("dwconv2d_wgrad");
  }
  const int blocks = dw_wgrad_blocks((long)n * oh * ow, ppb);
  if ((c & 3) == 0) {
    int cql = 1; while (cql < c / 4 && cql < 256) cql <<= 1;
    if (r == 3 && s == 3) { EMBNET_TRACE("embnet::dwconv_wgrad4_kernel", TRACE_BYTES, 4.0 * n * c * ((double)h * wd + (double)oh * ow), stream); dwconv_wgrad4_kernel<9, 3><<<blocks, 256, 0, S(stream)>>>(x, dy, g, cql, ppb, (float*)workspace); }
    else if (r == 5 && s == 5) { EMBNET_TRACE("embnet::dwconv_wgrad4_kernel", TRACE_BYTES, 4.0 * n * c * ((double)h * wd + (double)oh * ow), stream); dwconv_wgrad4_kernel<25, 5><<<blocks, 256, 0, S(stream)>>>(x, dy, g, cql, ppb, (float*)workspace); }
    else if (r * s <= 9) { EMBNET_TRACE("embnet::dwconv_wgrad4_kernel", TRACE_BYTES, 4.0 * n * c * ((double)h * wd + (double)oh * ow), stream); dwconv_wgrad4_kernel<9, 0><<<blocks, 256, 0, S(stream)>>>(x, dy, g, cql, ppb, (float*)workspace); }
    else if (r * s <= 25) { EMBNET_TRACE("embnet::dwconv_wgrad4_kernel", TRACE_BYTES, 4.0 * n * c * ((double)h * wd + (double)oh * ow), stream); dwconv_wgrad4_kernel<25, 0><<<blocks, 256, 0, S(stream)>>>(x, dy, g, cql, ppb, (float*)workspace); }
    else { EMBNET_TRACE("embnet::dwconv_wgrad4_kernel", TRACE_BYTES, 4.0 * n * c * ((double)h * wd + (double)oh * ow), stream); dwconv_wgrad4_kernel<49, 0><<<blocks, 256, 0, S(stream)>>>(x, dy, g, cql, ppb, (float*)workspace); }
  } else {
    { EMBNET_TRACE("embnet::dwconv_wgrad1_kernel", TRACE_BYTES, 4.0 * n * c * ((double)h * wd + (double)oh * ow), stream); dwconv_wgrad1_kernel<<<blocks, 256, 0, S(stream)>>>(x, dy, g, ppb, (float*)workspace); }
  }
  const long cnt = (long)r * s * c;
  { EMBNET_TRACE("embnet::dw_slab_sum_kernel", TRACE_BYTES, 4.0 * cnt * (blocks + 1), stream); dw_slab_sum_kernel<<<cdiv(cnt, 16), 256, 0, S(stream)>>>((const float*)workspace, blocks, cnt, dw); }
  return check_launch("dwconv2d_wgrad");
}

extern "C" int embnet_activation_fwd(const float* x, long total, int kind, float* y, void* stream) {
  EMBNET_CHECK_ARG(x && y && total > 0 && (kind == 0 || kind == 1), "activation_fwd: bad argument");
  { EMBNET_TRACE("embnet::act_fwd_kernel", TRACE_BYTES, 8.0 * total, stream); act_fwd_kernel<<<ew_blocks(total), 256, 0, S(stream)>>>(x, total, kind, y); }
  return check_launch("activation_fwd");
}

extern "C" int embnet_activation_bwd(const float* x, const float* dy, long total, int kind, float* dx, void* stream) {
  EMBNET_CHECK_ARG(x && dy && dx && total > 0 && (kind == 0 || kind == 1), "activation_bwd: bad argument");
  { EMBNET_TRACE("embnet::act_bwd_kernel", TRACE_BYTES, 12.0 * total, stream); act_bwd_kernel<<<ew_blocks(total), 256, 0, S(stream)>>>(x, dy, total, kind, dx); }
  return check_launch("activation_bwd");
}

extern "C" int embnet_channel_scale_fwd(const float* x, const float* s, int n, int hw, int c, float* y, void* stream) {
  EMBNET_CHECK_ARG(x && s && y && n > 0 && hw > 0 && c > 0, "channel_scale_fwd: bad argument");
  const long total = (long)n * hw * c;
  { EMBNET_TRACE("embnet::chscale_fwd_kernel", TRACE_BYTES, 8.0 * total, stream); chscale_fwd_kernel<<<ew_blocks(total), 256, 0, S(stream)>>>(x, s, total, hw, c, y); }
  return check_launch("channel_scale_fwd");
}

extern "C" int embnet_channel_scale_bwd(const float* x, const float* s, const float* dy, int n, int hw, int c, float* dx,
                                        float* ds, void* stream) {
  EMBNET_CHECK_ARG(x && s && dy && dx && ds && n > 0 && hw > 0 && c > 0, "channel_scale_bwd: bad argument");
  if ((c & 3) == 0) { EMBNET_TRACE("embnet::chscale_bwd4_kernel", TRACE_BYTES, 12.0 * n * hw * c, stream); chscale_bwd4_kernel<<<dim3(cdiv(c / 4, 16), n), 256, 0, S(stream)>>>(x, s, dy, hw, c / 4, dx, ds); }
  else { EMBNET_TRACE("embnet::chscale_bwd_kernel", TRACE_BYTES, 12.0 * n * hw * c, stream); chscale_bwd_kernel<<<dim3(cdiv(c, 256), n), 256, 0, S(stream)>>>(x, s, dy, hw, c, dx, ds); }
  return check_launch("channel_scale_bwd");
}

extern "C" int embnet_channel_scale_dgate(const float* x, const float* dy, int n, int hw, int c, float* ds, void* stream) {
  EMBNET_CHECK_ARG(x && dy && ds && n > 0 && hw > 0 && c > 0 && (c & 3) == 0, "channel_scale_dgate: bad argument (c %% 4 == 0)");
  { EMBNET_TRACE("embnet::chscale_dgate4_kernel", TRACE_BYTES, 8.0 * n * hw * c, stream); chscale_dgate4_kernel<<<dim3(cdiv(c / 4, 16), n), 256, 0, S(stream)>>>(x, dy, hw, c / 4, ds); }
  return check_launch("channel_scale_dgate");
}

extern "C" int embnet_sample_dropout(const float* x, long total, long per_sample, float rate, uint64_t seed,
                                     const uint64_t* seed_add_dev, float* y, void* stream) {
  EMBNET_CHECK_ARG(x && y && total > 0 && per_sample > 0, "sample_dropout: bad argument");
  EMBNET_CHECK_ARG(rate >= 0.f && rate < 1.f, "sample_dropout: rate %f outside [0,1)", rate);
  { EMBNET_TRACE("embnet::sample_dropout_kernel", TRACE_BYTES, 8.0 * total, stream); sample_dropout_kernel<<<ew_blocks(total), 256, 0, S(stream)>>>(x, total, per_sample, rate, seed, seed_add_dev, y); }
  return check_launch("sample_dropout");
}

extern "C" int embnet_absdiff_fwd(const float* a, const float* b, long total, float* y, void* stream) {
  EMBNET_CHECK_ARG(a && b && y && total > 0, "absdiff_fwd: bad argument");
  { EMBNET_TRACE("embnet::absdiff_fwd_kernel", TRACE_BYTES, 12.0 * total, stream); absdiff_fwd_kernel<<<ew_blocks(total), 256, 0, S(stream)>>>(a, b, total, y); }
  return check_launch("absdiff_fwd");
}

extern "C" int embnet_absdiff_bwd(const float* a, const float* b, const float* dy, long total, float* da, float* db,
                                  void* stream) {
  EMBNET_CHECK_ARG(a && b && dy && da && db && total > 0, "absdiff_bwd: bad argument");
  { EMBNET_TRACE("embnet::absdiff_bwd_kernel", TRACE_BYTES, 20.0 * total, stream); absdiff_bwd_kernel<<<ew_blocks(total), 256, 0, S(stream)>>>(a, b, dy, total, da, db); }
  return check_launch("absdiff_bwd");
}
