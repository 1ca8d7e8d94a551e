// HBM-bound pieces of the EfficientNet MBConv block (the `efficientnet` zoo package instantiated at
// /root/reference/embedding_net/backbones.py:84-98) and of the siamese 'l1' head (models.py:217-221):
// depthwise convolution fwd / dgrad / wgrad, swish, sigmoid, squeeze-excite channel scaling,
// per-sample drop-connect, |a-b|.  NHWC fp32.  The 1x1 expand/project convolutions and the SE
// dense layers run on the MFMA engine (conv.hip / dense.hip); nothing here is GEMM-shaped:
// a depthwise tap is one multiply per loaded element, so these are priced in bytes.
#include "common.h"
#include "../../include/embnet.h"

namespace embnet {

struct DwGeom { int N, H, W, C, R, S, stride, pad_t, pad_l, OH, OW; };

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
                                                             float rate, uint64_t seed, float* __restrict__ y) {
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

extern "C" int embnet_dwconv2d_fwd_f32(const float* x, const float* w, float* y, int n, int h, int wd, int c, int r,
                                       int s, int stride, int pad_t, int pad_l, int oh, int ow, void* stream) {
  EMBNET_CHECK_ARG(x && w && y, "dwconv2d_fwd: null pointer");
  DwGeom g;
  if (int rc = make_dw(g, n, h, wd, c, r, s, stride, pad_t, pad_l, oh, ow, "dwconv2d_fwd")) return rc;
  const long total = (long)n * oh * ow * c;
  const int grid4 = cdiv(total / 4, 256);
  if ((c & 3) == 0 && r == s && r == 3) { EMBNET_TRACE("embnet::dwconv_fwd4_sq_kernel", TRACE_BYTES, 4.0 * total + 4.0 * n * h * wd * c, stream); dwconv_fwd4_sq_kernel<3><<<grid4, 256, 0, S(stream)>>>(x, w, g, y); }
  else if ((c & 3) == 0 && r == s && r == 5) { EMBNET_TRACE("embnet::dwconv_fwd4_sq_kernel", TRACE_BYTES, 4.0 * total + 4.0 * n * h * wd * c, stream); dwconv_fwd4_sq_kernel<5><<<grid4, 256, 0, S(stream)>>>(x, w, g, y); }
  else if ((c & 3) == 0) { EMBNET_TRACE("embnet::dwconv_fwd_kernel", TRACE_BYTES, 4.0 * total + 4.0 * n * h * wd * c, stream); dwconv_fwd_kernel<4><<<grid4, 256, 0, S(stream)>>>(x, w, g, y); }
  else { EMBNET_TRACE("embnet::dwconv_fwd_kernel", TRACE_BYTES, 4.0 * total + 4.0 * n * h * wd * c, stream); dwconv_fwd_kernel<1><<<cdiv(total, 256), 256, 0, S(stream)>>>(x, w, g, y); }
  return check_launch("dwconv2d_fwd");
}

extern "C" int embnet_dwconv2d_dgrad_f32(const float* dy, const float* w, float* dx, int n, int h, int wd, int c, int r,
                                         int s, int stride, int pad_t, int pad_l, int oh, int ow, void* stream) {
  EMBNET_CHECK_ARG(dy && w && dx, "dwconv2d_dgrad: null pointer");
  DwGeom g;
  if (int rc = make_dw(g, n, h, wd, c, r, s, stride, pad_t, pad_l, oh, ow, "dwconv2d_dgrad")) return rc;
  const long total = (long)n * h * wd * c;
  const int grid4 = cdiv(total / 4, 256);
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

extern "C" size_t embnet_dwconv2d_wgrad_workspace_bytes(int n, int c, int r, int s, int oh, int ow) {
  if (n <= 0 || c <= 0 || r <= 0 || s <= 0 || oh <= 0 || ow <= 0) return 0;
  int ppb;
  return (size_t)dw_wgrad_blocks((long)n * oh * ow, ppb) * r * s * c * sizeof(float);
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

extern "C" int embnet_sample_dropout(const float* x, long total, long per_sample, float rate, uint64_t seed, float* y,
                                     void* stream) {
  EMBNET_CHECK_ARG(x && y && total > 0 && per_sample > 0, "sample_dropout: bad argument");
  EMBNET_CHECK_ARG(rate >= 0.f && rate < 1.f, "sample_dropout: rate %f outside [0,1)", rate);
  { EMBNET_TRACE("embnet::sample_dropout_kernel", TRACE_BYTES, 8.0 * total, stream); sample_dropout_kernel<<<ew_blocks(total), 256, 0, S(stream)>>>(x, total, per_sample, rate, seed, y); }
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
