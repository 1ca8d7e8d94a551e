// HBM-bound layers around the convolutions, NHWC fp32: BatchNorm (train / inference, optional
// fused ReLU), max pooling, global average pooling, ReLU/bias backward, residual add, dropout,
// L2 kernel regulariser.  Stand-ins for the Keras layers used at
// /root/reference/embedding_net/backbones.py:21-36 (MaxPool2D, Flatten/Dense bias+ReLU),
// :44-75 (BatchNormalization, Dropout), :110-116 (GlobalAveragePooling2D) and the zoo ResNet blocks.
// Roofline: HBM.  Algorithmic bytes per element are noted at each kernel.
#include "common.h"
#include "gemm_engine.h"        // split4: the three bf16 pieces of an fp32 value (planes for conv_patch.hip)
#include <stdlib.h>
#include "../../include/embnet.h"

namespace embnet {

// Inverted dropout riding on another pass (the BatchNormalization in front of a Dropout layer, simple2's bn3 / bn6,
// /root/reference/embedding_net/backbones.py:55,66): the mask of embnet_dropout — keep element i iff
// rng_u32(seed, i, 1) >= thr — applied to the value in registers; thr == 0 switches it off.
struct DropArg { uint64_t seed; const uint64_t* seed_add; uint32_t thr; float keep_scale; };
static DropArg drop_arg(float rate, uint64_t seed, const uint64_t* seed_add) {
  DropArg d{seed, seed_add, 0u, 1.f};
  if (rate > 0.f) { d.thr = (uint32_t)((double)rate * 4294967296.0); d.keep_scale = 1.f / (1.f - rate); }
  return d;
}
__device__ __forceinline__ uint64_t drop_seed(const DropArg& d) { return d.seed + (d.seed_add ? *d.seed_add : 0ull); }
__device__ __forceinline__ float4 drop4(const DropArg& d, uint64_t seed, long i4, float4 v) {   // elements 4*i4 .. 4*i4+3
  const uint64_t e = (uint64_t)i4 * 4;
  v.x = rng_u32(seed, e, 1) >= d.thr ? v.x * d.keep_scale : 0.f;
  v.y = rng_u32(seed, e + 1, 1) >= d.thr ? v.y * d.keep_scale : 0.f;
  v.z = rng_u32(seed, e + 2, 1) >= d.thr ? v.z * d.keep_scale : 0.f;
  v.w = rng_u32(seed, e + 3, 1) >= d.thr ? v.w * d.keep_scale : 0.f;
  return v;
}

static bool bn_scalar() { static const bool v = env_long("EMBNET_BN_SCALAR", 0) != 0; return v; }   // A/B knob

// ---------------------------------------------------------------- column reductions over [M, C]
// Layout of a 256-thread workgroup: cl channel lanes x rl row lanes (cl*rl = 256, cl a power of 2),
// consecutive threads on consecutive channels -> coalesced rows.  Each workgroup reduces a slab of
// rows and writes partial[block][2][C]; a finalize kernel adds the partials in double, fixed order.
struct ColGeom { int cl, rl, blocks, rows_per_block; };

static ColGeom col_geom(long m, int c) {
  ColGeom g;
  int cl = 1; while (cl < c && cl < 256) cl <<= 1;
  g.cl = cl; g.rl = 256 / cl;
  long blocks = (m + (long)g.rl * 16 - 1) / ((long)g.rl * 16);
  if (blocks < 512) {                                   // small tensors (7x7, 14x14 maps): fewer rows per thread rather than idle CUs —
    blocks = (m + (long)g.rl * 4 - 1) / ((long)g.rl * 4);  // 196 workgroups read a 25 MB pair of tensors at 2 TB/s, 512+ at 3-4
    if (blocks > 512) blocks = 512;
  }
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  g.blocks = (int)blocks;
  g.rows_per_block = (int)((m + blocks - 1) / blocks);
  return g;
}

template <class F>   // F(row, col, &v0, &v1) accumulates two sums for element (row, col)
__device__ __forceinline__ void col_reduce2(long m, int c, ColGeom g, float* __restrict__ partial, F f) {
  __shared__ float sh[2][256];
  const int ci = threadIdx.x % g.cl, ri = threadIdx.x / g.cl;
  const long r0 = (long)blockIdx.x * g.rows_per_block;
  const long r1 = min(r0 + g.rows_per_block, m);
  for (int c0 = 0; c0 < c; c0 += g.cl) {
    const int col = c0 + ci;
    float a = 0.f, b = 0.f;
    if (col < c)
      for (long r = r0 + ri; r < r1; r += g.rl) f(r, col, a, b);
    sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = b;
    __syncthreads();
    if (ri == 0 && col < c) {
      for (int k = 1; k < g.rl; ++k) { a += sh[0][k * g.cl + ci]; b += sh[1][k * g.cl + ci]; }
      partial[((long)blockIdx.x * 2 + 0) * c + col] = a;
      partial[((long)blockIdx.x * 2 + 1) * c + col] = b;
    }
    __syncthreads();
  }
}

// Finalize helper: one 256-thread workgroup per channel adds that channel's per-block partials in
// double (fixed order: thread-strided, then a shuffle tree) — thread 0 gets the totals.
// q_max (optional): the LARGEST second partial — for the forward statistics the largest per-band sum of squares, an upper bound of
// max x^2 over the tensor's elements of this channel (every element's square is a term of exactly one band's sum).
__device__ __forceinline__ void block_partial_sums(const float* __restrict__ partial, int blocks, int c, int col,
                                                   double& s_out, double& ss_out, bool by_channel = false, float* q_max = nullptr) {
  __shared__ double red[2][4];
  __shared__ float redq[4];
  double s = 0.0, ss = 0.0;
  float qm = 0.f;
  if (by_channel) {                  // [2][c][blocks]: this channel's partials are contiguous (conv epilogue layout)
    const float* p1 = partial + (long)col * blocks;
    const float* p2 = p1 + (long)c * blocks;
    if ((blocks & 3) == 0 && ((reinterpret_cast<uintptr_t>(partial) & 15) == 0)) {
      // thousands of row-band partials per channel on the early layers (56x56: 6272): 16-byte loads, four in flight
      const float4* q1 = reinterpret_cast<const float4*>(p1);
      const float4* q2 = reinterpret_cast<const float4*>(p2);
      const int nb4 = blocks >> 2;
#pragma unroll 4
      for (int b = threadIdx.x; b < nb4; b += 256) {
        const float4 u = q1[b], v = q2[b];
        s += ((double)u.x + (double)u.y) + ((double)u.z + (double)u.w);
        ss += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
        qm = fmaxf(fmaxf(qm, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
      }
    } else {
      for (int b = threadIdx.x; b < blocks; b += 256) { s += (double)p1[b]; ss += (double)p2[b]; qm = fmaxf(qm, p2[b]); }
    }
  } else {                           // [blocks][2][c]
    for (int b = threadIdx.x; b < blocks; b += 256) {
      const float q = partial[((long)b * 2 + 1) * c + col];
      s += (double)partial[((long)b * 2) * c + col];
      ss += (double)q;
      qm = fmaxf(qm, q);
    }
  }
  s = wave_sum(s); ss = wave_sum(ss); qm = wave_max(qm);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = ss; redq[threadIdx.x >> 6] = qm; }
  __syncthreads();
  s_out = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  ss_out = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  if (q_max) *q_max = fmaxf(fmaxf(redq[0], redq[1]), fmaxf(redq[2], redq[3]));
}

// ---- the range of a BatchNormalization's OUTPUT, known before the apply pass runs -----------------------------------
// The two-piece fp16 operand format (gemm_engine.h) needs an upper bound B of max |y| of the tensor BEFORE the pass that writes y's
// planes starts, and the three-product gather convs that read the fp32 y need it as their range slot.  The statistics partials hold
// it: max x^2 <= the largest per-band sum of squares q_c of channel c (block_partial_sums), so for y = act(scale_c x + shift_c) with
// act in {identity, ReLU, swish} (|act(z)| <= |z|):
//     |y| <= |scale_c| sqrt(q_c) + |shift_c| =: bound_c,        B = max_c bound_c.
// Never below the true maximum (the 2^-10 margin covers the roundings of q_c and of this arithmetic); above it by at most
// sqrt(rows per band) x (a band of 32 ... 96 rows from a conv epilogue: <= 3.3 binades; a statistics-pass block of up to a few
// thousand rows: <= 6) plus the share of a large |mean| — looseness costs precision only at the subnormal floor, 2^-39 B absolute.
// bn_finalize_kernel leaves bound_c per channel; the apply passes fold them (tensor_bound: every workgroup reads <= 2048 floats from
// L2 — no atomics, no extra launch, order-independent).
__device__ __forceinline__ float channel_bound(float sc, float sh, float q_max) {
  return (fabsf(sc) * sqrtf(q_max) + fabsf(sh)) * 1.0009765625f;
}
// max of bound[0 .. c) for every thread of a 256-thread workgroup (all threads call)
__device__ __forceinline__ float tensor_bound(const float* __restrict__ bound, int c) {
  __shared__ float tb_w[4];
  float m = 0.f;
  for (int i = threadIdx.x; i < c; i += 256) m = fmaxf(m, bound[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) tb_w[threadIdx.x >> 6] = m;
  __syncthreads();
  return fmaxf(fmaxf(tb_w[0], tb_w[1]), fmaxf(tb_w[2], tb_w[3]));
}

// ---------------------------------------------------------------- BatchNorm
// stats pass: reads x once (4 B/elem)
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, long m, int c, ColGeom g,
                                                       float* __restrict__ partial) {
  col_reduce2(m, c, g, partial, [&](long r, int col, float& a, float& b) {
    const float v = x[r * c + col]; a += v; b = fmaf(v, v, b);
  });
}

// Four channels per thread (C % 4 == 0): same two-stage reduction with 16-byte loads; lanes run over
// channel QUADS.  partial keeps the [block][2][C] layout, so the finalize kernels are shared.
struct NoPrep { __device__ __forceinline__ int operator()(int) const { return 0; } };
// F(row, quad, const K& k, float4& a, float4& b) with k = prep(quad): the per-channel constants of a column are fetched ONCE
// per thread and column block, not once per row (bn_bwd_reduce4 issued four 16-byte constant loads beside the two data
// loads of every element: 3.9 TB/s where the apply pass, two constant loads per element, reached 5.3)
template <class P, class F, bool MAX3 = false>
__device__ __forceinline__ void col_reduce2_v4p(long m, int c4, ColGeom g, float* __restrict__ partial, P prep, F f,
                                                float* __restrict__ pmax = nullptr) {
  // MAX3: F takes a third accumulator, a running per-element MAXIMUM (>= 0); the block's maxima go to pmax[block][c]
  __shared__ float4 sh4[MAX3 ? 3 : 2][256];
  const int ci = threadIdx.x % g.cl, ri = threadIdx.x / g.cl;
  const long r0 = (long)blockIdx.x * g.rows_per_block;
  const long r1 = min(r0 + g.rows_per_block, m);
  for (int q0 = 0; q0 < c4; q0 += g.cl) {
    const int q = q0 + ci;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, mx = a;
    if (q < c4) {                     // four rows per trip (same accumulation order): 8 x 16-byte loads in flight per lane
      const auto k = prep(q);
      long r = r0 + ri;
      const long rl = g.rl;
      if constexpr (MAX3) {
        for (; r + 3 * rl < r1; r += 4 * rl) { f(r, q, k, a, b, mx); f(r + rl, q, k, a, b, mx); f(r + 2 * rl, q, k, a, b, mx); f(r + 3 * rl, q, k, a, b, mx); }
        for (; r < r1; r += rl) f(r, q, k, a, b, mx);
      } else {
        for (; r + 3 * rl < r1; r += 4 * rl) { f(r, q, k, a, b); f(r + rl, q, k, a, b); f(r + 2 * rl, q, k, a, b); f(r + 3 * rl, q, k, a, b); }
        for (; r < r1; r += rl) f(r, q, k, a, b);
      }
    }
    // row-lanes of one wave first (xor butterfly over the lane bits above the column bits), then the <= 4 per-wave
    // (or per-row-lane, when a wave is one row-lane) sums through LDS: 2 LDS round trips instead of rl - 1 serial ones
    int groups = g.rl;                                   // partial sums per column left for the LDS step
    if (g.cl < 64) {
      for (int o = g.cl; o < 64; o <<= 1) {
        a.x += __shfl_xor(a.x, o, 64); a.y += __shfl_xor(a.y, o, 64); a.z += __shfl_xor(a.z, o, 64); a.w += __shfl_xor(a.w, o, 64);
        b.x += __shfl_xor(b.x, o, 64); b.y += __shfl_xor(b.y, o, 64); b.z += __shfl_xor(b.z, o, 64); b.w += __shfl_xor(b.w, o, 64);
        if constexpr (MAX3) {
          mx.x = fmaxf(mx.x, __shfl_xor(mx.x, o, 64)); mx.y = fmaxf(mx.y, __shfl_xor(mx.y, o, 64));
          mx.z = fmaxf(mx.z, __shfl_xor(mx.z, o, 64)); mx.w = fmaxf(mx.w, __shfl_xor(mx.w, o, 64));
        }
      }
      groups = 4;
      if ((threadIdx.x & 63) < g.cl) {
        sh4[0][(threadIdx.x >> 6) * g.cl + ci] = a; sh4[1][(threadIdx.x >> 6) * g.cl + ci] = b;
        if constexpr (MAX3) sh4[2][(threadIdx.x >> 6) * g.cl + ci] = mx;
      }
    } else {
      sh4[0][threadIdx.x] = a; sh4[1][threadIdx.x] = b;
      if constexpr (MAX3) sh4[2][threadIdx.x] = mx;
    }
    __syncthreads();
    if (ri == 0 && q < c4) {
      a = sh4[0][ci]; b = sh4[1][ci];
      if constexpr (MAX3) mx = sh4[2][ci];
      for (int k = 1; k < groups; ++k) {
        const float4 oa = sh4[0][k * g.cl + ci], ob = sh4[1][k * g.cl + ci];
        a.x += oa.x; a.y += oa.y; a.z += oa.z; a.w += oa.w;
        b.x += ob.x; b.y += ob.y; b.z += ob.z; b.w += ob.w;
        if constexpr (MAX3) {
          const float4 om = sh4[2][k * g.cl + ci];
          mx.x = fmaxf(mx.x, om.x); mx.y = fmaxf(mx.y, om.y); mx.z = fmaxf(mx.z, om.z); mx.w = fmaxf(mx.w, om.w);
        }
      }
      reinterpret_cast<float4*>(partial + ((long)blockIdx.x * 2 + 0) * c4 * 4)[q] = a;
      reinterpret_cast<float4*>(partial + ((long)blockIdx.x * 2 + 1) * c4 * 4)[q] = b;
      if constexpr (MAX3) reinterpret_cast<float4*>(pmax + (long)blockIdx.x * c4 * 4)[q] = mx;
    }
    __syncthreads();
  }
}
template <class F>   // F(row, quad, float4& a, float4& b)
__device__ __forceinline__ void col_reduce2_v4(long m, int c4, ColGeom g, float* __restrict__ partial, F f) {
  col_reduce2_v4p(m, c4, g, partial, NoPrep(), [&](long r, int q, int, float4& a, float4& b) { f(r, q, a, b); });
}

__global__ __launch_bounds__(256) void bn_stats4_kernel(const float* __restrict__ x, long m, int c4, ColGeom g,
                                                        float* __restrict__ partial) {
  col_reduce2_v4(m, c4, g, partial, [&](long r, int q, float4& a, float4& b) {
    const float4 v = reinterpret_cast<const float4*>(x)[r * c4 + q];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    b.x = fmaf(v.x, v.x, b.x); b.y = fmaf(v.y, v.y, b.y); b.z = fmaf(v.z, v.z, b.z); b.w = fmaf(v.w, v.w, b.w);
  });
}

// pmax [blocks][c]: the block's max |dz| per channel — what bounds the dx of the apply pass before it runs (dx_channel_bound)
__global__ __launch_bounds__(256) void bn_bwd_reduce4_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             long m, int c4, ColGeom g, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int relu,
                                                             float* __restrict__ partial, const DropArg drop, float* __restrict__ pmax) {
  struct K4 { float4 sc, sh, mu, rs; };
  const uint64_t dseed = drop.thr ? drop_seed(drop) : 0ull;
  auto prep = [&](int q) {
    return K4{reinterpret_cast<const float4*>(scale)[q], reinterpret_cast<const float4*>(shift)[q],
              reinterpret_cast<const float4*>(mean)[q], reinterpret_cast<const float4*>(rstd)[q]};
  };
  auto body = [&](long r, int q, const K4& k, float4& a, float4& b, float4& mx) {
    const float4 xv = reinterpret_cast<const float4*>(x)[r * c4 + q];
    float4 dz = reinterpret_cast<const float4*>(dy)[r * c4 + q];
    if (drop.thr) dz = drop4(drop, dseed, r * c4 + q, dz);   // dy of the Dropout behind this layer -> dy of the layer
    const float4 sc = k.sc, sh = k.sh, mu = k.mu, rs = k.rs;
    if (relu) {
      dz.x = act_grad(relu, fmaf(xv.x, sc.x, sh.x), dz.x); dz.y = act_grad(relu, fmaf(xv.y, sc.y, sh.y), dz.y);
      dz.z = act_grad(relu, fmaf(xv.z, sc.z, sh.z), dz.z); dz.w = act_grad(relu, fmaf(xv.w, sc.w, sh.w), dz.w);
    }
    a.x += dz.x; a.y += dz.y; a.z += dz.z; a.w += dz.w;
    b.x = fmaf(dz.x, (xv.x - mu.x) * rs.x, b.x); b.y = fmaf(dz.y, (xv.y - mu.y) * rs.y, b.y);
    b.z = fmaf(dz.z, (xv.z - mu.z) * rs.z, b.z); b.w = fmaf(dz.w, (xv.w - mu.w) * rs.w, b.w);
    mx.x = fmaxf(mx.x, fabsf(dz.x)); mx.y = fmaxf(mx.y, fabsf(dz.y)); mx.z = fmaxf(mx.z, fabsf(dz.z)); mx.w = fmaxf(mx.w, fabsf(dz.w));
  };
  if (pmax) col_reduce2_v4p<decltype(prep), decltype(body), true>(m, c4, g, partial, prep, body, pmax);
  else {
    auto body2 = [&](long r, int q, const K4& k, float4& a, float4& b) { float4 mx = make_float4(0.f, 0.f, 0.f, 0.f); body(r, q, k, a, b, mx); };
    col_reduce2_v4p(m, c4, g, partial, prep, body2);
  }
}

// MODE 0: as described.  MODE 1: dx_planes in the two-piece fp16 format, scaled by the s the slot holds.  MODE 2: the dry run in front
// of MODE 1 — the same arithmetic, nothing stored but the workgroup's max |dx| (floats 2 + blockIdx.x behind the slot); the scale
// kernel below turns the maxima into s.  (8 bytes per element read once more: the price of an exact range for the gradient's planes.)
// MODE 3: MODE 0 + the range of dx into `range_slot` (common.h range_emit) for the three-product gather convs that read dx (conv.hip Ranges).
// MODE 4: MODE 1 without the dry run: the planes' scale from `dx_bound` [c] (bn_bwd_finalize_kernel's dx_channel_bound), + the range of
// dx_add (`add_range`: the bound of the identity shortcut's gradient) where one is added; workgroup 0 leaves (s, 1 / s) in the planes'
// slot and — range_slot != NULL — the bound in the range slot of the fp32 dx.
template <int MODE = 0>
__global__ __launch_bounds__(256) void bn_bwd_apply4_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            long total4, int c4, float inv_m, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const float* __restrict__ dbeta,
                                                            const float* __restrict__ dgamma, int relu, int training,
                                                            const float* __restrict__ dx_add, float* __restrict__ dx,
                                                            unsigned short* __restrict__ dx_planes, uint32_t* __restrict__ range_slot = nullptr,
                                                            const float* __restrict__ dx_bound = nullptr, const uint32_t* __restrict__ add_range = nullptr) {
  const long stride = (long)gridDim.x * 256;
  // (the launcher makes the stride a multiple of c4 whenever c4 divides a power of two, so a thread keeps its channel
  // quad and the six per-channel constants are loaded once; otherwise they are re-read per element)
  const bool fixed = stride % c4 == 0;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 sc = z4, sh = z4, mu = z4, rs = z4, db = z4, dg = z4;
  auto consts = [&](int q) {
    sc = reinterpret_cast<const float4*>(scale)[q]; sh = reinterpret_cast<const float4*>(shift)[q];
    if (training) {
      mu = reinterpret_cast<const float4*>(mean)[q]; rs = reinterpret_cast<const float4*>(rstd)[q];
      db = reinterpret_cast<const float4*>(dbeta)[q]; dg = reinterpret_cast<const float4*>(dgamma)[q];
    }
  };
  if (fixed) consts((int)(((long)blockIdx.x * 256 + threadIdx.x) % c4));
  float amax = 0.f;
  float pscale = MODE == 1 ? planes_scale_slot(dx_planes, total4 * 4)[0] : 1.f;
  if (MODE == 4) {
    float b = tensor_bound(dx_bound, 4 * c4);
    if (dx_add) b += __uint_as_float(*add_range);
    const float2 sp = scale_pair(scale_exponent_of(b));
    pscale = sp.x;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      float* sl = planes_scale_slot(dx_planes, total4 * 4); sl[0] = sp.x; sl[1] = sp.y;
      if (range_slot) *range_slot = __float_as_uint(b);
    }
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
    if (!fixed) consts((int)(i % c4));
    const float4 xv = reinterpret_cast<const float4*>(x)[i];
    float4 dz = reinterpret_cast<const float4*>(dy)[i];
    if (relu) {
      dz.x = act_grad(relu, fmaf(xv.x, sc.x, sh.x), dz.x); dz.y = act_grad(relu, fmaf(xv.y, sc.y, sh.y), dz.y);
      dz.z = act_grad(relu, fmaf(xv.z, sc.z, sh.z), dz.z); dz.w = act_grad(relu, fmaf(xv.w, sc.w, sh.w), dz.w);
    }
    float4 o;
    if (training) {
      o.x = sc.x * (dz.x - db.x * inv_m - (xv.x - mu.x) * rs.x * dg.x * inv_m);
      o.y = sc.y * (dz.y - db.y * inv_m - (xv.y - mu.y) * rs.y * dg.y * inv_m);
      o.z = sc.z * (dz.z - db.z * inv_m - (xv.z - mu.z) * rs.z * dg.z * inv_m);
      o.w = sc.w * (dz.w - db.w * inv_m - (xv.w - mu.w) * rs.w * dg.w * inv_m);
    } else {
      o = make_float4(sc.x * dz.x, sc.y * dz.y, sc.z * dz.z, sc.w * dz.w);
    }
    if (dx_add) {                                        // gradient of the tensor's other consumer (identity shortcut)
      const float4 a = reinterpret_cast<const float4*>(dx_add)[i];
      o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    }
    if (MODE == 2) { amax = fmaxf(amax, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w)))); continue; }
    if (MODE == 3) amax = amax4(amax, o);
    if (dx) reinterpret_cast<float4*>(dx)[i] = o;        // (NULL: planes only — every consumer of dx reads the planes)
    if (dx_planes) {                                     // the same values as bf16 pieces, chunk-major: dy operand of the
      const long pix = i / c4; const int q = (int)(i - pix * c4);      // patch data gradient of the convolution in front
      const long e = ((long)(q >> 2) * (total4 / c4) + pix) * 16 + 4 * (q & 3);
      if (MODE == 1 || MODE == 4) {
        const Split4H s = split4h(o, pscale);
#pragma unroll
        for (int k = 0; k < 2; ++k) *reinterpret_cast<uint2*>(dx_planes + k * total4 * 4 + e) = s.p[k];
      } else {
        const Split4 s = split4(o);
#pragma unroll
        for (int k = 0; k < 3; ++k) *reinterpret_cast<uint2*>(dx_planes + k * total4 * 4 + e) = s.p[k];
      }
    }
  }
  if (MODE == 2) {
    amax = wave_max(amax);
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) planes_scale_slot(dx_planes, total4 * 4)[2 + blockIdx.x] = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
  }
  if (MODE == 3) range_emit_block(range_slot + 1 + blockIdx.x % RANGE_PARTIALS, amax);
}

// (s, 1 / s) of a planes tensor from the workgroup maxima a dry run left behind its slot: the largest |value| lands in [2^14, 2^15)
// range_out (conv.hip Ranges): the same maximum for the gather convs that read the fp32 copy of the tensor
__global__ __launch_bounds__(256) void planes_scale_kernel(float* __restrict__ slot, int blocks, uint32_t* __restrict__ range_out = nullptr) {
  float m = 0.f;
  for (int i = threadIdx.x; i < blocks; i += 256) m = fmaxf(m, slot[2 + i]);
  m = wave_max(m);
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    const float2 sp = scale_pair(scale_exponent_of(m));    // (zero, infinite or NaN: s = 1)
    slot[0] = sp.x; slot[1] = sp.y;
    if (range_out) *range_out = __float_as_uint(m);
  }
}

// BatchNorm backward of a layer whose output is also POOLED (squeeze-and-excite: layers._BNGapFn): the gradient is
//   dy_total[n,p,c] = dy[n,p,c] + dpool[n,c] / hw
// (embnet_gap_bwd's broadcast-add pass, 12 B per element on EfficientNet's 6C-wide tensors).  Here both BatchNorm passes form
// it while they read dy — dpool[n, quad] is a 16-byte L1-resident load — with embnet_gap_bwd's two roundings (the two-launch chain's result to the
// last bits), and the broadcast tensor is never written.
struct DivU { uint32_t mul, shift; };                     // n / d for 0 <= n < 2^31 (conv_geom.h FastDiv)
static DivU make_divu(uint32_t d) {
  uint32_t sft = 0; while ((1ull << sft) < d) ++sft;
  return DivU{(uint32_t)((((1ull << sft) - d) << 32) / d + 1), sft};
}
__device__ __forceinline__ uint32_t divu(uint32_t n, DivU f) { return (__umulhi(n, f.mul) + n) >> f.shift; }
__device__ __forceinline__ float4 add_pool4(float4 dy, float4 g, float inv) {
  return make_float4(__fadd_rn(__fmul_rn(g.x, inv), dy.x), __fadd_rn(__fmul_rn(g.y, inv), dy.y),
                     __fadd_rn(__fmul_rn(g.z, inv), dy.z), __fadd_rn(__fmul_rn(g.w, inv), dy.w));
}
// ... and, with gate != NULL, dy is the gradient of the GATED tensor y * gate[n,c] (embnet_channel_scale_fwd): the scaling's
// backward multiply (embnet_channel_scale_bwd's dx, one rounding) is applied here instead of being written and read back
__device__ __forceinline__ float4 gate4(float4 dy, const float* __restrict__ gate, long idx) {
  if (!gate) return dy;
  const float4 s = reinterpret_cast<const float4*>(gate)[idx];
  return make_float4(__fmul_rn(dy.x, s.x), __fmul_rn(dy.y, s.y), __fmul_rn(dy.z, s.z), __fmul_rn(dy.w, s.w));
}

__global__ __launch_bounds__(256) void bn_bwd_reduce4_gap_kernel(const float* __restrict__ dy, const float* __restrict__ dpool,
                                                                 const float* __restrict__ gate, DivU dhw, float inv_hw, const float* __restrict__ x, long m,
                                                                 int c4, ColGeom g, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, int relu,
                                                                 float* __restrict__ partial) {
  struct K4 { float4 sc, sh, mu, rs; };
  col_reduce2_v4p(m, c4, g, partial, [&](int q) {
    return K4{reinterpret_cast<const float4*>(scale)[q], reinterpret_cast<const float4*>(shift)[q],
              reinterpret_cast<const float4*>(mean)[q], reinterpret_cast<const float4*>(rstd)[q]};
  }, [&](long r, int q, const K4& k, float4& a, float4& b) {
    const float4 xv = reinterpret_cast<const float4*>(x)[r * c4 + q];
    const long nq = (long)divu((uint32_t)r, dhw) * c4 + q;
    const float4 gp = reinterpret_cast<const float4*>(dpool)[nq];
    float4 dz = add_pool4(gate4(reinterpret_cast<const float4*>(dy)[r * c4 + q], gate, nq), gp, inv_hw);
    const float4 sc = k.sc, sh = k.sh, mu = k.mu, rs = k.rs;
    if (relu) {
      dz.x = act_grad(relu, fmaf(xv.x, sc.x, sh.x), dz.x); dz.y = act_grad(relu, fmaf(xv.y, sc.y, sh.y), dz.y);
      dz.z = act_grad(relu, fmaf(xv.z, sc.z, sh.z), dz.z); dz.w = act_grad(relu, fmaf(xv.w, sc.w, sh.w), dz.w);
    }
    a.x += dz.x; a.y += dz.y; a.z += dz.z; a.w += dz.w;
    b.x = fmaf(dz.x, (xv.x - mu.x) * rs.x, b.x); b.y = fmaf(dz.y, (xv.y - mu.y) * rs.y, b.y);
    b.z = fmaf(dz.z, (xv.z - mu.z) * rs.z, b.z); b.w = fmaf(dz.w, (xv.w - mu.w) * rs.w, b.w);
  });
}

__global__ __launch_bounds__(256) void bn_bwd_apply4_gap_kernel(const float* __restrict__ dy, const float* __restrict__ dpool,
                                                                const float* __restrict__ gate, DivU dhwc4, float inv_hw, const float* __restrict__ x,
                                                                long total4, int c4, float inv_m, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* __restrict__ dbeta,
                                                                const float* __restrict__ dgamma, int relu, float* __restrict__ dx) {
  const long stride = (long)gridDim.x * 256;
  const bool fixed = stride % c4 == 0;                   // (as bn_bwd_apply4_kernel: the thread keeps its channel quad)
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 sc = z4, sh = z4, mu = z4, rs = z4, db = z4, dg = z4;
  auto consts = [&](int q) {
    sc = reinterpret_cast<const float4*>(scale)[q]; sh = reinterpret_cast<const float4*>(shift)[q];
    mu = reinterpret_cast<const float4*>(mean)[q]; rs = reinterpret_cast<const float4*>(rstd)[q];
    db = reinterpret_cast<const float4*>(dbeta)[q]; dg = reinterpret_cast<const float4*>(dgamma)[q];
  };
  if (fixed) consts((int)(((long)blockIdx.x * 256 + threadIdx.x) % c4));
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
    const int q = (int)(i % c4);
    if (!fixed) consts(q);
    const float4 xv = reinterpret_cast<const float4*>(x)[i];
    const long nq = (long)divu((uint32_t)i, dhwc4) * c4 + q;
    const float4 gp = reinterpret_cast<const float4*>(dpool)[nq];
    float4 dz = add_pool4(gate4(reinterpret_cast<const float4*>(dy)[i], gate, nq), gp, inv_hw);
    if (relu) {
      dz.x = act_grad(relu, fmaf(xv.x, sc.x, sh.x), dz.x); dz.y = act_grad(relu, fmaf(xv.y, sc.y, sh.y), dz.y);
      dz.z = act_grad(relu, fmaf(xv.z, sc.z, sh.z), dz.z); dz.w = act_grad(relu, fmaf(xv.w, sc.w, sh.w), dz.w);
    }
    float4 o;
    o.x = sc.x * (dz.x - db.x * inv_m - (xv.x - mu.x) * rs.x * dg.x * inv_m);
    o.y = sc.y * (dz.y - db.y * inv_m - (xv.y - mu.y) * rs.y * dg.y * inv_m);
    o.z = sc.z * (dz.z - db.z * inv_m - (xv.z - mu.z) * rs.z * dg.z * inv_m);
    o.w = sc.w * (dz.w - db.w * inv_m - (xv.w - mu.w) * rs.w * dg.w * inv_m);
    reinterpret_cast<float4*>(dx)[i] = o;
  }
}

// Per-(image, channel) sums over the pixels of an NHWC tensor (the squeeze-and-excite reductions): a workgroup = one image x
// `cls` channel quads (cls = 2^cls_log2 <= 16 consecutive threads -> up to 256 contiguous bytes per pixel), the other
// blockDim.x / cls thread groups stride the pixels.  256 threads on the small maps; 1024 where (channel blocks x images)
// alone would leave the chip with one 4-wave workgroup per CU (112^2 x 32: 256 workgroups for 411 MB — 1.2 TB/s with 256
// threads, half of them idle at c4 = 8).  Q sums per thread: pixel lanes of a wave by xor butterfly, then the <= 16 per-wave
// sums through LDS in wave order; thread q * cls + cl receives sum q of channel lane cl.
struct PixGeom { int cls_log2, xblocks, threads; };
static PixGeom pix_geom(int n, int hw, int c4) {
  PixGeom g;
  g.cls_log2 = 4;
  while (g.cls_log2 > 0 && (1 << (g.cls_log2 - 1)) >= c4) --g.cls_log2;
  g.xblocks = (c4 + (1 << g.cls_log2) - 1) >> g.cls_log2;
  static const long wide = env_long("EMBNET_PIX_WIDE", 1);               // A/B knob: 0 = 256 threads everywhere
  g.threads = (wide && (long)g.xblocks * n < 2048 && hw >= 256) ? 1024 : 256;
  return g;
}
template <int Q, class F>
__device__ __forceinline__ void pixel_lane_sums(float4 (&v)[Q], int cls, F store) {
  __shared__ float4 sh[Q][16][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6, cl = threadIdx.x & (cls - 1);
  for (int o = cls; o < 64; o <<= 1) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      v[q].x += __shfl_xor(v[q].x, o, 64); v[q].y += __shfl_xor(v[q].y, o, 64);
      v[q].z += __shfl_xor(v[q].z, o, 64); v[q].w += __shfl_xor(v[q].w, o, 64);
    }
  }
  if (lane < cls) {
#pragma unroll
    for (int q = 0; q < Q; ++q) sh[q][wave][cl] = v[q];
  }
  __syncthreads();
  if ((int)threadIdx.x < Q * cls) {
    const int q = threadIdx.x / cls;
    float4 a = sh[q][0][cl];
    for (int w = 1; w < nw; ++w) { const float4 o = sh[q][w][cl]; a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w; }
    store(q, a);
  }
}

// Squeeze-and-excite backward, first pass.  The block's tensor a = act(BN(x)) is gated: out = a * s[n,c], and pooled: s = f(mean_p a).
// With dg = d(out) this ONE pass over (dg, x) produces, per image and channel,
//   T0 = sum_p dg * a                (the gate's gradient: embnet_channel_scale_dgate's result, a recomputed from x)
//   S1 = sum_p a' dg   S2 = sum_p a'   S3 = sum_p a' dg xhat   S4 = sum_p a' xhat       (a' = act'(BN(x)), xhat = (x - mean) rstd)
// from which the BatchNorm-backward sums follow without a second pass over the tensors, because the layer's output gradient
// dz = a' (dg s[n,c] + dpool[n,c] / hw) is linear in the two per-(n,c) factors:
//   dbeta[c] = sum_n (s S1 + dpool/hw S2),   dgamma[c] = sum_n (s S3 + dpool/hw S4)      (se_bn_finalize_kernel, in double).
// Replaces embnet_channel_scale_dgate + bn_bwd_reduce4_gap (8 + 8 B per element) by one 8-byte pass.  out: [n][5][c].
__global__ __launch_bounds__(1024) void se_bn_sums4_kernel(const float* __restrict__ dg, const float* __restrict__ x, int hw, int c4,
                                                           int cls_log2, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                           float* __restrict__ out) {
  const int cls = 1 << cls_log2, n = blockIdx.y, cl = threadIdx.x & (cls - 1), pl = threadIdx.x >> cls_log2;
  const int npl = blockDim.x >> cls_log2;
  const int cq = blockIdx.x * cls + cl;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 v[5] = {z4, z4, z4, z4, z4};                      // T0, S1 .. S4
  if (cq < c4) {
    const float4 sc = reinterpret_cast<const float4*>(scale)[cq], sf = reinterpret_cast<const float4*>(shift)[cq];
    const float4 mu = reinterpret_cast<const float4*>(mean)[cq], rs = reinterpret_cast<const float4*>(rstd)[cq];
    const float4* dgi = reinterpret_cast<const float4*>(dg) + (long)n * hw * c4 + cq;
    const float4* xi = reinterpret_cast<const float4*>(x) + (long)n * hw * c4 + cq;
    auto one = [&](float dgv, float xq, float scq, float sfq, float muq, float rsq, float& T0, float& S1, float& S2, float& S3, float& S4) {
      const float z = fmaf(xq, scq, sfq);
      float av, ad;                                        // act(z) and act'(z); swish from ONE sigmoid (exp + rcp are quarter rate)
      if (act == 2) { const float sg = __frcp_rn(1.f + __expf(-z)); av = z * sg; ad = sg * fmaf(z, 1.f - sg, 1.f); }
      else if (act == 1) { av = fmaxf(z, 0.f); ad = z > 0.f ? 1.f : 0.f; }
      else { av = z; ad = 1.f; }
      T0 = fmaf(dgv, av, T0);
      const float xh = (xq - muq) * rsq, adg = ad * dgv;
      S1 += adg; S2 += ad; S3 = fmaf(adg, xh, S3); S4 = fmaf(ad, xh, S4);
    };
    auto quad = [&](const float4 d, const float4 xv) {
      one(d.x, xv.x, sc.x, sf.x, mu.x, rs.x, v[0].x, v[1].x, v[2].x, v[3].x, v[4].x);
      one(d.y, xv.y, sc.y, sf.y, mu.y, rs.y, v[0].y, v[1].y, v[2].y, v[3].y, v[4].y);
      one(d.z, xv.z, sc.z, sf.z, mu.z, rs.z, v[0].z, v[1].z, v[2].z, v[3].z, v[4].z);
      one(d.w, xv.w, sc.w, sf.w, mu.w, rs.w, v[0].w, v[1].w, v[2].w, v[3].w, v[4].w);
    };
    int p = pl;
    for (; p + npl < hw; p += 2 * npl) {                   // two pixels per trip: four 16-byte loads in flight per lane
      const float4 d0 = dgi[(long)p * c4], x0 = xi[(long)p * c4], d1 = dgi[(long)(p + npl) * c4], x1 = xi[(long)(p + npl) * c4];
      quad(d0, x0); quad(d1, x1);
    }
    if (p < hw) quad(dgi[(long)p * c4], xi[(long)p * c4]);
  }
  pixel_lane_sums<5>(v, cls, [&](int q, const float4 a) {
    if (cq < c4) reinterpret_cast<float4*>(out)[((long)n * 5 + q) * c4 + cq] = a;
  });
}

// dbeta / dgamma from the per-(n,c) sums (see se_bn_sums4_kernel): one WAVE per channel, lanes stride the images, the lane
// partials are added by a fixed butterfly in double (a serial loop over 256 images per thread was a 250 us latency chain)
__global__ __launch_bounds__(256) void se_bn_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ gate,
                                                             const float* __restrict__ dpool, int n, int c, float inv_hw,
                                                             float* __restrict__ dbeta, float* __restrict__ dgamma) {
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (col >= c) return;
  double b = 0.0, g = 0.0;
  for (int i = lane; i < n; i += 64) {
    const float* sp = sums + (long)i * 5 * c + col;
    const double s = (double)gate[(long)i * c + col], pq = (double)__fmul_rn(dpool[(long)i * c + col], inv_hw);
    b += s * (double)sp[c] + pq * (double)sp[2 * c];
    g += s * (double)sp[3 * c] + pq * (double)sp[4 * c];
  }
  b = wave_sum(b); g = wave_sum(g);
  if (lane == 0) { dbeta[col] = (float)b; dgamma[col] = (float)g; }
}

// BatchNorm backward apply for a BN whose INPUT is the output of a Conv2D / Dense with a fused ReLU (the small backbones'
// conv -> ReLU -> BN blocks, /root/reference/embedding_net/backbones.py:44-68): the ReLU's backward and the bias gradient in
// the same pass.  x >= 0 is the ReLU's output, so its mask is (x > 0):  dz = dx * [x > 0]  is what the producer's data /
// weight gradients need, and the bias gradient is the column sum of dz.  A column-reduction kernel (rows by block, as the
// statistics passes) that writes dz on the way: replaces bn_bwd_apply4 + relu_bwd_colsum (12 + 12 -> 16 bytes per element).
__global__ __launch_bounds__(256) void bn_bwd_apply_inrelu4_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                   long m, int c4, ColGeom g, float inv_m,
                                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                                   const float* __restrict__ dbeta, const float* __restrict__ dgamma,
                                                                   int relu, int training, float* __restrict__ dz_out,
                                                                   float* __restrict__ partial, const DropArg drop,
                                                                   uint32_t* __restrict__ range_slot) {
  struct K6 { float4 sc, sh, mu, rs, db, dg; };
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const uint64_t dseed = drop.thr ? drop_seed(drop) : 0ull;
  float amax = 0.f;                                     // max |dz| of this thread's elements (range_slot: conv.hip Ranges)
  col_reduce2_v4p(m, c4, g, partial, [&](int q) {
    K6 k{reinterpret_cast<const float4*>(scale)[q], reinterpret_cast<const float4*>(shift)[q], z4, z4, z4, z4};
    if (training) {
      k.mu = reinterpret_cast<const float4*>(mean)[q]; k.rs = reinterpret_cast<const float4*>(rstd)[q];
      k.db = reinterpret_cast<const float4*>(dbeta)[q]; k.dg = reinterpret_cast<const float4*>(dgamma)[q];
    }
    return k;
  }, [&](long r, int q, const K6& k, float4& a, float4&) {
    const float4 xv = reinterpret_cast<const float4*>(x)[r * c4 + q];
    float4 dz = reinterpret_cast<const float4*>(dy)[r * c4 + q];
    if (drop.thr) dz = drop4(drop, dseed, r * c4 + q, dz);
    const float4 sc = k.sc, sh = k.sh, mu = k.mu, rs = k.rs, db = k.db, dg = k.dg;
    if (relu) {
      dz.x = act_grad(relu, fmaf(xv.x, sc.x, sh.x), dz.x); dz.y = act_grad(relu, fmaf(xv.y, sc.y, sh.y), dz.y);
      dz.z = act_grad(relu, fmaf(xv.z, sc.z, sh.z), dz.z); dz.w = act_grad(relu, fmaf(xv.w, sc.w, sh.w), dz.w);
    }
    float4 o;
    if (training) {                                      // (the arithmetic of bn_bwd_apply4_kernel, term for term)
      o.x = sc.x * (dz.x - db.x * inv_m - (xv.x - mu.x) * rs.x * dg.x * inv_m);
      o.y = sc.y * (dz.y - db.y * inv_m - (xv.y - mu.y) * rs.y * dg.y * inv_m);
      o.z = sc.z * (dz.z - db.z * inv_m - (xv.z - mu.z) * rs.z * dg.z * inv_m);
      o.w = sc.w * (dz.w - db.w * inv_m - (xv.w - mu.w) * rs.w * dg.w * inv_m);
    } else {
      o = make_float4(sc.x * dz.x, sc.y * dz.y, sc.z * dz.z, sc.w * dz.w);
    }
    o.x = xv.x > 0.f ? o.x : 0.f; o.y = xv.y > 0.f ? o.y : 0.f; o.z = xv.z > 0.f ? o.z : 0.f; o.w = xv.w > 0.f ? o.w : 0.f;
    reinterpret_cast<float4*>(dz_out)[r * c4 + q] = o;
    a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
    amax = amax4(amax, o);
  });
  if (range_slot) range_emit_block(range_slot + 1 + blockIdx.x % RANGE_PARTIALS, amax);
}

// (A one-launch BatchNorm backward — sums, two spin barriers across a co-resident 1024-workgroup grid, then dx from the
// still-cached inputs — was built and measured in round 2: 144 us per layer against 27 + 31 us for the two passes
// (barrier latency, 16 instead of 32 waves per CU, 64-channel finalize on 64 workgroups).  Removed.)
// mean/var -> scale = gamma*rstd, shift = beta - mean*scale; moving stats updated in place.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int blocks, long m, int c,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, float momentum, float* __restrict__ mean_out,
                                                          float* __restrict__ rstd_out, float* __restrict__ scale,
                                                          float* __restrict__ shift, float* __restrict__ moving_mean,
                                                          float* __restrict__ moving_var, int by_channel,
                                                          float* __restrict__ bound = nullptr, float* __restrict__ xhat_bound = nullptr) {
  const int col = blockIdx.x;
  double s, ss;
  float qmax;
  block_partial_sums(partial, blocks, c, col, s, ss, by_channel != 0, &qmax);
  if (threadIdx.x) return;
  const double mean = s / (double)m;
  double var = ss / (double)m - mean * mean;            // biased, as Keras uses in training
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = (gamma ? gamma[col] : 1.f) * rstd;
  mean_out[col] = (float)mean; rstd_out[col] = rstd;
  const float sh = (beta ? beta[col] : 0.f) - (float)mean * sc;
  scale[col] = sc; shift[col] = sh;
  if (bound) bound[col] = channel_bound(sc, sh, qmax);     // |act(scale x + shift)| of this channel (see channel_bound)
  if (xhat_bound) xhat_bound[col] = (sqrtf(qmax) + fabsf((float)mean)) * rstd * 1.0009765625f;   // |x - mean| rstd <= this (backward: dx_channel_bound)
  if (moving_mean) moving_mean[col] = momentum * moving_mean[col] + (1.f - momentum) * (float)mean;
  if (moving_var) moving_var[col] = momentum * moving_var[col] + (1.f - momentum) * (float)var;
}

// inference: scale/shift from the moving statistics
__global__ __launch_bounds__(256) void bn_infer_prepare_kernel(int c, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               const float* __restrict__ moving_mean,
                                                               const float* __restrict__ moving_var, float eps,
                                                               float* __restrict__ scale, float* __restrict__ shift) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= c) return;
  const float sc = (gamma ? gamma[col] : 1.f) * rsqrtf(moving_var[col] + eps);
  scale[col] = sc; shift[col] = (beta ? beta[col] : 0.f) - moving_mean[col] * sc;
}

// y = [relu](x*scale[c] + shift[c])   (8 B/elem)
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, long total, int c,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         int relu, float* __restrict__ y, const DropArg drop,
                                                         const float* __restrict__ bound = nullptr, uint32_t* __restrict__ range_out = nullptr) {
  const long stride = (long)gridDim.x * 256;
  const uint64_t dseed = drop.thr ? drop_seed(drop) : 0ull;
  if (range_out && blockIdx.x == 0) {                      // the range slot of y for the convs that read it (conv.hip Ranges): B = max_c bound_c,
    const float b = tensor_bound(bound, c) * drop.keep_scale;   // x 1 / (1 - rate) behind a fused Dropout
    if (threadIdx.x == 0) *range_out = __float_as_uint(b);
  }
  if ((c & 3) == 0) {
    const long n4 = total >> 2;
    const int c4 = c >> 2;
    const bool fixed = stride % c4 == 0;                 // a thread keeps its channel quad: scale / shift loaded once
    float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = sc;
    if (fixed) {
      const int q = (int)(((long)blockIdx.x * 256 + threadIdx.x) % c4);
      sc = reinterpret_cast<const float4*>(scale)[q]; sh = reinterpret_cast<const float4*>(shift)[q];
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      if (!fixed) {
        const int col = (int)((i * 4) % c);
        sc = *reinterpret_cast<const float4*>(scale + col); sh = *reinterpret_cast<const float4*>(shift + col);
      }
      const float4 v = reinterpret_cast<const float4*>(x)[i];
      float4 o = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
      if (relu) { o.x = act_apply(relu, o.x); o.y = act_apply(relu, o.y); o.z = act_apply(relu, o.z); o.w = act_apply(relu, o.w); }
      if (drop.thr) o = drop4(drop, dseed, i, o);
      reinterpret_cast<float4*>(y)[i] = o;
    }
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
      const int col = (int)(i % c);
      float o = act_apply(relu, fmaf(x[i], scale[col], shift[col]));
      if (drop.thr) o = rng_u32(dseed, (uint64_t)i, 1) >= drop.thr ? o * drop.keep_scale : 0.f;
      y[i] = o;
    }
  }
}

// backward reductions: dbeta = sum dz, dgamma = sum dz*xhat, dz = dy * [x*scale+shift > 0]   (8 B/elem)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            long m, int c, ColGeom g, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int relu,
                                                            float* __restrict__ partial) {
  col_reduce2(m, c, g, partial, [&](long r, int col, float& a, float& b) {
    const float xv = x[r * c + col];
    float dz = dy[r * c + col];
    if (relu) dz = act_grad(relu, fmaf(xv, scale[col], shift[col]), dz);
    a += dz; b = fmaf(dz, (xv - mean[col]) * rstd[col], b);
  });
}

// The range of a BatchNorm backward's dx, known before its apply pass runs (replaces round 5's dry run of the pass):
//     dx = scale_c (dz - dbeta_c / m - xhat dgamma_c / m)   =>   |dx| <= |scale_c| (max |dz| + |dbeta_c| / m + max |xhat| |dgamma_c| / m)
// with max |dz| of channel c from the reduction pass (its third accumulator / the data-gradient epilogue's third plane) and
// max |xhat| from the forward statistics (bn_finalize_kernel's xhat_bound).  Above the true maximum by the two correction terms'
// share (they are O(1 / sqrt(m)) of the first for a gradient that does not correlate with the batch): typically within a binade.
__device__ __forceinline__ float dx_channel_bound(float sc, float mz, float db, float dg, float xh, float inv_m, int training) {
  const float corr = training ? (fabsf(db) + xh * fabsf(dg)) * inv_m : 0.f;
  return fabsf(sc) * (mz + corr) * 1.0009765625f;
}
// pmax (optional): max |dz| partials — [blocks][c] behind the own reduction's sums (by_channel = 0) or the third plane [c][blocks] of
// a data-gradient epilogue's partials (by_channel = 1); dx_bound [c] receives dx_channel_bound
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int blocks, int c,
                                                              float* __restrict__ dbeta, float* __restrict__ dgamma,
                                                              int by_channel = 0, uint32_t* __restrict__ zero_slot = nullptr,
                                                              const float* __restrict__ pmax = nullptr, const float* __restrict__ scale = nullptr,
                                                              const float* __restrict__ xhat_bound = nullptr, float inv_m = 0.f,
                                                              int training = 1, float* __restrict__ dx_bound = nullptr) {
  const int col = blockIdx.x;
  double s, ss;
  block_partial_sums(partial, blocks, c, col, s, ss, by_channel != 0);
  if (dx_bound) {
    __shared__ float redm[4];
    float mz = 0.f;
    if (by_channel) for (int b = threadIdx.x; b < blocks; b += 256) mz = fmaxf(mz, pmax[(long)col * blocks + b]);
    else for (int b = threadIdx.x; b < blocks; b += 256) mz = fmaxf(mz, pmax[(long)b * c + col]);
    mz = wave_max(mz);
    if ((threadIdx.x & 63) == 0) redm[threadIdx.x >> 6] = mz;
    __syncthreads();
    if (threadIdx.x == 0)
      dx_bound[col] = dx_channel_bound(scale[col], fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3])), (float)s, (float)ss,
                                       xhat_bound ? xhat_bound[col] : 0.f, inv_m, training && xhat_bound);
  }
  if (threadIdx.x == 0) { dbeta[col] = (float)s; dgamma[col] = (float)ss; }
  if (zero_slot && col == 0)                                           // the apply pass behind this kernel emits dx's range there
    for (int i = threadIdx.x; i < 1 + RANGE_PARTIALS; i += 256) zero_slot[i] = 0u;
}
__global__ __launch_bounds__(256) void range_zero_kernel(uint32_t* __restrict__ slot) {
  for (int i = threadIdx.x; i < 1 + RANGE_PARTIALS; i += 256) slot[i] = 0u;
}
// (instead of hipMemsetAsync: a memset node inside a captured HIP graph did not zero a range slot on replay — round 6, ROCm 7.2 —
// so nothing a captured step may run uses one)
__global__ __launch_bounds__(256) void zero2_kernel(float* __restrict__ a, float* __restrict__ b, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) { a[i] = 0.f; if (b) b[i] = 0.f; }
}
// word 0 of a range slot = the maximum of the workgroup partials behind it (common.h range_emit_block)
__global__ __launch_bounds__(256) void range_fold_kernel(uint32_t* __restrict__ slot) {
  uint32_t m = 0u;
  for (int i = threadIdx.x; i < RANGE_PARTIALS; i += 256) m = max(m, slot[1 + i]);
  range_emit_block(slot, __uint_as_float(m));                           // (bit patterns of non-negative floats order like the floats)
}

// A gradient's range slot (conv.hip Ranges) is embnet_range_slot_words() uint32 words: the range + the workgroup partials.  The BatchNorm
// backward entry points that write an fp32 dx take it as their `dx_range` ARGUMENT (embnet_bn_bwd_ex, embnet_bn_bwd_partials_ex,
// embnet_bn_act_maxpool_bwd_ex; ABI 21) and leave max |dx| in its first word; a call that cannot (planes-only dx, scalar kernels, no
// saved statistics) fails.  DEPRECATED: embnet_range_emit(slot) arms the same request for the NEXT non-_ex call of the calling
// thread (ABI 20; hidden per-thread state — see embnet_conv2d_ranges in conv.hip); every embnet_bn_bwd* entry point clears it.
extern "C" int embnet_range_slot_words(void) { return 1 + RANGE_PARTIALS; }
static thread_local uint32_t* t_emit_slot = nullptr;
extern "C" int embnet_range_emit(uint32_t* slot) {
  EMBNET_CHECK_ARG(!(reinterpret_cast<uintptr_t>(slot) & 3), "range_emit: the slot is 4-byte aligned");
  t_emit_slot = slot;
  return 0;
}
static uint32_t* take_emit_slot() { uint32_t* s = t_emit_slot; t_emit_slot = nullptr; return s; }

// dx = scale * (dz - dbeta/M - xhat*dgamma/M)     (12 B/elem)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           long total, int c, float inv_m, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ dbeta,
                                                           const float* __restrict__ dgamma, int relu, int training,
                                                           const float* __restrict__ dx_add, float* __restrict__ dx) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int col = (int)(i % c);
    const float xv = x[i];
    float dz = dy[i];
    if (relu) dz = act_grad(relu, fmaf(xv, scale[col], shift[col]), dz);
    float o;
    if (training) {
      const float xh = (xv - mean[col]) * rstd[col];
      o = scale[col] * (dz - dbeta[col] * inv_m - xh * dgamma[col] * inv_m);
    } else {
      o = scale[col] * dz;           // frozen statistics: plain affine
    }
    dx[i] = dx_add ? o + dx_add[i] : o;
  }
}

// ---------------------------------------------------------------- pooling
// y = max over the window; taps outside the image read as 0 (ZeroPadding2D semantics) and carry no
// gradient; first maximum in row-major tap order wins.  argmax: tap index or 255.
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, int n, int h, int w, int c,
                                                          int k, int stride, int pad, int oh, int ow,
                                                          float* __restrict__ y, uint8_t* __restrict__ argmax) {
  const long total = (long)n * oh * ow * c;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int col = (int)(i % c);
  long t = i / c;
  const int x_o = (int)(t % ow); t /= ow;
  const int y_o = (int)(t % oh);
  const int b = (int)(t / oh);
  float best = -INFINITY; int bi = 255;
  for (int dy = 0; dy < k; ++dy)
    for (int dx = 0; dx < k; ++dx) {
      const int ih = y_o * stride + dy - pad, iw = x_o * stride + dx - pad;
      const bool in = (unsigned)ih < (unsigned)h && (unsigned)iw < (unsigned)w;
      const float v = in ? x[(((long)b * h + ih) * w + iw) * c + col] : 0.f;
      if (v > best) { best = v; bi = in ? dy * k + dx : 255; }
    }
  y[i] = best; argmax[i] = (uint8_t)bi;
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ argmax,
                                                          int n, int h, int w, int c, int k, int stride, int pad,
                                                          int oh, int ow, float* __restrict__ dx) {
  const long total = (long)n * h * w * c;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int col = (int)(i % c);
  long t = i / c;
  const int iw = (int)(t % w); t /= w;
  const int ih = (int)(t % h);
  const int b = (int)(t / h);
  float g = 0.f;
  // windows (y_o,x_o) with y_o*stride - pad <= ih < y_o*stride - pad + k
  const int y_hi = min((ih + pad) / stride, oh - 1), x_hi = min((iw + pad) / stride, ow - 1);
  for (int y_o = y_hi; y_o >= 0 && y_o * stride - pad + k > ih; --y_o)
    for (int x_o = x_hi; x_o >= 0 && x_o * stride - pad + k > iw; --x_o) {
      const int tap = (ih - (y_o * stride - pad)) * k + (iw - (x_o * stride - pad));
      const long o = (((long)b * oh + y_o) * ow + x_o) * c + col;
      if (argmax[o] == tap) g += dy[o];
    }
  dx[i] = g;
}

// Four channels per thread (C % 4 == 0): 16-byte loads/stores, 4-byte arg-max words.
__global__ __launch_bounds__(256) void maxpool_fwd4_kernel(const float* __restrict__ x, int n, int h, int w, int c4,
                                                           int k, int stride, int pad, int oh, int ow,
                                                           float* __restrict__ y, uint8_t* __restrict__ argmax,
                                                           uint32_t* __restrict__ range_slot) {
  const long total = (long)n * oh * ow * c4;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  float amax = 0.f;
  if (i < total) {
    const int col = (int)(i % c4);
    long t = i / c4;
    const int x_o = (int)(t % ow); t /= ow;
    const int y_o = (int)(t % oh);
    const int b = (int)(t / oh);
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {255, 255, 255, 255};
    for (int dy = 0; dy < k; ++dy)
      for (int dx = 0; dx < k; ++dx) {
        const int ih = y_o * stride + dy - pad, iw = x_o * stride + dx - pad;
        const bool in = (unsigned)ih < (unsigned)h && (unsigned)iw < (unsigned)w;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in) v = reinterpret_cast<const float4*>(x)[(((long)b * h + ih) * w + iw) * c4 + col];
        const float vv[4] = {v.x, v.y, v.z, v.w};
        const int tap = in ? dy * k + dx : 255;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (vv[j] > best[j]) { best[j] = vv[j]; bi[j] = tap; }
      }
    const float4 o = make_float4(best[0], best[1], best[2], best[3]);
    reinterpret_cast<float4*>(y)[i] = o;
    reinterpret_cast<uint32_t*>(argmax)[i] = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) |
                                             ((uint32_t)bi[3] << 24);
    amax = amax4(0.f, o);
  }
  // the exact max |y| of the pooled tensor for the three-product conv that reads it (conv.hip Ranges; `simple`: conv -> ReLU -> pool)
  if (range_slot) range_emit_block(range_slot + 1 + blockIdx.x % RANGE_PARTIALS, amax);
}

__global__ __launch_bounds__(256) void maxpool_bwd4_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ argmax,
                                                           int n, int h, int w, int c4, int k, int stride, int pad,
                                                           int oh, int ow, float* __restrict__ dx) {
  const long total = (long)n * h * w * c4;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int col = (int)(i % c4);
  long t = i / c4;
  const int iw = (int)(t % w); t /= w;
  const int ih = (int)(t % h);
  const int b = (int)(t / h);
  float g[4] = {0.f, 0.f, 0.f, 0.f};
  const int y_hi = min((ih + pad) / stride, oh - 1), x_hi = min((iw + pad) / stride, ow - 1);
  for (int y_o = y_hi; y_o >= 0 && y_o * stride - pad + k > ih; --y_o)
    for (int x_o = x_hi; x_o >= 0 && x_o * stride - pad + k > iw; --x_o) {
      const uint32_t tap = (uint32_t)((ih - (y_o * stride - pad)) * k + (iw - (x_o * stride - pad)));
      const long o = (((long)b * oh + y_o) * ow + x_o) * c4 + col;
      // both loads unconditional: independent of each other and of the compare, so the (at most four)
      // windows of a pixel are all in flight together; dy lines are shared by neighbouring pixels (L1/L2)
      const uint32_t am = reinterpret_cast<const uint32_t*>(argmax)[o];
      const float4 d = reinterpret_cast<const float4*>(dy)[o];
      g[0] += (am & 0xff) == tap ? d.x : 0.f;
      g[1] += ((am >> 8) & 0xff) == tap ? d.y : 0.f;
      g[2] += ((am >> 16) & 0xff) == tap ? d.z : 0.f;
      g[3] += (am >> 24) == tap ? d.w : 0.f;
    }
  reinterpret_cast<float4*>(dx)[i] = make_float4(g[0], g[1], g[2], g[3]);
}

// conv -> ReLU -> MaxPool (the `simple` backbone, reference backbones.py:21-31): the pool's backward, the ReLU mask of the
// conv in front of it and that conv's bias gradient in one pass over the conv output.  Unfused the pooled gradient is
// scattered into a full-size tensor (4 B/elem written) that relu_bwd_colsum reads back beside y (12 B/elem): 16 + 5/s^2
// bytes per element of y; here 8 + 5/s^2.  y is the conv's ReLU output (= the pool's input); windows whose maximum is 0
// carry no gradient either way (the mask zeroes them), every other value is the same sum in the same window order.
__global__ __launch_bounds__(256) void maxpool_relu_bwd_colsum4_kernel(
    const float* __restrict__ dy, const uint8_t* __restrict__ argmax, const float* __restrict__ y, long m, int h, int w,
    DivU dw, DivU dh, int c4, int k, int stride, int pad, int oh, int ow, ColGeom g, float* __restrict__ dz,
    float* __restrict__ partial, uint32_t* __restrict__ range_slot) {
  float amax = 0.f;                                     // max |dz| of this thread's elements (range_slot: conv.hip Ranges)
  col_reduce2_v4(m, c4, g, partial, [&](long r, int q, float4& a, float4& b) {
    const uint32_t t = divu((uint32_t)r, dw);
    const int iw = (int)((uint32_t)r - t * (uint32_t)w);
    const uint32_t img = divu(t, dh);
    const int ih = (int)(t - img * (uint32_t)h);
    const float4 yv = reinterpret_cast<const float4*>(y)[r * c4 + q];
    float gr[4] = {0.f, 0.f, 0.f, 0.f};
    const int y_hi = min((ih + pad) / stride, oh - 1), x_hi = min((iw + pad) / stride, ow - 1);
    for (int y_o = y_hi; y_o >= 0 && y_o * stride - pad + k > ih; --y_o)
      for (int x_o = x_hi; x_o >= 0 && x_o * stride - pad + k > iw; --x_o) {
        const uint32_t tap = (uint32_t)((ih - (y_o * stride - pad)) * k + (iw - (x_o * stride - pad)));
        const long o = (((long)img * oh + y_o) * ow + x_o) * c4 + q;
        const uint32_t am = reinterpret_cast<const uint32_t*>(argmax)[o];
        const float4 d = reinterpret_cast<const float4*>(dy)[o];
        gr[0] += (am & 0xff) == tap ? d.x : 0.f;
        gr[1] += ((am >> 8) & 0xff) == tap ? d.y : 0.f;
        gr[2] += ((am >> 16) & 0xff) == tap ? d.z : 0.f;
        gr[3] += (am >> 24) == tap ? d.w : 0.f;
      }
    const float4 v = make_float4(yv.x > 0.f ? gr[0] : 0.f, yv.y > 0.f ? gr[1] : 0.f, yv.z > 0.f ? gr[2] : 0.f,
                                 yv.w > 0.f ? gr[3] : 0.f);
    reinterpret_cast<float4*>(dz)[r * c4 + q] = v;
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    amax = amax4(amax, v);
  });
  if (range_slot) range_emit_block(range_slot + 1 + blockIdx.x % RANGE_PARTIALS, amax);
}

// ---------------------------------------------------------------- BN-apply + activation + max-pool, fused
// The zoo ResNet stem is bn0 -> relu -> ZeroPadding2D(1) -> MaxPool(3,2) on the largest activation of the
// net (112x112x64 per image).  Unfused that is: BN apply (read x, write a), pool (read a, write y), and
// backward pool (write da), BN reduce (read da, x), BN apply (read da, x, write dx) — five passes over the
// big tensor.  Fused: the pool reads x and applies the affine + activation per tap (the padded zeros stay
// zeros: padding follows the activation in the reference), and backward works from the pooled gradient:
// the BN reductions run over the POOLED elements (each routes to the one input pixel its arg-max names,
// whose x is gathered), and the apply pass rebuilds da per input pixel from the <= 4 windows that cover
// it.  a and da never exist.  Bytes: fwd 4*in + 5*out; bwd reduce ~4*in + 9*out; bwd apply 8*in + 5*out.
__global__ __launch_bounds__(256) void affine_act_maxpool_fwd4_kernel(
    const float* __restrict__ x, int n, int h, int w, int c4, const float* __restrict__ scale,
    const float* __restrict__ shift, int act, int k, int stride, int pad, int oh, int ow, float* __restrict__ y,
    uint8_t* __restrict__ argmax, float* __restrict__ xwin) {
  const long total = (long)n * oh * ow * c4;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int col = (int)(i % c4);
  long t = i / c4;
  const int x_o = (int)(t % ow); t /= ow;
  const int y_o = (int)(t % oh);
  const int b = (int)(t / oh);
  const float4 sc = reinterpret_cast<const float4*>(scale)[col], sh = reinterpret_cast<const float4*>(shift)[col];
  float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  float bx[4] = {0.f, 0.f, 0.f, 0.f};                    // the BN input at the winning tap (backward's reduction reads it)
  int bi[4] = {255, 255, 255, 255};
  for (int dy = 0; dy < k; ++dy)
    for (int dx = 0; dx < k; ++dx) {
      const int ih = y_o * stride + dy - pad, iw = x_o * stride + dx - pad;
      const bool in = (unsigned)ih < (unsigned)h && (unsigned)iw < (unsigned)w;
      float vv[4] = {0.f, 0.f, 0.f, 0.f}, xx[4] = {0.f, 0.f, 0.f, 0.f};
      if (in) {
        const float4 v = reinterpret_cast<const float4*>(x)[(((long)b * h + ih) * w + iw) * c4 + col];
        xx[0] = v.x; xx[1] = v.y; xx[2] = v.z; xx[3] = v.w;
        vv[0] = act_apply(act, fmaf(v.x, sc.x, sh.x)); vv[1] = act_apply(act, fmaf(v.y, sc.y, sh.y));
        vv[2] = act_apply(act, fmaf(v.z, sc.z, sh.z)); vv[3] = act_apply(act, fmaf(v.w, sc.w, sh.w));
      }
      const int tap = in ? dy * k + dx : 255;
#pragma unroll
      for (int j = 0; j < 4; ++j) if (vv[j] > best[j]) { best[j] = vv[j]; bi[j] = tap; bx[j] = xx[j]; }
    }
  if (xwin) reinterpret_cast<float4*>(xwin)[i] = make_float4(bx[0], bx[1], bx[2], bx[3]);
  reinterpret_cast<float4*>(y)[i] = make_float4(best[0], best[1], best[2], best[3]);
  reinterpret_cast<uint32_t*>(argmax)[i] = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) |
                                           ((uint32_t)bi[3] << 24);
}

// dbeta / dgamma partial sums over the pooled elements (rows = n*oh*ow pooled pixels)
__global__ __launch_bounds__(256) void pool_bn_bwd_reduce4_kernel(
    const float* __restrict__ dy, const uint8_t* __restrict__ argmax, const float* __restrict__ x, long mp, int h, int w,
    int c4, int k, int stride, int pad, int oh, int ow, ColGeom g, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ scale, const float* __restrict__ shift, int act,
    const float* __restrict__ xwin, float* __restrict__ partial) {
  col_reduce2_v4(mp, c4, g, partial, [&](long r, int q, float4& a, float4& b) {
    const uint32_t am = reinterpret_cast<const uint32_t*>(argmax)[r * c4 + q];
    float4 xw4 = make_float4(0.f, 0.f, 0.f, 0.f);      // forward kept the winners' inputs: a streaming read, no gather
    if (xwin) xw4 = reinterpret_cast<const float4*>(xwin)[r * c4 + q];
    const float xwv[4] = {xw4.x, xw4.y, xw4.z, xw4.w};
    const float4 d = reinterpret_cast<const float4*>(dy)[r * c4 + q];
    const int x_o = (int)(r % ow); const long t = r / ow;
    const int y_o = (int)(t % oh), bb = (int)(t / oh);
    const float4 sc = reinterpret_cast<const float4*>(scale)[q], sh = reinterpret_cast<const float4*>(shift)[q];
    const float4 mu = reinterpret_cast<const float4*>(mean)[q], rs = reinterpret_cast<const float4*>(rstd)[q];
    const float dd[4] = {d.x, d.y, d.z, d.w}, scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
    const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, rsv[4] = {rs.x, rs.y, rs.z, rs.w};
    float av[4], bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t tap = (am >> (8 * j)) & 0xff;
      const bool on = tap != 255u;                       // window maximum was a padding zero: no gradient
      const uint32_t tp = on ? tap : 0u;
      const int ih = min(max(y_o * stride - pad + (int)(tp / (uint32_t)k), 0), h - 1);
      const int iw = min(max(x_o * stride - pad + (int)(tp % (uint32_t)k), 0), w - 1);
      const float xv = xwin ? xwv[j] : x[((((long)bb * h + ih) * w + iw) * c4 + q) * 4 + j];
      const float dz = on ? act_grad(act, fmaf(xv, scv[j], shv[j]), dd[j]) : 0.f;
      av[j] = dz; bv[j] = dz * ((xv - muv[j]) * rsv[j]);
    }
    a.x += av[0]; a.y += av[1]; a.z += av[2]; a.w += av[3];
    b.x += bv[0]; b.y += bv[1]; b.z += bv[2]; b.w += bv[3];
  });
}

// dx over the input pixels: da rebuilt from the covering windows, then the BN-backward formula
__global__ __launch_bounds__(256) void pool_bn_bwd_apply4_kernel(
    const float* __restrict__ dy, const uint8_t* __restrict__ argmax, const float* __restrict__ x, int n, int h, int w,
    int c4, int k, int stride, int pad, int oh, int ow, float inv_m, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ dbeta, const float* __restrict__ dgamma, int act, int training, float* __restrict__ dx,
    uint32_t* __restrict__ range_slot) {
  const long total = (long)n * h * w * c4;
  const long i0 = (long)blockIdx.x * 256 + threadIdx.x;
  const bool valid = i0 < total;                         // (no early return: every lane takes part in the range's wave maximum)
  const long i = valid ? i0 : total - 1;
  const int q = (int)(i % c4);
  long t = i / c4;
  const int iw = (int)(t % w); t /= w;
  const int ih = (int)(t % h);
  const int b = (int)(t / h);
  const float4 xv = reinterpret_cast<const float4*>(x)[i];
  float g[4] = {0.f, 0.f, 0.f, 0.f};
  const int y_hi = min((ih + pad) / stride, oh - 1), x_hi = min((iw + pad) / stride, ow - 1);
  if (k <= 2 * stride) {
    // at most two windows per dimension cover a pixel (3x3/2, 2x2/2): all four candidates are fetched unconditionally
    // (clamped addresses, masked afterwards) so the loads are in flight together; same accumulation order as the loops
    uint32_t am[4]; float4 d[4]; bool ok[4]; uint32_t tp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int y_o = y_hi - (j >> 1), x_o = x_hi - (j & 1);
      ok[j] = y_o >= 0 && x_o >= 0 && y_o * stride - pad + k > ih && x_o * stride - pad + k > iw;
      const int yc = max(y_o, 0), xc = max(x_o, 0);
      tp[j] = (uint32_t)((ih - (yc * stride - pad)) * k + (iw - (xc * stride - pad)));
      const long o = (((long)b * oh + yc) * ow + xc) * c4 + q;
      am[j] = reinterpret_cast<const uint32_t*>(argmax)[o];
      d[j] = reinterpret_cast<const float4*>(dy)[o];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!ok[j]) continue;
      g[0] += (am[j] & 0xff) == tp[j] ? d[j].x : 0.f;
      g[1] += ((am[j] >> 8) & 0xff) == tp[j] ? d[j].y : 0.f;
      g[2] += ((am[j] >> 16) & 0xff) == tp[j] ? d[j].z : 0.f;
      g[3] += (am[j] >> 24) == tp[j] ? d[j].w : 0.f;
    }
  } else
  for (int y_o = y_hi; y_o >= 0 && y_o * stride - pad + k > ih; --y_o)
    for (int x_o = x_hi; x_o >= 0 && x_o * stride - pad + k > iw; --x_o) {
      const uint32_t tap = (uint32_t)((ih - (y_o * stride - pad)) * k + (iw - (x_o * stride - pad)));
      const long o = (((long)b * oh + y_o) * ow + x_o) * c4 + q;
      const uint32_t am = reinterpret_cast<const uint32_t*>(argmax)[o];
      const float4 d = reinterpret_cast<const float4*>(dy)[o];
      g[0] += (am & 0xff) == tap ? d.x : 0.f;
      g[1] += ((am >> 8) & 0xff) == tap ? d.y : 0.f;
      g[2] += ((am >> 16) & 0xff) == tap ? d.z : 0.f;
      g[3] += (am >> 24) == tap ? d.w : 0.f;
    }
  const float4 sc = reinterpret_cast<const float4*>(scale)[q], sh = reinterpret_cast<const float4*>(shift)[q];
  float4 dz = make_float4(act_grad(act, fmaf(xv.x, sc.x, sh.x), g[0]), act_grad(act, fmaf(xv.y, sc.y, sh.y), g[1]),
                          act_grad(act, fmaf(xv.z, sc.z, sh.z), g[2]), act_grad(act, fmaf(xv.w, sc.w, sh.w), g[3]));
  float4 o;
  if (training) {
    const float4 mu = reinterpret_cast<const float4*>(mean)[q], rs = reinterpret_cast<const float4*>(rstd)[q];
    const float4 db = reinterpret_cast<const float4*>(dbeta)[q], dg = reinterpret_cast<const float4*>(dgamma)[q];
    o.x = sc.x * (dz.x - db.x * inv_m - (xv.x - mu.x) * rs.x * dg.x * inv_m);
    o.y = sc.y * (dz.y - db.y * inv_m - (xv.y - mu.y) * rs.y * dg.y * inv_m);
    o.z = sc.z * (dz.z - db.z * inv_m - (xv.z - mu.z) * rs.z * dg.z * inv_m);
    o.w = sc.w * (dz.w - db.w * inv_m - (xv.w - mu.w) * rs.w * dg.w * inv_m);
  } else {
    o = make_float4(sc.x * dz.x, sc.y * dz.y, sc.z * dz.z, sc.w * dz.w);
  }
  if (valid) reinterpret_cast<float4*>(dx)[i] = o;
  if (range_slot) range_emit_block(range_slot + 1 + blockIdx.x % RANGE_PARTIALS, valid ? amax4(0.f, o) : 0.f);
}

// y[n,c] = mean over hw.  Workgroup = one sample x 64 channels (16 channel-quad lanes x 16 pixel lanes).
__global__ __launch_bounds__(256) void gap_fwd4_kernel(const float* __restrict__ x, int hw, int c4, float* __restrict__ y) {
  __shared__ float4 sh[256];
  const int n = blockIdx.y, cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int cq = blockIdx.x * 16 + cl;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cq < c4)
    for (int p = pl; p < hw; p += 16) {
      const float4 v = reinterpret_cast<const float4*>(x)[((long)n * hw + p) * c4 + cq];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  sh[threadIdx.x] = acc;
  __syncthreads();
  if (pl == 0 && cq < c4) {
    for (int k = 1; k < 16; ++k) { const float4 o = sh[k * 16 + cl]; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
    const float inv = 1.f / (float)hw;
    reinterpret_cast<float4*>(y)[(long)n * c4 + cq] = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  }
}

// BatchNorm apply (+ activation) and GlobalAveragePooling of the result in one pass: y = act(x*scale + shift) is written
// and its per-image channel means come out of the same read (squeeze-and-excite pools the tensor it then scales).
// Same decomposition as gap_fwd4_kernel (16 channel quads x 16 pixel lanes per workgroup, grid = (c4/16, n)); four
// pixels per trip so four loads are in flight per lane.
__global__ __launch_bounds__(1024) void affine_act_gap4_kernel(const float* __restrict__ x, int hw, int c4, int cls_log2,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               int act, float* __restrict__ y, float* __restrict__ gap) {
  const int cls = 1 << cls_log2, n = blockIdx.y, cl = threadIdx.x & (cls - 1), pl = threadIdx.x >> cls_log2;
  const int npl = blockDim.x >> cls_log2;
  const int cq = blockIdx.x * cls + cl;
  float4 acc[1] = {make_float4(0.f, 0.f, 0.f, 0.f)};
  if (cq < c4) {
    const float4 sc = reinterpret_cast<const float4*>(scale)[cq], sf = reinterpret_cast<const float4*>(shift)[cq];
    const float4* xi = reinterpret_cast<const float4*>(x) + (long)n * hw * c4 + cq;
    float4* yi = reinterpret_cast<float4*>(y) + (long)n * hw * c4 + cq;
    auto one = [&](const float4 v, int p) {
      float4 o = make_float4(fmaf(v.x, sc.x, sf.x), fmaf(v.y, sc.y, sf.y), fmaf(v.z, sc.z, sf.z), fmaf(v.w, sc.w, sf.w));
      if (act) { o.x = act_apply(act, o.x); o.y = act_apply(act, o.y); o.z = act_apply(act, o.z); o.w = act_apply(act, o.w); }
      if (y) yi[(long)p * c4] = o;                        // y == NULL: pooled means only (the tensor is formed later, gated)
      acc[0].x += o.x; acc[0].y += o.y; acc[0].z += o.z; acc[0].w += o.w;
    };
    int p = pl;
    for (; p + 3 * npl < hw; p += 4 * npl) {
      const float4 v0 = xi[(long)p * c4], v1 = xi[(long)(p + npl) * c4], v2 = xi[(long)(p + 2 * npl) * c4], v3 = xi[(long)(p + 3 * npl) * c4];
      one(v0, p); one(v1, p + npl); one(v2, p + 2 * npl); one(v3, p + 3 * npl);
    }
    for (; p < hw; p += npl) one(xi[(long)p * c4], p);
  }
  const float inv = 1.f / (float)hw;
  pixel_lane_sums<1>(acc, cls, [&](int, const float4 a) {
    if (cq < c4) reinterpret_cast<float4*>(gap)[(long)n * c4 + cq] = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
  });
}

__global__ __launch_bounds__(256) void gap_fwd_kernel(const float* __restrict__ x, int n, int hw, int c,
                                                      float* __restrict__ y) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * c) return;
  const int col = i % c, b = i / c;
  float s = 0.f;
  for (int p = 0; p < hw; ++p) s += x[((long)b * hw + p) * c + col];
  y[i] = s / (float)hw;
}

__global__ __launch_bounds__(256) void gap_bwd_kernel(const float* __restrict__ dy, int n, int hw, int c,
                                                      float* __restrict__ dx) {
  const long total = (long)n * hw * c;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int col = (int)(i % c);
  const int b = (int)(i / ((long)hw * c));
  dx[i] = dy[b * c + col] / (float)hw;
}

// dx = dx_add + dy/hw broadcast, four channels per thread: the pooled tensor's OTHER consumer (squeeze-and-excite: the
// channel scaling) hands its gradient in and the sum is written once — no broadcast tensor, no accumulation pass
__global__ __launch_bounds__(256) void gap_bwd_add4_kernel(const float* __restrict__ dy, const float* __restrict__ dx_add,
                                                           long total4, int hw, int c4, float* __restrict__ dx) {
  const float inv = 1.f / (float)hw;
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
    const int q = (int)(i % c4);
    const long b = i / ((long)hw * c4);
    const float4 g = reinterpret_cast<const float4*>(dy)[b * c4 + q];
    float4 o = make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv);
    if (dx_add) {
      const float4 a = reinterpret_cast<const float4*>(dx_add)[i];
      o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    }
    reinterpret_cast<float4*>(dx)[i] = o;
  }
}

// ---------------------------------------------------------------- elementwise
// dz = dy * [y > 0];  optional column sums of dz -> bias gradient partials
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                       long total, float* __restrict__ dz) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) dz[i] = y[i] > 0.f ? dy[i] : 0.f;
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long m, int c, ColGeom g,
                                                     float* __restrict__ partial) {
  col_reduce2(m, c, g, partial, [&](long r, int col, float& a, float& b) { a += x[r * c + col]; });
}

// dz = dy * [y > 0] AND the column sums of dz (the bias gradient of a Conv2D / Dense with a fused ReLU) in one pass over
// the tensor: the separate relu_bwd + colsum pair read dz back (12 + 4 bytes per element -> 12)
__global__ __launch_bounds__(256) void relu_bwd_colsum_kernel(const float* __restrict__ dy, const float* __restrict__ y, long m, int c,
                                                              ColGeom g, float* __restrict__ dz, float* __restrict__ partial,
                                                              uint32_t* __restrict__ range_slot) {
  float amax = 0.f;
  col_reduce2(m, c, g, partial, [&](long r, int col, float& a, float& b) {
    const long i = r * c + col;
    const float v = y[i] > 0.f ? dy[i] : 0.f;
    dz[i] = v; a += v;
    amax = fmaxf(amax, fabsf(v));
  });
  if (range_slot) range_emit_block(range_slot + 1 + blockIdx.x % RANGE_PARTIALS, amax);
}

// Sum over several tensors of alpha_t * sum(x_t^2) — all the kernel_regularizer=l2(lambda) terms of a model
// (backbones.py:22-36) in one launch pair: chunk c = (tensor, 4096-element block) -> partial[c]; one workgroup adds the
// partials in double, in chunk order (reproducible).
struct SumsqTensor { const float* x; long n; float alpha; int pad; };
static_assert(sizeof(SumsqTensor) == 24, "descriptor layout is part of the ABI (include/embnet.h)");
__global__ __launch_bounds__(256) void sumsq_multi_kernel(const SumsqTensor* __restrict__ table, const int* __restrict__ chunks,
                                                          float* __restrict__ partial) {
  __shared__ float part[4];
  const SumsqTensor t = table[chunks[2 * blockIdx.x]];
  const long first = (long)chunks[2 * blockIdx.x + 1] * 4096, end = min(first + 4096, t.n);
  float s = 0.f;
  for (long i = first + threadIdx.x; i < end; i += 256) s = fmaf(t.x[i], t.x[i], s);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = t.alpha * (part[0] + part[1] + part[2] + part[3]);
}
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
  __shared__ double part[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = (float)(part[0] + part[1] + part[2] + part[3]);
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b, long total,
                                                  float* __restrict__ y) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) y[i] = a[i] + b[i];
}

__global__ __launch_bounds__(256) void scale_kernel(const float* __restrict__ x, long total, float alpha,
                                                    const float* __restrict__ alpha_dev, float* __restrict__ y) {
  const float a = alpha * (alpha_dev ? *alpha_dev : 1.f);
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) y[i] = a * x[i];
}

// y = act(x*scale + shift) written as fp32 (y, optional) AND as the three bf16 pieces of every value in the chunk-major
// layout conv_patch.hip consumes ([plane][C/16][pixels][16]): the BatchNormalization in front of a patch convolution
// produces the convolution's operand in its final form, once.  C % 16 == 0; one thread per (pixel, channel quad).
// F16: the two-piece fp16 format (gemm_engine.h).  Where the scale comes from (SRC):
//   0: `bound` [c] — the per-channel output bounds bn_finalize_kernel left (channel_bound): every workgroup folds them and derives
//      s itself; workgroup 0 writes (s, 1 / s) into the planes' slot and the bound's bit pattern into range_out (the range slot of
//      the fp32 y for the gather convs that read it);
//   1: the slot, filled by planes_scale_kernel from a dry run (SRC 2) — no statistics partials exist (inference mode: moving
//      statistics), so the exact maximum is taken by one more read of x;
//   2: the dry run: nothing stored but the workgroup's max |y| (floats 2 + blockIdx.x behind the slot).
template <bool F16, int SRC = 0>
__global__ __launch_bounds__(256) void affine_act_planes_kernel(const float* __restrict__ x, long pixels, int c4,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                int act, float* __restrict__ y, unsigned short* __restrict__ planes,
                                                                const float* __restrict__ bound = nullptr, uint32_t* __restrict__ range_out = nullptr) {
  const long total4 = pixels * c4, plane = total4 * 4, stride = (long)gridDim.x * 256;
  float ps = 1.f;
  if (F16 && SRC == 0) {
    const float b = tensor_bound(bound, 4 * c4);
    const float2 sp = scale_pair(scale_exponent_of(b));
    ps = sp.x;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      float* sl = planes_scale_slot(planes, plane); sl[0] = sp.x; sl[1] = sp.y;
      if (range_out) *range_out = __float_as_uint(b);
    }
  }
  if (F16 && SRC == 1) ps = planes_scale_slot(planes, plane)[0];
  const bool fixed = stride % c4 == 0;
  float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = sc;
  if (fixed) {
    const int q = (int)(((long)blockIdx.x * 256 + threadIdx.x) % c4);
    sc = reinterpret_cast<const float4*>(scale)[q]; sh = reinterpret_cast<const float4*>(shift)[q];
  }
  float amax = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
    const long pix = i / c4; const int q = (int)(i - pix * c4);
    if (!fixed) { sc = reinterpret_cast<const float4*>(scale)[q]; sh = reinterpret_cast<const float4*>(shift)[q]; }
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    float4 o = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
    if (act) { o.x = act_apply(act, o.x); o.y = act_apply(act, o.y); o.z = act_apply(act, o.z); o.w = act_apply(act, o.w); }
    if (SRC == 2) { amax = amax4(amax, o); continue; }
    if (y) reinterpret_cast<float4*>(y)[i] = o;
    const long e = ((long)(q >> 2) * pixels + pix) * 16 + 4 * (q & 3);
    if (F16) {
      const Split4H s = split4h(o, ps);
#pragma unroll
      for (int k = 0; k < 2; ++k) *reinterpret_cast<uint2*>(planes + k * plane + e) = s.p[k];
    } else {
      const Split4 s = split4(o);
#pragma unroll
      for (int k = 0; k < 3; ++k) *reinterpret_cast<uint2*>(planes + k * plane + e) = s.p[k];
    }
  }
  if (SRC == 2) {
    amax = wave_max(amax);
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) planes_scale_slot(planes, plane)[2 + blockIdx.x] = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
  }
}

// inverted dropout with a counter-based mask: y = x * keep / (1-rate); the same (seed, index) gives
// the same mask in backward.
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, long total, float rate,
                                                      uint64_t seed, const uint64_t* __restrict__ seed_add, float* __restrict__ y) {
  if (seed_add) seed += *seed_add;                         // graph replays: the step count since capture, from device memory
  const float keep_scale = 1.f / (1.f - rate);
  const uint32_t thr = (uint32_t)((double)rate * 4294967296.0);
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride)
    y[i] = rng_u32(seed, (uint64_t)i, 1) >= thr ? x[i] * keep_scale : 0.f;
}

// y[p, 0:cin] = x[p, :], y[p, cin:cout] = 0 — widens a 3-channel image batch to 4 channels so the stem
// convolution gathers 16 bytes per pixel (the zero channel adds no MACs that matter: 196 vs 147 taps*ch).
__global__ __launch_bounds__(256) void pad_channels_kernel(const float* __restrict__ x, long pixels, int cin, int cout,
                                                           float* __restrict__ y, uint32_t* __restrict__ range_slot) {
  const long total = pixels * cout;
  const long stride = (long)gridDim.x * 256;
  float amax = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const long p = i / cout;
    const int c = (int)(i - p * cout);
    const float v = c < cin ? x[p * cin + c] : 0.f;
    y[i] = v;
    amax = fmaxf(amax, fabsf(v));
  }
  if (range_slot) range_emit_block(range_slot + 1 + blockIdx.x % RANGE_PARTIALS, amax);     // the image's exact max |x| (conv.hip Ranges)
}

// sum of squares -> partial per block (double finalize on one thread)
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, long total,
                                                            float* __restrict__ partial) {
  __shared__ float part[4];
  float s = 0.f;
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) s = fmaf(x[i], x[i], s);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// out[c] = sum_{t,k} w[t][c][k] * s[t][k]   (t = filter tap): gradient of a per-channel input offset
// through a convolution, from the per-tap sums s of the output gradient.  One workgroup per channel.
__global__ __launch_bounds__(256) void tap_contract_kernel(const float* __restrict__ w, const float* __restrict__ s,
                                                           int taps, int c, int k, float* __restrict__ out) {
  __shared__ float part[4];
  const int ch = blockIdx.x;
  float acc = 0.f;
  for (int i = threadIdx.x; i < taps * k; i += 256) {
    const int t = i / k, kk = i - t * k;
    acc = fmaf(w[((long)t * c + ch) * k + kk], s[i], acc);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[ch] = part[0] + part[1] + part[2] + part[3];
}

// Per-tap sums of a convolution's output gradient when that gradient sums to ZERO over all pixels of every channel
// (it is the data gradient of a training-mode BatchNormalization: sum dx = 0 identically):
//   taps[r][s][k] = sum over pixels whose tap (r,s) lies INSIDE the image = - sum over pixels whose tap lies in the padding.
// Those pixels are border strips: with rows_out(r) = output rows whose tap row r is in the padding and cols_out(s)
// likewise, the sum is  sum_{y in rows_out(r)} R[y] + sum_{x in cols_out(s)} C[x] - sum_{y,x in both} Q[y][x]  with the
// line sums R[y] = sum_{n,x} dy, C[x] = sum_{n,y} dy, Q[y][x] = sum_n dy.  For the 7x7/2 ResNet stem that is 4 rows,
// 3 columns and 12 corners of a 112x112 map: 6 % of the tensor is read.  "Lines" are numbered rows, then columns,
// then corners (row-major); border rows are [0,yt) and [yb,oh), border columns [0,xl) and [xr,ow).
struct TapBorder { int oh, ow, k, r, s, stride, pad_t, pad_l, h, w, yt, yb, xl, xr; };
__device__ __forceinline__ int tb_nrow(const TapBorder& g) { return g.yt + (g.oh - g.yb); }
__device__ __forceinline__ int tb_ncol(const TapBorder& g) { return g.xl + (g.ow - g.xr); }
__device__ __forceinline__ int tb_row(const TapBorder& g, int i) { return i < g.yt ? i : g.yb + (i - g.yt); }
__device__ __forceinline__ int tb_col(const TapBorder& g, int i) { return i < g.xl ? i : g.xr + (i - g.xl); }

// partial[line][image][k]: one workgroup per (line, image); 64 channels x 4 pixel lanes
__global__ __launch_bounds__(256) void tap_border_lines_kernel(const float* __restrict__ dy, TapBorder g,
                                                               float* __restrict__ partial) {
  __shared__ float red[4][64];
  const int line = blockIdx.x, n = blockIdx.y, nimg = gridDim.y;
  const int nrow = tb_nrow(g), ncol = tb_ncol(g);
  int y0, x0, count, step;                                 // pixels (y0, x0) + j * step, j < count (step in pixels)
  if (line < nrow) { y0 = tb_row(g, line); x0 = 0; count = g.ow; step = 1; }
  else if (line < nrow + ncol) { y0 = 0; x0 = tb_col(g, line - nrow); count = g.oh; step = g.ow; }
  else { const int q = line - nrow - ncol; y0 = tb_row(g, q / ncol); x0 = tb_col(g, q % ncol); count = 1; step = 1; }
  const float* base = dy + (((long)n * g.oh + y0) * g.ow + x0) * g.k;
  const int lk = threadIdx.x & 63, pl = threadIdx.x >> 6;
  for (int k0 = 0; k0 < g.k; k0 += 64) {
    float acc = 0.f;
    if (k0 + lk < g.k)
      for (int j = pl; j < count; j += 4) acc += base[(long)j * step * g.k + k0 + lk];
    red[pl][lk] = acc;
    __syncthreads();
    if (pl == 0 && k0 + lk < g.k)
      partial[((long)line * nimg + n) * g.k + k0 + lk] = (red[0][lk] + red[1][lk]) + (red[2][lk] + red[3][lk]);
    __syncthreads();
  }
}

// L[line][k] = sum over images, in double, fixed order
__global__ __launch_bounds__(256) void tap_border_reduce_kernel(const float* __restrict__ partial, int nimg, int k,
                                                                float* __restrict__ lines) {
  __shared__ double red[4][64];
  const int line = blockIdx.x, lk = threadIdx.x & 63, pl = threadIdx.x >> 6;
  for (int k0 = 0; k0 < k; k0 += 64) {
    double acc = 0.0;
    if (k0 + lk < k)
      for (int n = pl; n < nimg; n += 4) acc += (double)partial[((long)line * nimg + n) * k + k0 + lk];
    red[pl][lk] = acc;
    __syncthreads();
    if (pl == 0 && k0 + lk < k) lines[(long)line * k + k0 + lk] = (float)((red[0][lk] + red[1][lk]) + (red[2][lk] + red[3][lk]));
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void tap_border_combine_kernel(const float* __restrict__ lines, TapBorder g,
                                                                 float* __restrict__ taps) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.r * g.s * g.k) return;
  const int kk = i % g.k, t = i / g.k, rr = t / g.s, ss = t - rr * g.s;
  const int nrow = tb_nrow(g), ncol = tb_ncol(g);
  float acc = 0.f;
  for (int a = 0; a < nrow; ++a) {
    const bool rout = (unsigned)(tb_row(g, a) * g.stride + rr - g.pad_t) >= (unsigned)g.h;
    if (rout) acc += lines[(long)a * g.k + kk];
    for (int b = 0; b < ncol; ++b) {
      const bool cout_ = (unsigned)(tb_col(g, b) * g.stride + ss - g.pad_l) >= (unsigned)g.w;
      if (a == 0 && cout_) acc += lines[(long)(nrow + b) * g.k + kk];
      if (rout && cout_) acc -= lines[(long)(nrow + ncol + a * ncol + b) * g.k + kk];
    }
  }
  if (nrow == 0)
    for (int b = 0; b < ncol; ++b)
      if ((unsigned)(tb_col(g, b) * g.stride + ss - g.pad_l) >= (unsigned)g.w) acc += lines[(long)b * g.k + kk];
  taps[i] = -acc;
}

__global__ void sum_finalize_kernel(const float* __restrict__ partial, int n, float alpha, float* __restrict__ out) {
  if (threadIdx.x || blockIdx.x) return;
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += partial[i];
  *out = (float)(alpha * s);
}

// fold_slot: a range slot whose workgroup partials the pass in front of this kernel filled — column 0's workgroup folds them into
// the slot's first word on the way (range_fold_kernel's job, without its launch)
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ partial, int blocks, int c,
                                                              float* __restrict__ out, uint32_t* __restrict__ fold_slot = nullptr) {
  const int col = blockIdx.x;
  double s, unused;
  block_partial_sums(partial, blocks, c, col, s, unused);
  if (threadIdx.x == 0) out[col] = (float)s;
  if (fold_slot && col == 0) {
    uint32_t m = 0u;
    for (int i = threadIdx.x; i < RANGE_PARTIALS; i += 256) m = max(m, fold_slot[1 + i]);
    range_emit_block(fold_slot, __uint_as_float(m));
  }
}

}  // namespace embnet

using namespace embnet;
#define S(stream) ((hipStream_t)(stream))
static inline int ew_blocks(long total) {
  static const long cap = env_long("EMBNET_EW_BLOCKS", 4096);      // grid cap of the grid-stride elementwise kernels (sweep: DESIGN 3.12)
  long b = (total + 255) / 256; return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}
// the same, rounded to a block count whose grid stride (blocks * 256 quads) is a multiple of c4: every thread of a
// channel-quad kernel then keeps ONE channel quad and loads its per-channel constants once
static inline int ew_blocks_c4(long total4, int c4) {
  int b = ew_blocks(total4);
  int gcd = c4, r = 256; while (r) { const int t = gcd % r; gcd = r; r = t; }
  const int unit = c4 / gcd;
  if (unit > 1) b = b < unit ? unit : b / unit * unit;
  return b;
}

extern "C" size_t embnet_bn_workspace_bytes(long m, int c) {
  if (m <= 0 || c <= 0) return 0;
  // per block: the two sums and (backward, four-channel kernels) the max |dz| row; + one row of per-channel dx bounds
  int blocks = col_geom(m, c).blocks;
  if ((c & 3) == 0) { const int b4 = col_geom(m, c / 4).blocks; if (b4 > blocks) blocks = b4; }
  return ((size_t)blocks * 3 * c + c) * sizeof(float);
}

static int bn_train_fwd_impl(const float* x, long m, int c, const float* gamma, const float* beta, float eps,
                             float momentum, int relu, float* y, float* save_mean, float* save_rstd,
                             float* scale, float* shift, float* moving_mean, float* moving_var,
                             const float* partial_in, int partial_rows, void* workspace, size_t workspace_bytes,
                             float* y_bound, uint32_t* y_range, void* stream, float* xhat_bound = nullptr) {
  EMBNET_CHECK_ARG(x && save_mean && save_rstd && scale && shift && workspace, "bn_train_fwd: null pointer");
  EMBNET_CHECK_ARG(m > 0 && c > 0, "bn_train_fwd: m=%ld c=%d", m, c);
  EMBNET_CHECK_ARG(!y_range || (y_bound && y && !(reinterpret_cast<uintptr_t>(y_range) & 3)),
                   "bn_train_fwd: y_range (the range slot of the fp32 y) goes with y_bound and y");
  if (workspace_bytes < embnet_bn_workspace_bytes(m, c))
    return fail(EMBNET_EWORKSPACE, "bn_train_fwd: workspace %zu < %zu", workspace_bytes, embnet_bn_workspace_bytes(m, c));
  const ColGeom g = col_geom(m, c);
  const float* partial = (const float*)workspace;
  int nblocks = g.blocks;
  if (partial_in) {                         // sum / sum-of-squares partials [2][c][rows] already produced (conv epilogue)
    EMBNET_CHECK_ARG(partial_rows > 0, "bn_train_fwd: partial_rows=%d", partial_rows);
    partial = partial_in; nblocks = partial_rows;
  } else if ((c & 3) == 0 && !bn_scalar()) {
    const ColGeom g4 = col_geom(m, c / 4);
    nblocks = g4.blocks;
    { EMBNET_TRACE("embnet::bn_stats4_kernel", TRACE_BYTES, 4.0 * m * c, stream); bn_stats4_kernel<<<g4.blocks, 256, 0, S(stream)>>>(x, m, c / 4, g4, (float*)workspace); }
  } else {
    { EMBNET_TRACE("embnet::bn_stats_kernel", TRACE_BYTES, 4.0 * m * c, stream); bn_stats_kernel<<<g.blocks, 256, 0, S(stream)>>>(x, m, c, g, (float*)workspace); }
  }
  { EMBNET_TRACE("embnet::bn_finalize_kernel", TRACE_BYTES, 8.0 * nblocks * c, stream); bn_finalize_kernel<<<c, 256, 0, S(stream)>>>(partial, nblocks, m, c, gamma, beta, eps, momentum, save_mean,
                                                          save_rstd, scale, shift, moving_mean, moving_var, partial_in != nullptr, y_bound, xhat_bound); }
  if (y)                                    // y == NULL: statistics + scale/shift only (a fused consumer applies them)
    { EMBNET_TRACE("embnet::affine_act_kernel", TRACE_BYTES, 8.0 * m * c, stream); affine_act_kernel<<<((c & 3) ? ew_blocks(m * c / 4 + 1) : ew_blocks_c4(m * c / 4, c / 4)), 256, 0, S(stream)>>>(x, m * c, c, scale, shift, relu, y, DropArg{0, nullptr, 0u, 1.f}, y_bound, y_range); }
  return check_launch("bn_train_fwd");
}
extern "C" int embnet_bn_train_fwd(const float* x, long m, int c, const float* gamma, const float* beta, float eps,
                                   float momentum, int relu, float* y, float* save_mean, float* save_rstd,
                                   float* scale, float* shift, float* moving_mean, float* moving_var,
                                   const float* partial_in, int partial_rows, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  return bn_train_fwd_impl(x, m, c, gamma, beta, eps, momentum, relu, y, save_mean, save_rstd, scale, shift, moving_mean, moving_var,
                           partial_in, partial_rows, workspace, workspace_bytes, nullptr, nullptr, stream);
}
extern "C" int embnet_bn_train_fwd_ex(const float* x, long m, int c, const float* gamma, const float* beta, float eps,
                                      float momentum, int relu, float* y, float* save_mean, float* save_rstd,
                                      float* scale, float* shift, float* moving_mean, float* moving_var,
                                      const float* partial_in, int partial_rows, void* workspace, size_t workspace_bytes,
                                      float* y_bound, uint32_t* y_range, float* xhat_bound, void* stream) {
  return bn_train_fwd_impl(x, m, c, gamma, beta, eps, momentum, relu, y, save_mean, save_rstd, scale, shift, moving_mean, moving_var,
                           partial_in, partial_rows, workspace, workspace_bytes, y_bound, y_range, stream, xhat_bound);
}

extern "C" int embnet_bn_infer_fwd(const float* x, long m, int c, const float* gamma, const float* beta,
                                   const float* moving_mean, const float* moving_var, float eps, int relu, float* y,
                                   float* scale, float* shift, void* stream) {
  EMBNET_CHECK_ARG(x && moving_mean && moving_var && scale && shift, "bn_infer_fwd: null pointer");
  EMBNET_CHECK_ARG(m > 0 && c > 0, "bn_infer_fwd: m=%ld c=%d", m, c);
  bn_infer_prepare_kernel<<<cdiv(c, 256), 256, 0, S(stream)>>>(c, gamma, beta, moving_mean, moving_var, eps, scale, shift);
  if (y) { EMBNET_TRACE("embnet::affine_act_kernel", TRACE_BYTES, 8.0 * m * c, stream); affine_act_kernel<<<((c & 3) ? ew_blocks(m * c / 4 + 1) : ew_blocks_c4(m * c / 4, c / 4)), 256, 0, S(stream)>>>(x, m * c, c, scale, shift, relu, y, DropArg{0, nullptr, 0u, 1.f}); }
  return check_launch("bn_infer_fwd");
}

extern "C" int embnet_affine_act(const float* x, long m, int c, const float* scale, const float* shift, int act, float* y,
                                 void* stream) {
  EMBNET_CHECK_ARG(x && scale && shift && y, "affine_act: null pointer");
  EMBNET_CHECK_ARG(m > 0 && c > 0, "affine_act: m=%ld c=%d", m, c);
  { EMBNET_TRACE("embnet::affine_act_kernel", TRACE_BYTES, 8.0 * m * c, stream); affine_act_kernel<<<((c & 3) ? ew_blocks(m * c / 4 + 1) : ew_blocks_c4(m * c / 4, c / 4)), 256, 0, S(stream)>>>(x, m * c, c, scale, shift, act, y, DropArg{0, nullptr, 0u, 1.f}); }
  return check_launch("affine_act");
}

extern "C" int embnet_affine_act_dropout(const float* x, long m, int c, const float* scale, const float* shift, int act, float rate,
                                         uint64_t seed, const uint64_t* seed_add_dev, float* y, void* stream) {
  EMBNET_CHECK_ARG(x && scale && shift && y, "affine_act_dropout: null pointer");
  EMBNET_CHECK_ARG(m > 0 && c > 0, "affine_act_dropout: m=%ld c=%d", m, c);
  EMBNET_CHECK_ARG(rate >= 0.f && rate < 1.f, "affine_act_dropout: rate %f outside [0,1)", rate);
  { EMBNET_TRACE("embnet::affine_act_kernel", TRACE_BYTES, 8.0 * m * c, stream); affine_act_kernel<<<((c & 3) ? ew_blocks(m * c / 4 + 1) : ew_blocks_c4(m * c / 4, c / 4)), 256, 0, S(stream)>>>(x, m * c, c, scale, shift, act, y, drop_arg(rate, seed, seed_add_dev)); }
  return check_launch("affine_act_dropout");
}

// The range slot of a tensor y with |y| <= factor * max_c bound_c (+ the range of another tensor): the outputs of the fused
// BatchNorm passes that are not plain applies — act(BN(x)) * gate (gate in (0, 1)), skip + drop_factor * BN(x) — for the
// three-product gather convs that read them (EfficientNet's project / expand convs, reference backbones.py:84-98).
__global__ __launch_bounds__(256) void range_from_bound_kernel(const float* __restrict__ bound, int c, float factor,
                                                               const uint32_t* __restrict__ add_range, uint32_t* __restrict__ out) {
  const float b = tensor_bound(bound, c) * factor + (add_range ? __uint_as_float(*add_range) : 0.f);
  if (threadIdx.x == 0) *out = __float_as_uint(b);
}
extern "C" int embnet_range_from_bound(const float* bound, int c, float factor, const uint32_t* add_range, uint32_t* out, void* stream) {
  EMBNET_CHECK_ARG(bound && out && c > 0 && factor >= 0.f && !((reinterpret_cast<uintptr_t>(add_range) | reinterpret_cast<uintptr_t>(out)) & 3),
                   "range_from_bound: bad argument");
  range_from_bound_kernel<<<1, 256, 0, S(stream)>>>(bound, c, factor, add_range, out);
  return check_launch("range_from_bound");
}

static int affine_act_planes_impl(const float* x, long m, int c, const float* scale, const float* shift, int act, float* y,
                                  void* planes, const float* y_bound, uint32_t* y_range, void* stream) {
  EMBNET_CHECK_ARG(x && scale && shift && planes, "affine_act_planes: null pointer");
  EMBNET_CHECK_ARG(m > 0 && c > 0 && (c & 15) == 0, "affine_act_planes: m=%ld c=%d (c %% 16 == 0)", m, c);
  EMBNET_CHECK_ARG((size_t)m * c * 2 < 0x7FFFFFF0ull / 3, "affine_act_planes: tensor too large");
  EMBNET_CHECK_ARG(!y_range || (y && !(reinterpret_cast<uintptr_t>(y_range) & 3)),
                   "affine_act_planes: y_range is the range slot of the fp32 y (4-byte aligned)");
  const long total4 = m * c / 4;
  const int blocks = ew_blocks_c4(total4, c / 4);
  unsigned short* pl = (unsigned short*)planes;
  if (!planes_f16()) {
    EMBNET_TRACE("void embnet::affine_act_planes_kernel<false>", TRACE_BYTES, ((y ? 8.0 : 4.0) + 6.0) * m * c, stream);
    affine_act_planes_kernel<false><<<blocks, 256, 0, S(stream)>>>(x, m, c / 4, scale, shift, act, y, pl);
    if (y_range) {                                          // (the three-piece planes need no scale; the fp32 y's readers want its range)
      EMBNET_CHECK_ARG(y_bound, "affine_act_planes: y_range without y_bound in the three-piece (bf16) planes format");
      range_from_bound_kernel<<<1, 256, 0, S(stream)>>>(y_bound, c, 1.f, nullptr, y_range);
    }
    return check_launch("affine_act_planes");
  }
  if (y_bound) {                                            // the range is known before the pass (bn_finalize_kernel's bounds)
    EMBNET_TRACE("void embnet::affine_act_planes_kernel<true>", TRACE_BYTES, ((y ? 8.0 : 4.0) + 4.0) * m * c, stream);
    affine_act_planes_kernel<true, 0><<<blocks, 256, 0, S(stream)>>>(x, m, c / 4, scale, shift, act, y, pl, y_bound, y_range);
    return check_launch("affine_act_planes");
  }
  // no bound (inference-mode statistics, or a caller without the BatchNorm's partials): the exact maximum from a dry run
  EMBNET_TRACE("void embnet::affine_act_planes_kernel<true, 1>", TRACE_BYTES, ((y ? 8.0 : 4.0) + 8.0) * m * c, stream);
  const int dry = (int)(2 * total4 - 2 < blocks ? 2 * total4 - 2 : blocks);
  affine_act_planes_kernel<true, 2><<<dry, 256, 0, S(stream)>>>(x, m, c / 4, scale, shift, act, nullptr, pl);
  planes_scale_kernel<<<1, 256, 0, S(stream)>>>(planes_scale_slot(planes, total4 * 4), dry, y_range);
  affine_act_planes_kernel<true, 1><<<blocks, 256, 0, S(stream)>>>(x, m, c / 4, scale, shift, act, y, pl);
  return check_launch("affine_act_planes");
}
extern "C" int embnet_affine_act_planes(const float* x, long m, int c, const float* scale, const float* shift, int act, float* y,
                                        void* planes, void* stream) {
  return affine_act_planes_impl(x, m, c, scale, shift, act, y, planes, nullptr, nullptr, stream);
}
extern "C" int embnet_affine_act_planes_ex(const float* x, long m, int c, const float* scale, const float* shift, int act, float* y,
                                           void* planes, const float* y_bound, uint32_t* y_range, void* stream) {
  return affine_act_planes_impl(x, m, c, scale, shift, act, y, planes, y_bound, y_range, stream);
}

// bn_bwd_apply4_kernel in the planes format of the process: with EMBNET_PLANES_F16 a dry run finds the gradient's range first
static void launch_bn_bwd_apply4(const float* dy, const float* x, long m, int c, const float* save_mean, const float* save_rstd,
                                 const float* scale, const float* shift, const float* dbeta, const float* dgamma, int relu,
                                 int training, const float* dx_add, float* dx, void* dx_planes, hipStream_t st,
                                 uint32_t* range_slot = nullptr, const float* dx_bound = nullptr, const uint32_t* add_range = nullptr) {
  const long total4 = m * c / 4;
  const int blocks = ew_blocks_c4(total4, c / 4);
  const float inv_m = 1.f / (float)m;
  unsigned short* pl = (unsigned short*)dx_planes;
  if (dx_planes && planes_f16() && dx_bound) {              // the range is known before the pass (bn_bwd_finalize_kernel's bounds): no dry run
    bn_bwd_apply4_kernel<4><<<blocks, 256, 0, st>>>(dy, x, total4, c / 4, inv_m, save_mean, save_rstd, scale, shift, dbeta, dgamma, relu,
                                                   training, dx_add, dx, pl, range_slot, dx_bound, add_range);
    return;
  }
  if (dx_planes && planes_f16()) {
    float* slot = planes_scale_slot(dx_planes, total4 * 4);
    // (the third plane's space — 2 * total4 floats, >= 8 — holds the slot and the dry run's workgroup maxima: fewer workgroups for a tiny tensor)
    const int dry = (int)(2 * total4 - 2 < blocks ? 2 * total4 - 2 : blocks);
    bn_bwd_apply4_kernel<2><<<dry, 256, 0, st>>>(dy, x, total4, c / 4, inv_m, save_mean, save_rstd, scale, shift, dbeta, dgamma,
                                                relu, training, dx_add, nullptr, pl);
    planes_scale_kernel<<<1, 256, 0, st>>>(slot, dry, range_slot);
    bn_bwd_apply4_kernel<1><<<blocks, 256, 0, st>>>(dy, x, total4, c / 4, inv_m, save_mean, save_rstd, scale, shift, dbeta, dgamma, relu,
                                                   training, dx_add, dx, pl);
    return;
  }
  if (range_slot && dx && !dx_planes) {
    bn_bwd_apply4_kernel<3><<<blocks, 256, 0, st>>>(dy, x, total4, c / 4, inv_m, save_mean, save_rstd, scale, shift, dbeta, dgamma, relu,
                                                   training, dx_add, dx, pl, range_slot);
    range_fold_kernel<<<1, 256, 0, st>>>(range_slot);
    return;
  }
  bn_bwd_apply4_kernel<0><<<blocks, 256, 0, st>>>(dy, x, total4, c / 4, inv_m, save_mean, save_rstd, scale, shift, dbeta, dgamma, relu,
                                                 training, dx_add, dx, pl);
}

static int bn_bwd_impl(const float* dy, const float* x, long m, int c, const float* save_mean,
                       const float* save_rstd, const float* scale, const float* shift, int relu, int training,
                       const float* dx_add, float* dx, float* dgamma, float* dbeta, void* dx_planes, void* workspace,
                       size_t workspace_bytes, uint32_t* emit, void* stream, const float* xhat_bound = nullptr,
                       const uint32_t* dx_add_range = nullptr) {
  EMBNET_CHECK_ARG(!emit || (dx && (!dx_planes || planes_f16()) && (c & 3) == 0 && !bn_scalar() && save_mean && save_rstd &&
                             !(reinterpret_cast<uintptr_t>(emit) & 3)),
                   "bn_bwd: a range of dx was requested but this call cannot emit one (an fp32 dx — beside planes only in the two-piece "
                   "format —, c %% 4 == 0, saved statistics, a 4-byte aligned slot)");
  EMBNET_CHECK_ARG(dy && x && scale && shift && (dx || dx_planes) && dgamma && dbeta && workspace, "bn_bwd: null pointer");
  EMBNET_CHECK_ARG(dx || ((c & 3) == 0 && !bn_scalar()), "bn_bwd: dx = NULL (planes only) needs the four-channel kernels");
  EMBNET_CHECK_ARG(!dx_planes || ((c & 15) == 0 && (size_t)m * c * 2 < 0x7FFFFFF0ull / 3), "bn_bwd: dx_planes needs c %% 16 == 0");
  EMBNET_CHECK_ARG(!training || (save_mean && save_rstd), "bn_bwd: training needs saved statistics");
  EMBNET_CHECK_ARG(m > 0 && c > 0, "bn_bwd: m=%ld c=%d", m, c);
  if (workspace_bytes < embnet_bn_workspace_bytes(m, c))
    return fail(EMBNET_EWORKSPACE, "bn_bwd: workspace %zu < %zu", workspace_bytes, embnet_bn_workspace_bytes(m, c));
  const ColGeom g = col_geom(m, c);
  float* partial = (float*)workspace;
  // inference-mode statistics: xhat uses the moving stats folded in scale/shift; dgamma then needs them too.
  // We only support parameter gradients in training mode; frozen BN returns dgamma = dbeta sums with xhat from
  // save_mean/save_rstd when given, else zeros.
  // planes in the two-piece format need dx's range before the apply pass: from the bound (the reduction's max |dz|, the forward's
  // max |xhat|, the range of what is added) where all of it is at hand, else from a dry run of the pass (launch_bn_bwd_apply4)
  static const bool no_dry = env_long("EMBNET_BN_BWD_BOUND", 1) != 0;          // 0: round 5's dry run (A/B)
  const bool bound = no_dry && dx_planes && planes_f16() && (c & 3) == 0 && !bn_scalar() && save_mean && save_rstd &&
                     (xhat_bound || !training) && (!dx_add || dx_add_range);
  float* dx_bound = nullptr;
  if (save_mean && save_rstd) {
    if ((c & 3) == 0 && !bn_scalar()) {
      const ColGeom g4 = col_geom(m, c / 4);
      float* pmax = bound ? partial + (size_t)g4.blocks * 2 * c : nullptr;
      if (bound) dx_bound = pmax + (size_t)g4.blocks * c;
      { EMBNET_TRACE("embnet::bn_bwd_reduce4_kernel", TRACE_BYTES, 8.0 * m * c, stream); bn_bwd_reduce4_kernel<<<g4.blocks, 256, 0, S(stream)>>>(dy, x, m, c / 4, g4, save_mean, save_rstd, scale, shift, relu, partial, DropArg{0, nullptr, 0u, 1.f}, pmax); }
      bn_bwd_finalize_kernel<<<c, 256, 0, S(stream)>>>(partial, g4.blocks, c, dbeta, dgamma, 0, bound ? nullptr : emit, pmax, scale, xhat_bound,
                                                       1.f / (float)m, training, dx_bound);
    } else {
      { EMBNET_TRACE("embnet::bn_bwd_reduce_kernel", TRACE_BYTES, 8.0 * m * c, stream); bn_bwd_reduce_kernel<<<g.blocks, 256, 0, S(stream)>>>(dy, x, m, c, g, save_mean, save_rstd, scale, shift, relu, partial); }
      bn_bwd_finalize_kernel<<<c, 256, 0, S(stream)>>>(partial, g.blocks, c, dbeta, dgamma);
    }
  } else {
    zero2_kernel<<<cdiv(c, 256), 256, 0, S(stream)>>>(dbeta, dgamma, c);
  }
  if ((c & 3) == 0 && !bn_scalar())
    { EMBNET_TRACE(dx_bound ? "void embnet::bn_bwd_apply4_kernel<4>" : emit && !dx_planes ? "void embnet::bn_bwd_apply4_kernel<3>" : dx_planes && planes_f16() ? "void embnet::bn_bwd_apply4_kernel<1>" : "void embnet::bn_bwd_apply4_kernel<0>", TRACE_BYTES, ((dx_add ? 16.0 : 12.0) + (dx_planes && planes_f16() && !dx_bound ? 8.0 : 0.0)) * m * c, stream);   // (<1>: + the dry run <2>; the names rocprofv3 prints)
      launch_bn_bwd_apply4(dy, x, m, c, save_mean, save_rstd, scale, shift, dbeta, dgamma, relu, training, dx_add, dx, dx_planes, S(stream), emit, dx_bound, dx_add_range); }
  else
    { EMBNET_TRACE("embnet::bn_bwd_apply_kernel", TRACE_BYTES, (dx_add ? 16.0 : 12.0) * m * c, stream); bn_bwd_apply_kernel<<<ew_blocks(m * c), 256, 0, S(stream)>>>(dy, x, m * c, c, 1.f / (float)m, save_mean, save_rstd,
                                                                 scale, shift, dbeta, dgamma, relu, training, dx_add, dx); }
  return check_launch("bn_bwd");
}
extern "C" int embnet_bn_bwd(const float* dy, const float* x, long m, int c, const float* save_mean,
                             const float* save_rstd, const float* scale, const float* shift, int relu, int training,
                             const float* dx_add, float* dx, float* dgamma, float* dbeta, void* dx_planes, void* workspace,
                             size_t workspace_bytes, void* stream) {
  return bn_bwd_impl(dy, x, m, c, save_mean, save_rstd, scale, shift, relu, training, dx_add, dx, dgamma, dbeta, dx_planes, workspace,
                     workspace_bytes, take_emit_slot(), stream);
}
extern "C" int embnet_bn_bwd_ex(const float* dy, const float* x, long m, int c, const float* save_mean,
                                const float* save_rstd, const float* scale, const float* shift, int relu, int training,
                                const float* dx_add, float* dx, float* dgamma, float* dbeta, void* dx_planes, void* workspace,
                                size_t workspace_bytes, uint32_t* dx_range, const float* xhat_bound, const uint32_t* dx_add_range,
                                void* stream) {
  (void)take_emit_slot();
  return bn_bwd_impl(dy, x, m, c, save_mean, save_rstd, scale, shift, relu, training, dx_add, dx, dgamma, dbeta, dx_planes, workspace,
                     workspace_bytes, dx_range, stream, xhat_bound, dx_add_range);
}

// A range slot (embnet_range_slot_words() words) filled by an elementwise pass: zeroed in front, the workgroup partials folded into word 0
// behind (the BatchNorm passes zero theirs in their finalize kernels).
static void range_begin(uint32_t* slot, void* stream) {      // (a kernel, not hipMemsetAsync: these passes run inside captured HIP graphs)
  if (slot) range_zero_kernel<<<1, 256, 0, S(stream)>>>(slot);
}
static void range_end(uint32_t* slot, void* stream) {
  if (slot) range_fold_kernel<<<1, 256, 0, S(stream)>>>(slot);
}
#define EMBNET_RANGE_ARG(slot, what) \
  EMBNET_CHECK_ARG(!(reinterpret_cast<uintptr_t>(slot) & 3), what ": the range slot is 4-byte aligned")

static int bn_bwd_inrelu_impl(const float* dy, const float* x, long m, int c, const float* save_mean,
                              const float* save_rstd, const float* scale, const float* shift, int relu, int training,
                              float* dz, float* dgamma, float* dbeta, float* dbias, void* workspace,
                              size_t workspace_bytes, void* stream, const DropArg drop, uint32_t* dz_range = nullptr) {
  EMBNET_CHECK_ARG(!take_emit_slot(), "bn_bwd_inrelu: the deprecated embnet_range_emit request is not served here: pass dz_range to the _ex form");
  EMBNET_RANGE_ARG(dz_range, "bn_bwd_inrelu");
  EMBNET_CHECK_ARG(dy && x && scale && shift && dz && dgamma && dbeta && dbias && workspace, "bn_bwd_inrelu: null pointer");
  EMBNET_CHECK_ARG(m > 0 && c > 0 && (c & 3) == 0, "bn_bwd_inrelu: m=%ld c=%d (c %% 4 == 0 required)", m, c);
  EMBNET_CHECK_ARG(!training || (save_mean && save_rstd), "bn_bwd_inrelu: training needs saved statistics");
  if (workspace_bytes < embnet_bn_workspace_bytes(m, c))
    return fail(EMBNET_EWORKSPACE, "bn_bwd_inrelu: workspace %zu < %zu", workspace_bytes, embnet_bn_workspace_bytes(m, c));
  float* partial = (float*)workspace;
  const ColGeom g4 = col_geom(m, c / 4);
  if (save_mean && save_rstd) {
    { EMBNET_TRACE("embnet::bn_bwd_reduce4_kernel", TRACE_BYTES, 8.0 * m * c, stream); bn_bwd_reduce4_kernel<<<g4.blocks, 256, 0, S(stream)>>>(dy, x, m, c / 4, g4, save_mean, save_rstd, scale, shift, relu, partial, drop, nullptr); }
    bn_bwd_finalize_kernel<<<c, 256, 0, S(stream)>>>(partial, g4.blocks, c, dbeta, dgamma, 0, dz_range);     // (zeroes the range slot)
  } else {
    zero2_kernel<<<cdiv(c, 256), 256, 0, S(stream)>>>(dbeta, dgamma, c);
    range_begin(dz_range, stream);
  }
  { EMBNET_TRACE("embnet::bn_bwd_apply_inrelu4_kernel", TRACE_BYTES, 12.0 * m * c, stream);
    bn_bwd_apply_inrelu4_kernel<<<g4.blocks, 256, 0, S(stream)>>>(dy, x, m, c / 4, g4, 1.f / (float)m, save_mean, save_rstd, scale, shift,
                                                               dbeta, dgamma, relu, training, dz, partial, drop, dz_range); }
  colsum_finalize_kernel<<<c, 256, 0, S(stream)>>>(partial, g4.blocks, c, dbias, dz_range);                     // (and folds it)
  return check_launch("bn_bwd_inrelu");
}

// BatchNorm backward on dy + dpool / hw (see bn_bwd_reduce4_gap_kernel): training statistics, c % 4 == 0, n * hw * c / 4 < 2^31
extern "C" int embnet_bn_bwd_gap(const float* dy, const float* dpool, const float* gate, int n, int hw, const float* x, int c, const float* save_mean,
                                 const float* save_rstd, const float* scale, const float* shift, int relu, float* dx,
                                 float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream) {
  EMBNET_CHECK_ARG(!take_emit_slot(), "bn_bwd_gap: this pass cannot emit the range requested by embnet_range_emit");
  EMBNET_CHECK_ARG(dy && dpool && x && save_mean && save_rstd && scale && shift && dx && dgamma && dbeta && workspace, "bn_bwd_gap: null pointer");
  EMBNET_CHECK_ARG(n > 0 && hw > 0 && c > 0 && (c & 3) == 0, "bn_bwd_gap: n=%d hw=%d c=%d (c %% 4 == 0)", n, hw, c);
  const long m = (long)n * hw;
  EMBNET_CHECK_ARG(m * (c / 4) < 0x7FFFFFFFl, "bn_bwd_gap: tensor too large");
  if (workspace_bytes < embnet_bn_workspace_bytes(m, c))
    return fail(EMBNET_EWORKSPACE, "bn_bwd_gap: workspace %zu < %zu", workspace_bytes, embnet_bn_workspace_bytes(m, c));
  float* partial = (float*)workspace;
  const ColGeom g4 = col_geom(m, c / 4);
  const float inv_hw = 1.f / (float)hw;
  { EMBNET_TRACE("embnet::bn_bwd_reduce4_gap_kernel", TRACE_BYTES, 8.0 * m * c, stream);
    bn_bwd_reduce4_gap_kernel<<<g4.blocks, 256, 0, S(stream)>>>(dy, dpool, gate, make_divu((uint32_t)hw), inv_hw, x, m, c / 4, g4, save_mean, save_rstd,
                                                                scale, shift, relu, partial); }
  bn_bwd_finalize_kernel<<<c, 256, 0, S(stream)>>>(partial, g4.blocks, c, dbeta, dgamma);
  { EMBNET_TRACE("embnet::bn_bwd_apply4_gap_kernel", TRACE_BYTES, 12.0 * m * c, stream);
    bn_bwd_apply4_gap_kernel<<<ew_blocks_c4(m * c / 4, c / 4), 256, 0, S(stream)>>>(dy, dpool, gate, make_divu((uint32_t)((long)hw * (c / 4))), inv_hw, x,
                                                                                    m * c / 4, c / 4, 1.f / (float)m, save_mean, save_rstd,
                                                                                    scale, shift, dbeta, dgamma, relu, dx); }
  return check_launch("bn_bwd_gap");
}

// squeeze-and-excite backward, pass 1 (se_bn_sums4_kernel): sums [n][5][c]; row 0 of every image = the gate's gradient
extern "C" int embnet_se_bn_sums(const float* dg, const float* x, int n, int hw, int c, const float* save_mean, const float* save_rstd,
                                 const float* scale, const float* shift, int act, float* sums, void* stream) {
  EMBNET_CHECK_ARG(dg && x && save_mean && save_rstd && scale && shift && sums, "se_bn_sums: null pointer");
  EMBNET_CHECK_ARG(n > 0 && hw > 0 && c > 0 && (c & 3) == 0 && act >= 0 && act <= 2, "se_bn_sums: n=%d hw=%d c=%d act=%d (c %% 4 == 0)", n, hw, c, act);
  EMBNET_TRACE("embnet::se_bn_sums4_kernel", TRACE_BYTES, 8.0 * n * hw * c, stream);
  const PixGeom g = pix_geom(n, hw, c / 4);
  se_bn_sums4_kernel<<<dim3(g.xblocks, n), g.threads, 0, S(stream)>>>(dg, x, hw, c / 4, g.cls_log2, save_mean, save_rstd, scale, shift, act, sums);
  return check_launch("se_bn_sums");
}

// BatchNorm backward on dy * gate + dpool / hw with the column sums taken from embnet_se_bn_sums (no reduction pass)
extern "C" int embnet_bn_bwd_gap_sums(const float* dy, const float* dpool, const float* gate, const float* sums, int n, int hw,
                                      const float* x, int c, const float* save_mean, const float* save_rstd, const float* scale,
                                      const float* shift, int relu, float* dx, float* dgamma, float* dbeta, void* stream) {
  EMBNET_CHECK_ARG(!take_emit_slot(), "bn_bwd_gap_sums: this pass cannot emit the range requested by embnet_range_emit");
  EMBNET_CHECK_ARG(dy && dpool && gate && sums && x && save_mean && save_rstd && scale && shift && dx && dgamma && dbeta, "bn_bwd_gap_sums: null pointer");
  EMBNET_CHECK_ARG(n > 0 && hw > 0 && c > 0 && (c & 3) == 0, "bn_bwd_gap_sums: n=%d hw=%d c=%d (c %% 4 == 0)", n, hw, c);
  const long m = (long)n * hw;
  EMBNET_CHECK_ARG(m * (c / 4) < 0x7FFFFFFFl, "bn_bwd_gap_sums: tensor too large");
  const float inv_hw = 1.f / (float)hw;
  se_bn_finalize_kernel<<<(c + 3) / 4, 256, 0, S(stream)>>>(sums, gate, dpool, n, c, inv_hw, dbeta, dgamma);
  { EMBNET_TRACE("embnet::bn_bwd_apply4_gap_kernel", TRACE_BYTES, 12.0 * m * c, stream);
    bn_bwd_apply4_gap_kernel<<<ew_blocks_c4(m * c / 4, c / 4), 256, 0, S(stream)>>>(dy, dpool, gate, make_divu((uint32_t)((long)hw * (c / 4))), inv_hw, x,
                                                                                    m * c / 4, c / 4, 1.f / (float)m, save_mean, save_rstd,
                                                                                    scale, shift, dbeta, dgamma, relu, dx); }
  return check_launch("bn_bwd_gap_sums");
}

// BatchNorm backward whose sums were produced by the data gradient of the conv behind it (embnet_conv2d_dgrad_bnsums_f32):
// finalize over the [2][c][rows] partials, then the apply pass of embnet_bn_bwd.  c % 4 == 0, training statistics.
static int bn_bwd_partials_impl(const float* dy, const float* x, long m, int c, const float* save_mean,
                                const float* save_rstd, const float* scale, const float* shift, int relu,
                                const float* partials, int rows, const float* dx_add, float* dx, float* dgamma,
                                float* dbeta, void* dx_planes, uint32_t* emit, void* stream, int partial_kinds = 2,
                                const float* xhat_bound = nullptr, const uint32_t* dx_add_range = nullptr) {
  EMBNET_CHECK_ARG(!emit || (dx && (!dx_planes || planes_f16()) && !(reinterpret_cast<uintptr_t>(emit) & 3)),
                   "bn_bwd_partials: a range of dx was requested but this call cannot emit one (see embnet_bn_bwd_ex)");
  EMBNET_CHECK_ARG(dy && x && save_mean && save_rstd && scale && shift && partials && (dx || dx_planes) && dgamma && dbeta, "bn_bwd_partials: null pointer");
  EMBNET_CHECK_ARG(m > 0 && c > 0 && (c & 3) == 0 && rows > 0, "bn_bwd_partials: m=%ld c=%d rows=%d (c %% 4 == 0)", m, c, rows);
  EMBNET_CHECK_ARG(!dx_planes || ((c & 15) == 0 && (size_t)m * c * 2 < 0x7FFFFFF0ull / 3), "bn_bwd_partials: dx_planes needs c %% 16 == 0");
  // the planes' scale from the bound where the partials carry the max |dz| plane (the per-channel bounds live behind the planes' slot)
  static const bool no_dry = env_long("EMBNET_BN_BWD_BOUND", 1) != 0;
  const long total4 = m * c / 4;
  const bool bound = no_dry && dx_planes && planes_f16() && partial_kinds == 3 && xhat_bound && (!dx_add || dx_add_range) && 2 * total4 >= 2 + c;
  float* dx_bound = bound ? planes_scale_slot(dx_planes, total4 * 4) + 2 : nullptr;
  bn_bwd_finalize_kernel<<<c, 256, 0, S(stream)>>>(partials, rows, c, dbeta, dgamma, 1, bound ? nullptr : emit,
                                                   bound ? partials + (size_t)2 * c * rows : nullptr, scale, xhat_bound, 1.f / (float)m, 1, dx_bound);
  { EMBNET_TRACE(dx_bound ? "void embnet::bn_bwd_apply4_kernel<4>" : emit && !dx_planes ? "void embnet::bn_bwd_apply4_kernel<3>" : dx_planes && planes_f16() ? "void embnet::bn_bwd_apply4_kernel<1>" : "void embnet::bn_bwd_apply4_kernel<0>", TRACE_BYTES, ((dx_add ? 16.0 : 12.0) + (dx_planes && planes_f16() && !dx_bound ? 8.0 : 0.0)) * m * c, stream);   // (<1>: + the dry run <2>; the names rocprofv3 prints)
    launch_bn_bwd_apply4(dy, x, m, c, save_mean, save_rstd, scale, shift, dbeta, dgamma, relu, 1, dx_add, dx, dx_planes, S(stream), emit, dx_bound, dx_add_range); }
  return check_launch("bn_bwd_partials");
}
extern "C" int embnet_bn_bwd_partials(const float* dy, const float* x, long m, int c, const float* save_mean,
                                      const float* save_rstd, const float* scale, const float* shift, int relu,
                                      const float* partials, int rows, const float* dx_add, float* dx, float* dgamma,
                                      float* dbeta, void* dx_planes, void* stream) {
  return bn_bwd_partials_impl(dy, x, m, c, save_mean, save_rstd, scale, shift, relu, partials, rows, dx_add, dx, dgamma, dbeta, dx_planes,
                              take_emit_slot(), stream);
}
extern "C" int embnet_bn_bwd_partials_ex(const float* dy, const float* x, long m, int c, const float* save_mean,
                                         const float* save_rstd, const float* scale, const float* shift, int relu,
                                         const float* partials, int rows, const float* dx_add, float* dx, float* dgamma,
                                         float* dbeta, void* dx_planes, uint32_t* dx_range, int partial_kinds, const float* xhat_bound,
                                         const uint32_t* dx_add_range, void* stream) {
  (void)take_emit_slot();
  EMBNET_CHECK_ARG(partial_kinds == 2 || partial_kinds == 3, "bn_bwd_partials: partial_kinds %d (2: the sums; 3: + max |dz|)", partial_kinds);
  return bn_bwd_partials_impl(dy, x, m, c, save_mean, save_rstd, scale, shift, relu, partials, rows, dx_add, dx, dgamma, dbeta, dx_planes,
                              dx_range, stream, partial_kinds, xhat_bound, dx_add_range);
}

extern "C" int embnet_bn_bwd_inrelu(const float* dy, const float* x, long m, int c, const float* save_mean,
                                    const float* save_rstd, const float* scale, const float* shift, int relu, int training,
                                    float* dz, float* dgamma, float* dbeta, float* dbias, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  return bn_bwd_inrelu_impl(dy, x, m, c, save_mean, save_rstd, scale, shift, relu, training, dz, dgamma, dbeta, dbias, workspace,
                            workspace_bytes, stream, DropArg{0, nullptr, 0u, 1.f});
}

extern "C" int embnet_bn_bwd_inrelu_dropout(const float* dy, const float* x, long m, int c, const float* save_mean,
                                            const float* save_rstd, const float* scale, const float* shift, int relu,
                                            int training, float rate, uint64_t seed, const uint64_t* seed_add_dev, float* dz,
                                            float* dgamma, float* dbeta, float* dbias, void* workspace, size_t workspace_bytes,
                                            void* stream) {
  EMBNET_CHECK_ARG(rate >= 0.f && rate < 1.f, "bn_bwd_inrelu_dropout: rate %f outside [0,1)", rate);
  return bn_bwd_inrelu_impl(dy, x, m, c, save_mean, save_rstd, scale, shift, relu, training, dz, dgamma, dbeta, dbias, workspace,
                            workspace_bytes, stream, drop_arg(rate, seed, seed_add_dev));
}
// ABI 22: the same two with `dz_range` (NULL or a RANGE SLOT): the exact max |dz| is left in its first word
extern "C" int embnet_bn_bwd_inrelu_ex(const float* dy, const float* x, long m, int c, const float* save_mean,
                                       const float* save_rstd, const float* scale, const float* shift, int relu, int training,
                                       float* dz, float* dgamma, float* dbeta, float* dbias, void* workspace,
                                       size_t workspace_bytes, uint32_t* dz_range, void* stream) {
  return bn_bwd_inrelu_impl(dy, x, m, c, save_mean, save_rstd, scale, shift, relu, training, dz, dgamma, dbeta, dbias, workspace,
                            workspace_bytes, stream, DropArg{0, nullptr, 0u, 1.f}, dz_range);
}
extern "C" int embnet_bn_bwd_inrelu_dropout_ex(const float* dy, const float* x, long m, int c, const float* save_mean,
                                               const float* save_rstd, const float* scale, const float* shift, int relu,
                                               int training, float rate, uint64_t seed, const uint64_t* seed_add_dev, float* dz,
                                               float* dgamma, float* dbeta, float* dbias, void* workspace, size_t workspace_bytes,
                                               uint32_t* dz_range, void* stream) {
  EMBNET_CHECK_ARG(rate >= 0.f && rate < 1.f, "bn_bwd_inrelu_dropout: rate %f outside [0,1)", rate);
  return bn_bwd_inrelu_impl(dy, x, m, c, save_mean, save_rstd, scale, shift, relu, training, dz, dgamma, dbeta, dbias, workspace,
                            workspace_bytes, stream, drop_arg(rate, seed, seed_add_dev), dz_range);
}

extern "C" int embnet_maxpool_fwd_ex(const float* x, int n, int h, int w, int c, int k, int stride, int pad, int oh,
                                     int ow, float* y, uint8_t* argmax, uint32_t* y_range, void* stream) {
  EMBNET_CHECK_ARG(x && y && argmax, "maxpool_fwd: null pointer");
  EMBNET_RANGE_ARG(y_range, "maxpool_fwd");
  EMBNET_CHECK_ARG(!y_range || (c & 3) == 0, "maxpool_fwd: the range of y needs c %% 4 == 0 (c = %d)", c);
  EMBNET_CHECK_ARG(n > 0 && h > 0 && w > 0 && c > 0 && k > 0 && k <= 15 && stride > 0 && pad >= 0 && oh > 0 && ow > 0,
                   "maxpool_fwd: bad geometry");
  EMBNET_CHECK_ARG((oh - 1) * stride - pad + k <= h + pad && (ow - 1) * stride - pad + k <= w + pad,
                   "maxpool_fwd: window leaves the padded image");
  const long total = (long)n * oh * ow * c;
  range_begin(y_range, stream);
  if ((c & 3) == 0)
    { EMBNET_TRACE("embnet::maxpool_fwd4_kernel", TRACE_BYTES, 4.0 * n * h * w * c + 5.0 * total, stream); maxpool_fwd4_kernel<<<cdiv(total / 4, 256), 256, 0, S(stream)>>>(x, n, h, w, c / 4, k, stride, pad, oh, ow, y, argmax, y_range); }
  else
    { EMBNET_TRACE("embnet::maxpool_fwd_kernel", TRACE_BYTES, 4.0 * n * h * w * c + 5.0 * total, stream); maxpool_fwd_kernel<<<cdiv(total, 256), 256, 0, S(stream)>>>(x, n, h, w, c, k, stride, pad, oh, ow, y, argmax); }
  range_end(y_range, stream);
  return check_launch("maxpool_fwd");
}
extern "C" int embnet_maxpool_fwd(const float* x, int n, int h, int w, int c, int k, int stride, int pad, int oh,
                                  int ow, float* y, uint8_t* argmax, void* stream) {
  return embnet_maxpool_fwd_ex(x, n, h, w, c, k, stride, pad, oh, ow, y, argmax, nullptr, stream);
}

extern "C" int embnet_maxpool_bwd(const float* dy, const uint8_t* argmax, int n, int h, int w, int c, int k,
                                  int stride, int pad, int oh, int ow, float* dx, void* stream) {
  EMBNET_CHECK_ARG(dy && argmax && dx, "maxpool_bwd: null pointer");
  EMBNET_CHECK_ARG(n > 0 && h > 0 && w > 0 && c > 0 && k > 0 && stride > 0 && oh > 0 && ow > 0, "maxpool_bwd: bad geometry");
  const long total = (long)n * h * w * c;
  if ((c & 3) == 0)
    { EMBNET_TRACE("embnet::maxpool_bwd4_kernel", TRACE_BYTES, 4.0 * total + 5.0 * n * oh * ow * c, stream); maxpool_bwd4_kernel<<<cdiv(total / 4, 256), 256, 0, S(stream)>>>(dy, argmax, n, h, w, c / 4, k, stride, pad, oh, ow, dx); }
  else
    { EMBNET_TRACE("embnet::maxpool_bwd_kernel", TRACE_BYTES, 4.0 * total + 5.0 * n * oh * ow * c, stream); maxpool_bwd_kernel<<<cdiv(total, 256), 256, 0, S(stream)>>>(dy, argmax, n, h, w, c, k, stride, pad, oh, ow, dx); }
  return check_launch("maxpool_bwd");
}

extern "C" int embnet_maxpool_relu_bwd_colsum_ex(const float* dy, const uint8_t* argmax, const float* y, int n, int h, int w,
                                                 int c, int k, int stride, int pad, int oh, int ow, float* dz, float* dbias,
                                                 void* workspace, size_t workspace_bytes, uint32_t* dz_range, void* stream) {
  EMBNET_CHECK_ARG(dy && argmax && y && dz && dbias && workspace, "maxpool_relu_bwd_colsum: null pointer");
  EMBNET_RANGE_ARG(dz_range, "maxpool_relu_bwd_colsum");
  EMBNET_CHECK_ARG(n > 0 && h > 0 && w > 0 && c > 0 && k > 0 && stride > 0 && oh > 0 && ow > 0 && pad >= 0,
                   "maxpool_relu_bwd_colsum: bad geometry");
  EMBNET_CHECK_ARG((c & 3) == 0, "maxpool_relu_bwd_colsum: channel count %d not a multiple of 4 (use maxpool_bwd + relu_bwd_colsum)", c);
  EMBNET_CHECK_ARG((oh - 1) * stride + k <= h + 2 * pad && (ow - 1) * stride + k <= w + 2 * pad,
                   "maxpool_relu_bwd_colsum: window leaves the padded image");
  const long m = (long)n * h * w;
  EMBNET_CHECK_ARG(m < (1l << 31), "maxpool_relu_bwd_colsum: %ld pixels (limit 2^31 - 1)", m);
  if (workspace_bytes < embnet_bn_workspace_bytes(m, c))
    return fail(EMBNET_EWORKSPACE, "maxpool_relu_bwd_colsum: workspace %zu < %zu", workspace_bytes, embnet_bn_workspace_bytes(m, c));
  const ColGeom g4 = col_geom(m, c / 4);
  range_begin(dz_range, stream);
  { EMBNET_TRACE("embnet::maxpool_relu_bwd_colsum4_kernel", TRACE_BYTES, 8.0 * m * c + 5.0 * n * oh * ow * c, stream);
    maxpool_relu_bwd_colsum4_kernel<<<g4.blocks, 256, 0, S(stream)>>>(dy, argmax, y, m, h, w, make_divu((uint32_t)w), make_divu((uint32_t)h),
                                                                       c / 4, k, stride, pad, oh, ow, g4, dz, (float*)workspace, dz_range); }
  colsum_finalize_kernel<<<c, 256, 0, S(stream)>>>((const float*)workspace, g4.blocks, c, dbias, dz_range);     // (folds the range partials)
  return check_launch("maxpool_relu_bwd_colsum");
}
extern "C" int embnet_maxpool_relu_bwd_colsum(const float* dy, const uint8_t* argmax, const float* y, int n, int h, int w,
                                              int c, int k, int stride, int pad, int oh, int ow, float* dz, float* dbias,
                                              void* workspace, size_t workspace_bytes, void* stream) {
  return embnet_maxpool_relu_bwd_colsum_ex(dy, argmax, y, n, h, w, c, k, stride, pad, oh, ow, dz, dbias, workspace, workspace_bytes,
                                           nullptr, stream);
}

extern "C" int embnet_bn_act_maxpool_fwd(const float* x, int n, int h, int w, int c, const float* scale, const float* shift,
                                         int act, int k, int stride, int pad, int oh, int ow, float* y, uint8_t* argmax,
                                         float* xwin, void* stream) {
  EMBNET_CHECK_ARG(x && scale && shift && y && argmax, "bn_act_maxpool_fwd: null pointer");
  EMBNET_CHECK_ARG(n > 0 && h > 0 && w > 0 && c > 0 && k > 0 && k * k < 255 && stride > 0 && pad >= 0 && oh > 0 && ow > 0,
                   "bn_act_maxpool_fwd: bad geometry");
  EMBNET_CHECK_ARG((c & 3) == 0, "bn_act_maxpool_fwd: channel count %d not a multiple of 4 (use bn + maxpool)", c);
  EMBNET_CHECK_ARG((oh - 1) * stride + k <= h + 2 * pad && (ow - 1) * stride + k <= w + 2 * pad,
                   "bn_act_maxpool_fwd: window leaves the padded image");
  const long total = (long)n * oh * ow * (c / 4);
  { EMBNET_TRACE("embnet::affine_act_maxpool_fwd4_kernel", TRACE_BYTES, 4.0 * n * h * w * c + (xwin ? 36.0 : 20.0) * total, stream); affine_act_maxpool_fwd4_kernel<<<cdiv(total, 256), 256, 0, S(stream)>>>(x, n, h, w, c / 4, scale, shift, act, k, stride,
                                                                          pad, oh, ow, y, argmax, xwin); }
  return check_launch("bn_act_maxpool_fwd");
}

extern "C" size_t embnet_bn_act_maxpool_bwd_workspace_bytes(int n, int oh, int ow, int c) {
  return embnet_bn_workspace_bytes((long)n * oh * ow, c);
}

static int bn_act_maxpool_bwd_impl(const float* dy, const uint8_t* argmax, const float* x, int n, int h, int w, int c,
                                   int k, int stride, int pad, int oh, int ow, const float* save_mean,
                                   const float* save_rstd, const float* scale, const float* shift, int act,
                                   int training, const float* xwin, float* dx, float* dgamma, float* dbeta,
                                   void* workspace, size_t workspace_bytes, uint32_t* emit, void* stream) {
  EMBNET_CHECK_ARG(!emit || (save_mean && save_rstd && !(reinterpret_cast<uintptr_t>(emit) & 3)),
                   "bn_act_maxpool_bwd: a range of dx was requested without saved statistics (or the slot is not 4-byte aligned)");
  EMBNET_CHECK_ARG(dy && argmax && x && scale && shift && dx && dgamma && dbeta && workspace, "bn_act_maxpool_bwd: null pointer");
  EMBNET_CHECK_ARG(!training || (save_mean && save_rstd), "bn_act_maxpool_bwd: training needs saved statistics");
  EMBNET_CHECK_ARG(n > 0 && h > 0 && w > 0 && c > 0 && k > 0 && stride > 0 && oh > 0 && ow > 0, "bn_act_maxpool_bwd: bad geometry");
  EMBNET_CHECK_ARG((c & 3) == 0, "bn_act_maxpool_bwd: channel count %d not a multiple of 4", c);
  const long mp = (long)n * oh * ow;
  if (workspace_bytes < embnet_bn_workspace_bytes(mp, c))
    return fail(EMBNET_EWORKSPACE, "bn_act_maxpool_bwd: workspace %zu < %zu", workspace_bytes, embnet_bn_workspace_bytes(mp, c));
  if (save_mean && save_rstd) {
    const ColGeom g4 = col_geom(mp, c / 4);
    { EMBNET_TRACE("embnet::pool_bn_bwd_reduce4_kernel", TRACE_BYTES, 9.0 * mp * c, stream); pool_bn_bwd_reduce4_kernel<<<g4.blocks, 256, 0, S(stream)>>>(dy, argmax, x, mp, h, w, c / 4, k, stride, pad, oh, ow, g4,
                                                                 save_mean, save_rstd, scale, shift, act, xwin, (float*)workspace); }
    bn_bwd_finalize_kernel<<<c, 256, 0, S(stream)>>>((const float*)workspace, g4.blocks, c, dbeta, dgamma, 0, emit);
  } else {
    zero2_kernel<<<cdiv(c, 256), 256, 0, S(stream)>>>(dbeta, dgamma, c);
  }
  const long total = (long)n * h * w * (c / 4);
  { EMBNET_TRACE("embnet::pool_bn_bwd_apply4_kernel", TRACE_BYTES, 32.0 * total + 5.0 * mp * c, stream); pool_bn_bwd_apply4_kernel<<<cdiv(total, 256), 256, 0, S(stream)>>>(dy, argmax, x, n, h, w, c / 4, k, stride, pad, oh, ow,
                                                                    1.f / (float)((long)n * h * w), save_mean, save_rstd,
                                                                    scale, shift, dbeta, dgamma, act, training, dx, emit); }
  if (emit) range_fold_kernel<<<1, 256, 0, S(stream)>>>(emit);
  return check_launch("bn_act_maxpool_bwd");
}
extern "C" int embnet_bn_act_maxpool_bwd(const float* dy, const uint8_t* argmax, const float* x, int n, int h, int w, int c,
                                         int k, int stride, int pad, int oh, int ow, const float* save_mean,
                                         const float* save_rstd, const float* scale, const float* shift, int act,
                                         int training, const float* xwin, float* dx, float* dgamma, float* dbeta,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  return bn_act_maxpool_bwd_impl(dy, argmax, x, n, h, w, c, k, stride, pad, oh, ow, save_mean, save_rstd, scale, shift, act, training,
                                 xwin, dx, dgamma, dbeta, workspace, workspace_bytes, take_emit_slot(), stream);
}
extern "C" int embnet_bn_act_maxpool_bwd_ex(const float* dy, const uint8_t* argmax, const float* x, int n, int h, int w, int c,
                                            int k, int stride, int pad, int oh, int ow, const float* save_mean,
                                            const float* save_rstd, const float* scale, const float* shift, int act,
                                            int training, const float* xwin, float* dx, float* dgamma, float* dbeta,
                                            void* workspace, size_t workspace_bytes, uint32_t* dx_range, void* stream) {
  (void)take_emit_slot();
  return bn_act_maxpool_bwd_impl(dy, argmax, x, n, h, w, c, k, stride, pad, oh, ow, save_mean, save_rstd, scale, shift, act, training,
                                 xwin, dx, dgamma, dbeta, workspace, workspace_bytes, dx_range, stream);
}

extern "C" int embnet_gap_fwd(const float* x, int n, int hw, int c, float* y, void* stream) {
  EMBNET_CHECK_ARG(x && y && n > 0 && hw > 0 && c > 0, "gap_fwd: bad argument");
  if ((c & 3) == 0) { EMBNET_TRACE("embnet::gap_fwd4_kernel", TRACE_BYTES, 4.0 * n * hw * c, stream); gap_fwd4_kernel<<<dim3(cdiv(c / 4, 16), n), 256, 0, S(stream)>>>(x, hw, c / 4, y); }
  else { EMBNET_TRACE("embnet::gap_fwd_kernel", TRACE_BYTES, 4.0 * n * hw * c, stream); gap_fwd_kernel<<<cdiv((long)n * c, 256), 256, 0, S(stream)>>>(x, n, hw, c, y); }
  return check_launch("gap_fwd");
}

extern "C" int embnet_affine_act_gap(const float* x, int n, int hw, int c, const float* scale, const float* shift, int act,
                                     float* y, float* gap, void* stream) {
  EMBNET_CHECK_ARG(x && scale && shift && gap && n > 0 && hw > 0 && c > 0, "affine_act_gap: bad argument");
  EMBNET_CHECK_ARG((c & 3) == 0, "affine_act_gap: channel count %d not a multiple of 4", c);
  EMBNET_TRACE("embnet::affine_act_gap4_kernel", TRACE_BYTES, (y ? 8.0 : 4.0) * n * hw * c, stream);
  const PixGeom g = pix_geom(n, hw, c / 4);
  affine_act_gap4_kernel<<<dim3(g.xblocks, n), g.threads, 0, S(stream)>>>(x, hw, c / 4, g.cls_log2, scale, shift, act, y, gap);
  return check_launch("affine_act_gap");
}

// y[n,p,c] = act(x*scale + shift) * gate[n,c]: the BatchNormalization apply and the squeeze-and-excite multiply in one pass over
// x (embnet_affine_act_gap(y = NULL) has produced the pooled means the gate was computed from): the activated tensor itself is
// never written.  The arithmetic of embnet_affine_act followed by embnet_channel_scale_fwd (one extra rounding each).
__global__ __launch_bounds__(256) void affine_act_scale4_kernel(const float* __restrict__ x, long total4, int c4, DivU dhwc4,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                int act, const float* __restrict__ gate, float* __restrict__ y) {
  const long stride = (long)gridDim.x * 256;
  const bool fixed = stride % c4 == 0;
  float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = sc;
  if (fixed) {
    const int q = (int)(((long)blockIdx.x * 256 + threadIdx.x) % c4);
    sc = reinterpret_cast<const float4*>(scale)[q]; sh = reinterpret_cast<const float4*>(shift)[q];
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
    const int q = (int)(i % c4);
    if (!fixed) { sc = reinterpret_cast<const float4*>(scale)[q]; sh = reinterpret_cast<const float4*>(shift)[q]; }
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 g = reinterpret_cast<const float4*>(gate)[(long)divu((uint32_t)i, dhwc4) * c4 + q];
    float4 o = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
    if (act) { o.x = act_apply(act, o.x); o.y = act_apply(act, o.y); o.z = act_apply(act, o.z); o.w = act_apply(act, o.w); }
    reinterpret_cast<float4*>(y)[i] = make_float4(__fmul_rn(o.x, g.x), __fmul_rn(o.y, g.y), __fmul_rn(o.z, g.z), __fmul_rn(o.w, g.w));
  }
}

extern "C" int embnet_affine_act_scale(const float* x, int n, int hw, int c, const float* scale, const float* shift, int act,
                                       const float* gate, float* y, void* stream) {
  EMBNET_CHECK_ARG(x && scale && shift && gate && y && n > 0 && hw > 0 && c > 0 && (c & 3) == 0, "affine_act_scale: bad argument (c %% 4 == 0)");
  const long total4 = (long)n * hw * (c / 4);
  EMBNET_CHECK_ARG(total4 < 0x7FFFFFFFl, "affine_act_scale: tensor too large");
  EMBNET_TRACE("embnet::affine_act_scale4_kernel", TRACE_BYTES, 8.0 * n * hw * c, stream);
  affine_act_scale4_kernel<<<ew_blocks_c4(total4, c / 4), 256, 0, S(stream)>>>(x, total4, c / 4, make_divu((uint32_t)((long)hw * (c / 4))), scale, shift,
                                                                               act, gate, y);
  return check_launch("affine_act_scale");
}

// y = drop_n(x*scale + shift) + skip: the BatchNormalization apply, the per-sample drop-connect (efficientnet's FixedDropout with
// noise_shape (None,1,1,1): embnet_sample_dropout's mask, rng_u32(seed, n, 2)) and the residual Add of an MBConv block in one
// pass (28 -> 12 bytes per element); drop_factor_kernel leaves factor[n,c] = 1/(1-rate) or 0 for the backward (embnet_bn_bwd_gap's gate).
// The three layers' arithmetic, rounding for rounding.  rate == 0: no drop (factor 1).
__global__ __launch_bounds__(256) void affine_drop_add4_kernel(const float* __restrict__ x, long total4, int c4, DivU dhwc4,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               float rate, uint64_t seed, const uint64_t* __restrict__ seed_add,
                                                               const float* __restrict__ skip, float* __restrict__ y) {
  if (seed_add) seed += *seed_add;
  const float keep_scale = 1.f / (1.f - rate);
  const uint32_t thr = (uint32_t)((double)rate * 4294967296.0);
  const long stride = (long)gridDim.x * 256;
  const bool fixed = stride % c4 == 0;
  float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = sc;
  if (fixed) {
    const int q = (int)(((long)blockIdx.x * 256 + threadIdx.x) % c4);
    sc = reinterpret_cast<const float4*>(scale)[q]; sh = reinterpret_cast<const float4*>(shift)[q];
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
    const int q = (int)(i % c4);
    if (!fixed) { sc = reinterpret_cast<const float4*>(scale)[q]; sh = reinterpret_cast<const float4*>(shift)[q]; }
    const uint32_t nimg = divu((uint32_t)i, dhwc4);
    const bool keep = rate <= 0.f || rng_u32(seed, (uint64_t)nimg, 2) >= thr;
    const float4 v = reinterpret_cast<const float4*>(x)[i], k = reinterpret_cast<const float4*>(skip)[i];
    float4 o = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
    if (rate > 0.f) {
      o.x = keep ? __fmul_rn(o.x, keep_scale) : 0.f; o.y = keep ? __fmul_rn(o.y, keep_scale) : 0.f;
      o.z = keep ? __fmul_rn(o.z, keep_scale) : 0.f; o.w = keep ? __fmul_rn(o.w, keep_scale) : 0.f;
    }
    reinterpret_cast<float4*>(y)[i] = make_float4(__fadd_rn(o.x, k.x), __fadd_rn(o.y, k.y), __fadd_rn(o.z, k.z), __fadd_rn(o.w, k.w));
  }
}

__global__ __launch_bounds__(256) void drop_factor_kernel(int n, int c, float rate, uint64_t seed, const uint64_t* __restrict__ seed_add,
                                                          float* __restrict__ factor) {
  if (seed_add) seed += *seed_add;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * c) return;
  const uint32_t thr = (uint32_t)((double)rate * 4294967296.0);
  factor[i] = (rate <= 0.f || rng_u32(seed, (uint64_t)(i / c), 2) >= thr) ? 1.f / (1.f - rate) : 0.f;
}

extern "C" int embnet_affine_drop_add(const float* x, int n, int hw, int c, const float* scale, const float* shift, float rate,
                                      uint64_t seed, const uint64_t* seed_add_dev, const float* skip, float* y, float* factor,
                                      void* stream) {
  EMBNET_CHECK_ARG(x && scale && shift && skip && y && factor && n > 0 && hw > 0 && c > 0 && (c & 3) == 0, "affine_drop_add: bad argument (c %% 4 == 0)");
  EMBNET_CHECK_ARG(rate >= 0.f && rate < 1.f, "affine_drop_add: rate %f outside [0,1)", rate);
  const long total4 = (long)n * hw * (c / 4);
  EMBNET_CHECK_ARG(total4 < 0x7FFFFFFFl, "affine_drop_add: tensor too large");
  drop_factor_kernel<<<(n * c + 255) / 256, 256, 0, S(stream)>>>(n, c, rate, seed, seed_add_dev, factor);
  EMBNET_TRACE("embnet::affine_drop_add4_kernel", TRACE_BYTES, 12.0 * n * hw * c, stream);
  affine_drop_add4_kernel<<<ew_blocks_c4(total4, c / 4), 256, 0, S(stream)>>>(x, total4, c / 4, make_divu((uint32_t)((long)hw * (c / 4))), scale, shift,
                                                                              rate, seed, seed_add_dev, skip, y);
  return check_launch("affine_drop_add");
}

extern "C" int embnet_gap_bwd(const float* dy, int n, int hw, int c, const float* dx_add, float* dx, void* stream) {
  EMBNET_CHECK_ARG(dy && dx && n > 0 && hw > 0 && c > 0, "gap_bwd: bad argument");
  EMBNET_CHECK_ARG(!dx_add || (c & 3) == 0, "gap_bwd: dx_add needs c %% 4 == 0 (got %d)", c);
  if ((c & 3) == 0) {
    const long total4 = (long)n * hw * (c / 4);
    EMBNET_TRACE("embnet::gap_bwd_add4_kernel", TRACE_BYTES, (dx_add ? 8.0 : 4.0) * n * hw * c, stream);
    gap_bwd_add4_kernel<<<ew_blocks(total4), 256, 0, S(stream)>>>(dy, dx_add, total4, hw, c / 4, dx);
    return check_launch("gap_bwd");
  }
  { EMBNET_TRACE("embnet::gap_bwd_kernel", TRACE_BYTES, 4.0 * n * hw * c, stream); gap_bwd_kernel<<<cdiv((long)n * hw * c, 256), 256, 0, S(stream)>>>(dy, n, hw, c, dx); }
  return check_launch("gap_bwd");
}

extern "C" int embnet_relu_bwd(const float* dy, const float* y, long total, float* dz, void* stream) {
  EMBNET_CHECK_ARG(dy && y && dz && total > 0, "relu_bwd: bad argument");
  { EMBNET_TRACE("embnet::relu_bwd_kernel", TRACE_BYTES, 12.0 * total, stream); relu_bwd_kernel<<<ew_blocks(total), 256, 0, S(stream)>>>(dy, y, total, dz); }
  return check_launch("relu_bwd");
}

extern "C" size_t embnet_colsum_workspace_bytes(long m, int c) { return embnet_bn_workspace_bytes(m, c); }

extern "C" int embnet_colsum(const float* x, long m, int c, float* out, void* workspace, size_t workspace_bytes,
                             void* stream) {
  EMBNET_CHECK_ARG(x && out && workspace && m > 0 && c > 0, "colsum: bad argument");
  if (workspace_bytes < embnet_bn_workspace_bytes(m, c))
    return fail(EMBNET_EWORKSPACE, "colsum: workspace %zu < %zu", workspace_bytes, embnet_bn_workspace_bytes(m, c));
  const ColGeom g = col_geom(m, c);
  { EMBNET_TRACE("embnet::colsum_kernel", TRACE_BYTES, 4.0 * m * c, stream); colsum_kernel<<<g.blocks, 256, 0, S(stream)>>>(x, m, c, g, (float*)workspace); }
  colsum_finalize_kernel<<<c, 256, 0, S(stream)>>>((const float*)workspace, g.blocks, c, out);
  return check_launch("colsum");
}

extern "C" int embnet_relu_bwd_colsum_ex(const float* dy, const float* y, long m, int c, float* dz, float* dbias, void* workspace,
                                         size_t workspace_bytes, uint32_t* dz_range, void* stream) {
  EMBNET_CHECK_ARG(dy && y && dz && dbias && workspace && m > 0 && c > 0, "relu_bwd_colsum: bad argument");
  EMBNET_RANGE_ARG(dz_range, "relu_bwd_colsum");
  if (workspace_bytes < embnet_bn_workspace_bytes(m, c))
    return fail(EMBNET_EWORKSPACE, "relu_bwd_colsum: workspace %zu < %zu", workspace_bytes, embnet_bn_workspace_bytes(m, c));
  const ColGeom g = col_geom(m, c);
  range_begin(dz_range, stream);
  { EMBNET_TRACE("embnet::relu_bwd_colsum_kernel", TRACE_BYTES, 12.0 * m * c, stream);
    relu_bwd_colsum_kernel<<<g.blocks, 256, 0, S(stream)>>>(dy, y, m, c, g, dz, (float*)workspace, dz_range); }
  colsum_finalize_kernel<<<c, 256, 0, S(stream)>>>((const float*)workspace, g.blocks, c, dbias, dz_range);      // (folds the range partials)
  return check_launch("relu_bwd_colsum");
}
extern "C" int embnet_relu_bwd_colsum(const float* dy, const float* y, long m, int c, float* dz, float* dbias, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  return embnet_relu_bwd_colsum_ex(dy, y, m, c, dz, dbias, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int embnet_add(const float* a, const float* b, long total, float* y, void* stream) {
  EMBNET_CHECK_ARG(a && b && y && total > 0, "add: bad argument");
  { EMBNET_TRACE("embnet::add_kernel", TRACE_BYTES, 12.0 * total, stream); add_kernel<<<ew_blocks(total), 256, 0, S(stream)>>>(a, b, total, y); }
  return check_launch("add");
}

extern "C" int embnet_scale(const float* x, long total, float alpha, const float* alpha_dev, float* y, void* stream) {
  EMBNET_CHECK_ARG(x && y && total > 0, "scale: bad argument");
  { EMBNET_TRACE("embnet::scale_kernel", TRACE_BYTES, 8.0 * total, stream); scale_kernel<<<ew_blocks(total), 256, 0, S(stream)>>>(x, total, alpha, alpha_dev, y); }
  return check_launch("scale");
}

extern "C" int embnet_dropout(const float* x, long total, float rate, uint64_t seed, const uint64_t* seed_add_dev, float* y,
                              void* stream) {
  EMBNET_CHECK_ARG(x && y && total > 0, "dropout: bad argument");
  EMBNET_CHECK_ARG(rate >= 0.f && rate < 1.f, "dropout: rate %f outside [0,1)", rate);
  { EMBNET_TRACE("embnet::dropout_kernel", TRACE_BYTES, 8.0 * total, stream); dropout_kernel<<<ew_blocks(total), 256, 0, S(stream)>>>(x, total, rate, seed, seed_add_dev, y); }
  return check_launch("dropout");
}

extern "C" int embnet_tap_contract(const float* w, const float* tap_sums, int taps, int c, int k, float* out,
                                   void* stream) {
  EMBNET_CHECK_ARG(w && tap_sums && out && taps > 0 && c > 0 && k > 0, "tap_contract: bad argument");
  tap_contract_kernel<<<c, 256, 0, S(stream)>>>(w, tap_sums, taps, c, k, out);
  return check_launch("tap_contract");
}

// border rows / columns of the output map: those with at least one tap in the padding
static bool make_tap_border(TapBorder& g, int oh, int ow, int k, int r, int s, int stride, int pad_t, int pad_l, int h, int w) {
  g = TapBorder{oh, ow, k, r, s, stride, pad_t, pad_l, h, w, 0, oh, 0, ow};
  auto row_out = [&](int y) { return y * stride - pad_t < 0 || y * stride + r - 1 - pad_t >= h; };
  auto col_out = [&](int x) { return x * stride - pad_l < 0 || x * stride + s - 1 - pad_l >= w; };
  while (g.yt < oh && row_out(g.yt)) ++g.yt;
  while (g.yb > g.yt && row_out(g.yb - 1)) --g.yb;
  while (g.xl < ow && col_out(g.xl)) ++g.xl;
  while (g.xr > g.xl && col_out(g.xr - 1)) --g.xr;
  for (int y = g.yt; y < g.yb; ++y) if (row_out(y)) return false;        // (cannot happen for a contiguous image)
  for (int x = g.xl; x < g.xr; ++x) if (col_out(x)) return false;
  return true;
}
static int tap_border_lines(const TapBorder& g) {
  const int nrow = g.yt + (g.oh - g.yb), ncol = g.xl + (g.ow - g.xr);
  return nrow + ncol + nrow * ncol;
}

extern "C" size_t embnet_tap_border_sums_workspace_bytes(int n, int oh, int ow, int k, int r, int s, int stride, int pad_t,
                                                         int pad_l, int h, int w) {
  if (n <= 0 || oh <= 0 || ow <= 0 || r <= 0 || s <= 0 || k <= 0 || stride <= 0) return 0;
  TapBorder g;
  if (!make_tap_border(g, oh, ow, k, r, s, stride, pad_t, pad_l, h, w)) return 0;
  return (size_t)tap_border_lines(g) * ((size_t)n + 1) * k * sizeof(float) + 16;
}

extern "C" int embnet_tap_border_sums(const float* dy, int n, int oh, int ow, int k, int r, int s, int stride, int pad_t,
                                      int pad_l, int h, int w, float* taps, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  EMBNET_CHECK_ARG(dy && taps, "tap_border_sums: null pointer");
  EMBNET_CHECK_ARG(n > 0 && oh > 0 && ow > 0 && k > 0 && r > 0 && s > 0 && stride > 0 && h > 0 && w > 0, "tap_border_sums: bad geometry");
  TapBorder g;
  EMBNET_CHECK_ARG(make_tap_border(g, oh, ow, k, r, s, stride, pad_t, pad_l, h, w), "tap_border_sums: border rows not contiguous");
  const int nlines = tap_border_lines(g);
  if (nlines == 0) {                                       // no padding at all: every tap sums the whole (zero-sum) map
    zero2_kernel<<<cdiv(r * s * k, 256), 256, 0, S(stream)>>>(taps, nullptr, r * s * k);
    return check_launch("tap_border_sums");
  }
  const size_t need = embnet_tap_border_sums_workspace_bytes(n, oh, ow, k, r, s, stride, pad_t, pad_l, h, w);
  EMBNET_CHECK_ARG(workspace, "tap_border_sums: null workspace");
  if (workspace_bytes < need) return fail(EMBNET_EWORKSPACE, "tap_border_sums: workspace %zu < %zu", workspace_bytes, need);
  float* partial = (float*)workspace;
  float* lines = partial + (size_t)nlines * n * k;
  {
    EMBNET_TRACE("embnet::tap_border_lines_kernel", TRACE_BYTES,
                 4.0 * n * k * ((double)(g.yt + oh - g.yb) * ow + (double)(g.xl + ow - g.xr) * oh), stream);
    tap_border_lines_kernel<<<dim3(nlines, n), 256, 0, S(stream)>>>(dy, g, partial);
  }
  {
    EMBNET_TRACE("embnet::tap_border_reduce_kernel", TRACE_BYTES, 4.0 * nlines * n * k, stream);
    tap_border_reduce_kernel<<<nlines, 256, 0, S(stream)>>>(partial, n, k, lines);
  }
  EMBNET_TRACE("embnet::tap_border_combine_kernel", TRACE_BYTES, 4.0 * r * s * k, stream);
  tap_border_combine_kernel<<<cdiv(r * s * k, 256), 256, 0, S(stream)>>>(lines, g, taps);
  return check_launch("tap_border_sums");
}

extern "C" int embnet_pad_channels_ex(const float* x, long pixels, int cin, int cout, float* y, uint32_t* y_range, void* stream) {
  EMBNET_CHECK_ARG(x && y && pixels > 0 && cin > 0 && cout >= cin, "pad_channels: bad argument");
  EMBNET_RANGE_ARG(y_range, "pad_channels");
  range_begin(y_range, stream);
  { EMBNET_TRACE("embnet::pad_channels_kernel", TRACE_BYTES, 4.0 * pixels * (cin + cout), stream); pad_channels_kernel<<<ew_blocks(pixels * cout), 256, 0, S(stream)>>>(x, pixels, cin, cout, y, y_range); }
  range_end(y_range, stream);
  return check_launch("pad_channels");
}
extern "C" int embnet_pad_channels(const float* x, long pixels, int cin, int cout, float* y, void* stream) {
  return embnet_pad_channels_ex(x, pixels, cin, cout, y, nullptr, stream);
}

extern "C" int embnet_sumsq_chunk_elems(void) { return 4096; }
extern "C" int embnet_sumsq_multi(const void* table, int n_tensors, const int32_t* chunks, int n_chunks, float* out, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  EMBNET_CHECK_ARG(table && chunks && out && workspace && n_tensors > 0 && n_chunks > 0, "sumsq_multi: bad argument");
  if (workspace_bytes < (size_t)n_chunks * sizeof(float)) return fail(EMBNET_EWORKSPACE, "sumsq_multi: workspace too small");
  { EMBNET_TRACE("embnet::sumsq_multi_kernel", TRACE_BYTES, 4.0 * 4096 * n_chunks, stream);
    sumsq_multi_kernel<<<n_chunks, 256, 0, S(stream)>>>((const SumsqTensor*)table, chunks, (float*)workspace); }
  sum_partials_kernel<<<1, 256, 0, S(stream)>>>((const float*)workspace, n_chunks, out);
  return check_launch("sumsq_multi");
}

extern "C" size_t embnet_sumsq_workspace_bytes(void) { return 1024 * sizeof(float); }

extern "C" int embnet_sumsq(const float* x, long total, float alpha, float* out, void* workspace,
                            size_t workspace_bytes, void* stream) {
  EMBNET_CHECK_ARG(x && out && workspace && total > 0, "sumsq: bad argument");
  if (workspace_bytes < embnet_sumsq_workspace_bytes()) return fail(EMBNET_EWORKSPACE, "sumsq: workspace too small");
  const int blocks = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
  { EMBNET_TRACE("embnet::sumsq_partial_kernel", TRACE_BYTES, 4.0 * total, stream); sumsq_partial_kernel<<<blocks, 256, 0, S(stream)>>>(x, total, (float*)workspace); }
  sum_finalize_kernel<<<1, 64, 0, S(stream)>>>((const float*)workspace, blocks, alpha, out);
  return check_launch("sumsq");
}
