// NHWC fp32 convolution as implicit GEMM on the fp32 MFMA engine: forward, data gradient,
// weight gradient.  Hand-written stand-in for the Keras Conv2D layers that
// /root/reference/embedding_net/backbones.py:21-31,44-68 and the zoo backbones (:84-104)
// instantiate (TensorFlow's cuDNN/Eigen kernels in the reference).
//
//   fwd   Y[(n,oh,ow)][k]  = sum_{(r,s,c)} X[n, oh*st+r-pt, ow*st+s-pl, c] * W[(r,s,c)][k]
//         A = im2col gather (KC tile: c contiguous), B = W as stored (KM tile), bias/ReLU epilogue
//   dgrad dX[(n,h,w)][c]   = sum_{(r,s,k)} dY[n,(h+pt-r)/st,(w+pl-s)/st,k] * W[r,s,c,k]
//         A = gather of dY (KC), B = W rows indexed by c (KC, k contiguous) — no weight transform
//   wgrad dW[(r,s,c)][k]   = sum_{(n,oh,ow)} X[n, oh*st+r-pt, ow*st+s-pl, c] * dY[(n,oh,ow)][k]
//         A = gather of X (KM: c contiguous), B = dY as stored (KM); split-K over workgroups into
//         fp32 slabs + a fixed-order reduce (bitwise reproducible, no float atomics)
// All gathers are branch-free buffer loads: padding taps, rows past the edge and k past the end get
// an out-of-range offset and read as 0 (gemm_engine.h).  VEC = 16-byte gathers (channel count % 4 == 0).
// Roofline: MFMA f32 (157.3 TFLOP/s); algorithmic FLOP = 2 * N*OH*OW * K * R*S*C per pass.
#include "gemm_engine.h"
#include "conv_geom.h"
#include "../../include/embnet.h"
#include <stdlib.h>

namespace embnet {

// Optional per-channel transform of the conv INPUT, applied in registers between the gather and LDS:
// a = act(x*scale[c] + shift[c]) — the BatchNormalization(+activation) in front of the conv — so the
// normalised tensor is never written.  Padding taps / out-of-range rows stay exactly 0 (the reference
// pads AFTER the activation).  16-byte-lane loaders only (channel count % 4 == 0).
struct InputTransform { const float* scale; const float* shift; int act; };
__device__ __forceinline__ float4 transform4(float4 v, bool ok, const float4& sc, const float4& sh, int act) {
  float4 z = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
  if (act == 1) {                                        // wave-uniform: the common case costs fma + max + select
    z = make_float4(fmaxf(z.x, 0.f), fmaxf(z.y, 0.f), fmaxf(z.z, 0.f), fmaxf(z.w, 0.f));
  } else if (act == 2) {
    z = make_float4(act_apply(2, z.x), act_apply(2, z.y), act_apply(2, z.z), act_apply(2, z.w));
  }
  return ok ? z : make_float4(0.f, 0.f, 0.f, 0.f);
}


// ---- forward A: rows = output pixels, k = (r,s,c) -------------------------------------------
template <int ROWS, bool VEC, bool TF = false>
struct LoadConvFwdA {
  using Tile = TileKC<ROWS>;
  static constexpr bool CAN_INTERLEAVE = VEC && !TF;
  Buf buf; int H, W, C, Kg; FastDiv dC, dS; int tid;
  unsigned base[Tile::PASSES]; int ih0[Tile::PASSES], iw0[Tile::PASSES];
  InputTransform tf; mutable unsigned okbits; mutable float4 tsc, tsh;
  __device__ void init(const float* x, const ConvGeom& g, int m0, int tid_, const InputTransform& tf_) {
    buf.init(x, (size_t)g.N * g.H * g.W * g.C * 4);
    H = g.H; W = g.W; C = g.C; Kg = g.R * g.S * g.C; dC = g.dC; dS = g.dS; tid = tid_; tf = tf_; okbits = 0;
    const int M = g.N * g.OH * g.OW;
#pragma unroll
    for (int p = 0; p < Tile::PASSES; ++p) {
      const int m = m0 + Tile::row_of(tid, p);
      uint32_t n, rem, oh, ow;
      g.dOHW.divmod((uint32_t)min(m, M - 1), n, rem); g.dOW.divmod(rem, oh, ow);
      base[p] = 4u * n * (unsigned)(g.H * g.W * g.C);
      ih0[p] = m < M ? (int)oh * g.stride - g.pad_t : ROW_INVALID;
      iw0[p] = (int)ow * g.stride - g.pad_l;
    }
  }
  __device__ __forceinline__ unsigned off(int p, int kk) const {
    int r, s, c; split_k(kk, dC, dS, r, s, c);
    const int ih = ih0[p] + r, iw = iw0[p] + s;
    const bool ok = kk < Kg && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    return ok ? base[p] + 4u * (unsigned)((ih * W + iw) * C + c) : OOB;
  }
  __device__ __forceinline__ void fix(float4 (&rg)[Tile::PASSES]) const {
    if (!VEC || !TF) return;
#pragma unroll
    for (int p = 0; p < Tile::PASSES; ++p) rg[p] = transform4(rg[p], (okbits >> p) & 1u, tsc, tsh, tf.act);
  }
  __device__ __forceinline__ void load_pass(int kt, int p, float4& rg) const {       // VEC && !TF (CAN_INTERLEAVE)
    const int kk = kt * BK + Tile::k_of(tid);
    int r, s, c; split_k(kk, dC, dS, r, s, c);
    const int ih = ih0[p] + r, iw = iw0[p] + s;
    const bool ok = kk < Kg && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    const unsigned o = opaque(base[p] + 4u * (unsigned)((ih * W + iw) * C + c));
    rg = buf.ld4(ok ? o : OOB);
  }
  __device__ __forceinline__ void load(int kt, float4 (&rg)[Tile::PASSES]) const {
    const int kk = kt * BK + Tile::k_of(tid);
    if (VEC) {
      int r, s, c; split_k(kk, dC, dS, r, s, c);
      const bool kin = kk < Kg;
      unsigned bits = 0;
#pragma unroll
      for (int p = 0; p < Tile::PASSES; ++p) {
        const int ih = ih0[p] + r, iw = iw0[p] + s;
        const bool ok = kin && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
        bits |= (unsigned)ok << p;
        const unsigned o = opaque(base[p] + 4u * (unsigned)((ih * W + iw) * C + c));
        rg[p] = buf.ld4(ok ? o : OOB);
      }
      if (TF) {                                           // c in [0,C) also past the K end (kk mod C)
        okbits = bits;
        tsc = *reinterpret_cast<const float4*>(tf.scale + c); tsh = *reinterpret_cast<const float4*>(tf.shift + c);
      }
    } else {
#pragma unroll
      for (int p = 0; p < Tile::PASSES; ++p)
        rg[p] = make_float4(buf.ld1(off(p, kk)), buf.ld1(off(p, kk + 1)), buf.ld1(off(p, kk + 2)),
                            buf.ld1(off(p, kk + 3)));
    }
  }
};

// ---- dgrad, by stride class -------------------------------------------------------------------
// With stride st an input pixel (h,w) only receives from taps r = (h+pad_t) mod st (mod st), so the
// gradient splits into st*st independent stride-1 correlations ("classes"), one per residue pair
// (ph,pw): pixels h = hoff + st*hc, taps r = ph + st*tr, and the output row is oh = hc + hb - tr.
// Each class is its own implicit GEMM (rows = its pixels, k = its taps x K): no structurally-zero
// taps are multiplied (a 3x3 stride-2 conv does 9 tap-GEMMs per 4 pixels instead of 36).
// stride 1 is the single class (0,0).
struct DgradClass {
  int hoff, woff, Hc, Wc, nR, nS, r0, s0, hb, wb;
  FastDiv dHWc, dWc, dnS;
};
constexpr int MAX_CLASSES = 9;          // stride <= 3

// A: rows = the class's input pixels, k = (tr, ts, kout)
template <int ROWS, bool VEC>
struct LoadConvDgradA {
  using Tile = TileKC<ROWS>;
  static constexpr bool CAN_INTERLEAVE = VEC;
  Buf buf; int OH, OW, K, Kg; FastDiv dK, dnS; int tid;
  unsigned base[Tile::PASSES]; int ih0[Tile::PASSES], iw0[Tile::PASSES];
  __device__ void init(const float* dy, const ConvGeom& g, const DgradClass& cg, int m0, int tid_) {
    buf.init(dy, (size_t)g.N * g.OH * g.OW * g.K * 4);
    OH = g.OH; OW = g.OW; K = g.K; Kg = cg.nR * cg.nS * g.K; dK = g.dK; dnS = cg.dnS; tid = tid_;
    const int M = g.N * cg.Hc * cg.Wc;
#pragma unroll
    for (int p = 0; p < Tile::PASSES; ++p) {
      const int m = m0 + Tile::row_of(tid, p);
      uint32_t n, rem, hc, wc;
      cg.dHWc.divmod((uint32_t)min(m, max(M - 1, 0)), n, rem); cg.dWc.divmod(rem, hc, wc);
      base[p] = 4u * n * (unsigned)(g.OH * g.OW * g.K);
      ih0[p] = m < M ? (int)hc + cg.hb : ROW_INVALID;
      iw0[p] = (int)wc + cg.wb;
    }
  }
  __device__ __forceinline__ void fix(float4 (&)[Tile::PASSES]) const {}
  __device__ __forceinline__ unsigned off(int p, int kk) const {
    uint32_t t, c, tr, ts; dK.divmod((uint32_t)kk, t, c); dnS.divmod(t, tr, ts);
    const int oh = ih0[p] - (int)tr, ow = iw0[p] - (int)ts;
    const bool ok = kk < Kg && (unsigned)oh < (unsigned)OH && (unsigned)ow < (unsigned)OW;
    return ok ? base[p] + 4u * (unsigned)((oh * OW + ow) * K + (int)c) : OOB;
  }
  __device__ __forceinline__ void load_pass(int kt, int p, float4& rg) const {       // VEC (CAN_INTERLEAVE)
    const int kk = kt * BK + Tile::k_of(tid);
    uint32_t t, c, tr, ts; dK.divmod((uint32_t)kk, t, c); dnS.divmod(t, tr, ts);
    const int oh = ih0[p] - (int)tr, ow = iw0[p] - (int)ts;
    const bool ok = kk < Kg && (unsigned)oh < (unsigned)OH && (unsigned)ow < (unsigned)OW;
    const unsigned o = opaque(base[p] + 4u * (unsigned)((oh * OW + ow) * K + (int)c));
    rg = buf.ld4(ok ? o : OOB);
  }
  __device__ __forceinline__ void load(int kt, float4 (&rg)[Tile::PASSES]) const {
    const int kk = kt * BK + Tile::k_of(tid);
    if (VEC) {
      uint32_t t, c, tr, ts; dK.divmod((uint32_t)kk, t, c); dnS.divmod(t, tr, ts);
      const bool kin = kk < Kg;
#pragma unroll
      for (int p = 0; p < Tile::PASSES; ++p) {
        const int oh = ih0[p] - (int)tr, ow = iw0[p] - (int)ts;
        const bool ok = kin && (unsigned)oh < (unsigned)OH && (unsigned)ow < (unsigned)OW;
        const unsigned o = opaque(base[p] + 4u * (unsigned)((oh * OW + ow) * K + (int)c));
        rg[p] = buf.ld4(ok ? o : OOB);
      }
    } else {
#pragma unroll
      for (int p = 0; p < Tile::PASSES; ++p)
        rg[p] = make_float4(buf.ld1(off(p, kk)), buf.ld1(off(p, kk + 1)), buf.ld1(off(p, kk + 2)),
                            buf.ld1(off(p, kk + 3)));
    }
  }
};

// B: rows = input channel c, k = (tr, ts, kout): W[((r*S+s)*C + c)*K + kout], r = r0 + st*tr, s = s0 + st*ts
template <int ROWS, bool VEC>
struct LoadConvDgradB {
  using Tile = TileKC<ROWS>;
  static constexpr bool CAN_INTERLEAVE = VEC;
  Buf buf; int C, K, S, Kg, st, r0, s0; FastDiv dK, dnS; int row0, tid;
  __device__ void init(const float* w, const ConvGeom& g, const DgradClass& cg, int n0, int tid_) {
    buf.init(w, (size_t)g.R * g.S * g.C * g.K * 4);
    C = g.C; K = g.K; S = g.S; Kg = cg.nR * cg.nS * g.K; st = g.stride; r0 = cg.r0; s0 = cg.s0;
    dK = g.dK; dnS = cg.dnS; row0 = n0; tid = tid_;
  }
  __device__ __forceinline__ void fix(float4 (&)[Tile::PASSES]) const {}
  __device__ __forceinline__ unsigned off(int c, int kk) const {
    uint32_t t, ko, tr, ts; dK.divmod((uint32_t)kk, t, ko); dnS.divmod(t, tr, ts);
    const unsigned rs = (unsigned)((r0 + st * (int)tr) * S + s0 + st * (int)ts);
    return (kk < Kg && c < C) ? 4u * ((rs * (unsigned)C + (unsigned)c) * (unsigned)K + ko) : OOB;
  }
  __device__ __forceinline__ void load_pass(int kt, int p, float4& rg) const {       // VEC (CAN_INTERLEAVE)
    rg = buf.ld4(off(row0 + Tile::row_of(tid, p), kt * BK + Tile::k_of(tid)));
  }
  __device__ __forceinline__ void load(int kt, float4 (&rg)[Tile::PASSES]) const {
    const int kk = kt * BK + Tile::k_of(tid);
#pragma unroll
    for (int p = 0; p < Tile::PASSES; ++p) {
      const int c = row0 + Tile::row_of(tid, p);
      if (VEC) rg[p] = buf.ld4(off(c, kk));
      else rg[p] = make_float4(buf.ld1(off(c, kk)), buf.ld1(off(c, kk + 1)), buf.ld1(off(c, kk + 2)),
                               buf.ld1(off(c, kk + 3)));
    }
  }
};


// ---- wgrad A: k = output pixel (n,oh,ow), rows = (r,s,c) ------------------------------------------
template <int ROWS, bool VEC, bool TF = false>
struct LoadConvWgradA {
  using Tile = TileKM<ROWS>;
  static constexpr bool CAN_INTERLEAVE = VEC && !TF;
  Buf buf; int H, W, C, Mrows, Kg, stride, pad_t, pad_l; FastDiv dOHW, dOW, dC, dS; unsigned HWC4; int tid;
  int r_[Tile::PASSES], s_[Tile::PASSES], c_[Tile::PASSES], m_[Tile::PASSES];    // first row of each pass
  InputTransform tf; mutable unsigned okbits;
  float4 tsc[Tile::PASSES], tsh[Tile::PASSES];            // rows (hence channels) of a pass never change
  __device__ __forceinline__ void fix(float4 (&rg)[Tile::PASSES]) const {
    if (!VEC || !TF) return;
#pragma unroll
    for (int p = 0; p < Tile::PASSES; ++p) rg[p] = transform4(rg[p], (okbits >> p) & 1u, tsc[p], tsh[p], tf.act);
  }
  __device__ void init(const float* x, const ConvGeom& g, int m0, int tid_, const InputTransform& tf_) {
    tf = tf_; okbits = 0;
    buf.init(x, (size_t)g.N * g.H * g.W * g.C * 4);
    H = g.H; W = g.W; C = g.C; Mrows = g.R * g.S * g.C; Kg = g.N * g.OH * g.OW;
    stride = g.stride; pad_t = g.pad_t; pad_l = g.pad_l; dOHW = g.dOHW; dOW = g.dOW; dC = g.dC; dS = g.dS;
    HWC4 = 4u * (unsigned)(g.H * g.W * g.C); tid = tid_;
#pragma unroll
    for (int p = 0; p < Tile::PASSES; ++p) {
      m_[p] = m0 + Tile::row_of(tid, p);
      split_k(min(m_[p], Mrows - 1), dC, dS, r_[p], s_[p], c_[p]);
      if (VEC && TF) {                                    // rows past the end are clamped to Mrows-1 (c = C-1): read quad 0, they are masked
        const int cq = m_[p] < Mrows ? c_[p] : 0;
        tsc[p] = *reinterpret_cast<const float4*>(tf.scale + cq); tsh[p] = *reinterpret_cast<const float4*>(tf.shift + cq);
      }
    }
  }
  __device__ __forceinline__ unsigned off(int m, unsigned n, int oh, int ow, bool kin) const {
    int r, s, c; split_k(min(m, Mrows - 1), dC, dS, r, s, c);
    const int ih = oh * stride + r - pad_t, iw = ow * stride + s - pad_l;
    const bool ok = kin && m < Mrows && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    return ok ? n * HWC4 + 4u * (unsigned)((ih * W + iw) * C + c) : OOB;
  }
  __device__ __forceinline__ void load_pass(int kt, int p, float4& rg) const {       // VEC && !TF (CAN_INTERLEAVE)
    const int kg = kt * BK + Tile::k_of(tid, p);
    const bool kin = kg < Kg;
    uint32_t n, rem, oh, ow;
    dOHW.divmod((uint32_t)(kin ? kg : 0), n, rem); dOW.divmod(rem, oh, ow);
    const int ih = (int)oh * stride + r_[p] - pad_t, iw = (int)ow * stride + s_[p] - pad_l;
    const bool ok = kin && m_[p] < Mrows && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    const unsigned o = opaque(n * HWC4 + 4u * (unsigned)((ih * W + iw) * C + c_[p]));
    rg = buf.ld4(ok ? o : OOB);
  }
  __device__ __forceinline__ void load(int kt, float4 (&rg)[Tile::PASSES]) const {
    unsigned bits = 0;
#pragma unroll
    for (int p = 0; p < Tile::PASSES; ++p) {
      const int kg = kt * BK + Tile::k_of(tid, p);
      const bool kin = kg < Kg;
      uint32_t n, rem, oh, ow;
      dOHW.divmod((uint32_t)(kin ? kg : 0), n, rem); dOW.divmod(rem, oh, ow);
      if (VEC) {
        const int ih = (int)oh * stride + r_[p] - pad_t, iw = (int)ow * stride + s_[p] - pad_l;
        const bool ok = kin && m_[p] < Mrows && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
        bits |= (unsigned)ok << p;
        const unsigned o = opaque(n * HWC4 + 4u * (unsigned)((ih * W + iw) * C + c_[p]));
        rg[p] = buf.ld4(ok ? o : OOB);
      } else {
        rg[p] = make_float4(buf.ld1(off(m_[p], n, oh, ow, kin)), buf.ld1(off(m_[p] + 1, n, oh, ow, kin)),
                            buf.ld1(off(m_[p] + 2, n, oh, ow, kin)), buf.ld1(off(m_[p] + 3, n, oh, ow, kin)));
      }
    }
    if (TF) okbits = bits;
  }
};

// ---------------------------------------------------------------------------------------------
// Arithmetic of the convolution main loops: 1 = fp32 products as six bf16 MFMA terms of an exact three-way split
// (gemm_engine.h, "bf16x6"; error vs fp64 no larger than the fp32 MFMA's), 0 = v_mfma_f32_32x32x2_f32.
#ifndef EMBNET_CONV_SPLIT
#define EMBNET_CONV_SPLIT 1
#endif
template <class TA, class TB>
constexpr int CONV_MAIN_FLOATS = EMBNET_CONV_SPLIT ? (MAIN3_BYTES<TA, TB> + 3) / 4 : MAIN_FLOATS<TA, TB>;
template <class G, class TA, class TB>
constexpr int SMEM_FLOATS = CONV_MAIN_FLOATS<TA, TB> > EPI_FLOATS<G> ? CONV_MAIN_FLOATS<TA, TB> : EPI_FLOATS<G>;

// Workgroups per CU the LDS image allows (one wave of each on every SIMD) -> the register budget hipcc must keep
// for the 16-byte-load kernels (the scalar-load variants for odd channel counts would spill)
// (__launch_bounds__'s second argument is waves per SIMD): 128x128 and 192x64 two, 128x64 three, smaller tiles four.
#ifndef EMBNET_OCC_12864
#define EMBNET_OCC_12864 3
#endif
template <class G>
constexpr int CONV_OCC = !EMBNET_CONV_SPLIT ? 1 : (G::BM * G::BN >= 192 * 64 ? 2 : (G::BM * G::BN >= 128 * 64 ? EMBNET_OCC_12864 : 4));

// the three-product ("H") kernels hold two LDS planes and two-piece fragments: their own register budget (build-time knobs for A/B)
#ifndef EMBNET_OCC_H_128128
#define EMBNET_OCC_H_128128 2
#endif
#ifndef EMBNET_OCC_H_12864
#define EMBNET_OCC_H_12864 EMBNET_OCC_12864
#endif
template <class G>
constexpr int CONV_OCC_H = G::BM * G::BN >= 192 * 64 ? (G::BM == 128 && G::BN == 128 ? EMBNET_OCC_H_128128 : 2)
                                                      : (G::BM * G::BN >= 128 * 64 ? EMBNET_OCC_H_12864 : 4);

template <class G, class TA, class TB, class LA, class LB, bool H = false>
__device__ __forceinline__ void conv_mainloop(const LA& la, const LB& lb, int kt_begin, int kt_end, float* smem,
                                              f32x16 (&acc)[G::TM][G::TN], bool fair = false, bool zero_acc = true,
                                              float sa = 1.f, float sb = 1.f) {
#if EMBNET_CONV_SPLIT
  gemm_mainloop3<G, TA, TB, LA, LB, H>(la, lb, kt_begin, kt_end, reinterpret_cast<unsigned char*>(smem), acc, fair, zero_acc, sa, sb);
#else
  gemm_mainloop<G, TA, TB>(la, lb, kt_begin, kt_end, smem, acc, fair, zero_acc);
#endif
}

// ---- three products per fp32 product for the gather kernels ("H" instantiations) -------------------------
// An operand RANGE SLOT is one uint32 in device memory holding the bit pattern of a float B >= max |element| of a tensor, left
// there by the tensor's producer: the exact maximum for kernels (embnet_range_multi, once per optimizer step) and for gradients
// (the BatchNorm backward passes, `dx_range` of embnet_bn_bwd_ex), an upper bound for activations (the BatchNorm forward,
// `y_range` of embnet_bn_train_fwd_ex / embnet_affine_act_planes_ex: nn_kernels.hip bn_finalize_kernel).  A conv launched with
// BOTH operands' slots (the *_ex entry points) splits each fp32 element, on the fly, into the two fp16 pieces of x * s — s the
// power of two that puts B into [2^14, 2^15) — keeps three of the four piece products and multiplies the sums by 1 / s and
// 1 / s' (exact) in front of its epilogue: the planes kernels' two-piece format (gemm_engine.h, PRECISION).  A conv that lacks
// either slot runs the six-term bf16 kernels: there is no default scale.
struct Ranges { const uint32_t* a; const uint32_t* b; };
__device__ __forceinline__ float2 range_scale(const uint32_t* slot) {
  return scale_pair(scale_exponent_of(__uint_as_float(__builtin_nontemporal_load(slot))));
}
template <class G>
__device__ __forceinline__ void scale_acc(f32x16 (&acc)[G::TM][G::TN], float fa, float fb) {   // (one factor after the other)
#pragma unroll
  for (int i = 0; i < G::TM; ++i)
#pragma unroll
    for (int j = 0; j < G::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = (acc[i][j][r] * fb) * fa;
}

// Remainder split ("tail"): workgroups finish in waves of 256 (one per CU), so `tiles mod 256` left-over
// tiles would keep a few CUs busy for a whole extra tile time (784 tiles = 3.06 per CU ran 22 % longer
// than 768).  The first n_full tiles are computed whole; each left-over tile is cut along K into `parts`
// equal pieces computed by separate workgroups (dispatched last, so they fill the slots the full tiles
// leave free) into raw fp32 partial tiles in the workspace, and tail_fixup_kernel sums the pieces in a
// fixed order and applies the epilogue.  Bitwise reproducible; no atomics.
struct SplitTail { int n_full, parts, kt_part; float* ws; };

struct ConvFwdParams { const float* x; const float* w; const float* bias; float* y; ConvGeom g; int relu; SplitTail tail; const float* residual; InputTransform tf; float* stats; int stats_rows; int fair_from; Ranges rg; };

// Decode blockIdx -> (tile, k range, partial destination).  Full tiles keep the XCD-aware order.
template <class G>
__device__ __forceinline__ void tail_decode(const SplitTail& t, int bid, int kt_total, int& tile, int& k0, int& k1,
                                            float*& part) {
  if (bid < t.n_full) { tile = xcd_remap(bid, t.n_full); k0 = 0; k1 = kt_total; part = nullptr; return; }
  const int j = bid - t.n_full;
  tile = t.n_full + j / t.parts;
  k0 = (j % t.parts) * t.kt_part; k1 = min(kt_total, k0 + t.kt_part);
  part = t.ws + (long)j * (G::BM * G::BN);
}

// raw accumulator tile -> part[BM][BN] (whole tile: rows/cols past the matrix edge hold zeros)
template <class G>
__device__ __forceinline__ void store_partial(const f32x16 (&acc)[G::TM][G::TN], float* smem, float* part) {
  for_each_acc_row4<G>(acc, smem, [&](int r, int c, float4 v) {
    *reinterpret_cast<float4*>(part + r * G::BN + c) = v;
  });
}

// out[(m0+r)*ld + n0+c] = act(sum_parts partial + bias) (+ residual) for the left-over tiles; one float4 per
// thread.  A workgroup covers one band of `wtm` rows (the row band a wave owns in the main kernel) x 1024/wtm
// columns of a tile, so that it can also emit that band's per-channel sum / sum of squares (stats != NULL)
// in the same [2][cols][tile_m * bm/wtm + band] layout the main kernel's epilogue writes.
// BatchNorm-backward sums riding on a data gradient.  The conv's INPUT was act(BN(x)): the BatchNormalization's backward needs
// dbeta = sum dz and dgamma = sum dz * xhat over the pixels, dz = d(conv input) * act'(BN(x)) — a pass over two tensors
// (bn_bwd_reduce4, 8 bytes per element).  The data-gradient epilogue holds d(conv input) in registers, so it reads x (4 bytes
// per element) and emits the partial sums per row band in the forward statistics' layout [2][C][rows]; the BatchNormalization
// backward then starts at its finalize kernel (embnet_bn_bwd_partials).  x == NULL: off.
// (struct BnSums: conv_geom.h)

__device__ __forceinline__ void bn_sums_add(int act, float4 v, float4 xq, float4 sc, float4 sh, float4 mu, float4 rs,
                                            float4& s1, float4& s2, float4& s3) {      // the arithmetic of bn_bwd_reduce4_kernel
  float4 dz = v;
  if (act) {
    dz.x = act_grad(act, fmaf(xq.x, sc.x, sh.x), v.x); dz.y = act_grad(act, fmaf(xq.y, sc.y, sh.y), v.y);
    dz.z = act_grad(act, fmaf(xq.z, sc.z, sh.z), v.z); dz.w = act_grad(act, fmaf(xq.w, sc.w, sh.w), v.w);
  }
  s3.x = fmaxf(s3.x, fabsf(dz.x)); s3.y = fmaxf(s3.y, fabsf(dz.y)); s3.z = fmaxf(s3.z, fabsf(dz.z)); s3.w = fmaxf(s3.w, fabsf(dz.w));
  s1.x += dz.x; s1.y += dz.y; s1.z += dz.z; s1.w += dz.w;
  s2.x = fmaf(dz.x, (xq.x - mu.x) * rs.x, s2.x); s2.y = fmaf(dz.y, (xq.y - mu.y) * rs.y, s2.y);
  s2.z = fmaf(dz.z, (xq.z - mu.z) * rs.z, s2.z); s2.w = fmaf(dz.w, (xq.w - mu.w) * rs.w, s2.w);
}

__global__ __launch_bounds__(256) void tail_fixup_kernel(const float* __restrict__ ws, int parts, int bm, int bn, int wtm,
                                                         int n_full, int tiles_n, long m, int cols,
                                                         const float* __restrict__ bias, int relu,
                                                         const float* __restrict__ residual, float* __restrict__ out,
                                                         float* __restrict__ stats, int stats_rows, const BnSums bsum) {
  __shared__ float4 red[3][256];
  const int cpb = 1024 / wtm, qpb = cpb / 4;              // columns / column quads per workgroup
  const int blocks_per_tile = bm * bn / 1024, col_groups = bn / cpb;
  const int t = blockIdx.x / blocks_per_tile, bi = blockIdx.x % blocks_per_tile;
  const int band = bi / col_groups, cg = bi % col_groups;
  const int rr = threadIdx.x / qpb, r = band * wtm + rr, c = cg * cpb + (threadIdx.x % qpb) * 4;
  const int tile = n_full + t;
  const long row = (long)(tile / tiles_n) * bm + r;
  const int col = (tile % tiles_n) * bn + c;
  const bool in = row < m && col < cols;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (in) {
    const float* src = ws + (long)t * parts * bm * bn + r * bn + c;
    a = *reinterpret_cast<const float4*>(src);
    for (int k = 1; k < parts; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(src + (long)k * bm * bn);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + col); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    if (relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
    if (residual) { const float4 q = *reinterpret_cast<const float4*>(residual + row * cols + col); a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w; }
    *reinterpret_cast<float4*>(out + row * cols + col) = a;
  }
  if (bsum.x) {                                            // a data gradient's fix-up: the BatchNorm-backward sums of its tile rows
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1, s3 = s1;
    if (in)
      bn_sums_add(bsum.act, a, *reinterpret_cast<const float4*>(bsum.x + row * cols + col),
                  *reinterpret_cast<const float4*>(bsum.scale + col), *reinterpret_cast<const float4*>(bsum.shift + col),
                  *reinterpret_cast<const float4*>(bsum.mean + col), *reinterpret_cast<const float4*>(bsum.rstd + col), s1, s2, s3);
    red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2; red[2][threadIdx.x] = s3;
    stats = bsum.partial; stats_rows = bsum.rows;
  } else {
    if (!stats) return;
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = make_float4(a.x * a.x, a.y * a.y, a.z * a.z, a.w * a.w);
    red[2][threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  for (int s = wtm / 2; s >= 1; s >>= 1) {                // fixed tree over the band's rows
    if (rr < s) {
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const float4 o = red[w][threadIdx.x + s * qpb];
        float4& d = red[w][threadIdx.x];
        d.x += o.x; d.y += o.y; d.z += o.z; d.w += o.w;
      }
      const float4 o = red[2][threadIdx.x + s * qpb];
      float4& d = red[2][threadIdx.x];
      d.x = fmaxf(d.x, o.x); d.y = fmaxf(d.y, o.y); d.z = fmaxf(d.z, o.z); d.w = fmaxf(d.w, o.w);
    }
    __syncthreads();
  }
  if (rr == 0 && col < cols) {
    const long prow = (long)(tile / tiles_n) * (bm / wtm) + band, P = stats_rows;
    const float4 s1 = red[0][threadIdx.x], s2 = red[1][threadIdx.x];
    float* d1 = stats + (long)col * P + prow;
    float* d2 = d1 + (long)cols * P;
    d1[0] = s1.x; d1[P] = s1.y; d1[2 * P] = s1.z; d1[3 * P] = s1.w;
    d2[0] = s2.x; d2[P] = s2.y; d2[2 * P] = s2.z; d2[3 * P] = s2.w;
    if (bsum.x && bsum.kinds == 3) {
      const float4 s3 = red[2][threadIdx.x];
      float* d3 = d2 + (long)cols * P;
      d3[0] = s3.x; d3[P] = s3.y; d3[2 * P] = s3.z; d3[3 * P] = s3.w;
    }
  }
}

template <class G, bool VEC, bool TF, bool H = false>
__device__ __forceinline__ void conv_fwd_body(const ConvFwdParams& p) {
  using TA = TileKC<G::BM>;
  using TB = TileKM<G::BN>;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS<G, TA, TB>];
  prio_hi();                                             // prologue at raised issue priority (gemm_engine.h)
  const int M = p.g.N * p.g.OH * p.g.OW, Kg = p.g.R * p.g.S * p.g.C;
  const int tiles_n = (p.g.K + G::BN - 1) / G::BN;
  int id, k0, k1; float* part;
  tail_decode<G>(p.tail, blockIdx.x, (Kg + BK - 1) / BK, id, k0, k1, part);
  const int m0 = (id / tiles_n) * G::BM, n0 = (id % tiles_n) * G::BN;
  stamp(0);
  LoadConvFwdA<G::BM, VEC, TF> la; la.init(p.x, p.g, m0, threadIdx.x, p.tf);
  LoadRowsKM<G::BN, VEC> lb; lb.init(p.w, p.g.K, p.g.K, Kg, n0, threadIdx.x);
  f32x16 acc[G::TM][G::TN];
  stamp(2);
  if constexpr (H) {                                     // x = operand a, the kernel = operand b
    const float2 sa = range_scale(p.rg.a), sb = range_scale(p.rg.b);
    conv_mainloop<G, TA, TB, decltype(la), decltype(lb), true>(la, lb, k0, k1, smem, acc, (int)blockIdx.x >= p.fair_from, true, sa.x, sb.x);
    scale_acc<G>(acc, sa.y, sb.y);
  } else {
    conv_mainloop<G, TA, TB>(la, lb, k0, k1, smem, acc, (int)blockIdx.x >= p.fair_from);
  }
  stamp(4);
  if (part) { store_partial<G>(acc, smem, part); stamp(5); return; }
  if (VEC) {                                             // K % 4 == 0: 16-byte row stores
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;  // this lane's column quad: sum, sum of squares over its rows
    // what the epilogue adds, fetched for all of the lane's rows before the staging loop (gemm_engine.h, EpiIdx)
    constexpr int NJ = EpiIdx<G>::NJ;
    const int ecol = n0 + epi_col<G>();
    const bool cok = ecol < p.g.K;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && cok) bv = *reinterpret_cast<const float4*>(p.bias + ecol);
    float4 res[G::TM][NJ];
    if (p.residual) {                                      // the Add layer behind the conv (residual units)
#pragma unroll
      for (int im = 0; im < G::TM; ++im)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          res[im][j] = *reinterpret_cast<const float4*>(p.residual + (long)min(m0 + epi_row<G>(im, j), M - 1) * p.g.K + (cok ? ecol : 0));
#pragma unroll
      for (int im = 0; im < G::TM; ++im)
#pragma unroll
        for (int j = 0; j < NJ; ++j) settle(res[im][j]);
    }
    settle(bv);
    auto emit = [&](int im, int j, int row, int col, float4 v) {
      if (p.bias) { v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
      if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (p.residual) { const float4 q = res[im][j]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
      *reinterpret_cast<float4*>(p.y + (long)row * p.g.K + col) = v;
      s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
      s2.x = fmaf(v.x, v.x, s2.x); s2.y = fmaf(v.y, v.y, s2.y); s2.z = fmaf(v.z, v.z, s2.z); s2.w = fmaf(v.w, v.w, s2.w);
    };
    if (m0 + G::BM <= M && n0 + G::BN <= p.g.K) {          // interior tile (wave-uniform): straight-line stores, no edge tests
      for_each_acc_row4_idx<G>(acc, smem, [&](int im, int j, int r, int c, float4 v) { emit(im, j, m0 + r, n0 + c, v); });
    } else {
      for_each_acc_row4_idx<G>(acc, smem, [&](int im, int j, int r, int c, float4 v) {
        const int row = m0 + r, col = n0 + c;
        if (row < M && col < p.g.K) emit(im, j, row, col, v);
      });
    }
    if (p.stats) {
      // BatchNorm statistics of the layer that follows, while the tile is in registers: every wave writes the
      // column sums of its WTM-row band as partial (tile_m*WAVES_M + wave_m) of stats[2][K][P]; the BN
      // finalize kernel adds the P rows per channel in double.  (lanes l, l+LPR, l+2*LPR.. hold the same columns)
      constexpr int LPR = G::WTN / 4;
      s1 = colquad_sum<LPR>(s1); s2 = colquad_sum<LPR>(s2);
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      const int col = n0 + (wave % G::WAVES_N) * G::WTN + lane * 4;
      if (lane < LPR && col < p.g.K) {
        const long prow = (long)(m0 / G::BM) * G::WAVES_M + wave / G::WAVES_N, P = p.stats_rows;
        float* d1 = p.stats + (long)col * P + prow;          // [2][K][P]: a channel's partials are contiguous for
        float* d2 = d1 + (long)p.g.K * P;                    // the finalize kernel (it reads P floats per channel)
        d1[0] = s1.x; d1[P] = s1.y; d1[2 * P] = s1.z; d1[3 * P] = s1.w;
        d2[0] = s2.x; d2[P] = s2.y; d2[2 * P] = s2.z; d2[3 * P] = s2.w;
      }
    }
    stamp(5);
    return;
  }
  for_each_acc<G>(acc, [&](int r, int c, float v) {
    const int row = m0 + r, col = n0 + c;
    if (row < M && col < p.g.K) {
      if (p.bias) v += p.bias[col];
      if (p.relu) v = fmaxf(v, 0.f);
      if (p.residual) v += p.residual[(long)row * p.g.K + col];
      p.y[(long)row * p.g.K + col] = v;
    }
  });
}

template <class G, bool VEC>
__global__ __launch_bounds__(256, VEC ? CONV_OCC<G> : 1) void conv_fwd_kernel(ConvFwdParams p) { conv_fwd_body<G, VEC, false>(p); }
// same, reading act(x*in_scale + in_shift) (InputTransform); its own symbol so the plain kernel keeps its code
template <class G, bool VEC>
__global__ __launch_bounds__(256, CONV_OCC<G>) void conv_fwd_tf_kernel(ConvFwdParams p) { conv_fwd_body<G, true, true>(p); }
// the three-product form (16-byte loads only)
template <class G>
__global__ __launch_bounds__(256, CONV_OCC_H<G>) void conv_fwd_h_kernel(ConvFwdParams p) { conv_fwd_body<G, true, false, true>(p); }

struct ConvDgradParams { const float* dy; const float* w; float* dx; ConvGeom g; DgradClass cls[MAX_CLASSES]; SplitTail tail; int accumulate; const float* add_src; int fair_from; BnSums bn; Ranges rg; };

template <class G, bool VEC, bool H = false>
__device__ __forceinline__ void conv_dgrad_body(const ConvDgradParams& p) {
  using TA = TileKC<G::BM>;
  using TB = TileKC<G::BN>;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS<G, TA, TB>];
  prio_hi();                                             // prologue at raised issue priority (gemm_engine.h)
  const DgradClass& cg = p.cls[blockIdx.y];
  const int M = p.g.N * cg.Hc * cg.Wc, Kg = cg.nR * cg.nS * p.g.K;
  const int tiles_n = (p.g.C + G::BN - 1) / G::BN;
  int id, k0, k1; float* part;                           // the tail split is only planned for stride 1 (one class)
  tail_decode<G>(p.tail, blockIdx.x, (Kg + BK - 1) / BK, id, k0, k1, part);
  const int m0 = (id / tiles_n) * G::BM, n0 = (id % tiles_n) * G::BN;
  if (m0 >= M) return;                                   // classes differ in size by a row/column
  if (Kg == 0 && p.accumulate && p.add_src == p.dx) return;   // a class no tap reaches adds nothing (1x1 stride-2: 3 of 4)
  stamp(0);
  LoadConvDgradA<G::BM, VEC> la; la.init(p.dy, p.g, cg, m0, threadIdx.x);
  LoadConvDgradB<G::BN, VEC> lb; lb.init(p.w, p.g, cg, n0, threadIdx.x);
  f32x16 acc[G::TM][G::TN];
  stamp(2);
  if constexpr (H) {                                     // dy = operand a, the kernel = operand b
    const float2 sa = range_scale(p.rg.a), sb = range_scale(p.rg.b);
    conv_mainloop<G, TA, TB, decltype(la), decltype(lb), true>(la, lb, k0, k1, smem, acc, (int)(blockIdx.y * gridDim.x + blockIdx.x) >= p.fair_from,
                                                               true, sa.x, sb.x);
    scale_acc<G>(acc, sa.y, sb.y);
  } else {
    conv_mainloop<G, TA, TB>(la, lb, k0, k1, smem, acc, (int)(blockIdx.y * gridDim.x + blockIdx.x) >= p.fair_from);
  }
  stamp(4);
  if (part) { store_partial<G>(acc, smem, part); stamp(5); return; }
  const int st = p.g.stride;
  if ((p.g.C & 3) == 0) {                                // 16-byte row stores
    // output addresses of the lane's rows and — the gradient through the tensor's other consumer (in dx or apart) — what is
    // added to them, fetched before the staging loop (gemm_engine.h, EpiIdx)
    constexpr int NJ = EpiIdx<G>::NJ;
    const int ecol = n0 + epi_col<G>();
    const bool cok = ecol < p.g.C;
    long base[G::TM][NJ];
    float4 res[G::TM][NJ];                                 // the other gradient (accumulate) or the BatchNorm input (bn): never both
    const bool bnon = p.bn.x != nullptr;
    const float* pre = bnon ? p.bn.x : p.add_src;
#pragma unroll
    for (int im = 0; im < G::TM; ++im)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        uint32_t n, rem, hc, wc;
        cg.dHWc.divmod((uint32_t)min(m0 + epi_row<G>(im, j), M - 1), n, rem); cg.dWc.divmod(rem, hc, wc);
        base[im][j] = (((long)n * p.g.H + cg.hoff + st * (int)hc) * p.g.W + cg.woff + st * (int)wc) * p.g.C;
        if (p.accumulate || bnon) res[im][j] = *reinterpret_cast<const float4*>(pre + base[im][j] + (cok ? ecol : 0));
      }
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 bsc = z4, bsh = z4, bmu = z4, brs = z4, s1 = z4, s2 = z4, s3 = z4;
    if (bnon && cok) {
      bsc = *reinterpret_cast<const float4*>(p.bn.scale + ecol); bsh = *reinterpret_cast<const float4*>(p.bn.shift + ecol);
      bmu = *reinterpret_cast<const float4*>(p.bn.mean + ecol); brs = *reinterpret_cast<const float4*>(p.bn.rstd + ecol);
    }
    if (p.accumulate || bnon) {
#pragma unroll
      for (int im = 0; im < G::TM; ++im)
#pragma unroll
        for (int j = 0; j < NJ; ++j) settle(res[im][j]);
      settle(bsc); settle(bsh); settle(bmu); settle(brs);
    }
    auto emit = [&](int im, int j, int col, float4 v) {
      if (p.accumulate) { const float4 o = res[im][j]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
      *reinterpret_cast<float4*>(p.dx + base[im][j] + col) = v;
      if (bnon) bn_sums_add(p.bn.act, v, res[im][j], bsc, bsh, bmu, brs, s1, s2, s3);
    };
    if (m0 + G::BM <= M && n0 + G::BN <= p.g.C) {          // interior tile (wave-uniform): straight-line stores, no edge tests
      for_each_acc_row4_idx<G>(acc, smem, [&](int im, int j, int, int c, float4 v) { emit(im, j, n0 + c, v); });
    } else {
      for_each_acc_row4_idx<G>(acc, smem, [&](int im, int j, int r, int c, float4 v) {
        if (m0 + r < M && n0 + c < p.g.C) emit(im, j, n0 + c, v);
      });
    }
    if (bnon) {                                            // partial (tile_m * WAVES_M + wave_m) of [2][C][rows], as the forward statistics
      constexpr int LPR = G::WTN / 4;
      s1 = colquad_sum<LPR>(s1); s2 = colquad_sum<LPR>(s2); s3 = colquad_max<LPR>(s3);
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      const int col = n0 + (wave % G::WAVES_N) * G::WTN + lane * 4;
      if (lane < LPR && col < p.g.C) {
        const long prow = (long)(m0 / G::BM) * G::WAVES_M + wave / G::WAVES_N, P = p.bn.rows;
        float* d1 = p.bn.partial + (long)col * P + prow;
        float* d2 = d1 + (long)p.g.C * P;
        d1[0] = s1.x; d1[P] = s1.y; d1[2 * P] = s1.z; d1[3 * P] = s1.w;
        d2[0] = s2.x; d2[P] = s2.y; d2[2 * P] = s2.z; d2[3 * P] = s2.w;
        if (p.bn.kinds == 3) {                               // max |dz| of the band: the bound of the BatchNorm backward's dx
          float* d3 = d2 + (long)p.g.C * P;
          d3[0] = s3.x; d3[P] = s3.y; d3[2 * P] = s3.z; d3[3 * P] = s3.w;
        }
      }
    }
    stamp(5);
    return;
  }
  int last_r = -1; long row_base = 0;
  for_each_acc<G>(acc, [&](int r, int c, float v) {
    const int row = m0 + r, col = n0 + c;
    if (r != last_r) {                                   // rows repeat across this lane's column tiles
      last_r = r;
      uint32_t n, rem, hc, wc;
      cg.dHWc.divmod((uint32_t)min(row, M - 1), n, rem); cg.dWc.divmod(rem, hc, wc);
      row_base = (((long)n * p.g.H + cg.hoff + st * (int)hc) * p.g.W + cg.woff + st * (int)wc) * p.g.C;
    }
    if (row < M && col < p.g.C) p.dx[row_base + col] = p.accumulate ? p.add_src[row_base + col] + v : v;
  });
}
template <class G, bool VEC>
__global__ __launch_bounds__(256, VEC ? CONV_OCC<G> : 1) void conv_dgrad_kernel(ConvDgradParams p) { conv_dgrad_body<G, VEC>(p); }
template <class G>
__global__ __launch_bounds__(256, CONV_OCC_H<G>) void conv_dgrad_h_kernel(ConvDgradParams p) { conv_dgrad_body<G, true, true>(p); }

struct ConvWgradParams { const float* x; const float* dy; float* out; ConvGeom g; int kt_per_split, splits, xcd_order; InputTransform tf; int fair_from; int stagger; Ranges rg; };

// The weight gradient on the 16x16x32 MFMA shape (gemm_engine.h, "K32"): same tiles, loaders, slabs and summation order per
// output element over k tiles; inside a k tile the 32 pixels are summed by one instruction per term instead of two.
template <class G, bool TF>
__device__ __forceinline__ void conv_wgrad_k32_body(const ConvWgradParams& p) {
  using TA = TileKM<G::BM>;
  using TB = TileKM<G::BN>;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS<G, TA, TB>];
  prio_hi();
  const int M = p.g.R * p.g.S * p.g.C, Kg = p.g.N * p.g.OH * p.g.OW;
  const int tiles_n = (p.g.K + G::BN - 1) / G::BN;
  const int tiles = ((M + G::BM - 1) / G::BM) * tiles_n;
  const int split = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  if (split >= p.splits) return;
  const int m0 = (tile / tiles_n) * G::BM, n0 = (tile % tiles_n) * G::BN;
  const int kt_total = (Kg + BK - 1) / BK;
  const int kt0 = split * p.kt_per_split, kt1 = min(kt0 + p.kt_per_split, kt_total);
  LoadConvWgradA<G::BM, true, TF> la; la.init(p.x, p.g, m0, threadIdx.x, p.tf);
  LoadRowsKM<G::BN, true> lb; lb.init(p.dy, p.g.K, p.g.K, Kg, n0, threadIdx.x);
  f32x4 acc[G::TM][G::TN][4];
  gemm_mainloop3_k32<G, TA, TB>(la, lb, kt0, kt1, reinterpret_cast<unsigned char*>(smem), acc, (int)blockIdx.x >= p.fair_from);
  float* out = p.out + (long)split * M * p.g.K;
  for_each_acc16_row4<G>(acc, smem, [&](int r, int c, float4 v) {
    const int row = m0 + r, col = n0 + c;
    if (row < M && col < p.g.K) *reinterpret_cast<float4*>(out + (long)row * p.g.K + col) = v;
  });
}
template <class G>
__global__ __launch_bounds__(256, CONV_OCC<G>) void conv_wgrad_k32_kernel(ConvWgradParams p) { conv_wgrad_k32_body<G, false>(p); }
// (the input-transform form: x is the INPUT of the BatchNormalization in front, act(x*scale + shift) applied in the loader)
template <class G>
__global__ __launch_bounds__(256) void conv_wgrad_k32_tf_kernel(ConvWgradParams p) { conv_wgrad_k32_body<G, true>(p); }

// VA: 16-byte gathers of X (C % 4 == 0); VB: 16-byte loads of dY (K % 4 == 0)
template <class G, bool VA, bool VB, bool TF, bool H = false>
__device__ __forceinline__ void conv_wgrad_body(const ConvWgradParams& p) {
  using TA = TileKM<G::BM>;
  using TB = TileKM<G::BN>;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS<G, TA, TB>];
  prio_hi();                                             // prologue at raised issue priority (gemm_engine.h)
  const int M = p.g.R * p.g.S * p.g.C, Kg = p.g.N * p.g.OH * p.g.OW;
  const int tiles_n = (p.g.K + G::BN - 1) / G::BN;
  // Two workgroup orders (speed only; results identical).  Default: tiles of one K-split on consecutive
  // ids, i.e. spread over the 8 XCDs.  xcd_order=1 (EMBNET_WGRAD_XCD=1) puts all tiles of a split on ONE
  // XCD so its L2 serves the shared X/dY pixels — HBM traffic drops ~4x on the 3x3 layers, but measured
  // 20-25 % SLOWER (A/B in one process, tools/kernel_bench.py): the tiles hit the same L2 lines at the
  // same instant; spreading them over eight L2s + the 256 MB Infinity Cache is faster.  Kept as a knob.
  const int tiles = ((M + G::BM - 1) / G::BM) * tiles_n;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int split = p.xcd_order ? (slot / tiles) * 8 + xcd : blockIdx.x / tiles;
  const int tile = p.xcd_order ? slot % tiles : blockIdx.x % tiles;
  if (split >= p.splits) return;
  const int m0 = (tile / tiles_n) * G::BM, n0 = (tile % tiles_n) * G::BN;
  const int kt_total = (Kg + BK - 1) / BK;
  const int kt0 = split * p.kt_per_split, kt1 = min(kt0 + p.kt_per_split, kt_total);
  stamp(0);
  LoadConvWgradA<G::BM, VA, TF> la; la.init(p.x, p.g, m0, threadIdx.x, p.tf);
  LoadRowsKM<G::BN, VB> lb; lb.init(p.dy, p.g.K, p.g.K, Kg, n0, threadIdx.x);
  f32x16 acc[G::TM][G::TN];
  stamp(2);
  // stagger > 0 (with xcd_order): the tiles of one split, co-resident on one XCD, start their K ranges `stagger`
  // tiles apart (wrapping around), so they want different L2 lines at any instant but the same ones within a few
  // tiles' time.  Two passes over the rotated range; the accumulators carry over.
  const int rot = p.stagger > 0 ? min((tile * p.stagger) % max(kt1 - kt0, 1), kt1 - kt0) : 0;
  if constexpr (H) {                                     // x = operand a, dy = operand b
    const float2 sa = range_scale(p.rg.a), sb = range_scale(p.rg.b);
    conv_mainloop<G, TA, TB, decltype(la), decltype(lb), true>(la, lb, kt0 + rot, kt1, smem, acc, (int)blockIdx.x >= p.fair_from, true, sa.x, sb.x);
    if (rot > 0) { prio_lo(); conv_mainloop<G, TA, TB, decltype(la), decltype(lb), true>(la, lb, kt0, kt0 + rot, smem, acc, false, false, sa.x, sb.x); }
    scale_acc<G>(acc, sa.y, sb.y);
  } else {
    conv_mainloop<G, TA, TB>(la, lb, kt0 + rot, kt1, smem, acc, (int)blockIdx.x >= p.fair_from);
    if (rot > 0) { prio_lo(); conv_mainloop<G, TA, TB>(la, lb, kt0, kt0 + rot, smem, acc, false, false); }
  }
  stamp(4);
  float* out = p.out + (long)split * M * p.g.K;
  if (VB) {                                              // K % 4 == 0
    for_each_acc_row4<G>(acc, smem, [&](int r, int c, float4 v) {
      const int row = m0 + r, col = n0 + c;
      if (row < M && col < p.g.K) *reinterpret_cast<float4*>(out + (long)row * p.g.K + col) = v;
    });
    stamp(5);
    return;
  }
  for_each_acc<G>(acc, [&](int r, int c, float v) {
    const int row = m0 + r, col = n0 + c;
    if (row < M && col < p.g.K) out[(long)row * p.g.K + col] = v;
  });
}

template <class G, bool VA, bool VB>
__global__ __launch_bounds__(256, (VA && VB) ? CONV_OCC<G> : 1) void conv_wgrad_kernel(ConvWgradParams p) { conv_wgrad_body<G, VA, VB, false>(p); }
template <class G, bool VA, bool VB>
__global__ __launch_bounds__(256, CONV_OCC<G>) void conv_wgrad_tf_kernel(ConvWgradParams p) { conv_wgrad_body<G, true, true, true>(p); }
template <class G>
__global__ __launch_bounds__(256, CONV_OCC_H<G>) void conv_wgrad_h_kernel(ConvWgradParams p) { conv_wgrad_body<G, true, true, false, true>(p); }

// out[i] = sum_s slabs[s][i], fixed order.  A workgroup owns 32 float4 columns (512 contiguous bytes of
// every slab) and spreads the slabs over 8 thread groups (slab s goes to group s % 8, each group keeping
// two independent accumulators), then combines the groups through LDS in a fixed tree — so a [576x64]
// gradient with 153 slabs still has ~300 workgroups x 8 x 2 sixteen-byte loads in flight instead of one
// column walk per thread.  bytes = splits * n * 4 read once (L2 / Infinity-Cache resident: the slabs were
// just written).  Bitwise reproducible.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, int splits, long n,
                                                          float* __restrict__ out) {
  if (n & 3) {                       // slabs not 16-byte aligned to each other: scalar path
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
      float s = 0.f;
      for (int k = 0; k < splits; ++k) s += slabs[(long)k * n + i];
      out[i] = s;
    }
    return;
  }
  __shared__ float4 part[8][32];
  const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const long n4 = n >> 2, i4 = (long)blockIdx.x * 32 + col;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  if (i4 < n4) {
    const float4* src = reinterpret_cast<const float4*>(slabs) + i4;
    int k = grp;
    for (; k + 8 < splits; k += 16) {
      const float4 u = src[(long)k * n4], v = src[(long)(k + 8) * n4];
      a0.x += u.x; a0.y += u.y; a0.z += u.z; a0.w += u.w;
      a1.x += v.x; a1.y += v.y; a1.z += v.z; a1.w += v.w;
    }
    if (k < splits) { const float4 u = src[(long)k * n4]; a0.x += u.x; a0.y += u.y; a0.z += u.z; a0.w += u.w; }
  }
  part[grp][col] = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
  __syncthreads();
  if (grp == 0 && i4 < n4) {
    float4 r[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) r[g] = part[g][col];
#define EMBNET_ADD4(p, q) make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w)
    const float4 s01 = EMBNET_ADD4(r[0], r[1]), s23 = EMBNET_ADD4(r[2], r[3]), s45 = EMBNET_ADD4(r[4], r[5]),
                 s67 = EMBNET_ADD4(r[6], r[7]);
    const float4 lo = EMBNET_ADD4(s01, s23), hi = EMBNET_ADD4(s45, s67);
    reinterpret_cast<float4*>(out)[i4] = EMBNET_ADD4(lo, hi);
#undef EMBNET_ADD4
  }
}


// The same sum for MANY weight gradients in one launch (embnet_slab_reduce_multi): a training step's backward leaves the
// split-K slabs of every conv layer in buffers of their own and adds them up once, before the optimizer (or before a
// bucket's all-reduce), instead of with one ~8 us launch per layer (ResNet18: 21, simple2: 7).  chunks = (tensor index,
// block of 32 float4 columns) per workgroup; per element the summation order is slab_reduce_kernel's, so the result is
// bit-identical to the per-layer launches.
struct SlabTensor { const float* slabs; float* out; long n; int splits; int first_block; };
static_assert(sizeof(SlabTensor) == 32, "descriptor layout is part of the ABI (include/embnet.h)");
constexpr int SLAB_BATCH = 112;                      // descriptors per launch: they travel as the kernel argument (< 4 KiB), so
struct SlabBatch { SlabTensor t[SLAB_BATCH]; int n; int pad; };    // a captured HIP graph holds them by value — no table in memory
__global__ __launch_bounds__(256) void slab_reduce_multi_kernel(const SlabBatch) {
  // read the descriptors straight from the kernel-argument segment (uniform addresses: scalar loads); indexing the by-value
  // argument dynamically would make the compiler copy all of it to scratch first
  typedef const SlabBatch __attribute__((address_space(4)))* kargb;
  kargb pb = (kargb)__builtin_amdgcn_kernarg_segment_ptr();
  int lo = 0, hi = pb->n - 1;                        // the tensor whose block range holds this workgroup
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((int)blockIdx.x >= pb->t[mid].first_block) lo = mid; else hi = mid - 1; }
  SlabTensor t;
  t.slabs = pb->t[lo].slabs; t.out = pb->t[lo].out; t.n = pb->t[lo].n; t.splits = pb->t[lo].splits; t.first_block = pb->t[lo].first_block;
  const int blk = (int)blockIdx.x - t.first_block;
  const long n = t.n; const int splits = t.splits;
  if (n & 3) {                       // scalar path: this workgroup owns 128 consecutive elements
    const long i = (long)blk * 128 + (threadIdx.x & 127);
    if (threadIdx.x < 128 && i < n) {
      float s = 0.f;
      for (int k = 0; k < splits; ++k) s += t.slabs[(long)k * n + i];
      t.out[i] = s;
    }
    return;
  }
  __shared__ float4 part[8][32];
  const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const long n4 = n >> 2, i4 = (long)blk * 32 + col;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  if (i4 < n4) {
    const float4* src = reinterpret_cast<const float4*>(t.slabs) + i4;
    int k = grp;
    for (; k + 8 < splits; k += 16) {
      const float4 u = src[(long)k * n4], v = src[(long)(k + 8) * n4];
      a0.x += u.x; a0.y += u.y; a0.z += u.z; a0.w += u.w;
      a1.x += v.x; a1.y += v.y; a1.z += v.z; a1.w += v.w;
    }
    if (k < splits) { const float4 u = src[(long)k * n4]; a0.x += u.x; a0.y += u.y; a0.z += u.z; a0.w += u.w; }
  }
  part[grp][col] = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
  __syncthreads();
  if (grp == 0 && i4 < n4) {
    float4 r[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) r[g] = part[g][col];
#define EMBNET_ADD4(p, q) make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w)
    const float4 s01 = EMBNET_ADD4(r[0], r[1]), s23 = EMBNET_ADD4(r[2], r[3]), s45 = EMBNET_ADD4(r[4], r[5]),
                 s67 = EMBNET_ADD4(r[6], r[7]);
    const float4 lo = EMBNET_ADD4(s01, s23), hi = EMBNET_ADD4(s45, s67);
    reinterpret_cast<float4*>(t.out)[i4] = EMBNET_ADD4(lo, hi);
#undef EMBNET_ADD4
  }
}

}  // namespace embnet

using namespace embnet;

// host helper for conv_patch.hip: the fix-up pass over `rem` left-over tiles cut into `parts` partial tiles (SplitTail)
namespace embnet {
void launch_tail_fixup(const float* ws, int parts, int bm, int bn, int wtm, int n_full, int rem, int tiles_n, long m, int cols,
                       const float* bias, int relu, const float* residual, float* out, float* stats, int stats_rows,
                       const BnSums& bsum, hipStream_t st) {
  EMBNET_TRACE("embnet::tail_fixup_kernel", TRACE_BYTES, 4.0 * rem * bm * bn * (parts + 1 + (residual ? 1 : 0) + (bsum.x ? 1 : 0)), st);
  tail_fixup_kernel<<<rem * (bm * bn / 1024), 256, 0, st>>>(ws, parts, bm, bn, wtm, n_full, tiles_n, m, cols, bias, relu,
                                                           residual, out, stats, stats_rows, bsum);
}
// host helper for conv_wgrad_planes.hip: the fixed-order sum of split-K slabs into the gradient
void launch_slab_reduce(const float* slabs, int splits, long n, float* out, hipStream_t st) {
  EMBNET_TRACE("embnet::slab_reduce_kernel", TRACE_BYTES, 4.0 * n * (splits + 1), st);
  slab_reduce_kernel<<<(n & 3) ? cdiv(n, 256) : cdiv(n / 4, 32), 256, 0, st>>>(slabs, splits, n, out);
}
}  // namespace embnet

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

#if EMBNET_STAMPS
// diagnostic build only (not in include/embnet.h): point the conv kernels' stamp() at a device buffer of
// 8 x uint64 per workgroup (NULL = off)
extern "C" int embnet_debug_set_stamps(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(embnet::g_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

using G128x128 = Geom<128, 128, 2, 2>;
using G128x64 = Geom<128, 64, 2, 2>;
using G128x32 = Geom<128, 32, 4, 1>;
using G64x64 = Geom<64, 64, 2, 2>;
using G192x64 = Geom<192, 64, 2, 2>;        // wgrad only: 3x3xC64 kernels have 576 = 3*192 rows (4.5 tiles of 128)
// three-product wgrad only (EMBNET_WGRAD_256=1, measured slower, off): a gradient of 129 .. 256 rows x <= 64 filters as ONE row tile —
// the ResNet stem's 7x7x4 x 64 = 196 rows takes two 128-row tiles, each streaming all of dy: 1 136 MB per launch against 514 MB
// algorithmic (profiles/r05_pmc_traffic_c2.txt)
using G256x64 = Geom<256, 64, 4, 1>;


// pick the widest N tile that the channel count fills, shrink M tile when the grid would not cover the chip
static int pick_tile(long m, int ncols, bool strided_dgrad = false, long kdepth = 0) {
  const int forced = (int)env_long("EMBNET_CONV_TILE", -1);    // tuning aid (tools/exp/tile_sweep.py), read per call
  if (forced >= 0) return forced;
  // Measured on ResNet18 shapes (tools/kernel_bench.py, EMBNET_CONV_TILE sweep): workgroups all take the
  // same time, so what matters is how evenly the grid fills the 256 CUs x (3..6 resident workgroups);
  // 128x64 wins once it gives >= 4 workgroups per CU, 64x64 below that, 128x128 only for many rounds.
  if (ncols <= 32) return 2;
#if EMBNET_CONV_SPLIT
  // Split arithmetic (tools/exp/tile_sweep.py, profiles/r02_tile_sweep_split.txt): the split VALU work and the loads
  // per MFMA fall with the tile size, two 128x128 workgroups fit a CU, and the matrix pipe is 2.7x faster, so the
  // big tile wins as soon as it gives every CU 1.5 tiles (28x28x128 -> 256: 95 us vs 103 / 114 for 128x64 / 64x64).
  // In the training step (bench.py, EMBNET_BENCH_DETAIL, per layer): 14x14x256 -> 256 195 us vs 221 forward and
  // 198 vs 235 data gradient, 28x28x128 s2 -> 256 105 vs 117; 128-wide layers are neutral, and the stride classes of
  // a strided data gradient (unequal K per class) lose with the big tile (162 vs 114 us) -> they keep the old rule.
  // short reductions (EfficientNet's 1x1 convs at 14x14 / 7x7: 3-6 K tiles): the loop is a few gather round trips whatever the
  // tile; EMBNET_CONV_SHORTK = 1 / 3 picks 128x64 / 64x64 tiles for them (more workgroups per CU to hide the trips) — experiment
  static const int shortk = (int)env_long("EMBNET_CONV_SHORTK", 0);
  static const long shortk_max = env_long("EMBNET_CONV_SHORTK_MAX", 8);
  if (shortk && kdepth > 0 && kdepth <= shortk_max * BK && !strided_dgrad && ncols >= 64) return shortk;
  static const long t128_min = env_long("EMBNET_CONV_T128_MIN", 384);
  const long t128 = cdiv(m, 128) * cdiv(ncols, 128);
  if ((ncols >= 256 && !strided_dgrad && t128 >= t128_min) || (ncols >= 128 && t128 >= 4 * 768)) return 0;
#else
  if (cdiv(m, 128) * cdiv(ncols, 128) >= 4 * 768 && ncols >= 128) return 0;
#endif
  // long reductions (7x7 / 10x10 kernels of the small backbones: 100-200 K tiles) amortise the bigger tile's operand traffic
  // over more MFMAs, so 128x64 pays from fewer tiles: `simple`@105 conv2 (882 tiles, K = 3136 / 6272) 1.648 -> 1.587 ms/step
  const long need = (kdepth >= 96 * BK && !strided_dgrad) ? 512 : 1024;
  // a launch of fewer than 512 small tiles gets its K range cut over workgroups anyway (plan_tail): with >= 32 K tiles to cut,
  // half as many 128x64 tiles, each cut finer, beat 64x64 tiles (`simple`@105 conv3 forward 61 -> 41 us, data gradient 67 -> 53)
  if (!strided_dgrad && kdepth >= 32 * BK && cdiv(m, 64) * cdiv(ncols, 64) < 512) return 1;
  return (cdiv(m, 128) * cdiv(ncols, 64) >= need) ? 1 : 3;
}
static const int TILE_BM[6] = {128, 128, 128, 64, 192, 256}, TILE_BN[6] = {128, 64, 32, 64, 64, 64}, TILE_WTM[6] = {64, 64, 32, 32, 96, 64};
// workgroups of each tile type a CU holds at once (registers / LDS; measured with in-kernel stamps)
#if EMBNET_CONV_SPLIT
static const int TILE_RESIDENT[6] = {2, 3, 4, 4, 2, 2};
#else
static const int TILE_RESIDENT[6] = {3, 4, 5, 7, 3, 3};
#endif
// Progress-ordered priority (gemm_engine.h: `fair`) for launches whose workgroups are all resident at once.  Measured
// A/B, one process (tools/exp/ab_conv.py, ResNet18 layers at batch 128): 64x64-tile forward / data-gradient launches
// that fit one round +1..+3 % (7 waves per SIMD end together instead of one after the other), split-K weight-gradient
// launches +1 % in sum; 128x64 tiles (4 per CU) 7 % SLOWER when forced into lock-step, and any multi-round launch
// slower (the last round's high-priority newcomers starve the round before).  Returns the first "fair" workgroup id.
#ifndef EMBNET_FAIR_DEFAULT
#define EMBNET_FAIR_DEFAULT 1
#endif
static int fair_from(long grid, int tile, bool wgrad) {
  static const int knob = (int)env_long("EMBNET_FAIR_SINGLE_ROUND", EMBNET_FAIR_DEFAULT);
  const bool single_round = grid <= 256L * TILE_RESIDENT[tile];
  return (knob && single_round && (wgrad || tile == 3)) ? 0 : 0x7fffffff;
}

// Plan the remainder split for `tiles` output tiles of bm x bn with kt K-tiles each (see SplitTail).
// Model: a CU retires one tile per t_tile; whole tiles cost ceil(tiles/256) of those, the split costs
// floor(tiles/256) + ceil(rem*parts/256)/parts plus the fix-up pass over rem*parts partial tiles.
static void plan_tail(long tiles, int kt, int bm, int bn, size_t ws_bytes, SplitTail& t, size_t* need = nullptr) {
  t.n_full = (int)tiles; t.parts = 1; t.kt_part = kt; t.ws = nullptr;
  if (need) *need = 0;
  static const int knob = (int)env_long("EMBNET_CONV_TAIL", 1);
  static const int slots_knob = (int)env_long("EMBNET_TAIL_SLOTS", 512);   // granularity of a "round"; 0: 256 x resident(tile)
  int NUM_CUS = slots_knob;
  if (slots_knob == 0) {
    int res = 2;
    for (int t = 0; t < 5; ++t) if (TILE_BM[t] == bm && TILE_BN[t] == bn) res = TILE_RESIDENT[t];
    NUM_CUS = 256 * res;
  }
  const int rem = (int)(tiles % NUM_CUS);
  if (knob && tiles < NUM_CUS) {
    // A launch that does not even fill one round (the small backbones' layers at batch 32, the 7x7 maps): every CU runs at
    // most a workgroup or two, and the launch lasts as long as ONE workgroup's K chain — kt tiles at the gather's round trip
    // (~1.5 us) each, with nothing to overlap it with.  Cut every tile's K range into parts (more, shorter chains side by
    // side; raw partial tiles + the fix-up launch).  EMBNET_SMALL_SPLIT=0: off (A/B).
    static const int small = (int)env_long("EMBNET_SMALL_SPLIT", 1);
    static const int small_wgs = (int)env_long("EMBNET_SMALL_SPLIT_WGS", 512);
    if (!small || kt < 6 || bm * bn > 128 * 64) return;
    int parts = (int)(small_wgs / tiles);
    if (parts > kt / 3) parts = kt / 3;
    if (parts > 32) parts = 32;
    if (parts < 2) return;
    const int kt_part = cdiv(kt, parts);
    parts = cdiv(kt, kt_part);
    if (parts < 2) return;
    const size_t bytes = (size_t)tiles * parts * bm * bn * 4;
    if (need) *need = bytes;
    if (bytes > ws_bytes) return;
    t.n_full = 0; t.parts = parts; t.kt_part = kt_part;
    return;
  }
  if (!knob || rem == 0 || tiles < NUM_CUS) return;
  const double t_tile = (double)kt * bm * bn * BK * 2.0 / (0.85 * 146e12 / NUM_CUS);        // seconds
  const double tile_bytes = (double)bm * bn * 4;
  double best = 1.0 * t_tile; int best_parts = 1;                                         // the left-over round, whole tiles
  for (int parts = 2; parts <= 32 && parts <= kt / 2; ++parts) {
    const int kt_part = cdiv(kt, parts), real = cdiv(kt, kt_part);
    if (real != parts) continue;
    const double cost = (double)cdiv((long)rem * parts, NUM_CUS) / parts * t_tile + 4e-6 +
                        2.0 * rem * parts * tile_bytes / 3e12 + rem * tile_bytes / 3e12;
    if (cost < best * 0.97) { best = cost; best_parts = parts; }
  }
  if (best_parts == 1) return;
  const size_t bytes = (size_t)rem * best_parts * bm * bn * 4;
  if (need) *need = bytes;
  if (bytes > ws_bytes) return;                          // no (or too small a) workspace: whole tiles
  t.n_full = (int)(tiles - rem); t.parts = best_parts; t.kt_part = cdiv(kt, best_parts);
}

// kernel name as rocprofv3 prints it, for the trace log (and embnet_conv2d_kernel_name)
static const char* GEOM_NAME[6] = {"128, 128, 2, 2", "128, 64, 2, 2", "128, 32, 4, 1", "64, 64, 2, 2", "192, 64, 2, 2", "256, 64, 4, 1"};
static const char* conv_kernel_name(const char* kernel, const char* params, int tile, const char* flags) {
  static thread_local char buf[160];
  snprintf(buf, sizeof buf, "void embnet::%s<embnet::Geom<%s>, %s>(embnet::%s)", kernel, GEOM_NAME[tile], flags, params);
  return buf;
}

#define LAUNCH_TILED(KERNEL, VECARGS, tile, grid, st, p)                              \
  switch (tile) {                                                                     \
    case 0: KERNEL<G128x128, VECARGS><<<grid, 256, 0, st>>>(p); break;                \
    case 1: KERNEL<G128x64, VECARGS><<<grid, 256, 0, st>>>(p); break;                 \
    case 2: KERNEL<G128x32, VECARGS><<<grid, 256, 0, st>>>(p); break;                 \
    default: KERNEL<G64x64, VECARGS><<<grid, 256, 0, st>>>(p); break;                 \
  }

#define LAUNCH_TILED_H(KERNEL, tile, grid, st, p)                                     \
  switch (tile) {                                                                     \
    case 0: KERNEL<G128x128><<<grid, 256, 0, st>>>(p); break;                         \
    case 1: KERNEL<G128x64><<<grid, 256, 0, st>>>(p); break;                          \
    case 2: KERNEL<G128x32><<<grid, 256, 0, st>>>(p); break;                          \
    default: KERNEL<G64x64><<<grid, 256, 0, st>>>(p); break;                          \
  }
#define LAUNCH_WGRAD_H(KERNEL, tile, grid, st, p)          /* + the 192-row tile only the weight gradient plans */ \
  switch (tile) {                                                                     \
    case 4: KERNEL<G192x64><<<grid, 256, 0, st>>>(p); break;                          \
    case 5: KERNEL<G256x64><<<grid, 256, 0, st>>>(p); break;                          \
    default: LAUNCH_TILED_H(KERNEL, tile, grid, st, p)                                \
  }
static const char* conv_h_kernel_name(const char* kernel, const char* params, int tile) {
  static thread_local char buf[160];
  snprintf(buf, sizeof buf, "void embnet::%s<embnet::Geom<%s> >(embnet::%s)", kernel, GEOM_NAME[tile], params);
  return buf;
}

// DEPRECATED (ABI 20 form, kept for one round): operand ranges of the NEXT embnet_conv2d_{fwd, dgrad, dgrad_bnsums, wgrad,
// wgrad_slabs}_f32 call of the calling thread.  New code passes the slots as ARGUMENTS of the *_ex entry points (ABI 21): a
// per-thread "applies to the next call" request is hidden state — a binding in another language, a second stream on one thread or
// an exception between the two calls gets the wrong arithmetic silently.  The shim clears itself
// on every conv entry, whatever that call does; a NULL operand now means UNKNOWN (six-term kernels), no longer "scale 1".
static thread_local Ranges t_ranges{nullptr, nullptr};
extern "C" int embnet_conv2d_ranges(const uint32_t* a, const uint32_t* b) {
  EMBNET_CHECK_ARG(!((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 3), "conv2d_ranges: slots are 4-byte aligned");
  t_ranges = Ranges{a, b};
  return 0;
}
static Ranges take_ranges() {
  const Ranges r = t_ranges;
  t_ranges = Ranges{nullptr, nullptr};
  return r;
}
static bool ranges_ok(const Ranges& r) {
  return EMBNET_CONV_SPLIT && r.a && r.b && !((reinterpret_cast<uintptr_t>(r.a) | reinterpret_cast<uintptr_t>(r.b)) & 3);
}

// Ranges of many tensors in two launches (the kernels of a model's gather convs, once per optimizer step): table rows
// (tensor, elements, slot), chunks = (row, chunk of 4096 elements) pairs — the layout of embnet_conv_weight_planes' lists.
struct RangeTensor { const float* x; long n; uint32_t* slot; };
static_assert(sizeof(RangeTensor) == 24, "descriptor layout is part of the ABI (include/embnet.h)");
__global__ __launch_bounds__(256) void range_zero_kernel(const RangeTensor* __restrict__ table, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) *table[i].slot = 0u;
}
__global__ __launch_bounds__(256) void range_multi_kernel(const RangeTensor* __restrict__ table, const int* __restrict__ chunks) {
  const RangeTensor t = table[chunks[2 * blockIdx.x]];
  const long e0 = (long)chunks[2 * blockIdx.x + 1] * 4096;
  float m = 0.f;
#pragma unroll 4
  for (int u = 0; u < 16; ++u) {
    const long e = e0 + u * 256 + threadIdx.x;
    m = fmaxf(m, e < t.n ? fabsf(t.x[e]) : 0.f);
  }
  range_emit_block(t.slot, m);                           // (a kernel has a few hundred chunks at most: one word takes them)
}
extern "C" int embnet_range_chunk_elems(void) { return 4096; }
extern "C" int embnet_range_multi(const void* table, int n_tensors, const int32_t* chunks, int n_chunks, void* stream) {
  EMBNET_CHECK_ARG(table && chunks && n_tensors > 0 && n_chunks > 0, "range_multi: bad argument");
  hipStream_t st = (hipStream_t)stream;
  EMBNET_TRACE("embnet::range_multi_kernel", TRACE_BYTES, 4.0 * 4096 * n_chunks, stream);
  range_zero_kernel<<<cdiv(n_tensors, 256), 256, 0, st>>>((const RangeTensor*)table, n_tensors);
  range_multi_kernel<<<n_chunks, 256, 0, st>>>((const RangeTensor*)table, chunks);
  return check_launch("range_multi");
}

// Off by default: 1-6 % faster per layer back to back (profiles/r04_exp_wgrad_k32.txt), 2 % SLOWER inside the training step
// (same box, alternating runs: 123.9-124.9 vs 121.7-122.1 us for the 128x128 tile, step time equal) — see DESIGN 3.12.
static bool wgrad_k32() { static const bool v = env_long("EMBNET_WGRAD_K32", 0) != 0; return v; }

extern "C" int embnet_conv_mfma_terms(void) { return EMBNET_CONV_SPLIT ? 6 : 1; }
// ... of the kernels that read pre-split planes (conv_patch.hip, conv_wgrad_planes.hip): 3 in the two-piece fp16 format
extern "C" int embnet_conv_planes_mfma_terms(void) { return planes_f16() ? 3 : 6; }

extern "C" size_t embnet_conv2d_fwd_workspace_bytes(int n, int c, int r, int s, int k, int oh, int ow) {
  if (n <= 0 || c <= 0 || r <= 0 || s <= 0 || k <= 0 || oh <= 0 || ow <= 0 || ((c | k) & 3)) return 0;
  const long M = (long)n * oh * ow;
  const int tile = pick_tile(M, k, false, (long)r * s * c);
  SplitTail t; size_t need;
  plan_tail((long)cdiv(M, TILE_BM[tile]) * cdiv(k, TILE_BN[tile]), cdiv((long)r * s * c, BK), TILE_BM[tile], TILE_BN[tile],
            0, t, &need);
  return need;
}

// P of the stats[2][k][P] buffer conv2d_fwd fills (0: this geometry cannot emit statistics)
extern "C" int embnet_conv2d_fwd_stats_rows(int n, int c, int r, int s, int k, int oh, int ow) {
  if (n <= 0 || c <= 0 || r <= 0 || s <= 0 || k <= 0 || oh <= 0 || ow <= 0 || ((c | k) & 3)) return 0;
  const long M = (long)n * oh * ow;
  if (r == 1 && s == 1 && thin_gemm_applies(c, k)) return thin_gemm_stats_rows(M, k);       // conv_thin.hip
  const int tile = pick_tile(M, k, false, (long)r * s * c);
  return cdiv(M, TILE_BM[tile]) * (TILE_BM[tile] / TILE_WTM[tile]);
}

static int conv2d_fwd_impl(const float* x, const float* w, const float* bias, float* y, int n, int h,
                           int wd, int c, int r, int s, int k, int stride, int pad_t, int pad_l, int oh,
                           int ow, int relu, const float* residual, const float* in_scale,
                           const float* in_shift, int in_act, float* stats, void* workspace,
                           size_t workspace_bytes, void* stream, const Ranges rg) {
  const bool ranged = ranges_ok(rg);
  EMBNET_CHECK_ARG(x && w && y, "conv2d_fwd: null pointer");
  EMBNET_CHECK_ARG(!in_scale == !in_shift, "conv2d_fwd: in_scale and in_shift go together");
  EMBNET_CHECK_ARG(aligned16(y) && aligned16(workspace) && aligned16(residual),
                   "conv2d_fwd: output, residual and workspace must be 16-byte aligned");
  ConvFwdParams p{x, w, bias, y, {}, relu, {}, residual};
  if (int rc = make_geom(p.g, n, h, wd, c, r, s, k, stride, pad_t, pad_l, oh, ow, "conv2d_fwd")) return rc;
  const long M = (long)n * oh * ow;
  hipStream_t st = (hipStream_t)stream;
  if (r == 1 && s == 1 && thin_gemm_applies(c, k)) {
    // thin reduction: an HBM stream, not a matrix problem (conv_thin.hip); the statistics' row count follows this path
    const bool can = pad_t == 0 && pad_l == 0 && !in_scale && aligned16(x) && aligned16(w) && aligned16(bias) && aligned16(stats);
    if (can) return launch_thin_gemm(x, w, 0, M, c, k, y, stats, bias, relu, residual, stride, n, h, wd, oh, ow, st);
    EMBNET_CHECK_ARG(!stats, "conv2d_fwd: epilogue statistics of a thin 1x1 conv go with pad 0 and no input transform");
  }
  const bool vec = (c & 3) == 0 && (k & 3) == 0 && aligned16(x) && aligned16(w) && (!bias || aligned16(bias));
  const bool hform = ranged && vec && !in_scale;
  int tile = pick_tile(M, k, false, (long)r * s * c);
  // EMBNET_FWD_256=1 (experiment, off): a short reduction into <= 64 filters over very many pixels (the zoo ResNets' 7x7x4 stem: 7 K
  // tiles, 12 544 tiles of 128 x 64 at batch 128) on 256 x 64 tiles — if the launch were (rounds) x (K tiles) x (a gather round trip),
  // half the rounds would halve it; measured 244 -> 271 us: it is the rate at which the L1 takes the 49 16-byte requests per output
  // pixel, and fewer, fatter workgroups hide less of it.  Only the three-product form has the instantiation; M % 256 == 0 keeps the
  // statistics' row count what embnet_conv2d_fwd_stats_rows says.
  static const int fwd256 = (int)env_long("EMBNET_FWD_256", 0);     // measured slower (244 -> 271 us, profiles/r06_exp_fwd256.txt): off
  if (fwd256 && hform && tile == 1 && k <= 64 && M % 256 == 0 && cdiv((long)r * s * c, BK) <= 8 && M / 256 >= 4 * 512) tile = 5;
  const long tiles = (long)cdiv(M, TILE_BM[tile]) * cdiv(k, TILE_BN[tile]);
  EMBNET_CHECK_ARG(!in_scale || (vec && aligned16(in_scale) && aligned16(in_shift)),
                   "conv2d_fwd: the fused input transform needs channel counts that are multiples of 4 and aligned pointers");
  EMBNET_CHECK_ARG(!stats || (vec && aligned16(stats)),
                   "conv2d_fwd: epilogue statistics need channel counts that are multiples of 4 (see conv2d_fwd_stats_rows)");
  p.tf = InputTransform{in_scale, in_shift, in_act};
  p.stats = stats;
  p.stats_rows = cdiv(M, TILE_BM[tile]) * (TILE_BM[tile] / TILE_WTM[tile]);
  plan_tail(tiles, cdiv((long)r * s * c, BK), TILE_BM[tile], TILE_BN[tile], (vec && workspace) ? workspace_bytes : 0, p.tail);
  p.tail.ws = (float*)workspace;
  const int grid = p.tail.n_full + (int)(tiles - p.tail.n_full) * p.tail.parts;
  p.fair_from = fair_from(grid, tile, false);
  const double flop = 2.0 * M * k * r * s * c;
  p.rg = rg;
  {
    EMBNET_TRACE_FLOP(hform ? conv_h_kernel_name("conv_fwd_h_kernel", "ConvFwdParams", tile) :
                      conv_kernel_name(in_scale ? "conv_fwd_tf_kernel" : "conv_fwd_kernel", "ConvFwdParams", tile,
                                       (vec || in_scale) ? "true" : "false"), flop,
                      4.0 * ((double)n * h * wd * c + (double)r * s * c * k + (double)M * k * (residual ? 2 : 1)), st);
    if (hform && tile == 5) conv_fwd_h_kernel<G256x64><<<grid, 256, 0, st>>>(p);
    else if (hform) { LAUNCH_TILED_H(conv_fwd_h_kernel, tile, grid, st, p) }
    else if (in_scale) { LAUNCH_TILED(conv_fwd_tf_kernel, true, tile, grid, st, p) }
    else if (vec) { LAUNCH_TILED(conv_fwd_kernel, true, tile, grid, st, p) }
    else { LAUNCH_TILED(conv_fwd_kernel, false, tile, grid, st, p) }
  }
  if (p.tail.parts > 1) {
    const int rem = (int)(tiles - p.tail.n_full), bm = TILE_BM[tile], bn = TILE_BN[tile];
    EMBNET_TRACE("embnet::tail_fixup_kernel", TRACE_BYTES, 4.0 * rem * bm * bn * (p.tail.parts + 1 + (residual ? 1 : 0)), st);
    tail_fixup_kernel<<<rem * (bm * bn / 1024), 256, 0, st>>>(p.tail.ws, p.tail.parts, bm, bn, TILE_WTM[tile], p.tail.n_full,
                                                             cdiv(k, bn), M, k, bias, relu, residual, y, stats, p.stats_rows, BnSums{nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0});
  }
  return check_launch("conv2d_fwd");
}
extern "C" int embnet_conv2d_fwd_f32(const float* x, const float* w, const float* bias, float* y, int n, int h,
                                     int wd, int c, int r, int s, int k, int stride, int pad_t, int pad_l, int oh,
                                     int ow, int relu, const float* residual, const float* in_scale,
                                     const float* in_shift, int in_act, float* stats, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  return conv2d_fwd_impl(x, w, bias, y, n, h, wd, c, r, s, k, stride, pad_t, pad_l, oh, ow, relu, residual, in_scale, in_shift, in_act,
                         stats, workspace, workspace_bytes, stream, take_ranges());
}
extern "C" int embnet_conv2d_fwd_f32_ex(const float* x, const float* w, const float* bias, float* y, int n, int h,
                                        int wd, int c, int r, int s, int k, int stride, int pad_t, int pad_l, int oh,
                                        int ow, int relu, const float* residual, const float* in_scale,
                                        const float* in_shift, int in_act, float* stats, void* workspace,
                                        size_t workspace_bytes, const uint32_t* x_range, const uint32_t* w_range, void* stream) {
  (void)take_ranges();
  return conv2d_fwd_impl(x, w, bias, y, n, h, wd, c, r, s, k, stride, pad_t, pad_l, oh, ow, relu, residual, in_scale, in_shift, in_act,
                         stats, workspace, workspace_bytes, stream, Ranges{x_range, w_range});
}

extern "C" size_t embnet_conv2d_dgrad_workspace_bytes(int n, int h, int wd, int c, int r, int s, int k, int stride) {
  if (n <= 0 || h <= 0 || wd <= 0 || c <= 0 || r <= 0 || s <= 0 || k <= 0 || stride != 1 || ((c | k) & 3)) return 0;
  const long M = (long)n * h * wd;
  const int tile = pick_tile(M, c, false, (long)r * s * k);
  SplitTail t; size_t need;
  plan_tail((long)cdiv(M, TILE_BM[tile]) * cdiv(c, TILE_BN[tile]), cdiv((long)r * s * k, BK), TILE_BM[tile], TILE_BN[tile],
            0, t, &need);
  return need;
}

static int conv2d_dgrad_impl(const float* dy, const float* w, float* dx, int n, int h, int wd, int c,
                             int r, int s, int k, int stride, int pad_t, int pad_l, int oh, int ow,
                             int accumulate, const float* dx_add, void* workspace, size_t workspace_bytes,
                             void* stream, const BnSums bn, const Ranges rg) {
  const bool ranged = ranges_ok(rg);
  EMBNET_CHECK_ARG(dy && w && dx, "conv2d_dgrad: null pointer");
  EMBNET_CHECK_ARG(aligned16(dx) && aligned16(workspace), "conv2d_dgrad: output and workspace must be 16-byte aligned");
  EMBNET_CHECK_ARG(!(accumulate && dx_add), "conv2d_dgrad: accumulate (into dx) or dx_add (another tensor), not both");
  EMBNET_CHECK_ARG(aligned16(dx_add), "conv2d_dgrad: dx_add must be 16-byte aligned");
  ConvDgradParams p{dy, w, dx, {}, {}, {}, (accumulate != 0 || dx_add) ? 1 : 0, dx_add ? dx_add : dx};
  p.bn = bn;
  if (int rc = make_geom(p.g, n, h, wd, c, r, s, k, stride, pad_t, pad_l, oh, ow, "conv2d_dgrad")) return rc;
  if (r == 1 && s == 1 && stride == 1 && pad_t == 0 && pad_l == 0 && !accumulate && !dx_add && !bn.partial && aligned16(dy) &&
      aligned16(w) && thin_gemm_applies(k, c))
    // dx[m, c] = dy[m, k] * W[c, k]^T with a thin k: an HBM stream whose output is wide (conv_thin.hip)
    return launch_thin_gemm(dy, w, 1, (long)n * h * wd, k, c, dx, nullptr, nullptr, 0, nullptr, 1, n, h, wd, h, wd, (hipStream_t)stream);
  EMBNET_CHECK_ARG(stride * stride <= MAX_CLASSES, "conv2d_dgrad: stride %d > 3 unsupported", stride);
  long max_m = 0;
  for (int ph = 0; ph < stride; ++ph)
    for (int pw = 0; pw < stride; ++pw) {
      DgradClass& cg = p.cls[ph * stride + pw];
      cg.r0 = ph; cg.s0 = pw;
      cg.hoff = ((ph - pad_t) % stride + stride) % stride; cg.woff = ((pw - pad_l) % stride + stride) % stride;
      cg.Hc = cg.hoff < h ? (h - cg.hoff + stride - 1) / stride : 0;
      cg.Wc = cg.woff < wd ? (wd - cg.woff + stride - 1) / stride : 0;
      cg.nR = ph < r ? (r - ph + stride - 1) / stride : 0;
      cg.nS = pw < s ? (s - pw + stride - 1) / stride : 0;
      cg.hb = (cg.hoff + pad_t - ph) / stride; cg.wb = (cg.woff + pad_l - pw) / stride;
      cg.dHWc = FastDiv::make(cg.Hc * cg.Wc > 0 ? cg.Hc * cg.Wc : 1);
      cg.dWc = FastDiv::make(cg.Wc > 0 ? cg.Wc : 1);
      cg.dnS = FastDiv::make(cg.nS > 0 ? cg.nS : 1);
      const long m = (long)n * cg.Hc * cg.Wc;
      if (m > max_m) max_m = m;
    }
  hipStream_t st = (hipStream_t)stream;
  const int tile = pick_tile(max_m * stride * stride, c, stride > 1, (long)r * s * k);
  const long tiles = (long)cdiv(max_m, TILE_BM[tile]) * cdiv(c, TILE_BN[tile]);
  const bool vec = (k & 3) == 0 && aligned16(dy) && aligned16(w);
  // stride 1 = one class whose rows are the input pixels in order, so the fix-up writes dx[row*C + col]
  const bool can_split = vec && stride == 1 && (c & 3) == 0 && workspace;
  plan_tail(tiles, cdiv((long)p.cls[0].nR * p.cls[0].nS * k, BK), TILE_BM[tile], TILE_BN[tile],
            can_split ? workspace_bytes : 0, p.tail);
  p.tail.ws = (float*)workspace;
  const dim3 grid(p.tail.n_full + (int)(tiles - p.tail.n_full) * p.tail.parts, stride * stride);
  p.fair_from = fair_from((long)grid.x * grid.y, tile, false);
  const bool hform = ranged && vec;
  p.rg = rg;
  {
    EMBNET_TRACE_FLOP(hform ? conv_h_kernel_name("conv_dgrad_h_kernel", "ConvDgradParams", tile) :
                      conv_kernel_name("conv_dgrad_kernel", "ConvDgradParams", tile, vec ? "true" : "false"),
                      2.0 * n * oh * ow * (double)k * r * s * c,
                      4.0 * ((double)n * oh * ow * k + (double)r * s * c * k + (double)n * h * wd * c * (p.accumulate ? 2 : 1)), st);
    if (hform) { LAUNCH_TILED_H(conv_dgrad_h_kernel, tile, grid, st, p) }
    else if (vec) { LAUNCH_TILED(conv_dgrad_kernel, true, tile, grid, st, p) }
    else { LAUNCH_TILED(conv_dgrad_kernel, false, tile, grid, st, p) }
  }
  if (p.tail.parts > 1) {
    const int rem = (int)(tiles - p.tail.n_full), bm = TILE_BM[tile], bn = TILE_BN[tile];
    EMBNET_TRACE("embnet::tail_fixup_kernel", TRACE_BYTES, 4.0 * rem * bm * bn * (p.tail.parts + 1 + (p.accumulate ? 1 : 0)), st);
    tail_fixup_kernel<<<rem * (bm * bn / 1024), 256, 0, st>>>(p.tail.ws, p.tail.parts, bm, bn, TILE_WTM[tile], p.tail.n_full,
                                                             cdiv(c, bn), max_m, c, nullptr, 0, p.accumulate ? p.add_src : nullptr, dx, nullptr, 0, p.bn);
  }
  return check_launch("conv2d_dgrad");
}

extern "C" int embnet_conv2d_dgrad_f32(const float* dy, const float* w, float* dx, int n, int h, int wd, int c,
                                       int r, int s, int k, int stride, int pad_t, int pad_l, int oh, int ow,
                                       int accumulate, const float* dx_add, void* workspace, size_t workspace_bytes,
                                       void* stream) {
  return conv2d_dgrad_impl(dy, w, dx, n, h, wd, c, r, s, k, stride, pad_t, pad_l, oh, ow, accumulate, dx_add, workspace,
                           workspace_bytes, stream, BnSums{nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0}, take_ranges());
}
extern "C" int embnet_conv2d_dgrad_f32_ex(const float* dy, const float* w, float* dx, int n, int h, int wd, int c,
                                          int r, int s, int k, int stride, int pad_t, int pad_l, int oh, int ow,
                                          int accumulate, const float* dx_add, void* workspace, size_t workspace_bytes,
                                          const uint32_t* dy_range, const uint32_t* w_range, void* stream) {
  (void)take_ranges();
  return conv2d_dgrad_impl(dy, w, dx, n, h, wd, c, r, s, k, stride, pad_t, pad_l, oh, ow, accumulate, dx_add, workspace,
                           workspace_bytes, stream, BnSums{nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0}, Ranges{dy_range, w_range});
}

// rows of the [2][C][rows] partial sums embnet_conv2d_dgrad_bnsums_f32 writes for this geometry; 0: not available
// (stride 1, C % 4 == 0 and K % 4 == 0 only: one class of rows = the input pixels in order, 16-byte epilogue)
extern "C" int embnet_conv2d_dgrad_bnsums_rows(int n, int h, int wd, int c, int r, int s, int k, int stride) {
  if (n <= 0 || h <= 0 || wd <= 0 || c <= 0 || r <= 0 || s <= 0 || k <= 0 || stride != 1 || ((c | k) & 3)) return 0;
  const long M = (long)n * h * wd;
  const int tile = pick_tile(M, c, false, (long)r * s * k);
  return cdiv(M, TILE_BM[tile]) * (TILE_BM[tile] / TILE_WTM[tile]);
}

static int conv2d_dgrad_bnsums_impl(const float* dy, const float* w, float* dx, int n, int h, int wd, int c, int r,
                                    int s, int k, int stride, int pad_t, int pad_l, int oh, int ow,
                                    const float* bn_x, const float* bn_scale, const float* bn_shift,
                                    const float* bn_mean, const float* bn_rstd, int bn_act, float* bn_partial,
                                    int bn_rows, void* workspace, size_t workspace_bytes, void* stream, const Ranges rg, int kinds = 2) {
  EMBNET_CHECK_ARG(bn_x && bn_scale && bn_shift && bn_mean && bn_rstd && bn_partial, "conv2d_dgrad_bnsums: null pointer");
  EMBNET_CHECK_ARG(bn_rows > 0 && bn_rows == embnet_conv2d_dgrad_bnsums_rows(n, h, wd, c, r, s, k, stride),
                   "conv2d_dgrad_bnsums: rows %d for this geometry (see embnet_conv2d_dgrad_bnsums_rows)", bn_rows);
  EMBNET_CHECK_ARG(aligned16(bn_x) && aligned16(bn_scale) && aligned16(bn_shift) && aligned16(bn_mean) && aligned16(bn_rstd) &&
                   aligned16(dy) && aligned16(w), "conv2d_dgrad_bnsums: operands must be 16-byte aligned");
  EMBNET_CHECK_ARG(bn_act >= 0 && bn_act <= 2, "conv2d_dgrad_bnsums: activation code %d", bn_act);
  return conv2d_dgrad_impl(dy, w, dx, n, h, wd, c, r, s, k, stride, pad_t, pad_l, oh, ow, 0, nullptr, workspace, workspace_bytes,
                           stream, BnSums{bn_x, bn_scale, bn_shift, bn_mean, bn_rstd, bn_act, bn_partial, bn_rows, kinds}, rg);
}
extern "C" int embnet_conv2d_dgrad_bnsums_f32(const float* dy, const float* w, float* dx, int n, int h, int wd, int c, int r,
                                              int s, int k, int stride, int pad_t, int pad_l, int oh, int ow,
                                              const float* bn_x, const float* bn_scale, const float* bn_shift,
                                              const float* bn_mean, const float* bn_rstd, int bn_act, float* bn_partial,
                                              int bn_rows, void* workspace, size_t workspace_bytes, void* stream) {
  return conv2d_dgrad_bnsums_impl(dy, w, dx, n, h, wd, c, r, s, k, stride, pad_t, pad_l, oh, ow, bn_x, bn_scale, bn_shift, bn_mean,
                                  bn_rstd, bn_act, bn_partial, bn_rows, workspace, workspace_bytes, stream, take_ranges());
}
extern "C" int embnet_conv2d_dgrad_bnsums_f32_ex(const float* dy, const float* w, float* dx, int n, int h, int wd, int c, int r,
                                                 int s, int k, int stride, int pad_t, int pad_l, int oh, int ow,
                                                 const float* bn_x, const float* bn_scale, const float* bn_shift,
                                                 const float* bn_mean, const float* bn_rstd, int bn_act, float* bn_partial,
                                                 int bn_rows, void* workspace, size_t workspace_bytes,
                                                 const uint32_t* dy_range, const uint32_t* w_range, void* stream) {
  (void)take_ranges();
  return conv2d_dgrad_bnsums_impl(dy, w, dx, n, h, wd, c, r, s, k, stride, pad_t, pad_l, oh, ow, bn_x, bn_scale, bn_shift, bn_mean,
                                  bn_rstd, bn_act, bn_partial, bn_rows, workspace, workspace_bytes, stream, Ranges{dy_range, w_range}, 3);
}

// wgrad tiling: rows = R*S*C, cols = K; split the (n,oh,ow) reduction so the grid covers the chip
static void wgrad_plan(int rows, int k, long kg, int& tile, int& splits, int& kt_per_split, bool one_by_one = false) {
  if (one_by_one && thin_wgrad_applies(rows, k)) {      // conv_thin.hip: a 1x1 conv with a thin side is a stream, not a GEMM
    tile = -1; splits = thin_wgrad_splits(kg, rows, k); kt_per_split = 0;
    return;
  }
  // few gradient rows (1x1 convs on 16..64 channels, EfficientNet): 64-row tiles, whatever the width
  tile = (rows <= 64 && k > 32) ? 3 : (k <= 32 ? 2 : (k <= 64 ? 1 : 0));
  // 128-row tiles waste the last half tile of a 576-row (3x3x64) gradient; 192-row tiles fit it exactly
  static const bool no192 = env_long("EMBNET_WGRAD_NO192", 0) != 0;
  if (tile == 1 && rows % 192 == 0 && rows % 128 != 0 && !no192) tile = 4;
  const int forced_tile = (int)env_long("EMBNET_WGRAD_TILE", -1);
  if (forced_tile >= 0) tile = forced_tile;
  const long tiles = (long)cdiv(rows, TILE_BM[tile]) * cdiv(k, TILE_BN[tile]);
  const int kt_total = cdiv(kg, BK);
  // measured (EMBNET_WGRAD_BLOCKS sweep on ResNet18 shapes): with few output tiles one round of 3
  // workgroups per CU is best; with many tiles shorter K ranges in 2-3 rounds balance better
#if EMBNET_CONV_SPLIT
  // two workgroups per CU are resident: whole rounds of 512
  // (tools/exp/wgrad_rounds.py, profiles/r02_wgrad_rounds.txt: with whole rounds, ONE round wins up to 72 tiles, two from 144)
  long target = tiles >= 100 ? 1024 : (tiles <= 2 ? 768 : 512);   // stem (2 tiles, 50k K tiles): 768
#else
  long target = tiles >= 100 ? 2048 : (tiles >= 30 ? 1536 : 768);
#endif
  const long forced_blocks = env_long("EMBNET_WGRAD_BLOCKS", 0);   // tuning aids, read per call
  if (forced_blocks > 0) target = forced_blocks;
#if EMBNET_CONV_SPLIT
  // whole rounds: tiles * splits must not spill a few workgroups into an extra round (36 tiles x 15 splits = 540 ran as
  // two rounds of 512 slots) -> round the split count DOWN
  long want = target / tiles;
#else
  long want = (target + tiles - 1) / tiles;
#endif
  if (want > kt_total / 4) want = kt_total / 4;       // at least 4 k-tiles per split
  if (want < 1) want = 1;
  kt_per_split = cdiv(kt_total, want);
  splits = cdiv(kt_total, kt_per_split);
}

extern "C" size_t embnet_conv2d_wgrad_workspace_bytes(int n, int c, int r, int s, int k, int oh, int ow) {
  if (n <= 0 || c <= 0 || r <= 0 || s <= 0 || k <= 0 || oh <= 0 || ow <= 0) return 0;
  int tile, splits, ktps;
  wgrad_plan(r * s * c, k, (long)n * oh * ow, tile, splits, ktps, r == 1 && s == 1);
  return splits > 1 ? (size_t)splits * r * s * c * k * sizeof(float) : 0;
}

#define LAUNCH_WGRAD(KERNEL, VA, VB)                                                \
  switch (tile) {                                                                       \
    case 0: KERNEL<G128x128, VA, VB><<<grid, 256, 0, st>>>(p); break;                   \
    case 1: KERNEL<G128x64, VA, VB><<<grid, 256, 0, st>>>(p); break;                    \
    case 2: KERNEL<G128x32, VA, VB><<<grid, 256, 0, st>>>(p); break;                    \
    case 4: KERNEL<G192x64, VA, VB><<<grid, 256, 0, st>>>(p); break;                    \
    default: KERNEL<G64x64, VA, VB><<<grid, 256, 0, st>>>(p); break;                    \
  }

static int wgrad_impl(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes, int n, int h,
                      int wd, int c, int r, int s, int k, int stride, int pad_t, int pad_l, int oh, int ow,
                      const float* in_scale, const float* in_shift, int in_act, void* stream, bool do_main,
                      bool do_reduce, const Ranges rg) {
  const bool ranged = ranges_ok(rg);
  EMBNET_CHECK_ARG(x && dy && dw, "conv2d_wgrad: null pointer");
  EMBNET_CHECK_ARG(aligned16(dw) && aligned16(workspace), "conv2d_wgrad: dw and workspace must be 16-byte aligned");
  ConvWgradParams p{x, dy, dw, {}, 0, 1, 0};
#ifndef EMBNET_WGRAD_XCD_DEFAULT
#define EMBNET_WGRAD_XCD_DEFAULT 0
#endif
#ifndef EMBNET_WGRAD_STAGGER_DEFAULT
#define EMBNET_WGRAD_STAGGER_DEFAULT 0
#endif
  static const int xcd_order = (int)env_long("EMBNET_WGRAD_XCD", EMBNET_WGRAD_XCD_DEFAULT);
  static const int stagger = (int)env_long("EMBNET_WGRAD_STAGGER", EMBNET_WGRAD_STAGGER_DEFAULT);
  p.xcd_order = xcd_order;
  p.stagger = xcd_order ? stagger : 0;
  if (int rc = make_geom(p.g, n, h, wd, c, r, s, k, stride, pad_t, pad_l, oh, ow, "conv2d_wgrad")) return rc;
  int tile;
  const int rows = r * s * c;
  wgrad_plan(rows, k, (long)n * oh * ow, tile, p.splits, p.kt_per_split, r == 1 && s == 1);
  const size_t need = embnet_conv2d_wgrad_workspace_bytes(n, c, r, s, k, oh, ow);
  if (need > workspace_bytes || (need && !workspace))
    return fail(EMBNET_EWORKSPACE, "conv2d_wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
  if (p.splits > 1) p.out = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  if (tile < 0) {
    // thin 1x1 weight gradient (the split count every caller planned with is this path's): one slab per workgroup + the slab sum
    EMBNET_CHECK_ARG(pad_t == 0 && pad_l == 0 && !in_scale && aligned16(x) && aligned16(dy),
                     "conv2d_wgrad: a 1x1 conv with a thin side takes pad 0, no fused input transform and 16-byte aligned operands "
                     "(EMBNET_CONV_THIN_WGRAD=0 restores the implicit-GEMM kernel)");
    if (do_main)
      if (int rc = launch_thin_wgrad(x, dy, p.splits > 1 ? (float*)workspace : dw, (long)n * oh * ow, c, k, stride, n, h, wd, oh, ow, st))
        return rc;
    if (p.splits > 1 && do_reduce) {
      const long cnt = (long)rows * k;
      EMBNET_TRACE("embnet::slab_reduce_kernel", TRACE_BYTES, 4.0 * cnt * (p.splits + 1), st);
      slab_reduce_kernel<<<(cnt & 3) ? cdiv(cnt, 256) : cdiv(cnt / 4, 32), 256, 0, st>>>((const float*)workspace, p.splits, cnt, dw);
    }
    return check_launch("conv2d_wgrad");
  }
  dim3 grid(cdiv(rows, TILE_BM[tile]) * cdiv(k, TILE_BN[tile]) * ((p.splits + 7) / 8 * 8));
  p.fair_from = fair_from(grid.x, tile, true);
  const bool va = (c & 3) == 0 && aligned16(x), vb = (k & 3) == 0 && aligned16(dy);
  EMBNET_CHECK_ARG(!in_scale == !in_shift, "conv2d_wgrad: in_scale and in_shift go together");
  EMBNET_CHECK_ARG(!in_scale || (va && vb && aligned16(in_scale) && aligned16(in_shift)),
                   "conv2d_wgrad: the fused input transform needs channel counts that are multiples of 4 and aligned pointers");
  p.tf = InputTransform{in_scale, in_shift, in_act};
  if (do_main) {
    // 16x16x32 MFMA shape for the weight gradient (both operands k-major: same fragment reads and matrix cycles as
    // 32x32x16): EMBNET_WGRAD_K32=1 (experiment; see wgrad_k32()).
    // (not the 128x64 tile: at three workgroups per CU its K32 form spills 9 registers in the loop — the ResNet stem's
    // weight gradient ran 455-472 us on it against 300-320, profiles/r04_bench_kernel_stats.md)
    const bool k32 = wgrad_k32() && tile != 1;
    const bool hform = ranged && va && vb && !in_scale && !(k32 && !p.xcd_order);
    p.rg = rg;
    // one 256-row tile where two 128-row tiles would each stream all of dy (three-product kernels only; the split count — the slab
    // layout every caller planned with — stays the plan's).  EMBNET_WGRAD_256=1; OFF by default: the stem's weight gradient moves half
    // the bytes (1 130 -> 554 MB per launch, profiles/r06_pmc_traffic_c2_xcd0.txt) and takes 256 -> 321-330 us: half as many
    // workgroups, and the loop is bound by its gather round trips, not by bytes (profiles/r06_exp_xcd_rows.txt)
    static const bool use256 = env_long("EMBNET_WGRAD_256", 0) != 0;
    if (hform && tile == 1 && rows > 128 && rows <= 256 && use256) {
      tile = 5;
      grid = dim3(cdiv(k, TILE_BN[5]) * ((p.splits + 7) / 8 * 8));
      p.fair_from = fair_from(grid.x, tile, true);
    }
    char k32name[160];
    snprintf(k32name, sizeof k32name, "void embnet::conv_wgrad_k32_kernel<embnet::Geom<%s> >(embnet::ConvWgradParams)", GEOM_NAME[tile]);
    if (in_scale) snprintf(k32name, sizeof k32name, "void embnet::conv_wgrad_k32_tf_kernel<embnet::Geom<%s> >(embnet::ConvWgradParams)", GEOM_NAME[tile]);
    EMBNET_TRACE_FLOP(hform ? conv_h_kernel_name("conv_wgrad_h_kernel", "ConvWgradParams", tile) :
                      ((in_scale || (va && vb)) && k32 && !p.xcd_order) ? k32name :
                      conv_kernel_name(in_scale ? "conv_wgrad_tf_kernel" : "conv_wgrad_kernel", "ConvWgradParams", tile,
                                       (in_scale || (va && vb)) ? "true, true" : (vb ? "false, true" : "false, false")),
                      2.0 * n * oh * ow * (double)k * rows,
                      4.0 * ((double)n * h * wd * c + (double)n * oh * ow * k + (double)rows * k * p.splits), st);
    if (hform) { LAUNCH_WGRAD_H(conv_wgrad_h_kernel, tile, grid, st, p) }
    else if (in_scale && k32 && !p.xcd_order) {         // same arithmetic as the plain K32 kernels: deferred BN stays bit-identical
      switch (tile) {
        case 0: conv_wgrad_k32_tf_kernel<G128x128><<<grid, 256, 0, st>>>(p); break;
        case 2: conv_wgrad_k32_tf_kernel<G128x32><<<grid, 256, 0, st>>>(p); break;
        case 4: conv_wgrad_k32_tf_kernel<G192x64><<<grid, 256, 0, st>>>(p); break;
        default: conv_wgrad_k32_tf_kernel<G64x64><<<grid, 256, 0, st>>>(p); break;
      }
    }
    else if (in_scale) { LAUNCH_WGRAD(conv_wgrad_tf_kernel, true, true) }
    else if (va && vb && k32 && !p.xcd_order) {
      switch (tile) {
        case 0: conv_wgrad_k32_kernel<G128x128><<<grid, 256, 0, st>>>(p); break;
        case 2: conv_wgrad_k32_kernel<G128x32><<<grid, 256, 0, st>>>(p); break;
        case 4: conv_wgrad_k32_kernel<G192x64><<<grid, 256, 0, st>>>(p); break;
        default: conv_wgrad_k32_kernel<G64x64><<<grid, 256, 0, st>>>(p); break;
      }
    }
    else if (va && vb) { LAUNCH_WGRAD(conv_wgrad_kernel, true, true) }
    else if (vb) { LAUNCH_WGRAD(conv_wgrad_kernel, false, true) }
    else { LAUNCH_WGRAD(conv_wgrad_kernel, false, false) }
  }
  if (p.splits > 1 && do_reduce) {
    const long cnt = (long)rows * k;
    EMBNET_TRACE("embnet::slab_reduce_kernel", TRACE_BYTES, 4.0 * cnt * (p.splits + 1), st);
    slab_reduce_kernel<<<(cnt & 3) ? cdiv(cnt, 256) : cdiv(cnt / 4, 32), 256, 0, st>>>((const float*)workspace, p.splits, cnt, dw);
  }
  return check_launch("conv2d_wgrad");
}

extern "C" int embnet_conv2d_wgrad_f32(const float* x, const float* dy, float* dw, void* workspace,
                                       size_t workspace_bytes, int n, int h, int wd, int c, int r, int s, int k,
                                       int stride, int pad_t, int pad_l, int oh, int ow, const float* in_scale, const float* in_shift, int in_act,
        void* stream) {
  return wgrad_impl(x, dy, dw, workspace, workspace_bytes, n, h, wd, c, r, s, k, stride, pad_t, pad_l, oh, ow, in_scale, in_shift, in_act,
                    stream, true, true, take_ranges());
}
extern "C" int embnet_conv2d_wgrad_f32_ex(const float* x, const float* dy, float* dw, void* workspace,
                                       size_t workspace_bytes, int n, int h, int wd, int c, int r, int s, int k,
                                       int stride, int pad_t, int pad_l, int oh, int ow, const float* in_scale, const float* in_shift, int in_act,
        const uint32_t* x_range, const uint32_t* dy_range, void* stream) {
  (void)take_ranges();
  return wgrad_impl(x, dy, dw, workspace, workspace_bytes, n, h, wd, c, r, s, k, stride, pad_t, pad_l, oh, ow, in_scale, in_shift, in_act,
                    stream, true, true, Ranges{x_range, dy_range});
}

extern "C" int embnet_conv2d_wgrad_splits(int n, int c, int r, int s, int k, int oh, int ow) {
  if (n <= 0 || c <= 0 || r <= 0 || s <= 0 || k <= 0 || oh <= 0 || ow <= 0) return 0;
  int tile, splits, ktps;
  wgrad_plan(r * s * c, k, (long)n * oh * ow, tile, splits, ktps, r == 1 && s == 1);
  return splits;
}

extern "C" int embnet_slab_reduce_multi(const void* host_table, int n_tensors, void* stream) {
  EMBNET_CHECK_ARG(host_table && n_tensors > 0, "slab_reduce_multi: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const SlabTensor* src = (const SlabTensor*)host_table;
  for (int first = 0; first < n_tensors; first += SLAB_BATCH) {
    SlabBatch b{};
    b.n = n_tensors - first < SLAB_BATCH ? n_tensors - first : SLAB_BATCH;
    int blocks = 0; double bytes = 0;
    for (int i = 0; i < b.n; ++i) {
      b.t[i] = src[first + i];
      EMBNET_CHECK_ARG(b.t[i].slabs && b.t[i].out && b.t[i].n > 0 && b.t[i].splits > 0, "slab_reduce_multi: bad descriptor %d", first + i);
      EMBNET_CHECK_ARG((b.t[i].n & 3) || !((((uintptr_t)b.t[i].slabs) | ((uintptr_t)b.t[i].out)) & 15),
                       "slab_reduce_multi: descriptor %d needs 16-byte aligned slabs and out", first + i);
      b.t[i].first_block = blocks;
      blocks += (int)((b.t[i].n + 127) / 128);
      bytes += 4.0 * b.t[i].n * (b.t[i].splits + 1);
    }
    EMBNET_TRACE("embnet::slab_reduce_multi_kernel", TRACE_BYTES, bytes, st);
    slab_reduce_multi_kernel<<<blocks, 256, 0, st>>>(b);
  }
  return check_launch("slab_reduce_multi");
}

// The same in two calls — split-K GEMM into the slabs, then the fixed-order slab sum — so a caller can
// time the MFMA kernel alone (bench.py's roofline leg).  When the plan has a single split the first call
// writes dw directly and the second is a no-op.
extern "C" int embnet_conv2d_wgrad_slabs_f32(const float* x, const float* dy, float* dw, void* workspace,
                                       size_t workspace_bytes, int n, int h, int wd, int c, int r, int s, int k,
                                       int stride, int pad_t, int pad_l, int oh, int ow, const float* in_scale, const float* in_shift, int in_act,
        void* stream) {
  return wgrad_impl(x, dy, dw, workspace, workspace_bytes, n, h, wd, c, r, s, k, stride, pad_t, pad_l, oh, ow, in_scale, in_shift, in_act,
                    stream, true, false, take_ranges());
}
extern "C" int embnet_conv2d_wgrad_slabs_f32_ex(const float* x, const float* dy, float* dw, void* workspace,
                                       size_t workspace_bytes, int n, int h, int wd, int c, int r, int s, int k,
                                       int stride, int pad_t, int pad_l, int oh, int ow, const float* in_scale, const float* in_shift, int in_act,
        const uint32_t* x_range, const uint32_t* dy_range, void* stream) {
  (void)take_ranges();
  return wgrad_impl(x, dy, dw, workspace, workspace_bytes, n, h, wd, c, r, s, k, stride, pad_t, pad_l, oh, ow, in_scale, in_shift, in_act,
                    stream, true, false, Ranges{x_range, dy_range});
}
extern "C" int embnet_conv2d_wgrad_reduce_f32(const float* x, const float* dy, float* dw, void* workspace,
                                       size_t workspace_bytes, int n, int h, int wd, int c, int r, int s, int k,
                                       int stride, int pad_t, int pad_l, int oh, int ow, const float* in_scale, const float* in_shift, int in_act,
        void* stream) {
  return wgrad_impl(x, dy, dw, workspace, workspace_bytes, n, h, wd, c, r, s, k, stride, pad_t, pad_l, oh, ow, in_scale, in_shift, in_act,
                    stream, false, true, take_ranges());
}

// Name (as rocprofv3 prints the template) of the kernel the entry points above launch for a geometry,
// so a caller can attribute its own HIP-event timings to the symbol the profiler reports.
extern "C" const char* embnet_conv2d_kernel_name(int kind, int n, int h, int wd, int c, int r, int s, int k,
                                                 int oh, int ow) {
  static thread_local char buf[160];
  const char* const* geoms = GEOM_NAME;
  const char* t = (c & 3) == 0 ? "true" : "false";
  const char* tk = (k & 3) == 0 ? "true" : "false";
  if (kind == 0) {
    snprintf(buf, sizeof buf, "void embnet::conv_fwd_kernel<embnet::Geom<%s>, %s>(embnet::ConvFwdParams)",
             geoms[pick_tile((long)n * oh * ow, k, false, (long)r * s * c)], ((c | k) & 3) == 0 ? "true" : "false");
  } else if (kind == 1) {
    long max_m = (long)n * ((h + 0) / 1) * wd;         // same tile choice as the launcher (all classes together)
    snprintf(buf, sizeof buf, "void embnet::conv_dgrad_kernel<embnet::Geom<%s>, %s>(embnet::ConvDgradParams)",
             geoms[pick_tile(max_m, c, oh < h, (long)r * s * k)], tk);
  } else if (kind == 2) {
    int tile, sp, kt;
    wgrad_plan(r * s * c, k, (long)n * oh * ow, tile, sp, kt, r == 1 && s == 1);
    if (tile < 0) snprintf(buf, sizeof buf, "embnet::thinw::thin_wgrad_kernel");
    else if (((c | k) & 3) == 0 && wgrad_k32() && tile != 1)
      snprintf(buf, sizeof buf, "void embnet::conv_wgrad_k32_kernel<embnet::Geom<%s> >(embnet::ConvWgradParams)", geoms[tile]);
    else
      snprintf(buf, sizeof buf, "void embnet::conv_wgrad_kernel<embnet::Geom<%s>, %s, %s>(embnet::ConvWgradParams)",
               geoms[tile], ((k & 3) == 0) ? t : "false", tk);
  } else {
    buf[0] = 0;
  }
  return buf;
}
