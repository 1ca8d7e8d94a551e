// Convolution geometry shared by conv.hip and conv_patch.hip (and the experiment tools/exp/conv_planes.hip).
#pragma once
#include "common.h"

namespace embnet {

struct FastDiv {           // exact n / d for 0 <= n < 2^31, d >= 1
  uint32_t mul, shift, d;
  static FastDiv make(uint32_t d) {
    FastDiv f; f.d = d;
    uint32_t s = 0; while ((1ull << s) < d) ++s;
    f.shift = s;
    f.mul = (uint32_t)((((1ull << s) - d) << 32) / d + 1);
    return f;
  }
  __device__ __forceinline__ uint32_t div(uint32_t n) const { return (__umulhi(n, mul) + n) >> shift; }
  __device__ __forceinline__ void divmod(uint32_t n, uint32_t& q, uint32_t& r) const { q = div(n); r = n - q * d; }
};

struct ConvGeom {
  int N, H, W, C, R, S, K, stride, pad_t, pad_l, OH, OW;
  FastDiv dOHW, dOW, dHW, dW, dC, dK, dS;
};

constexpr int ROW_INVALID = -(1 << 28);
constexpr size_t CONV_MAX_OPERAND_BYTES = 0x7FFFFFF0ull;   // buffer-addressed operands: 32-bit byte offsets, top bit = "reads as zero"


// gemm-k index kk -> (r, s, inner) with `dinner` the divider of the inner size (C for fwd, K for dgrad)
__device__ __forceinline__ void split_k(int kk, const FastDiv& dinner, const FastDiv& dS, int& r, int& s, int& c) {
  uint32_t rs, cc; dinner.divmod((uint32_t)kk, rs, cc);
  uint32_t rr, ss; dS.divmod(rs, rr, ss);
  r = (int)rr; s = (int)ss; c = (int)cc;
}

// BatchNorm-backward sums riding on a data gradient (conv.hip's conv_dgrad_kernel / tail_fixup_kernel, conv_patch.hip's
// epilogue): see the comment at conv.hip's bn_sums_add.  x == NULL: off.
// kinds: 3 = partial has a THIRD plane [C][rows]: max |dz| per row band (the bound of the BatchNorm backward's dx, nn_kernels.hip
// dx_channel_bound); 0 / 2 = the two sums only
struct BnSums { const float* x; const float* scale; const float* shift; const float* mean; const float* rstd; int act; float* partial; int rows; int kinds; };

// conv.hip: fix-up pass over left-over tiles computed as K-split partial tiles (used by conv.hip and conv_patch.hip)
void launch_tail_fixup(const float* ws, int parts, int bm, int bn, int wtm, int n_full, int rem, int tiles_n, long m, int cols,
                       const float* bias, int relu, const float* residual, float* out, float* stats, int stats_rows,
                       const BnSums& bsum, hipStream_t st);

// conv_thin.hip: 1x1 convolutions with a thin reduction as an HBM stream (EfficientNet's expand forward / project data gradient)
bool thin_gemm_applies(int red, int ncols);
int thin_gemm_stats_rows(long m, int ncols);
int launch_thin_gemm(const float* in, const float* w, int w_transposed, long m, int red, int ncols, float* out, float* stats,
                     const float* bias, int relu, const float* residual, int stride, int n, int h, int wd, int oh, int ow, hipStream_t st);

bool thin_wgrad_applies(int c, int k);
int thin_wgrad_splits(long m, int c, int k);
int launch_thin_wgrad(const float* x, const float* dy, float* slabs, long m, int c, int k, int stride, int n, int h, int wd, int oh,
                      int ow, hipStream_t st);

inline int make_geom(ConvGeom& g, int n, int h, int w, int c, int r, int s, int k, int stride, int pad_t,
                     int pad_l, int oh, int ow, const char* who) {
  EMBNET_CHECK_ARG(n > 0 && h > 0 && w > 0 && c > 0 && r > 0 && s > 0 && k > 0 && stride > 0 && oh > 0 && ow > 0,
                   "%s: non-positive dimension", who);
  EMBNET_CHECK_ARG(pad_t >= 0 && pad_l >= 0, "%s: negative padding", who);
  EMBNET_CHECK_ARG((oh - 1) * stride + 1 - pad_t <= h && (ow - 1) * stride + 1 - pad_l <= w,
                   "%s: output %dx%d reaches outside the %dx%d input", who, oh, ow, h, w);
  EMBNET_CHECK_ARG((size_t)n * h * w * c * 4 <= CONV_MAX_OPERAND_BYTES && (size_t)n * oh * ow * k * 4 <= CONV_MAX_OPERAND_BYTES &&
                   (size_t)r * s * c * k * 4 <= CONV_MAX_OPERAND_BYTES,
                   "%s: a tensor exceeds 2 GiB (buffer-addressed operands): split the batch", who);
  g.N = n; g.H = h; g.W = w; g.C = c; g.R = r; g.S = s; g.K = k; g.stride = stride; g.pad_t = pad_t; g.pad_l = pad_l;
  g.OH = oh; g.OW = ow;
  g.dOHW = FastDiv::make(oh * ow); g.dOW = FastDiv::make(ow); g.dHW = FastDiv::make(h * w); g.dW = FastDiv::make(w);
  g.dC = FastDiv::make(c); g.dK = FastDiv::make(k); g.dS = FastDiv::make(s);
  return 0;
}


}  // namespace embnet
