// Depthwise-convolution geometry shared by mbconv_kernels.hip (per-thread row kernels) and dwconv_tile.hip (LDS-tile kernels).
#pragma once
#include "common.h"

namespace embnet {

struct DwGeom { int N, H, W, C, R, S, stride, pad_t, pad_l, OH, OW; int img_major; int L; };   // img_major: dwconv_row4x2_kernel's thread order; L: 256-thread chunks per workgroup (statistics variants; 0 = 1)

// the BatchNormalization in front of a depthwise layer (its input was act(BN(e))): what a data-gradient kernel needs to emit
// that layer's backward sums (sum dz, sum dz * ehat)
struct DwBn { const float* e; const float* scale; const float* shift; const float* mean; const float* rstd; int act; };

__device__ __forceinline__ void dw_bn_sums_add(const DwBn& bn, float4 v, float4 xq, float4 sc, float4 sh, float4 mu, float4 rs,
                                               float4& s1, float4& s2) {          // the arithmetic of bn_bwd_reduce4_kernel
  float4 dz = v;
  if (bn.act) {
    dz.x = act_grad(bn.act, fmaf(xq.x, sc.x, sh.x), v.x); dz.y = act_grad(bn.act, fmaf(xq.y, sc.y, sh.y), v.y);
    dz.z = act_grad(bn.act, fmaf(xq.z, sc.z, sh.z), v.z); dz.w = act_grad(bn.act, fmaf(xq.w, sc.w, sh.w), v.w);
  }
  s1.x += dz.x; s1.y += dz.y; s1.z += dz.z; s1.w += dz.w;
  s2.x = fmaf(dz.x, (xq.x - mu.x) * rs.x, s2.x); s2.y = fmaf(dz.y, (xq.y - mu.y) * rs.y, s2.y);
  s2.z = fmaf(dz.z, (xq.z - mu.z) * rs.z, s2.z); s2.w = fmaf(dz.w, (xq.w - mu.w) * rs.w, s2.w);
}

namespace dwt {
// Stride-1, same-size depthwise correlation y = corr(x, w) (g: the forward-shaped geometry, pads as the correlation applies them;
// flip: the data gradient's flipped kernel) on the LDS-tile kernel.  stats_kind: 0 none, 1 the output's per-channel sum / sum of
// squares, 2 the BatchNorm-backward sums of `bn`; the partials are [2][C][tile_stats_rows].
bool tile_applies(const DwGeom& g, int stats_kind);
int tile_stats_rows(const DwGeom& g, int stats_kind);
void launch_tile(const float* x, const float* w, const DwGeom& g, bool flip, float* y, float* stats, const DwBn* bn, hipStream_t st);
// the weight gradient of the same layers: partial slabs [tile_wgrad_slabs][R*S*C] (0: does not apply), summed by the caller
int tile_wgrad_slabs(const DwGeom& g);
void launch_tile_wgrad(const float* x, const float* dy, const DwGeom& g, float* slabs, hipStream_t st);
}  // namespace dwt

}  // namespace embnet
