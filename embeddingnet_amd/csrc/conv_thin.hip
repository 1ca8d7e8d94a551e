// 1x1 convolutions with a THIN reduction (<= 40 channels): EfficientNet's expand convs forward (16 -> 96, 24 -> 144, 40 -> 240:
// /root/reference/embedding_net/backbones.py:84-98 via the efficientnet package's MBConv) and its project convs' data gradient
// (dy with 16 / 24 / 40 channels -> dx with 96 / 144 / 240).
//
// Why not the MFMA kernels of conv.hip: these layers are HBM streams — 23 FLOP per output byte at 16 -> 96 — whose output is
// 6x their input.  On the implicit-GEMM kernel a workgroup lives for ONE K tile (half of it padding), pays a prologue round
// trip and writes its 128 x 64 tile as 256-byte row pieces: round 5 measured the expand convs at 2.7 - 2.8 TB/s and the project
// data gradients at 2.0 - 3.0 (profiles/r05_c5_conv_launches.txt: 509 us for 1.44 GB at 112^2), against 4.2 - 4.9 for their
// read-dominated counterparts.  Here the output is written as ONE contiguous float4 stream:
//   out[m][n] = sum_{r < R} in[m][r] * B[r][n]          (N = 4 NQ columns)
//  * thread (pixel slot ps, column quad q) owns PIX pixels ps, ps + TP, ... of its workgroup's group and one float4 of each
//    output row: consecutive threads write consecutive 16-byte pieces of the dense [M][N] output;
//  * B (<= 48 KB) sits in LDS, one ds_read_b128 per (r, thread) serves PIX pixels; the input tile of a group is fetched by the
//    whole workgroup with coalesced 16-byte loads one group ahead and read back as LDS broadcasts (a first version loaded each
//    pixel's quads per thread, four dependent round trips per group: 2.5 TB/s, slower than the MFMA kernel);
//    R x 4 x PIX fp32 FMAs per thread — the VALU needs 60 us where HBM needs 290 at 16 -> 96, 112^2, batch 256;
//  * exact fp32 FMA chains over r in ascending order (the MFMA kernels' six-term products agree with them within fp32 rounding;
//    the whole-net parity tests run on this path);
//  * a workgroup walks GROUPS groups of TP x PIX pixels and, when asked, writes the per-channel sum and sum of squares of
//    everything it produced as ONE row of the BatchNorm statistics partials [2][N][P] (the layout of conv.hip's epilogues).
#include "common.h"
#include "conv_geom.h"
#include "../../include/embnet.h"

namespace embnet {
namespace thin {

constexpr int PIX = 8;
constexpr int MAX_B_FLOATS = 12288;          // 48 KB of LDS for B
constexpr int MAX_TILE4 = 1024;              // float4s of one input tile (TP * PIX pixels x R / 4): at most 4 per thread

struct Params {
  const float* in; const float* w; float* out; float* stats; const float* bias; int relu; const float* residual;
  long m; int R, NQ, TP, groups, wt;         // wt: 0 = w is [R][N], 1 = w is [N][R] (data gradient: B = W^T)
  int stats_P;
  // strided 1x1 forward: output pixel (n, oh, ow) reads input pixel (n, oh*st, ow*st); st == 1: same index
  int st, H, W, OH, OW; FastDiv dOHW, dOW, dR4;
};

// The input tile of a group — GP = TP * PIX consecutive output pixels x R channels, GP * R / 4 float4s, contiguous in memory at
// stride 1 — is fetched by the whole workgroup with coalesced 16-byte loads, one group AHEAD of the arithmetic (registers ->
// LDS behind a barrier pair), so a workgroup always has a tile in flight; the arithmetic reads x as LDS broadcasts.
__global__ __launch_bounds__(256, 3) void thin_gemm_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, N = 4 * p.NQ, R = p.R, R4 = R >> 2;
  const int GP = p.TP * PIX, T4 = GP * R4;                       // pixels / float4s of a group's input tile
  float4* const xs0 = reinterpret_cast<float4*>(lds + (size_t)R * N);
  if (p.wt) {                                                    // B = W^T: W is [N][R]
    for (int n = tid; n < N; n += 256)
      for (int r = 0; r < R; ++r) lds[r * N + n] = p.w[(long)n * R + r];
  } else {
    for (int i = tid; i < R * N; i += 256) lds[i] = p.w[i];
  }
  const int ps = tid / p.NQ, q = tid - ps * p.NQ;
  const bool active = ps < p.TP;
  const float4* in4 = reinterpret_cast<const float4*>(p.in);
  const float4* b4 = reinterpret_cast<const float4*>(lds);
  float4* out4 = reinterpret_cast<float4*>(p.out);
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 s1 = z4, s2 = z4;
  const long first = (long)blockIdx.x * p.groups * GP;
  float4 pre[4];
  // (branch-free: every slot loads from a clamped, valid address and is zeroed afterwards — a load under a branch makes hipcc
  // wait for it on the spot, which turned the prefetch into one memory round trip per slot)
  auto fetch = [&](int g) {                                      // group g's tile -> registers (zeros past the end)
    const long base = first + (long)g * GP;
    long addr[4]; bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + u * 256;
      uint32_t pl, r4; p.dR4.divmod((uint32_t)(i < T4 ? i : 0), pl, r4);      // pixel of the tile, channel quad
      long pix = base + pl;
      ok[u] = i < T4 && g < p.groups && pix < p.m;
      if (!ok[u]) pix = 0;
      if (p.st != 1) {
        uint32_t n, rem, oh, ow;
        p.dOHW.divmod((uint32_t)pix, n, rem); p.dOW.divmod(rem, oh, ow);
        pix = ((long)n * p.H + (long)oh * p.st) * p.W + (long)ow * p.st;
      }
      addr[u] = pix * R4 + r4;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) pre[u] = in4[addr[u]];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (!ok[u]) pre[u] = z4;
  };
  fetch(0);
  for (int g = 0; g < p.groups; ++g) {
    const long base = first + (long)g * GP;
    if (base >= p.m) break;
    __syncthreads();                                             // the previous group's reads of the tile (and B's fill) are done
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = tid + u * 256; if (i < T4) xs0[i] = pre[u]; }
    __syncthreads();
    fetch(g + 1);                                                // in flight while this group is computed
    float4 acc[PIX];
#pragma unroll
    for (int j = 0; j < PIX; ++j) acc[j] = z4;
    const int psa = active ? ps : 0, qa = active ? q : 0;
#pragma unroll 1
    for (int r4 = 0; r4 < R4; ++r4) {
      float4 xv[PIX];
#pragma unroll
      for (int j = 0; j < PIX; ++j) xv[j] = xs0[(psa + j * p.TP) * R4 + r4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float4 w4 = b4[(4 * r4 + e) * p.NQ + qa];
#pragma unroll
        for (int j = 0; j < PIX; ++j) {
          const float xsv = e == 0 ? xv[j].x : (e == 1 ? xv[j].y : (e == 2 ? xv[j].z : xv[j].w));
          acc[j].x = fmaf(xsv, w4.x, acc[j].x); acc[j].y = fmaf(xsv, w4.y, acc[j].y);
          acc[j].z = fmaf(xsv, w4.z, acc[j].z); acc[j].w = fmaf(xsv, w4.w, acc[j].w);
        }
      }
    }
    if (p.bias || p.relu) {
      const float4 bv = p.bias ? reinterpret_cast<const float4*>(p.bias)[qa] : z4;
#pragma unroll
      for (int j = 0; j < PIX; ++j) {
        acc[j].x += bv.x; acc[j].y += bv.y; acc[j].z += bv.z; acc[j].w += bv.w;
        if (p.relu) { acc[j].x = fmaxf(acc[j].x, 0.f); acc[j].y = fmaxf(acc[j].y, 0.f); acc[j].z = fmaxf(acc[j].z, 0.f); acc[j].w = fmaxf(acc[j].w, 0.f); }
      }
    }
#pragma unroll
    for (int j = 0; j < PIX; ++j) {
      const long po = base + ps + (long)j * p.TP;
      if (active && po < p.m) {
        if (p.residual) {                                        // the Add behind the conv (the statistics are of the sum)
          const float4 rv = reinterpret_cast<const float4*>(p.residual)[po * p.NQ + q];
          acc[j].x += rv.x; acc[j].y += rv.y; acc[j].z += rv.z; acc[j].w += rv.w;
        }
        out4[po * p.NQ + q] = acc[j];
        s1.x += acc[j].x; s1.y += acc[j].y; s1.z += acc[j].z; s1.w += acc[j].w;
        s2.x = fmaf(acc[j].x, acc[j].x, s2.x); s2.y = fmaf(acc[j].y, acc[j].y, s2.y);
        s2.z = fmaf(acc[j].z, acc[j].z, s2.z); s2.w = fmaf(acc[j].w, acc[j].w, s2.w);
      }
    }
  }
  if (p.stats) {                       // one row of the [2][N][P] partials per workgroup: pixel slots summed in slot order
    __syncthreads();                   // B and the tile are dead: reuse LDS
    float4* sc = reinterpret_cast<float4*>(lds);
    if (active) { sc[ps * p.NQ + q] = s1; sc[(p.TP + ps) * p.NQ + q] = s2; }
    __syncthreads();
    if (tid < p.NQ) {
      float4 a = z4, b = z4;
      for (int t = 0; t < p.TP; ++t) {
        const float4 u = sc[t * p.NQ + tid], v = sc[(p.TP + t) * p.NQ + tid];
        a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w; b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
      }
      const long P = p.stats_P;
      float* d1 = p.stats + (long)(4 * tid) * P + blockIdx.x;
      float* d2 = d1 + (long)N * P;
      d1[0] = a.x; d1[P] = a.y; d1[2 * P] = a.z; d1[3 * P] = a.w;
      d2[0] = b.x; d2[P] = b.y; d2[2 * P] = b.z; d2[3 * P] = b.w;
    }
  }
}

}  // namespace thin

// ---- weight gradient of the same layers: T[r][n] = sum_m A[m][r] * Bw[m][n], A the THIN operand (x of an expand conv, dy of a
// project conv), Bw the wide one — a reduction over pixels whose inputs are streams and whose output is tiny.  conv.hip's split-K
// kernel moves it at 2.9-4.2 TB/s (one 32-pixel K tile in flight per workgroup).  Here a workgroup walks tiles of TPIX pixels
// (both operands, up to 32 KB, fetched with coalesced 16-byte loads one tile AHEAD: registers -> LDS behind a barrier pair);
// thread (pixel slot, row group, column quad) keeps RQ x 4 rows x one float4 of T in registers and adds its slot's pixels (x
// as LDS broadcasts); the slots' sums meet in LDS in slot order; one fp32 slab per workgroup, summed in fixed order by
// conv.hip's slab_reduce (bitwise reproducible).  Exact fp32 FMA chains over the pixels of a slot.
namespace thinw {

constexpr int PRE = 8;                       // float4s a thread prefetches: tiles of up to 256 * 8 float4 = 32 KB

struct Params {
  const float* a; const float* b; float* slabs;
  long m; int R, NQ, RG, TP, TPIX, tiles_per_block, transposed;
  // the conv INPUT (x) may be strided against the output pixels that index dy: a_is_x says which operand x is
  int a_is_x, st, H, W; FastDiv dOHW, dOW, dR4, dNQ;
};

template <int RQ>
__global__ __launch_bounds__(256, RQ <= 1 ? 3 : 2) void thin_wgrad_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, R = p.R, R4 = R >> 2, NQ = p.NQ, N = 4 * NQ;
  const int TA4 = p.TPIX * R4, T4 = TA4 + p.TPIX * NQ;
  float4* const as4 = reinterpret_cast<float4*>(lds);
  float4* const bs4 = as4 + TA4;
  const float4* a4g = reinterpret_cast<const float4*>(p.a);
  const float4* b4g = reinterpret_cast<const float4*>(p.b);
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int q = tid % NQ, t2 = tid / NQ, rg = t2 % p.RG, ps = t2 / p.RG;
  const bool active = ps < p.TP;
  const long first = (long)blockIdx.x * p.tiles_per_block * p.TPIX;
  float4 pre[PRE];
  auto fetch = [&](int t) {                                      // branch-free, as thin_gemm_kernel's
    const long base = first + (long)t * p.TPIX;
    const float4* src[PRE]; bool ok[PRE];
#pragma unroll
    for (int u = 0; u < PRE; ++u) {
      const int i = tid + u * 256;
      const bool is_a = i < TA4;
      uint32_t pl, qd;
      if (is_a) p.dR4.divmod((uint32_t)i, pl, qd); else p.dNQ.divmod((uint32_t)(i < T4 ? i - TA4 : 0), pl, qd);
      long pix = base + pl;
      ok[u] = i < T4 && t < p.tiles_per_block && pix < p.m;
      if (!ok[u]) pix = 0;
      if (p.st != 1 && (is_a == (p.a_is_x != 0))) {              // this operand is the strided conv input
        uint32_t n, rem, oh, ow;
        p.dOHW.divmod((uint32_t)pix, n, rem); p.dOW.divmod(rem, oh, ow);
        pix = ((long)n * p.H + (long)oh * p.st) * p.W + (long)ow * p.st;
      }
      src[u] = is_a ? a4g + (pix * R4 + qd) : b4g + (pix * NQ + qd);
    }
#pragma unroll
    for (int u = 0; u < PRE; ++u) pre[u] = *src[u];
#pragma unroll
    for (int u = 0; u < PRE; ++u) if (!ok[u]) pre[u] = z4;
  };
  float4 acc[4 * RQ];
#pragma unroll
  for (int i = 0; i < 4 * RQ; ++i) acc[i] = z4;
  fetch(0);
  for (int t = 0; t < p.tiles_per_block; ++t) {
    if (first + (long)t * p.TPIX >= p.m) break;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PRE; ++u) { const int i = tid + u * 256; if (i < T4) as4[i] = pre[u]; }
    __syncthreads();
    fetch(t + 1);
    if (active) {
      for (int pl = ps; pl < p.TPIX; pl += p.TP) {
        const float4 bq = bs4[pl * NQ + q];
#pragma unroll
        for (int r4 = 0; r4 < RQ; ++r4) {
          const float4 av = as4[pl * R4 + rg * RQ + r4];
          acc[4 * r4 + 0].x = fmaf(av.x, bq.x, acc[4 * r4 + 0].x); acc[4 * r4 + 0].y = fmaf(av.x, bq.y, acc[4 * r4 + 0].y);
          acc[4 * r4 + 0].z = fmaf(av.x, bq.z, acc[4 * r4 + 0].z); acc[4 * r4 + 0].w = fmaf(av.x, bq.w, acc[4 * r4 + 0].w);
          acc[4 * r4 + 1].x = fmaf(av.y, bq.x, acc[4 * r4 + 1].x); acc[4 * r4 + 1].y = fmaf(av.y, bq.y, acc[4 * r4 + 1].y);
          acc[4 * r4 + 1].z = fmaf(av.y, bq.z, acc[4 * r4 + 1].z); acc[4 * r4 + 1].w = fmaf(av.y, bq.w, acc[4 * r4 + 1].w);
          acc[4 * r4 + 2].x = fmaf(av.z, bq.x, acc[4 * r4 + 2].x); acc[4 * r4 + 2].y = fmaf(av.z, bq.y, acc[4 * r4 + 2].y);
          acc[4 * r4 + 2].z = fmaf(av.z, bq.z, acc[4 * r4 + 2].z); acc[4 * r4 + 2].w = fmaf(av.z, bq.w, acc[4 * r4 + 2].w);
          acc[4 * r4 + 3].x = fmaf(av.w, bq.x, acc[4 * r4 + 3].x); acc[4 * r4 + 3].y = fmaf(av.w, bq.y, acc[4 * r4 + 3].y);
          acc[4 * r4 + 3].z = fmaf(av.w, bq.z, acc[4 * r4 + 3].z); acc[4 * r4 + 3].w = fmaf(av.w, bq.w, acc[4 * r4 + 3].w);
        }
      }
    }
  }
  // the pixel slots' sums meet in LDS, slot TP-1 first, slot 0 adds them up in that order and writes the workgroup's slab
  float4* sc = reinterpret_cast<float4*>(lds);                   // [RG * 4 RQ rows][NQ]
  const int row0 = rg * 4 * RQ;
  for (int s = p.TP - 1; s >= 1; --s) {
    __syncthreads();
    if (active && ps == s) {
#pragma unroll
      for (int i = 0; i < 4 * RQ; ++i) sc[(row0 + i) * NQ + q] = acc[i];
    }
    __syncthreads();
    if (active && ps == 0) {
#pragma unroll
      for (int i = 0; i < 4 * RQ; ++i) {
        const float4 v = sc[(row0 + i) * NQ + q];
        acc[i].x += v.x; acc[i].y += v.y; acc[i].z += v.z; acc[i].w += v.w;
      }
    }
  }
  if (active && ps == 0) {
    float* slab = p.slabs + (long)blockIdx.x * R * N;
#pragma unroll
    for (int i = 0; i < 4 * RQ; ++i) {
      const int r = row0 + i;
      if (!p.transposed) *reinterpret_cast<float4*>(slab + (long)r * N + 4 * q) = acc[i];
      else { slab[(long)(4 * q + 0) * R + r] = acc[i].x; slab[(long)(4 * q + 1) * R + r] = acc[i].y;
             slab[(long)(4 * q + 2) * R + r] = acc[i].z; slab[(long)(4 * q + 3) * R + r] = acc[i].w; }
    }
  }
}

}  // namespace thinw

// ---- host side (used by conv.hip's forward / data-gradient entry points) ---------------------------------------------------
static bool thin_enabled() { static const bool v = env_long("EMBNET_CONV_THIN", 1) != 0; return v; }

// a 1x1 product out[m, ncols] = in[m, red] * B with a thin reduction: shape-only decision (the statistics' row count must be
// known from the same arguments embnet_conv2d_fwd_stats_rows gets)
bool thin_gemm_applies(int red, int ncols) {
  return thin_enabled() && red >= 4 && red <= 40 && (red & 3) == 0 && (ncols & 3) == 0 && ncols >= 8 && ncols <= 1024 &&
         (long)red * ncols <= thin::MAX_B_FLOATS && (256 / (ncols / 4)) * thin::PIX * (red / 4) <= thin::MAX_TILE4;
}
static void thin_plan(long m, int ncols, int& tp, int& groups, int& blocks) {
  const int nq = ncols / 4;
  tp = 256 / nq;
  const long per_group = (long)tp * thin::PIX;
  // three workgroups are resident per CU (168 registers): ~768 long-lived workgroups, each filling B once and then streaming
  // groups with a tile in flight (a first plan of ~8k short workgroups spent a 28x28 layer's time filling B: 140 vs 79 us)
  static const long target = env_long("EMBNET_THIN_BLOCKS", 768);
  long g = (m + per_group * target - 1) / (per_group * target);
  groups = (int)(g < 1 ? 1 : g);
  blocks = (int)((m + per_group * groups - 1) / (per_group * groups));
}
int thin_gemm_stats_rows(long m, int ncols) { int tp, g, b; thin_plan(m, ncols, tp, g, b); return b; }

int launch_thin_gemm(const float* in, const float* w, int w_transposed, long m, int red, int ncols, float* out, float* stats,
                     const float* bias, int relu, const float* residual, int stride, int n, int h, int wd, int oh, int ow, hipStream_t st) {
  thin::Params p{};
  p.in = in; p.w = w; p.out = out; p.stats = stats; p.m = m; p.R = red; p.NQ = ncols / 4; p.wt = w_transposed;
  p.bias = bias; p.relu = relu; p.residual = residual;
  int blocks; thin_plan(m, ncols, p.TP, p.groups, blocks);
  p.stats_P = blocks;
  p.st = stride; p.H = h; p.W = wd; p.OH = oh; p.OW = ow;
  p.dOHW = FastDiv::make((uint32_t)(oh * ow)); p.dOW = FastDiv::make((uint32_t)ow); p.dR4 = FastDiv::make((uint32_t)(red / 4));
  size_t lds = (size_t)red * ncols * sizeof(float) + (size_t)p.TP * thin::PIX * red * sizeof(float);
  const size_t sred = (size_t)2 * p.TP * p.NQ * 16;
  if (lds < sred) lds = sred;
  EMBNET_TRACE("embnet::thin::thin_gemm_kernel", TRACE_BYTES, 4.0 * ((double)m * red + (double)m * ncols + (double)red * ncols), st);
  thin::thin_gemm_kernel<<<blocks, 256, lds, st>>>(p);
  return check_launch("thin_gemm");
}


// weight gradient dW[c][k] of a 1x1 conv with min(c, k) thin: shape-only decision (embnet_conv2d_wgrad_splits must know it)
static bool thinw_groups(int thin, int& rg, int& rq) {
  switch (thin) {
    case 4: case 8: case 12: case 16: case 20: rg = 1; rq = thin / 4; return true;
    case 24: case 32: case 40: rg = 2; rq = thin / 8; return true;
    default: return false;
  }
}
bool thin_wgrad_applies(int c, int k) {
  static const bool on = env_long("EMBNET_CONV_THIN_WGRAD", 1) != 0;
  const int thin_dim = c < k ? c : k, wide = c < k ? k : c;
  int rg, rq;
  if (!on || !thin_enabled() || !thinw_groups(thin_dim, rg, rq) || (wide & 3) || wide < 8 || wide > 1024) return false;
  if ((long)thin_dim * wide > thin::MAX_B_FLOATS) return false;
  const int nq = wide / 4;
  if (nq * rg > 256) return false;
  return (256 * thinw::PRE) / (thin_dim / 4 + nq) >= 8;          // at least 8 pixels per tile
}
static void thinw_plan(long m, int thin_dim, int wide, int& tpix, int& tiles_per_block, int& blocks) {
  tpix = (256 * thinw::PRE) / (thin_dim / 4 + wide / 4);
  if (tpix > 128) tpix = 128;
  tpix &= ~3;
  const long tiles = (m + tpix - 1) / tpix;
  // whole rounds of resident workgroups: three per CU with 4 accumulator rows (<= 168 registers), two beyond
  static const long forced = env_long("EMBNET_THIN_WGRAD_BLOCKS", 0);
  int rg, rq; thinw_groups(thin_dim, rg, rq);
  const long target = forced > 0 ? forced : (rq <= 1 ? 768 : 512);
  long per = (tiles + target - 1) / target;
  if (per < 4) per = 4;                                           // a workgroup's slab and its fill are worth at least four tiles
  tiles_per_block = (int)per;
  blocks = (int)((tiles + per - 1) / per);
}
int thin_wgrad_splits(long m, int c, int k) {
  int tpix, per, blocks; thinw_plan(m, c < k ? c : k, c < k ? k : c, tpix, per, blocks); return blocks;
}

// slabs [blocks][c*k] in dW's layout [c][k]; returns the launch status.  x [*, c] (strided by `stride` against dy's pixels), dy [m, k]
int launch_thin_wgrad(const float* x, const float* dy, float* slabs, long m, int c, int k, int stride, int n, int h, int wd, int oh,
                      int ow, hipStream_t st) {
  thinw::Params p{};
  const bool x_thin = c < k;
  const int thin_dim = x_thin ? c : k, wide = x_thin ? k : c;
  int rq; thinw_groups(thin_dim, p.RG, rq);
  p.a = x_thin ? x : dy; p.b = x_thin ? dy : x; p.slabs = slabs; p.m = m; p.R = thin_dim; p.NQ = wide / 4;
  p.TP = 256 / (p.NQ * p.RG);
  int blocks; thinw_plan(m, thin_dim, wide, p.TPIX, p.tiles_per_block, blocks);
  if (p.TP > p.TPIX) p.TP = p.TPIX;
  p.transposed = x_thin ? 0 : 1;                                  // T is [thin][wide]; dW is [c][k]
  p.a_is_x = x_thin ? 1 : 0; p.st = stride; p.H = h; p.W = wd;
  p.dOHW = FastDiv::make((uint32_t)(oh * ow)); p.dOW = FastDiv::make((uint32_t)ow);
  p.dR4 = FastDiv::make((uint32_t)(thin_dim / 4)); p.dNQ = FastDiv::make((uint32_t)p.NQ);
  size_t lds = (size_t)p.TPIX * (thin_dim + wide) * sizeof(float);
  const size_t red = (size_t)thin_dim * wide * sizeof(float);
  if (lds < red) lds = red;
  EMBNET_TRACE("embnet::thinw::thin_wgrad_kernel", TRACE_BYTES, 4.0 * ((double)m * (c + k) + (double)c * k * blocks), st);
  switch (rq) {
    case 1: thinw::thin_wgrad_kernel<1><<<blocks, 256, lds, st>>>(p); break;
    case 2: thinw::thin_wgrad_kernel<2><<<blocks, 256, lds, st>>>(p); break;
    case 3: thinw::thin_wgrad_kernel<3><<<blocks, 256, lds, st>>>(p); break;
    case 4: thinw::thin_wgrad_kernel<4><<<blocks, 256, lds, st>>>(p); break;
    default: thinw::thin_wgrad_kernel<5><<<blocks, 256, lds, st>>>(p); break;
  }
  return check_launch("thin_wgrad");
}

}  // namespace embnet

extern "C" int embnet_conv1x1_thin_supported(int red, int ncols) { return embnet::thin_gemm_applies(red, ncols) ? 1 : 0; }
