// The zoo ResNets' stem convolution — 7x7, stride 2, 3 (padded to 4) -> 64 channels on the 224x224 image
// (/root/reference/embedding_net/backbones.py:99-104 via image-classifiers: bn_data -> ZeroPadding2D(3) -> conv0) — forward, as a
// kernel of its own.
//
// Why: on the implicit-GEMM loop (conv.hip) this layer is 12 544 tiles of 128 pixels x 64 filters with a reduction of seven 32-deep K
// tiles; every output pixel issues 49 sixteen-byte gathers for its 196 multiply-adds x 64 filters, the loop runs at the rate the L1
// takes those requests (244 us at batch 128, 0.16 of what the matrix pipe could do), and neither more resident workgroups nor fatter
// tiles help (EMBNET_FWD_256).  Here every input pixel is fetched ONCE per tile:
//  * a workgroup (persistent) takes 16 x 16 output pixels of one image x all 64 filters; the input PATCH those outputs read
//    (37 x 37 pixels x 16 B, fp32, zero outside the image) reaches LDS by LDS-DMA from a loader wave, two tiles ahead (three
//    buffers, one s_barrier per tile);
//  * the kernel — 7 x (7 + 1 zero tap) x 4 channels = 224 reduction elements x 64 filters — is split into its two fp16 pieces by the
//    workgroup itself, once, into LDS (56 KB; planes-kernel row layout), so there is no weight-planes tensor to keep current;
//  * a k-step of the 32x32x16 matrix instruction is (kernel row r, four taps, four channels): a lane's eight reduction elements are
//    TWO ADJACENT input pixels, 32 contiguous bytes of the patch row 2 oy + r — two ds_read_b128 — which the matrix waves split into
//    the two fp16 pieces of x s (gemm_engine.h split4h; s from the activation's range slot) behind the other wave's matrix
//    instructions.  Patch pixel column q is stored at column q ^ ((q >> 4) & 1): the sixteen lanes of an output row read every
//    second pixel (stride 32 B), and the flip puts lanes 8..15 on the odd 16-byte slots — conflict-free;
//  * eight matrix waves, each two output rows (32 pixels) x 64 filters: 14 k-steps x 6 matrix instructions; three fp16 products per
//    fp32 product, the sums x 1 / s_w x 1 / s_x in the epilogue; the epilogue also leaves the BatchNorm statistics of the layer
//    behind (sum, sum of squares per filter and 32-pixel band) like the other conv epilogues.
// Algorithmic work 2 * M * 196 * 64 (the zero tap and the pad channel are not counted); HBM: the image once + the output once.
// Where its time goes (batch 128, back to back with the statistics, tools/exp/stem_bench.py on build variants; in the C2 step the
// kernel takes 171 us against 244 for the gather loop): 61 us with neither the matrix loop nor the output stores (the loader's
// round trips), + 67 us for the stores (411 MB: the HBM write rate), + 108 us for the matrix loop with its LDS reads and splits —
// and the three ADD UP (227 us): the eight waves of a workgroup run loop and epilogue one after the other, in step with each other.
// Tried to overlap them, measured, not kept (tools/exp/conv_stem_two_groups.diff): two groups of four waves on alternate tiles
// half a tile apart (233 us: one multiplying wave per SIMD leaves the matrix pipe 60 % idle, and its issue slots compete with the
// storing wave's); two workgroups per CU on 8 x 16 tiles with one patch buffer each (255 us); a bare s_barrier instead of
// __syncthreads() (which also waits for the stores' acknowledgement) and branch-free interior-tile stores are in (230 -> 222 us);
// the previous tile's stores issued four per k-step behind the next tile's matrix instructions (tools/exp/conv_stem_store_behind.diff):
// 245 us — a store in the loop holds the wave's issue slot longer than it saves.
#include "gemm_engine.h"
#include "conv_geom.h"
#include "../../include/embnet.h"

namespace embnet {
namespace stem {

typedef __attribute__((address_space(3))) void* lds_ptr;
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned char* lds, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)lds, 16, (int)voff, 0, 0, 0);
}

constexpr int R = 7, S = 7, C = 4, K = 64, TH = 16, TW = 16;
constexpr int PR = 2 * TH + R - 2, PC = 2 * TW + S - 1;       // 37 rows x (37 + 1: the zero tap's column must hold finite values) input pixels
constexpr int PWL = 40;                                       // patch row pitch in pixels (even: the column flip stays inside a row)
constexpr int PIECES = (PR * PWL + 63) / 64;                  // 64-pixel DMA pieces per patch (24)
constexpr int PATCH_BYTES = PIECES * 64 * 16;
constexpr int STEPS = R * 2;                                  // k-steps: (kernel row, taps 0..3 | 4..6 + a zero tap)
constexpr int W_BYTES = STEPS * 2 * K * 32;
constexpr int NBUF = 3;                                       // patch buffers: the loader runs two tiles ahead
constexpr int LDS_BYTES = W_BYTES + NBUF * PATCH_BYTES;

struct StemParams {
  const float* x; const float* w; float* y; float* stats;
  int N, H, W, OH, OW, pad_t, pad_l, tiles_x, tiles_y, tiles, stats_rows;
  const uint32_t* a_range; const uint32_t* w_range;
  unsigned x_bytes;
};

__device__ __forceinline__ int flip(int q) { return q ^ ((q >> 4) & 1); }

__global__ __launch_bounds__(576) void conv_stem_kernel(const StemParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const wl = smem;                       // [STEPS][2 pieces][64 filters][32 B]
  unsigned char* const patch0 = smem + W_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_mine = ((int)blockIdx.x < p.tiles) ? (p.tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  if (n_mine == 0) return;
  const float2 sw = scale_pair(scale_exponent_of(__uint_as_float(*p.w_range)));
  const float2 sa = scale_pair(scale_exponent_of(__uint_as_float(*p.a_range)));

  // ---- the kernel's two fp16 pieces, once per workgroup: element j of step t = (r, g) is tap s = 4 g + j / 4, channel j % 4 ------
  for (int idx = tid; idx < STEPS * 16 * K; idx += 576) {
    const int f = idx % K, j = (idx / K) % 16, t = idx / (K * 16);
    const int r = t >> 1, s = 4 * (t & 1) + (j >> 2), ch = j & 3;
    const float v = (s < S ? p.w[((r * S + s) * C + ch) * K + f] : 0.f) * sw.x;
    const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
    unsigned char* row = wl + ((t * 2) * K + f) * 32 + ((((j >> 3) ^ ((f >> 3) & 1))) << 4) + (j & 7) * 2;
    *reinterpret_cast<_Float16*>(row) = hi;
    *reinterpret_cast<_Float16*>(row + K * 32) = lo;
  }
  __syncthreads();

  if (wave == 8) {
    // ---- patch loader: buffer (i % 3) <- the patch of this workgroup's tile i, TWO tiles ahead of the matrix waves (a DMA round
    // trip is ~2.5 us under load, a tile ~4) ------------------------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    int rel[PIECES], rc[PIECES];                                 // per lane and piece: pixel offset inside the patch window, (row, column) packed
#pragma unroll
    for (int pc = 0; pc < PIECES; ++pc) {
      const int slot = pc * 64 + lane, prow = slot / PWL, pcl = flip(slot - prow * PWL);
      const bool in = prow < PR && pcl < PC;
      rel[pc] = prow * p.W + pcl;
      rc[pc] = in ? (prow << 16) | pcl : -1;
    }
    auto issue = [&](int i) {
      const int tile = (int)blockIdx.x + i * (int)gridDim.x;
      const int n = tile / (p.tiles_x * p.tiles_y), rem = tile - n * (p.tiles_x * p.tiles_y);
      const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
      const int ih0 = 2 * TH * ty - p.pad_t, iw0 = 2 * TW * tx - p.pad_l;
      const int base = (n * p.H + ih0) * p.W + iw0;
      unsigned char* buf = patch0 + (i % NBUF) * PATCH_BYTES;
#pragma unroll
      for (int pc = 0; pc < PIECES; ++pc) {
        const int ih = ih0 + (rc[pc] >> 16), iw = iw0 + (rc[pc] & 0xffff);
        const bool ok = rc[pc] >= 0 && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        dma16(xr, buf + pc * 1024, ok ? 16u * (unsigned)(base + rel[pc]) : OOB);
      }
    };
    issue(0);
    if (n_mine > 1) issue(1);
    for (int i = 0; i < n_mine; ++i) {
      // the patch of tile i has landed (the next tile's pieces may be in flight)
      if (i + 1 < n_mine) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PIECES) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                              // #i: the matrix waves are done with tile i - 1 (buffer (i + 2) % 3)
      if (i + 2 < n_mine) issue(i + 2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ---- matrix waves: wave w = output rows 2 w, 2 w + 1 of the tile (32 pixels) x 64 filters ------------------------------------
  const int pl = lane & 31, oy_l = 2 * wave + (pl >> 4), ox_l = pl & 15;
  int colb[2][2];                                       // byte offset of this lane's two pixels of tap group g inside a patch row
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int j = 0; j < 2; ++j) colb[g][j] = flip(2 * ox_l + 4 * g + 2 * h + j) * 16;
  const int rowb = 2 * oy_l * PWL * 16;
  int boff[2];
#pragma unroll
  for (int in = 0; in < 2; ++in) { const int f = in * 32 + pl; boff[in] = f * 32 + ((h ^ ((f >> 3) & 1)) << 4); }
  for (int i = 0; i < n_mine; ++i) {
    // barrier #i: the patch of tile i is in LDS.  (A bare s_barrier: __syncthreads() would also wait for the acknowledgement of
    // this wave's output stores; what must be complete is its LDS reads of the previous patch, and the matrix instructions consumed them.)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned char* buf = patch0 + (i % NBUF) * PATCH_BYTES + rowb;
    f32x16 acc[1][2];
#pragma unroll
    for (int in = 0; in < 2; ++in)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) acc[0][in][rr] = 0.f;
    float4 raw[2];
    f16x8 fa[2][1][2], fb[2][2][2];
    auto load = [&](int t, f16x8 (&bb)[2][2]) {
      const int r = t >> 1, g = t & 1;
      raw[0] = *reinterpret_cast<const float4*>(buf + r * (PWL * 16) + colb[g][0]);
      raw[1] = *reinterpret_cast<const float4*>(buf + r * (PWL * 16) + colb[g][1]);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int in = 0; in < 2; ++in) bb[in][q] = *reinterpret_cast<const f16x8*>(wl + ((t * 2 + q) * K) * 32 + boff[in]);
    };
    auto split = [&](f16x8 (&a)[1][2]) {
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      const Split4H s0 = split4h(raw[0], sa.x), s1 = split4h(raw[1], sa.x);
      a[0][0] = __builtin_bit_cast(f16x8, u32x4{s0.p[0].x, s0.p[0].y, s1.p[0].x, s1.p[0].y});
      a[0][1] = __builtin_bit_cast(f16x8, u32x4{s0.p[1].x, s0.p[1].y, s1.p[1].x, s1.p[1].y});
    };
    load(0, fb[0]);
    split(fa[0]);
#ifndef STEM_NO_MFMA                                         // (STEM_NO_*: timing variants for tools/exp/stem_bench.py, never in the library)
#pragma unroll
    for (int t = 0; t < STEPS; ++t) {
      if (t + 1 < STEPS) load(t + 1, fb[(t + 1) & 1]);
      mfma_step_h<1, 2>(fa[t & 1], fb[t & 1], acc);
      if (t + 1 < STEPS) split(fa[(t + 1) & 1]);
    }
#endif
    // ---- epilogue: 1 / (s_w s_x), the outputs, the statistics of this wave's 32-pixel band -------------------------------------
    // register rr of a 32x32 block holds block row (rr & 3) + 8 (rr >> 2) + 4 h = output row (rr >> 3), column
    // 8 ((rr >> 2) & 1) + 4 h + (rr & 3) of this wave's two output rows; lanes 0..31 are 32 consecutive filters (128 B per store)
    const int tile = (int)blockIdx.x + i * (int)gridDim.x;
    const int n = tile / (p.tiles_x * p.tiles_y), rem = tile - n * (p.tiles_x * p.tiles_y);
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    const int oy0 = TH * ty + 2 * wave, ox0 = TW * tx;
    const bool full = oy0 + 2 <= p.OH && ox0 + TW <= p.OW;          // (wave-uniform) no edge tests on interior tiles
    float* const yb = p.y + ((long)(n * p.OH + oy0) * p.OW + ox0 + 4 * h) * K + pl;
#pragma unroll
    for (int in = 0; in < 2; ++in) {
      float s1 = 0.f, s2 = 0.f;
      if (full) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const float v = (acc[0][in][rr] * sw.y) * sa.y;
#ifndef STEM_NO_STORE
          yb[((long)(rr >> 3) * p.OW + 8 * ((rr >> 2) & 1) + (rr & 3)) * K + in * 32] = v;
#endif
          s1 += v; s2 = fmaf(v, v, s2);
        }
      } else {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const int oy = oy0 + (rr >> 3), ox = ox0 + 8 * ((rr >> 2) & 1) + 4 * h + (rr & 3);
          const float v = (acc[0][in][rr] * sw.y) * sa.y;
          if (oy < p.OH && ox < p.OW) {
#ifndef STEM_NO_STORE
            yb[((long)(rr >> 3) * p.OW + 8 * ((rr >> 2) & 1) + (rr & 3)) * K + in * 32] = v;
#endif
            s1 += v; s2 = fmaf(v, v, s2);
          }
        }
      }
      if (p.stats) {
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (h == 0) {
          const long prow = (long)tile * 8 + wave, P = p.stats_rows;
          p.stats[(long)(in * 32 + pl) * P + prow] = s1;
          p.stats[((long)K + in * 32 + pl) * P + prow] = s2;
        }
      }
    }
  }
}

}  // namespace stem
}  // namespace embnet

using namespace embnet;

static bool stem_ok(int n, int h, int wd, int c, int r, int s, int k, int stride, int pad_t, int pad_l, int oh, int ow) {
  static const int enabled = (int)env_long("EMBNET_CONV_STEM", 1);
  if (!enabled || embnet_conv_planes_mfma_terms() != 3) return false;
  if (c != stem::C || r != stem::R || s != stem::S || k != stem::K || stride != 2 || n <= 0 || oh <= 0 || ow <= 0) return false;
  if (pad_t < 0 || pad_l < 0 || pad_t >= stem::R || pad_l >= stem::S) return false;       // (taps outside the image read zeros, as padding does)
  return (size_t)n * h * wd * 16 < 0x7FFFFFF0ull && (size_t)n * oh * ow * stem::K * 4 < 0x7FFFFFFF0ull;
}
extern "C" int embnet_conv2d_stem_supported(int n, int h, int wd, int c, int r, int s, int k, int stride, int pad_t, int pad_l, int oh, int ow) {
  return stem_ok(n, h, wd, c, r, s, k, stride, pad_t, pad_l, oh, ow) ? 1 : 0;
}
extern "C" int embnet_conv2d_stem_stats_rows(int n, int oh, int ow) {
  return n * cdiv(oh, stem::TH) * cdiv(ow, stem::TW) * 8;
}
extern "C" int embnet_conv2d_stem_f32(const float* x, const float* w, float* y, int n, int h, int wd, int pad_t, int pad_l, int oh, int ow,
                                      float* stats, const uint32_t* x_range, const uint32_t* w_range, void* stream) {
  EMBNET_CHECK_ARG(x && w && y && x_range && w_range, "conv2d_stem: null pointer (both range slots are required)");
  EMBNET_CHECK_ARG(!((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(y)) & 15) &&
                   !((reinterpret_cast<uintptr_t>(x_range) | reinterpret_cast<uintptr_t>(w_range)) & 3), "conv2d_stem: alignment");
  EMBNET_CHECK_ARG(stem_ok(n, h, wd, stem::C, stem::R, stem::S, stem::K, 2, pad_t, pad_l, oh, ow),
                   "conv2d_stem: unsupported geometry (embnet_conv2d_stem_supported)");
  stem::StemParams p{};
  p.x = x; p.w = w; p.y = y; p.stats = stats;
  p.N = n; p.H = h; p.W = wd; p.OH = oh; p.OW = ow; p.pad_t = pad_t; p.pad_l = pad_l;
  p.tiles_x = cdiv(ow, stem::TW); p.tiles_y = cdiv(oh, stem::TH); p.tiles = n * p.tiles_x * p.tiles_y;
  p.stats_rows = p.tiles * 8;
  p.a_range = x_range; p.w_range = w_range;
  p.x_bytes = (unsigned)((size_t)n * h * wd * 16);
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void*)stem::conv_stem_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    once = true;
  }
  const long M = (long)n * oh * ow;
  const int grid = p.tiles < 256 ? p.tiles : 256;
  hipStream_t st = (hipStream_t)stream;
  EMBNET_TRACE_FLOP("embnet::stem::conv_stem_kernel", 2.0 * M * stem::K * stem::R * stem::S * stem::C,
                    4.0 * ((double)n * h * wd * stem::C + (double)M * stem::K), st);
  stem::conv_stem_kernel<<<grid, 576, stem::LDS_BYTES, st>>>(p);
  return check_launch("conv2d_stem");
}
