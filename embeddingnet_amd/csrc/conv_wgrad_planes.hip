// Weight gradient of the 3x3 stride-1 'same' convolutions on PRE-SPLIT operands: every operand byte is fetched once per
// (64 channels x 64 filters) tile, all nine taps are accumulated from ONE window of the input held in LDS.
// Layers: the 3x3 convs of image-classifiers' residual units (/root/reference/embedding_net/backbones.py:99-104).
//
// Why (profiles/r04_pmc_traffic_c2.txt): conv.hip's conv_wgrad_kernel is an implicit GEMM with rows
// (r, s, c): every 128-row tile re-gathers the fp32 input of ITS tap(s) and all of dy, 3.1-5.5 x the algorithmic bytes per
// launch, and the fp32 copy of every patch-conv operand exists only because this kernel reads it.
//
// Form.  dW[r,s,c,k] = sum over output pixels of x[n, oh+r-1, ow+s-1, c] * dy[n, oh, ow, k].  Number the positions of the
// zero-padded image with SHARED padding — pitch OW + 1, OH + 1 rows per image: the right padding of row i is the left
// padding of row i + 1, the bottom padding row of image n the top one of image n + 1 (all of them zeros, so the aliases
// agree) —  q(n, py, px) = (n (OH+1) + py)(OW+1) + px.  With dy placed at q(n, oh, ow) (zero elsewhere) and x at
// q(n, ih+1, iw+1) (zero elsewhere), tap (r, s) is ONE shifted 1-D correlation:
//     dW[r,s] = sum_q  x_pos[q + r (OW+1) + s]  (outer)  dy_pos[q].
//  * operands are the chunk-major bf16 planes the patch convolution already consumes (conv_patch.hip): [3][C/16][pixel][16];
//    LDS-DMA (buffer_load ... lds) places 32 consecutive POSITIONS of one (plane, 16-channel chunk) per instruction — the
//    per-lane source offset does the position -> pixel gather, padding positions are out-of-range offsets (zeros);
//  * a workgroup owns (64 channels) x (64 filters) x 9 taps and a range of positions (split-K over positions, fp32 slabs +
//    conv.hip's fixed-order slab sum).  A STAGE is 32 positions: the dy rows of a stage go to one of NS slots, the x rows to
//    a ring of 256 positions that slides with the stages (a stage needs x rows q .. q + 31 + 2 (OW+1) + 2);  unit u of the
//    load stream = (x block u, dy stage u - HB) is requested D stages before the stage that needs it;
//  * 8 waves = 4 (32-channel, 32-filter) blocks x 2 halves of the stage (16 positions = one K step of
//    v_mfma_f32_32x32x16_bf16 each); a wave holds its block's NINE tap accumulators (144 registers) and per stage reads one
//    dy fragment and nine shifted x fragments, both with ds_read_b64_tr_b16 (positions are the reduction index: the
//    hardware transpose delivers 8 consecutive positions of one channel) — 60 reads for 54 MFMAs;  the two halves' sums meet
//    in LDS after the loop;
//  * waves 0-3 issue the x DMAs, waves 4-7 the dy DMAs (3 per wave and stage, counted s_waitcnt vmcnt); one s_barrier per stage.
// LDS: (plane, chunk) images are 32-byte position rows; the two chunks of a 32-channel block lie 128 (mod 256) bytes apart,
// so the two 16-lane groups of a transposed read's 32-lane pass cover all 64 banks at any tap shift.
// Same pieces and six-term products as gemm_mainloop3 (fp32-exact split, include/embnet.h); the summation order over the
// positions differs from conv_wgrad_kernel's (and zeros are added at the padding positions): equal within fp32 rounding.
#include "gemm_engine.h"
#include "conv_geom.h"
#include "../../include/embnet.h"

namespace embnet {
void launch_slab_reduce(const float* slabs, int splits, long n, float* out, hipStream_t st);   // conv.hip
namespace wgp {

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef int i32x4 __attribute__((ext_vector_type(4)));
// buffer_load_dwordx4 ... lds: lane l's 16 bytes at (rsrc base + soff + voff) -> LDS byte lds + 16 l  (M0 = LDS base).
// (M0 is a reserved register to the compiler — it cannot be named as a clobber; nothing else in this kernel uses it: gfx950's
// ds_read / ds_write take no M0, and there is no movrel, GWS or message instruction here.)
__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned lds, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

constexpr int XR = 256;                      // x ring: positions
constexpr int XCH = XR * 32 + 128;           // bytes of one (plane, chunk) image of the ring (128 mod 256: see header)
constexpr int NS = 4, D = 3;                 // dy slots, prefetch distance in stages
constexpr int DCH = 32 * 32 + 128;           // bytes of one (plane, chunk) image of a dy slot
constexpr int X_BYTES = 12 * XCH, DY_SLOT = 12 * DCH;
constexpr int LDS_BYTES = X_BYTES + NS * DY_SLOT;        // 155 136
static_assert(4 * 144 * 64 * 4 <= LDS_BYTES, "the half sums of four waves fit the operand buffers");

struct Params {
  const unsigned short* xp;    // [3][C/16][N*H*W][16]
  const unsigned short* dyp;   // [3][K/16][N*OH*OW][16]
  float* out;                  // [splits][9*C*K] slabs (or dw itself when splits == 1)
  int N, H, W, C, K, OH, OW, PWs;
  FastDiv dImg, dRow;          // (OH+1)(OW+1), OW+1
  unsigned x_plane_bytes, dy_plane_bytes, x_chunk_bytes, dy_chunk_bytes;
  int tiles_k, tiles, splits, stages_total, stages_per_split, HB;
  long slab_elems;
  int knobs;                   // experiments (EMBNET_WGP_KNOBS): bit 0 = s_setprio 1 for waves 4-7, bit 1 = for the x loaders (0-3), bit 2 = stagger
};

// F16: the planes hold two fp16 pieces + a scale (gemm_engine.h, EMBNET_PLANES_F16): a loader wave places two (plane, chunk) images
// per unit instead of three (the LDS layout keeps its twelve-image pitch), three matrix products per tap instead of six, the sums
// x 1 / (s_x s_dy) on the way out
template <bool STAG, bool F16 = false>
__global__ __launch_bounds__(512) void conv_wgrad_planes_kernel(const Params p) {
  constexpr int NP = F16 ? 2 : 3;                           // planes = (plane, chunk) images per loader wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // blocks b, b + 8, ... share an XCD (observed placement; speed only): the tiles of one position range stay on one L2
  const int per_xcd = gridDim.x >> 3;
  const int id = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (id >= p.tiles * p.splits) return;
  const int tile = id % p.tiles, split = id / p.tiles;
  const int c0 = (tile / p.tiles_k) * 64, k0 = (tile % p.tiles_k) * 64;
  const int sb = split * p.stages_per_split;
  const int nst = min(p.stages_per_split, p.stages_total - sb);
  const int HB = p.HB;

  // ---- load stream ------------------------------------------------------------------------------------------------------
  // (LDS-DMA by inline asm: after the builtin form the compiler orders every later ds_read of the SAME wave behind
  // s_waitcnt vmcnt(0) — it cannot tell the ring block being filled from the ones being read — which would wait for the
  // request just issued.  What orders reads behind the DMA here is the counted wait + barrier at the top of each stage.)
  const bool is_x = wave < 4;
  const int id0 = NP * (wave & 3);                         // this wave's NP (plane, chunk) images: id = plane * 4 + chunk
  const unsigned plane_b = is_x ? p.x_plane_bytes : p.dy_plane_bytes;
  const unsigned chunk_b = is_x ? p.x_chunk_bytes : p.dy_chunk_bytes;
  const uint64_t gbase = (uint64_t)(is_x ? p.xp : p.dyp);
  const i32x4 rs = {__builtin_amdgcn_readfirstlane((int)(uint32_t)gbase),
                    __builtin_amdgcn_readfirstlane((int)((gbase >> 32) & 0xffffu)),
                    __builtin_amdgcn_readfirstlane((int)(3u * plane_b)), 0x00020000};
  const int DH = is_x ? p.H : p.OH, DW = is_x ? p.W : p.OW, dp = is_x ? 1 : 0;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem;
  unsigned soff[NP], loff[NP];                             // per image: source offset of (plane, chunk), LDS offset in a block / slot
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int im = id0 + i;
    soff[i] = (unsigned)(im >> 2) * plane_b + ((unsigned)((is_x ? c0 : k0) >> 4) + (unsigned)(im & 3)) * chunk_b;
    loff[i] = lds0 + (is_x ? im * XCH : X_BYTES + im * DCH);
  }
  auto issue = [&](int u) {
    // x waves: positions of block sb + u -> ring rows 32 (u & 7);   dy waves: stage t = u - HB -> slot t & 3
    const int t = is_x ? u : u - HB;
    if (t < 0) return;
    const bool live = t < nst + (is_x ? HB : 0);
    const uint32_t q = 32u * (uint32_t)(sb + t) + (uint32_t)(lane >> 1);
    uint32_t n, rem, py, px;
    p.dImg.divmod(q, n, rem); p.dRow.divmod(rem, py, px);
    const int iy = (int)py - dp, ix = (int)px - dp;
    const bool ok = live && (int)n < p.N && (unsigned)iy < (unsigned)DH && (unsigned)ix < (unsigned)DW;
    const unsigned vo = ok ? 32u * (unsigned)(((int)n * DH + iy) * DW + ix) + 16u * (lane & 1) : OOB;
    const unsigned slot = is_x ? (unsigned)(u & 7) * 1024u : (unsigned)(t & (NS - 1)) * (unsigned)DY_SLOT;
#pragma unroll
    for (int i = 0; i < NP; ++i) dma16(rs, loff[i] + slot, vo, soff[i]);
  };

  // ---- this wave's block and fragment addresses -------------------------------------------------------------------------
  const int cb = (wave & 3) >> 1, kb = wave & 1, par = wave >> 2;
  const int g = lane >> 4, p4 = lane & 3, q4 = (lane & 15) >> 2, hh = g >> 1;
  const int PWs = p.PWs;
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
  const unsigned char* const xa = smem + (2 * cb + (g & 1)) * XCH + p4 * 8;
  const unsigned char* const ba = smem + X_BYTES + (2 * kb + (g & 1)) * DCH + (16 * par + 8 * hh + q4) * 32 + p4 * 8;
  const int row_in_stage = 16 * par + 8 * hh + q4;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  if (((p.knobs & 1) && wave >= 4) || ((p.knobs & 2) && wave < 4)) __builtin_amdgcn_s_setprio(1);

  // STAG: waves 4-7 run HALF A STAGE behind waves 0-3 (the partner waves of a SIMD then alternate between the barrier / DMA-issue
  // / fragment-read phase and the MFMA phase instead of meeting in both: microarchitecture guide, 'Two waves per SIMD', item 9).
  // A stage becomes two ticks with a barrier each; the early waves' stage start is the late waves' stage middle.  The x loaders
  // (early) issue at their stage MIDDLE (the block they overwrite is read by the late waves until then), the dy loaders (late) at
  // their stage START; each group waits for its own requests in front of the barrier that starts the OTHER group's stage.
  const bool late = STAG && wave >= 4;
  for (int u = 0; u < HB + D; ++u) issue(u);
  if (late) {                                                    // tick 0 belongs to the early waves' first half stage
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NP * (D - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
  }
  for (int j = 0; j < nst; ++j) {
    if (!late) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NP * (D - 1)) : "memory");   // unit j + HB has landed (this wave's part)
    __builtin_amdgcn_s_barrier();                                            // ... everybody's; stage j - 1 is done everywhere
    if (!STAG || late) issue(j + HB + D);
    const unsigned char* bs = ba + (j & (NS - 1)) * DY_SLOT;
    bf16x8 b[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(bs + q * 4 * DCH));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(bs + q * 4 * DCH + 4 * 32));
      const s16x8 w = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      b[q] = __builtin_bit_cast(bf16x8, w);
    }
    const int base_row = 32 * j + row_in_stage;
    auto load_a = [&](int tap, bf16x8 (&a)[NP]) {
      const int sh = (tap / 3) * PWs + tap % 3;
      const unsigned char* lo_p = xa + ((base_row + sh) & (XR - 1)) * 32;
      const unsigned char* hi_p = xa + ((base_row + sh + 4) & (XR - 1)) * 32;
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(lo_p + q * 4 * XCH));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(hi_p + q * 4 * XCH));
        const s16x8 w = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        a[q] = __builtin_bit_cast(bf16x8, w);
      }
    };
    bf16x8 a[2][NP];
    load_a(0, a[0]);
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (STAG && t == 5) {                                      // the stage's middle = the other group's stage start
        if (late) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NP * (D - 1)) : "memory");
        __builtin_amdgcn_s_barrier();
        if (!late) issue(j + HB + D);
      }
      if (t + 1 < 9) load_a(t + 1, a[(t + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);          // the next tap's reads go out BEFORE this tap's MFMAs
      if constexpr (F16) {
        constexpr int FA[3] = {0, 1, 0}, FB[3] = {1, 0, 0};      // smallest first
#pragma unroll
        for (int e = 0; e < 3; ++e)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[t & 1][FA[e]]), __builtin_bit_cast(f16x8, b[FB[e]]), acc[t], 0, 0, 0);
      } else {
#pragma unroll
        for (int e = 6 - EMBNET_EXP_TERMS; e < 6; ++e)      // (EMBNET_EXP_TERMS: gemm_engine.h; 6 in the product)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t & 1][PA[e]], b[PB[e]], acc[t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);          // one tap's fragments ahead, not all nine (registers)
    }
  }
  if (STAG && !late) __builtin_amdgcn_s_barrier();               // the late waves' last half stage

  // ---- the two halves meet in LDS; half 0 stores --------------------------------------------------------------------------
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // no DMA (the run-ahead requests past the end) lands after this
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem) + (wave & 3) * (144 * 64) + lane;
  if (par == 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(t * 16 + r) * 64] = acc[t][r];
  }
  __syncthreads();
  if (par == 0) {
    const int h = lane >> 5;
    // (the two factors one after the other: each is a normal number, their product need not be — gemm_engine.h scale_exponent_of)
    const float osx = F16 ? planes_scale_slot(p.xp, (long)(p.x_plane_bytes >> 1))[1] : 1.f, osd = F16 ? planes_scale_slot(p.dyp, (long)(p.dy_plane_bytes >> 1))[1] : 1.f;
    float* o = p.out + (long)split * p.slab_elems + (long)(c0 + 32 * cb + 4 * h) * p.K + k0 + 32 * kb + (lane & 31);
    const long tap_stride = (long)p.C * p.K;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        o[t * tap_stride + (long)((r & 3) + 8 * (r >> 2)) * p.K] = F16 ? ((acc[t][r] + red[(t * 16 + r) * 64]) * osx) * osd : acc[t][r] + red[(t * 16 + r) * 64];
  }
}

}  // namespace wgp
}  // namespace embnet

using namespace embnet;
using namespace embnet::wgp;

struct WgpPlan { int tiles_k, tiles, splits, stages_total, stages_per_split, HB; };

static bool wgp_plan(int n, int h, int wd, int c, int r, int s, int k, int stride, int pad_t, int pad_l, int oh, int ow, WgpPlan& pl) {
  static const int enabled = (int)env_long("EMBNET_WGRAD_PLANES", 1);
  if (!enabled || r != 3 || s != 3 || stride != 1 || pad_t != 1 || pad_l != 1 || oh != h || ow != wd) return false;
  if (n <= 0 || h <= 0 || wd <= 0 || (c & 63) || (k & 63) || c <= 0 || k <= 0) return false;
  const long positions = (long)n * (oh + 1) * (ow + 1);
  if (positions + 512 >= (1l << 31) / 32) return false;
  if ((size_t)n * h * wd * c * 2 >= 0x7FFFFFF0ull / 3 || (size_t)n * oh * ow * k * 2 >= 0x7FFFFFF0ull / 3) return false;
  const int halo = 2 * (ow + 1) + 2;
  pl.HB = (31 + halo) >> 5;
  if (pl.HB + D + 1 > XR / 32) return false;             // the ring holds the window plus the run-ahead
  pl.tiles_k = k / 64; pl.tiles = (c / 64) * pl.tiles_k;
  pl.stages_total = (int)((positions + 31) / 32);
  static const long target = env_long("EMBNET_WGRAD_PLANES_BLOCKS", 256);       // one workgroup per CU (155 KB of LDS)
  long want = target / pl.tiles; if (want < 1) want = 1;
  if (want > pl.stages_total / 4) want = pl.stages_total / 4;                   // at least four stages per split
  if (want < 1) want = 1;
  pl.stages_per_split = cdiv(pl.stages_total, want);
  pl.splits = cdiv(pl.stages_total, pl.stages_per_split);
  return true;
}

extern "C" int embnet_conv2d_wgrad_planes_supported(int n, int h, int wd, int c, int r, int s, int k, int stride, int pad_t,
                                                    int pad_l, int oh, int ow) {
  WgpPlan pl; return wgp_plan(n, h, wd, c, r, s, k, stride, pad_t, pad_l, oh, ow, pl) ? 1 : 0;
}
extern "C" int embnet_conv2d_wgrad_planes_splits(int n, int h, int wd, int c, int k) {
  WgpPlan pl; return wgp_plan(n, h, wd, c, 3, 3, k, 1, 1, 1, h, wd, pl) ? pl.splits : 0;
}
extern "C" size_t embnet_conv2d_wgrad_planes_workspace_bytes(int n, int h, int wd, int c, int k) {
  WgpPlan pl;
  if (!wgp_plan(n, h, wd, c, 3, 3, k, 1, 1, 1, h, wd, pl)) return 0;
  return pl.splits > 1 ? (size_t)pl.splits * 9 * c * k * sizeof(float) : 0;
}

extern "C" int embnet_conv2d_wgrad_planes_f32(const void* x_planes, const void* dy_planes, float* dw, void* workspace,
                                              size_t workspace_bytes, int n, int h, int wd, int c, int k, int reduce, void* stream) {
  EMBNET_CHECK_ARG(x_planes && dy_planes && dw, "conv2d_wgrad_planes: null pointer");
  EMBNET_CHECK_ARG(!(((uintptr_t)dw | (uintptr_t)workspace) & 15), "conv2d_wgrad_planes: dw and workspace must be 16-byte aligned");
  WgpPlan pl;
  EMBNET_CHECK_ARG(wgp_plan(n, h, wd, c, 3, 3, k, 1, 1, 1, h, wd, pl),
                   "conv2d_wgrad_planes: unsupported geometry (see embnet_conv2d_wgrad_planes_supported)");
  const size_t need = pl.splits > 1 ? (size_t)pl.splits * 9 * c * k * sizeof(float) : 0;
  if (need > workspace_bytes || (need && !workspace))
    return fail(EMBNET_EWORKSPACE, "conv2d_wgrad_planes: workspace %zu < %zu bytes", workspace_bytes, need);
  Params p{};
  p.xp = (const unsigned short*)x_planes; p.dyp = (const unsigned short*)dy_planes;
  p.out = pl.splits > 1 ? (float*)workspace : dw;
  p.N = n; p.H = h; p.W = wd; p.C = c; p.K = k; p.OH = h; p.OW = wd; p.PWs = wd + 1;
  p.dImg = FastDiv::make((uint32_t)((h + 1) * (wd + 1))); p.dRow = FastDiv::make((uint32_t)(wd + 1));
  p.x_plane_bytes = (unsigned)((size_t)n * h * wd * c * 2); p.dy_plane_bytes = (unsigned)((size_t)n * h * wd * k * 2);
  p.x_chunk_bytes = p.dy_chunk_bytes = (unsigned)((size_t)n * h * wd * 32);
  p.tiles_k = pl.tiles_k; p.tiles = pl.tiles; p.splits = pl.splits; p.stages_total = pl.stages_total;
  p.stages_per_split = pl.stages_per_split; p.HB = pl.HB; p.slab_elems = 9l * c * k;
  static const int knobs = (int)env_long("EMBNET_WGP_KNOBS", 0);
  p.knobs = knobs;
  hipStream_t st = (hipStream_t)stream;
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_planes_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_wgrad_planes_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_wgrad_planes_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    once = true;
  }
  const int grid = (pl.tiles * pl.splits + 7) / 8 * 8;
  {
    const double m = (double)n * h * wd;
    EMBNET_TRACE_FLOP(planes_f16() ? "void embnet::wgp::conv_wgrad_planes_kernel<false, true>(embnet::wgp::Params)" :
                      (knobs & 4) ? "void embnet::wgp::conv_wgrad_planes_kernel<true>(embnet::wgp::Params)"
                                  : "void embnet::wgp::conv_wgrad_planes_kernel<false>(embnet::wgp::Params)", 2.0 * m * k * 9.0 * c,
                      (planes_f16() ? 4.0 : 6.0) * m * (c + k) + 4.0 * 9.0 * c * k * pl.splits, st);
    if (planes_f16()) conv_wgrad_planes_kernel<false, true><<<grid, 512, LDS_BYTES, st>>>(p);
    else if (knobs & 4) conv_wgrad_planes_kernel<true><<<grid, 512, LDS_BYTES, st>>>(p);
    else conv_wgrad_planes_kernel<false><<<grid, 512, LDS_BYTES, st>>>(p);
  }
  if (pl.splits > 1 && reduce) launch_slab_reduce((const float*)workspace, pl.splits, 9l * c * k, dw, st);
  return check_launch("conv2d_wgrad_planes");
}
