// Convolution main loop on PRE-SPLIT operands: the three bf16 pieces of every fp32 value (gemm_engine.h, "bf16x6")
// are produced once by the tensor's producer and stored as three bf16 planes, and the implicit-GEMM kernels bring
// operand tiles into LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no register staging, no split arithmetic and no
// ds_write in the loop), two LDS stages, one barrier per K tile.  Same pieces, same six terms in the same order as
// gemm_mainloop3, so results are bit-identical to the in-loop-split kernels of conv.hip.
//
// LDS image of a K tile (32 k): per operand three planes [rows][64 bytes], the four 16-byte chunks of a row
// XOR-swizzled with (row >> 2) & 3 (TileKC3).  One DMA wave-instruction writes 1 KiB = 16 rows of one plane, lane l at
// base + 16 l, so lane l fetches row (l >> 2), LOGICAL chunk (l & 3) ^ ((l >> 4) & 3): the swizzle sits on the source
// address (the LDS side of an LDS-DMA is lane-linear).  Rows past the edge, k past K and padding taps get an
// out-of-range buffer offset: the DMA then writes zeros.
#include "gemm_engine.h"
#include "conv_geom.h"
#include "../../include/embnet.h"

namespace embnet {

template <int BM_, int BN_, int WM_, int WN_>
struct GeomN {
  static constexpr int BM = BM_, BN = BN_, WAVES_M = WM_, WAVES_N = WN_, NW = WM_ * WN_, NT = 64 * NW;
  static constexpr int WTM = BM / WM_, WTN = BN / WN_, TM = WTM / 32, TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1 && BM % (16 * NW) == 0 && BN % 16 == 0, "tile shape");
};

typedef __attribute__((address_space(3))) void* lds_ptr;

// Diagnostic build only (-DEMBNET_PLANES_STAMPS=1, tools/exp): per-workgroup time stamps (s_memrealtime, 100 MHz) into a
// buffer of their own: 0 entry, 1 first K tile landed, 2 main loop done, 3 exit.  The product build compiles them out.
#ifndef EMBNET_PLANES_STAMPS
#define EMBNET_PLANES_STAMPS 0
#endif
#if EMBNET_PLANES_STAMPS
static __device__ unsigned long long* g_pstamps = nullptr;
__device__ __forceinline__ void pstamp(int slot) {
  if (threadIdx.x != 0 || !g_pstamps) return;
  g_pstamps[(size_t)blockIdx.x * 8 + slot] = __builtin_amdgcn_s_memrealtime();
  if (slot == 0) g_pstamps[(size_t)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_getreg(63492);
}
#else
__device__ __forceinline__ void pstamp(int) {}
#endif

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned char* lds, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)lds, 16, (int)voff, (int)soff, 0, 0);
}

// fp32 -> three bf16 planes (truncation split, gemm_engine.h split4), four elements per thread
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, long n4, long plane_elems,
                                                           unsigned short* __restrict__ planes) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const Split4 s = split4(reinterpret_cast<const float4*>(x)[i]);
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<uint2*>(planes + q * plane_elems + 4 * i) = s.p[q];
  }
}

struct ConvPlanesFwdParams {
  const unsigned short* xp;      // [3][N,H,W,C] bf16 pieces of the conv input
  const unsigned short* wp;      // [3][K][R*S*C] bf16 pieces of the kernel, output channel major
  float* y; const float* bias; const float* residual; float* stats; int stats_rows; int relu;
  ConvGeom g; unsigned x_plane_bytes, w_plane_bytes; int n_tiles;
};

template <class G>
constexpr int PLANES_STAGE_BYTES = 3 * 64 * (G::BM + G::BN);
template <class G>
constexpr int PLANES_EPI_BYTES = 4 * G::NW * 32 * (G::WTN + 4);
template <class G>
constexpr int PLANES_SMEM_BYTES = 2 * PLANES_STAGE_BYTES<G> > PLANES_EPI_BYTES<G> ? 2 * PLANES_STAGE_BYTES<G> : PLANES_EPI_BYTES<G>;

template <class G>
__global__ __launch_bounds__(G::NT, (PLANES_SMEM_BYTES<G> <= 80 * 1024 ? 2 : 1) * G::NW / 4)
void conv_fwd_planes_kernel(ConvPlanesFwdParams p) {
  constexpr int NW = G::NW, PLA = G::BM * 64, PLB = G::BN * 64, STAGE = PLANES_STAGE_BYTES<G>;
  constexpr int GA = G::BM / 16 / NW, NGB = G::BN / 16, GB = (NGB + NW - 1) / NW;
  __shared__ __attribute__((aligned(16))) unsigned char smem[PLANES_SMEM_BYTES<G>];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  pstamp(0);
  const ConvGeom& g = p.g;
  const int M = g.N * g.OH * g.OW, Kg = g.R * g.S * g.C;
  const int tiles_n = (g.K + G::BN - 1) / G::BN;
  const int id = xcd_remap(blockIdx.x, p.n_tiles);
  const int m0 = (id / tiles_n) * G::BM, n0 = (id % tiles_n) * G::BN;
  const int kt_total = (Kg + BK - 1) / BK;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.xp), 0,
                                                                      3u * p.x_plane_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.wp), 0,
                                                                      3u * p.w_plane_bytes, 0x00020000);
  const int lc = (lane & 3) ^ ((lane >> 4) & 3);          // this lane's logical 8-k chunk of every row it fetches
  unsigned abase[GA]; int ih0[GA], iw0[GA];
#pragma unroll
  for (int j = 0; j < GA; ++j) {
    const int m = m0 + (wave + j * NW) * 16 + (lane >> 2);
    uint32_t n, rem, oh, ow;
    g.dOHW.divmod((uint32_t)min(m, M - 1), n, rem); g.dOW.divmod(rem, oh, ow);
    abase[j] = 2u * n * (unsigned)(g.H * g.W * g.C);
    ih0[j] = m < M ? (int)oh * g.stride - g.pad_t : ROW_INVALID;
    iw0[j] = (int)ow * g.stride - g.pad_l;
  }
  unsigned brow[GB];
#pragma unroll
  for (int j = 0; j < GB; ++j) {
    const int row = n0 + (wave + j * NW) * 16 + (lane >> 2);
    brow[j] = row < g.K ? 2u * (unsigned)row * (unsigned)Kg : OOB;
  }
  auto issue = [&](int kt, int stage) {
    const int kk = kt * BK + lc * 8;
    int r, s, c; split_k(kk, g.dC, g.dS, r, s, c);
    const bool kin = kk < Kg;
    unsigned char* sa = smem + stage * STAGE + wave * 1024;
#pragma unroll
    for (int j = 0; j < GA; ++j) {
      const int ih = ih0[j] + r, iw = iw0[j] + s;
      const bool ok = kin && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
      const unsigned off = ok ? abase[j] + 2u * (unsigned)((ih * g.W + iw) * g.C + c) : OOB;
#pragma unroll
      for (int q = 0; q < 3; ++q) dma16(xr, sa + j * NW * 1024 + q * PLA, off, q * p.x_plane_bytes);
    }
    unsigned char* sb = smem + stage * STAGE + 3 * PLA + wave * 1024;
#pragma unroll
    for (int j = 0; j < GB; ++j) {
      if (wave + j * NW < NGB) {
        const unsigned off = (kin && brow[j] != OOB) ? brow[j] + 2u * (unsigned)kk : OOB;
#pragma unroll
        for (int q = 0; q < 3; ++q) dma16(wr, sb + j * NW * 1024 + q * PLB, off, q * p.w_plane_bytes);
      }
    }
  };

  f32x16 acc[G::TM][G::TN];
#pragma unroll
  for (int i = 0; i < G::TM; ++i)
#pragma unroll
    for (int j = 0; j < G::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int wm = (wave / G::WAVES_N) * G::WTM, wn = (wave % G::WAVES_N) * G::WTN;
  using SA = TileKC3<G::BM>;
  using SB = TileKC3<G::BN>;

  issue(0, 0);
  for (int kt = 0; kt < kt_total; ++kt) {
    __syncthreads();                 // vmcnt(0) + barrier: tile kt has landed for every wave, tile kt-1 is no longer read
    if (kt == 0) pstamp(1);
    if (kt + 1 < kt_total) issue(kt + 1, (kt + 1) & 1);
    const unsigned char* sA = smem + (kt & 1) * STAGE;
    const unsigned char* sB = sA + 3 * PLA;
#pragma unroll
    for (int st = 0; st < BK / 16; ++st) {
      bf16x8 a[G::TM][3], b[G::TN][3];
#pragma unroll
      for (int i = 0; i < G::TM; ++i) SA::frag(sA, wm + 32 * i, st, lane, a[i]);
#pragma unroll
      for (int i = 0; i < G::TN; ++i) SB::frag(sB, wn + 32 * i, st, lane, b[i]);
      mfma_step3<G>(a, b, acc);
    }
  }

  pstamp(2);
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  for_each_acc_row4<G>(acc, reinterpret_cast<float*>(smem), [&](int r, int c, float4 v) {
    const int row = m0 + r, col = n0 + c;
    if (row < M && col < g.K) {
      if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + col); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
      if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (p.residual) {
        const float4 q = *reinterpret_cast<const float4*>(p.residual + (long)row * g.K + col);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
      }
      *reinterpret_cast<float4*>(p.y + (long)row * g.K + col) = v;
      s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
      s2.x = fmaf(v.x, v.x, s2.x); s2.y = fmaf(v.y, v.y, s2.y); s2.z = fmaf(v.z, v.z, s2.z); s2.w = fmaf(v.w, v.w, s2.w);
    }
  });
  if (p.stats) {
    constexpr int LPR = G::WTN / 4;
    s1 = colquad_sum<LPR>(s1); s2 = colquad_sum<LPR>(s2);
    const int col = n0 + (wave % G::WAVES_N) * G::WTN + lane * 4;
    if (lane < LPR && col < g.K) {
      const long prow = (long)(m0 / G::BM) * G::WAVES_M + wave / G::WAVES_N, P = p.stats_rows;
      float* d1 = p.stats + (long)col * P + prow;
      float* d2 = d1 + (long)g.K * P;
      d1[0] = s1.x; d1[P] = s1.y; d1[2 * P] = s1.z; d1[3 * P] = s1.w;
      d2[0] = s2.x; d2[P] = s2.y; d2[2 * P] = s2.z; d2[3 * P] = s2.w;
    }
  }
  pstamp(3);
}

}  // namespace embnet

using namespace embnet;

#if EMBNET_PLANES_STAMPS
extern "C" int embnet_debug_set_planes_stamps(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(embnet::g_pstamps), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

// planes[3][n] bf16 <- x[n] fp32 (n % 4 == 0)
extern "C" int embnet_split_planes_f32(const float* x, long n, void* planes, void* stream) {
  EMBNET_CHECK_ARG(x && planes && n > 0 && (n & 3) == 0, "split_planes: need n %% 4 == 0");
  const long n4 = n / 4;
  const int grid = (int)(n4 / 256 < 1 ? 1 : (n4 / 256 > 4096 ? 4096 : n4 / 256));
  EMBNET_TRACE("embnet::split_planes_kernel", TRACE_BYTES, 10.0 * n, stream);
  split_planes_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, n4, n, (unsigned short*)planes);
  return check_launch("split_planes");
}

using P128x64 = GeomN<128, 64, 2, 2>;
using P256x64 = GeomN<256, 64, 4, 2>;
using P256x128 = GeomN<256, 128, 4, 2>;
using P128x128 = GeomN<128, 128, 2, 2>;

// experimental entry point (tools/exp/ab_planes.py): tile 0 = 128x64 / 4 waves, 1 = 256x64 / 8, 2 = 256x128 / 8, 3 = 128x128 / 4
extern "C" int embnet_conv2d_fwd_planes(const void* xp, const void* wp, float* y, int n, int h, int wd, int c, int r,
                                        int s, int k, int stride, int pad_t, int pad_l, int oh, int ow,
                                        const float* residual, float* stats, int tile, void* stream) {
  EMBNET_CHECK_ARG(xp && wp && y, "conv2d_fwd_planes: null pointer");
  EMBNET_CHECK_ARG((c & 7) == 0 && (k & 3) == 0, "conv2d_fwd_planes: c %% 8 and k %% 4");
  ConvPlanesFwdParams p{(const unsigned short*)xp, (const unsigned short*)wp, y, nullptr, residual, stats, 0, 0};
  if (int rc = make_geom(p.g, n, h, wd, c, r, s, k, stride, pad_t, pad_l, oh, ow, "conv2d_fwd_planes")) return rc;
  const long M = (long)n * oh * ow;
  static const int BMs[4] = {128, 256, 256, 128}, BNs[4] = {64, 64, 128, 128}, WMs[4] = {2, 4, 4, 2};
  EMBNET_CHECK_ARG(tile >= 0 && tile < 4, "conv2d_fwd_planes: tile");
  p.x_plane_bytes = (unsigned)((size_t)n * h * wd * c * 2);
  p.w_plane_bytes = (unsigned)((size_t)r * s * c * k * 2);
  p.n_tiles = cdiv(M, BMs[tile]) * cdiv(k, BNs[tile]);
  p.stats_rows = cdiv(M, BMs[tile]) * WMs[tile];
  hipStream_t st = (hipStream_t)stream;
  static const char* names[4] = {"conv_fwd_planes<128x64>", "conv_fwd_planes<256x64>", "conv_fwd_planes<256x128>",
                                 "conv_fwd_planes<128x128>"};
  EMBNET_TRACE_FLOP(names[tile], 2.0 * M * k * r * s * c,
                    6.0 * ((double)n * h * wd * c + (double)r * s * c * k) + 4.0 * (double)M * k * (residual ? 2 : 1), st);
  switch (tile) {
    case 0: conv_fwd_planes_kernel<P128x64><<<p.n_tiles, P128x64::NT, 0, st>>>(p); break;
    case 1: conv_fwd_planes_kernel<P256x64><<<p.n_tiles, P256x64::NT, 0, st>>>(p); break;
    case 2: conv_fwd_planes_kernel<P256x128><<<p.n_tiles, P256x128::NT, 0, st>>>(p); break;
    default: conv_fwd_planes_kernel<P128x128><<<p.n_tiles, P128x128::NT, 0, st>>>(p); break;
  }
  return check_launch("conv2d_fwd_planes");
}
