// EXPERIMENT (round 3; not part of libembnet_hip.so — tools/build_variant.sh links it into build_variants/NAME.so when
// WITH_PLANES=1; driven by tools/exp/ab_planes.py, planes_timeline.py, patch_phases.py; results: DESIGN.md §3.9,
// profiles/r03_exp_planes_*.txt).
//
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
#include "../../embeddingnet_amd/csrc/gemm_engine.h"
#include "../../embeddingnet_amd/csrc/conv_geom.h"
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

// ---------------------------------------------------------------------------------------------------------------------
// "Patch" convolution (stride 1): every input value is brought on chip ONCE per output tile.
// The K loop of the plain implicit GEMM re-gathers the same input pixels for each of the R*S taps; measured
// (tools/exp/planes_timeline.py) its K-tile period is the LDS-DMA round trip, i.e. the loop runs at the CU's gather rate
// from L2 / Infinity Cache (30-70 GB/s per CU), not at the matrix rate.  Here a workgroup (8 waves, 256 output pixels x BN
// channels) keeps a PATCH of the padded input in LDS — the contiguous run of padded-image positions its 256 pixels'
// R x S windows cover, 16 channels at a time — and every tap reads its A fragments from the patch at a shifted row:
//   padded position of output pixel (n, oh, ow), tap (r, s):  n*PH*PW + (oh + r)*PW + (ow + s),   PH = OH+R-1, PW = OW+S-1
// so a tap is the constant row offset r*PW + s.  Padding positions are written as zeros by the DMA (out-of-range buffer
// offsets), image borders and image-to-image seams need no special case.
// LDS rows are 32 bytes (16 channels) per plane; the two 16-byte halves of a row are swapped when (row >> 3) & 1, which
// makes any 16 consecutive rows conflict-free for ds_read_b128 whatever the tap shift.
// Loop: chunks of 16 input channels; per chunk R steps (one kernel row = S taps each): 6*S*TM*TN MFMAs per wave and step.
// Per step one barrier; the weights of the next step (S taps x BN x 16 channels x 3 planes) and one plane of the next
// chunk's patch are in flight (two weight slots, two patch buffers).  The workgroup is persistent: it walks its output
// tiles with the next tile's first patch and weights requested before the current tile's epilogue; left-over tiles
// (tiles mod grid) are cut along the channel chunks into equal pieces, one per workgroup (partial tiles + tail_fixup_kernel).
struct ConvPatchParams {
  const unsigned short* xp;      // [3][N,H,W,C] bf16 pieces of the input
  const unsigned short* wp;      // [3][K][R*S*C] bf16 pieces of the kernel, output channel major
  float* y; const float* bias; const float* residual; float* stats; int stats_rows; int relu;
  ConvGeom g; unsigned x_plane_bytes, w_plane_bytes;
  int PH, PW; FastDiv dPHW, dPW;
  int LR;                        // LDS patch rows (multiple of 32)
  int n_full, parts, cc_part, n_pieces, grid; float* ws;
};

template <int TM>
struct PatchTile {
  int m0, n0, cc_b, cc_e, tile_m; float* part;
  unsigned poff[2]; int rowidx[TM];
};

// per-tile set-up: runs once per output tile and re-reads the kernel arguments where it needs them (kept in registers, its
// dividers and geometry words stayed live across the main loop: > 100 spilled scalar registers; as an out-of-line call it
// forced the accumulators to be saved around it)
template <int BN, int TM>
__device__ __forceinline__ void patch_setup(PatchTile<TM>& t, int item, int n_mine, int wave, int lane, int wm) {
  constexpr int NW = 8, KC = 16;
  // re-read the arguments from the kernel-argument segment here (scalar loads; ConvPatchParams is the kernel's only
  // argument) instead of keeping them in registers across the main loop
  typedef const ConvPatchParams __attribute__((address_space(4)))* kargp;
  kargp pp = (kargp)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(pp));
  const ConvPatchParams __attribute__((address_space(4)))& p = *pp;
  const ConvGeom __attribute__((address_space(4)))& g = p.g;
  const int b = blockIdx.x, NCC = g.C / KC, NG = p.LR / 32, PHW = p.PH * p.PW;
  const int M = g.N * g.OH * g.OW, tiles_n = (g.K + BN - 1) / BN;
  const int dhalf = (lane & 1) ^ ((lane >> 4) & 1);
  FastDiv dOHW, dOW, dPHW, dPW;                       // copies of the dividers out of the argument segment
  dOHW.mul = g.dOHW.mul; dOHW.shift = g.dOHW.shift; dOHW.d = g.dOHW.d; dOW.mul = g.dOW.mul; dOW.shift = g.dOW.shift; dOW.d = g.dOW.d;
  dPHW.mul = p.dPHW.mul; dPHW.shift = p.dPHW.shift; dPHW.d = p.dPHW.d; dPW.mul = p.dPW.mul; dPW.shift = p.dPW.shift; dPW.d = p.dPW.d;
  const int PWl = p.PW;
  auto base_of = [&](int m) {
    uint32_t n, rem, oh, ow;
    dOHW.divmod((uint32_t)m, n, rem); dOW.divmod(rem, oh, ow);
    return (int)n * PHW + (int)oh * PWl + (int)ow;
  };
  int id;
  if (item < n_mine) { id = b + item * p.grid; t.cc_b = 0; t.cc_e = NCC; t.part = nullptr; }
  else {
    id = p.n_full + b / p.parts;
    t.cc_b = (b % p.parts) * p.cc_part; t.cc_e = min(NCC, t.cc_b + p.cc_part);
    t.part = p.ws + (long)b * (256 * BN);
  }
  t.tile_m = id / tiles_n;
  t.m0 = t.tile_m * 256; t.n0 = (id % tiles_n) * BN;
  const int P0 = base_of(t.m0);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int grp = wave + j * NW;
    const int idx = P0 + 32 * grp + (lane >> 1);
    uint32_t n, rem, py, px;
    dPHW.divmod((uint32_t)idx, n, rem); dPW.divmod(rem, py, px);
    const int ih = (int)py - g.pad_t, iw = (int)px - g.pad_l;
    const bool ok = grp < NG && (int)n < g.N && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
    t.poff[j] = ok ? 2u * (unsigned)((((int)n * g.H + ih) * g.W + iw) * g.C) + 16u * dhalf : OOB;
  }
#pragma unroll
  for (int im = 0; im < TM; ++im) {
    const int m = t.m0 + wm + im * 32 + (lane & 31);
    t.rowidx[im] = m < M ? base_of(m) - P0 : 0;
  }
}

template <int BN, int R, int S>
__global__ __launch_bounds__(512, 2) void conv_patch_kernel(const ConvPatchParams p) {
  using G = GeomN<256, BN, 4, 2>;
  constexpr int NW = 8, KC = 16, TM = G::TM, TN = G::TN;
  constexpr int SBY = S * 3 * BN * 32;                   // one weight slot
  constexpr int NBI = S * 3 * (BN / 32), JMAX = (NBI + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave / G::WAVES_N) * G::WTM, wn = (wave % G::WAVES_N) * G::WTN;
  const int LR = p.LR, PLP = LR * 32, PB = 3 * PLP;      // patch plane / patch buffer bytes
  unsigned char* const bslot0 = smem + 2 * PB;
  const int C = p.g.C, K = p.g.K, PW = p.PW;
  const int M = p.g.N * p.g.OH * p.g.OW, Kg = R * S * C, NG = LR / 32;
  const unsigned xpb = p.x_plane_bytes, wpb = p.w_plane_bytes;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.xp), 0, 3u * xpb, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.wp), 0, 3u * wpb, 0x00020000);
  const int dhalf = (lane & 1) ^ ((lane >> 4) & 1);      // logical 16-byte half this lane's DMA piece holds

  // work items of this workgroup: full tiles b, b + grid, ... then (b < n_pieces) one piece of a left-over tile
  const int b = blockIdx.x;
  const int n_mine = b < p.n_full ? (p.n_full - b + p.grid - 1) / p.grid : 0;
  const int n_items = n_mine + (b < p.n_pieces ? 1 : 0);
  if (n_items == 0) return;
  using PT = PatchTile<TM>;
  // weights of step (n0, cc, r) -> weight slot: S taps x 3 planes x BN rows x 32 bytes; ok = false: nothing to fetch
  auto dma_b = [&](int n0, int cc, int r, bool ok, unsigned char* slot) {
    const unsigned kofs = 2u * (unsigned)(r * S * C + cc * KC + dhalf * 8);
#pragma unroll
    for (int jj = 0; jj < JMAX; ++jj) {
      const int j = wave + jj * NW;
      if (jj * NW + NW <= NBI || j < NBI) {
        const int gb = j % (BN / 32), tq = j / (BN / 32), q = tq % 3, s = tq / 3;
        const int row = n0 + gb * 32 + (lane >> 1);
        const unsigned off = (ok && row < K) ? 2u * (unsigned)row * (unsigned)Kg + kofs + 2u * (unsigned)(s * C) : OOB;
        dma16(wr, slot + ((s * 3 + q) * BN + gb * 32) * 32, off, q * wpb);
      }
    }
  };
  auto dma_patch = [&](const unsigned (&poff)[2], int cc, int q, bool ok, unsigned char* buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int grp = wave + j * NW;
      if (grp < NG) dma16(xr, buf + q * PLP + grp * 1024, ok ? poff[j] + (unsigned)(cc * KC * 2) : OOB, q * xpb);
    }
  };

#if EMBNET_PLANES_STAMPS
  unsigned long long t_wait = 0, t_issue = 0, t_comp = 0, t_epi = 0, t_steps = 0;
#define PSTAMP(var) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); var += now_ - t_last; t_last = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
  unsigned long long t_last = __builtin_amdgcn_s_memtime();
  const unsigned long long t_begin = t_last;
#else
#define PSTAMP(var) do {} while (0)
#endif
  PT cur, nxt;
  patch_setup<BN, TM>(cur, 0, n_mine, wave, lane, wm);
  nxt = cur;
#pragma unroll
  for (int q = 0; q < 3; ++q) dma_patch(cur.poff, cur.cc_b, q, true, smem);
  dma_b(cur.n0, cur.cc_b, 0, true, bslot0);
  int gs = 0, gc = 0;
  f32x16 acc[TM][TN];
  for (int item = 0; item < n_items; ++item) {
    const bool has_next = item + 1 < n_items;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int cc = cur.cc_b; cc < cur.cc_e; ++cc) {
      const bool lastc = cc == cur.cc_e - 1;
      if (lastc && has_next) patch_setup<BN, TM>(nxt, item + 1, n_mine, wave, lane, wm);
      const unsigned char* pbuf = smem + (gc & 1) * PB;
      unsigned char* nbuf = smem + ((gc + 1) & 1) * PB;
      // the chunk after this one: of this tile, or the first of the next tile
      const bool pok = !lastc || has_next;
      const int pcc = lastc ? nxt.cc_b : cc + 1;
      unsigned npoff[2];
      npoff[0] = lastc ? nxt.poff[0] : cur.poff[0]; npoff[1] = lastc ? nxt.poff[1] : cur.poff[1];
#pragma unroll 1
      for (int r = 0; r < R; ++r) {
        PSTAMP(t_comp);
        __syncthreads();             // vmcnt(0) + barrier: this step's weights and patch planes have landed everywhere
        PSTAMP(t_wait);
        {                            // next step's weights; one plane (R = 3) of the next chunk's patch
          const bool wrap = r + 1 == R;
          dma_b((wrap && lastc) ? nxt.n0 : cur.n0, wrap ? pcc : cc, wrap ? 0 : r + 1, !wrap || pok,
                bslot0 + ((gs + 1) & 1) * SBY);
          if constexpr (R == 3) dma_patch(npoff, pcc, r, pok, nbuf);
          else {
#pragma unroll
            for (int q = 0; q < 3; ++q) if (q % R == r) dma_patch(npoff, pcc, q, pok, nbuf);
          }
        }
        PSTAMP(t_issue);
#if EMBNET_PLANES_STAMPS
        ++t_steps;
#endif
        const unsigned char* bs = bslot0 + (gs & 1) * SBY;
#pragma unroll
        for (int s = 0; s < S; ++s) {
          bf16x8 a[TM][3], bb[TN][3];
#pragma unroll
          for (int im = 0; im < TM; ++im) {
            const int idx = cur.rowidx[im] + r * PW + s;
            const unsigned char* ap = pbuf + idx * 32 + ((h ^ ((idx >> 3) & 1)) << 4);
#pragma unroll
            for (int q = 0; q < 3; ++q) a[im][q] = *reinterpret_cast<const bf16x8*>(ap + q * PLP);
          }
#pragma unroll
          for (int in = 0; in < TN; ++in) {
            const int row = wn + in * 32 + (lane & 31);
            const unsigned char* bp = bs + (s * 3 * BN + row) * 32 + ((h ^ ((row >> 3) & 1)) << 4);
#pragma unroll
            for (int q = 0; q < 3; ++q) bb[in][q] = *reinterpret_cast<const bf16x8*>(bp + q * BN * 32);
          }
          mfma_step3<G>(a, bb, acc);
          __builtin_amdgcn_sched_barrier(0);     // one tap's fragments at a time (all taps' reads hoisted: 256 registers + spills)
        }
        ++gs;
      }
      ++gc;
    }
    PSTAMP(t_comp);
    // epilogue straight from the accumulators: register rr of a 32x32 block holds row (rr&3) + 8*(rr>>2) + 4*h, column
    // lane & 31, so a store instruction writes two 128-byte row segments
    if (cur.part) {
#pragma unroll
      for (int im = 0; im < TM; ++im)
#pragma unroll
        for (int in = 0; in < TN; ++in)
#pragma unroll
          for (int rr = 0; rr < 16; ++rr)
            cur.part[(wm + im * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h) * BN + wn + in * 32 + (lane & 31)] = acc[im][in][rr];
    } else {
      const bool inner = cur.m0 + 256 <= M && cur.n0 + BN <= K;      // wave-uniform: no edge tests on interior tiles
#pragma unroll
      for (int in = 0; in < TN; ++in) {
        const int col = cur.n0 + wn + in * 32 + (lane & 31);
        const bool cok = col < K;
        const float bv = (p.bias && cok) ? p.bias[col] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int im = 0; im < TM; ++im) {
          const long o0 = (long)(cur.m0 + wm + im * 32 + 4 * h) * K + col;
          float v[16];
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) { v[rr] = acc[im][in][rr] + bv; if (p.relu) v[rr] = fmaxf(v[rr], 0.f); }
          if (inner) {
            if (p.residual) {
#pragma unroll
              for (int rr = 0; rr < 16; ++rr) v[rr] += p.residual[o0 + (long)((rr & 3) + 8 * (rr >> 2)) * K];
            }
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
              p.y[o0 + (long)((rr & 3) + 8 * (rr >> 2)) * K] = v[rr];
              s1 += v[rr]; s2 = fmaf(v[rr], v[rr], s2);
            }
          } else {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
              const int row = cur.m0 + wm + im * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
              if (row < M && cok) {
                const long o = o0 + (long)((rr & 3) + 8 * (rr >> 2)) * K;
                if (p.residual) v[rr] += p.residual[o];
                p.y[o] = v[rr];
                s1 += v[rr]; s2 = fmaf(v[rr], v[rr], s2);
              }
            }
          }
        }
        if (p.stats) {
          s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
          if (h == 0 && cok) {
            const long prow = (long)cur.tile_m * G::WAVES_M + wave / G::WAVES_N, P = p.stats_rows;
            p.stats[(long)col * P + prow] = s1;
            p.stats[((long)K + col) * P + prow] = s2;
          }
        }
      }
    }
    cur = nxt;
    PSTAMP(t_epi);
  }
#if EMBNET_PLANES_STAMPS
  if (g_pstamps && lane == 0) {
    unsigned long long* d = g_pstamps + ((size_t)blockIdx.x * 8 + wave) * 8;
    d[0] = t_wait; d[1] = t_issue; d[2] = t_comp; d[3] = t_epi; d[4] = t_steps; d[5] = __builtin_amdgcn_s_memtime() - t_begin;
  }
#endif
}

// kernel [R,S,C,K] fp32 -> planes [3][rows][R*S*inner] bf16, four consecutive elements of a row per thread.
// flip = 0 (forward):        rows = K, row k,  column (r,s,c)  <- w[r,s,c,k]
// flip = 1 (data gradient):  rows = C, row c,  column (r,s,k)  <- w[R-1-r, S-1-s, c, k]   (the stride-1 data gradient is the
//                            correlation of dy with the flipped kernel, channels swapped)
__global__ __launch_bounds__(256) void prep_weight_planes_kernel(const float* __restrict__ w, int R, int S, int C, int K, int flip,
                                                                 unsigned short* __restrict__ out) {
  const int rows = flip ? C : K, inner = flip ? K : C, cols = R * S * inner;
  const long total4 = (long)rows * cols / 4, plane = (long)rows * cols;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    const long e = 4 * i;
    const int row = (int)(e / cols), col = (int)(e % cols);
    float v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int cc = col + t, rs = cc / inner, in = cc % inner, r = rs / S, s = rs % S;
      v[t] = flip ? w[((long)((R - 1 - r) * S + (S - 1 - s)) * C + row) * K + in] : w[((long)(r * S + s) * C + in) * K + row];
    }
    const Split4 sp = split4(make_float4(v[0], v[1], v[2], v[3]));
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<uint2*>(out + q * plane + e) = sp.p[q];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Patch convolution, second version: what the first one's stamps asked for (profiles/r03_exp_patch_phases.txt: per step
// 2 500 cycles of MFMA work, 830 cycles of DMA ISSUE with the matrix pipe idle, 1 600 cycles waiting for DMA that had
// been requested a whole step earlier).
//  * Operand layouts in HBM made for the consumer: activations CHUNK-MAJOR [plane][C/16][pixels][16] so that consecutive
//    patch rows are consecutive 32-byte pieces (a 1 KiB DMA = 8 whole cache lines instead of 32 scattered ones), weights
//    STEP-MAJOR [plane][r][C/16][s][K][16] so that a step's weights are one contiguous block per tap.
//  * Two LOADER waves (wave 8: weights, wave 9: patches) beside the 8 MFMA waves: the MFMA waves never issue a DMA and
//    never wait on vmcnt; the loaders' queues hold nothing but their own DMAs, so their counted waits are constants.
//  * Weight ring of NBS slots, requested NBS - 1 steps ahead; the next chunk's patch is requested at the first step of the
//    current chunk (R*S/TPS steps ahead).
// A step is TPS taps of one 16-channel chunk (TPS = 3: a kernel row; TPS = 1: one tap).
struct ConvPatch2Params {
  const unsigned short* xp;      // [3][C/16][N*H*W][16] bf16 pieces of the input
  const unsigned short* wp;      // [3][R][C/16][S][K][16] bf16 pieces of the kernel
  float* y; const float* bias; const float* residual; float* stats; int stats_rows; int relu;
  ConvGeom g; unsigned x_plane_bytes, w_plane_bytes;
  int PH, PW; FastDiv dPHW, dPW;
  int LR;                        // LDS patch rows (multiple of 32)
  int n_full, parts, cc_part, n_pieces, grid; float* ws;
};

struct Patch2Item { int m0, n0, cc_b, cc_e, tile_m; float* part; };

template <int BN>
__device__ __forceinline__ Patch2Item patch2_item(int item, int n_mine) {
  typedef const ConvPatch2Params __attribute__((address_space(4)))* kargp;
  kargp pp = (kargp)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(pp));
  const int b = blockIdx.x, NCC = pp->g.C / 16, tiles_n = (pp->g.K + BN - 1) / BN;
  Patch2Item t; int id;
  if (item < n_mine) { id = b + item * pp->grid; t.cc_b = 0; t.cc_e = NCC; t.part = nullptr; }
  else {
    id = pp->n_full + b / pp->parts;
    t.cc_b = (b % pp->parts) * pp->cc_part; t.cc_e = min(NCC, t.cc_b + pp->cc_part);
    t.part = pp->ws + (long)b * (256 * BN);
  }
  t.tile_m = id / tiles_n; t.m0 = t.tile_m * 256; t.n0 = (id % tiles_n) * BN;
  return t;
}

__device__ __forceinline__ int patch2_base(int m) {            // padded-image position of output pixel m
  typedef const ConvPatch2Params __attribute__((address_space(4)))* kargp;
  kargp pp = (kargp)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(pp));
  FastDiv dOHW, dOW;
  dOHW.mul = pp->g.dOHW.mul; dOHW.shift = pp->g.dOHW.shift; dOHW.d = pp->g.dOHW.d;
  dOW.mul = pp->g.dOW.mul; dOW.shift = pp->g.dOW.shift; dOW.d = pp->g.dOW.d;
  uint32_t n, rem, oh, ow;
  dOHW.divmod((uint32_t)m, n, rem); dOW.divmod(rem, oh, ow);
  return (int)n * (pp->PH * pp->PW) + (int)oh * pp->PW + (int)ow;
}

template <int BN, int R, int S, int TPS, int NBS>
__global__ __launch_bounds__(640) void conv_patch2_kernel(const ConvPatch2Params p) {
  using G = GeomN<256, BN, 4, 2>;
  constexpr int TM = G::TM, TN = G::TN, SPC = R * S / TPS, D = NBS - 1;
  constexpr int SBY = TPS * 3 * BN * 32;                 // one weight slot: TPS taps x 3 planes x BN rows x 32 bytes
  constexpr int NBI = TPS * 3 * (BN / 32);               // DMA instructions per weight slot
  static_assert((R * S) % TPS == 0 && SPC >= 2, "steps per chunk");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int LR = p.LR, PLP = LR * 32, PB = 3 * PLP;
  unsigned char* const bslot0 = smem + 2 * PB;
  const int b = blockIdx.x;
  const int n_mine = b < p.n_full ? (p.n_full - b + p.grid - 1) / p.grid : 0;
  const int n_items = n_mine + (b < p.n_pieces ? 1 : 0);
  if (n_items == 0) return;
  const int dhalf = (lane & 1) ^ ((lane >> 4) & 1);

  if (wave == 8) {
    // ---- weight loader: slot (gs % NBS) <- weights of step gs, D steps ahead of the MFMA waves ----------------------
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.wp), 0, 3u * p.w_plane_bytes, 0x00020000);
    const int K = p.g.K, NCC = p.g.C / 16;
    const unsigned wpb = p.w_plane_bytes;
    int it = 0; Patch2Item t = patch2_item<BN>(0, n_mine);       // the step being REQUESTED: (it, cc, st)
    int cc = t.cc_b, st = 0; bool live = true;
    auto issue = [&](int gs) {
      unsigned char* slot = bslot0 + (gs % NBS) * SBY;
#pragma unroll
      for (int j = 0; j < NBI; ++j) {
        const int gb = j % (BN / 32), tq = j / (BN / 32), q = tq % 3, tp = tq / 3;
        const int tap = st * TPS + tp, r = tap / S, s = tap % S;
        const int row = t.n0 + gb * 32 + (lane >> 1);
        const unsigned off = (live && row < K) ? 32u * (unsigned)row + 16u * dhalf : OOB;
        const unsigned so = q * wpb + 32u * (unsigned)(((r * NCC + cc) * S + s) * K);
        dma16(wr, slot + ((tp * 3 + q) * BN + gb * 32) * 32, off, so);
      }
      if (live && ++st == SPC) {
        st = 0;
        if (++cc == t.cc_e) {
          if (++it < n_items) { t = patch2_item<BN>(it, n_mine); cc = t.cc_b; } else live = false;
        }
      }
    };
    int total = 0;                                               // steps of this workgroup
    for (int i = 0; i < n_items; ++i) { const Patch2Item q = patch2_item<BN>(i, n_mine); total += (q.cc_e - q.cc_b) * SPC; }
    for (int gs = 0; gs < D; ++gs) issue(gs);
    for (int gs = 0; gs < total; ++gs) {
      // the D - 1 youngest requests may still be in flight: the weights of step gs have landed
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D - 1) * NBI) : "memory");
      __builtin_amdgcn_s_barrier();                              // #gs: step gs - 1 is done everywhere -> its slot is free
      issue(gs + D);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  if (wave == 9) {
    // ---- patch loader: buffer (gc & 1) <- patch of chunk gc, requested at the first step of chunk gc - 1 -------------
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.xp), 0, 3u * p.x_plane_bytes, 0x00020000);
    const int NG = LR / 32;
    const unsigned xpb = p.x_plane_bytes, chunk_bytes = 32u * (unsigned)(p.g.N * p.g.H * p.g.W);
    unsigned poff[16];
    auto tile_offsets = [&](const Patch2Item& t) {
      typedef const ConvPatch2Params __attribute__((address_space(4)))* kargp;
      kargp pp = (kargp)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(pp));
      FastDiv dPHW, dPW;
      dPHW.mul = pp->dPHW.mul; dPHW.shift = pp->dPHW.shift; dPHW.d = pp->dPHW.d;
      dPW.mul = pp->dPW.mul; dPW.shift = pp->dPW.shift; dPW.d = pp->dPW.d;
      const int P0 = patch2_base(t.m0), N = pp->g.N, H = pp->g.H, W = pp->g.W, pt = pp->g.pad_t, pl = pp->g.pad_l;
#pragma unroll
      for (int gI = 0; gI < 16; ++gI) {
        const int idx = P0 + 32 * gI + (lane >> 1);
        uint32_t n, rem, py, px;
        dPHW.divmod((uint32_t)idx, n, rem); dPW.divmod(rem, py, px);
        const int ih = (int)py - pt, iw = (int)px - pl;
        const bool ok = gI < NG && (int)n < N && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
        poff[gI] = ok ? 32u * (unsigned)(((int)n * H + ih) * W + iw) + 16u * dhalf : OOB;
      }
    };
    auto issue = [&](int cc, int gc) {
      unsigned char* buf = smem + (gc & 1) * PB;
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int gI = 0; gI < 16; ++gI)
          if (gI < NG) dma16(xr, buf + q * PLP + gI * 1024, poff[gI], q * xpb + (unsigned)cc * chunk_bytes);
    };
    int it = 0; Patch2Item t = patch2_item<BN>(0, n_mine);       // the chunk being REQUESTED
    int cc = t.cc_b; bool live = true;
    tile_offsets(t);
    auto advance = [&]() {
      if (++cc == t.cc_e) {
        if (++it < n_items) { t = patch2_item<BN>(it, n_mine); cc = t.cc_b; tile_offsets(t); } else live = false;
      }
    };
    issue(cc, 0); advance();
    int total_chunks = 0;
    for (int i = 0; i < n_items; ++i) { const Patch2Item q = patch2_item<BN>(i, n_mine); total_chunks += q.cc_e - q.cc_b; }
    for (int gc = 0; gc < total_chunks; ++gc) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // patch gc has landed (requested a chunk ago)
      __builtin_amdgcn_s_barrier();                              // first step of chunk gc: chunk gc - 1 is done -> its buffer is free
      if (live) { issue(cc, gc + 1); advance(); }
#pragma unroll 1
      for (int s2 = 1; s2 < SPC; ++s2) __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ---- MFMA waves ---------------------------------------------------------------------------------------------------
  const int wm = (wave / G::WAVES_N) * G::WTM, wn = (wave % G::WAVES_N) * G::WTN;
  const int K = p.g.K, PW = p.PW, M = p.g.N * p.g.OH * p.g.OW;
  int gs = 0, gc = 0;
  f32x16 acc[TM][TN];
#if EMBNET_PLANES_STAMPS
  unsigned long long t_wait = 0, t_comp = 0, t_epi = 0, t_steps = 0;
  unsigned long long t_last = __builtin_amdgcn_s_memtime();
  const unsigned long long t_begin = t_last;
#endif
  for (int item = 0; item < n_items; ++item) {
    const Patch2Item cur = patch2_item<BN>(item, n_mine);
    int rowidx[TM];
    {
      const int P0 = patch2_base(cur.m0);
#pragma unroll
      for (int im = 0; im < TM; ++im) {
        const int m = cur.m0 + wm + im * 32 + (lane & 31);
        rowidx[im] = m < M ? patch2_base(m) - P0 : 0;
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int cc = cur.cc_b; cc < cur.cc_e; ++cc) {
      const unsigned char* pbuf = smem + (gc & 1) * PB;
#pragma unroll 1
      for (int st = 0; st < SPC; ++st) {
        PSTAMP(t_comp);
        __syncthreads();             // barrier #gs: this step's weights (and, at st = 0, this chunk's patch) are in LDS
        PSTAMP(t_wait);
#if EMBNET_PLANES_STAMPS
        ++t_steps;
#endif
        const unsigned char* bs = bslot0 + (gs % NBS) * SBY;
#pragma unroll
        for (int tp = 0; tp < TPS; ++tp) {
          const int tap = st * TPS + tp, r = tap / S, s = tap % S;
          bf16x8 a[TM][3], bb[TN][3];
#pragma unroll
          for (int im = 0; im < TM; ++im) {
            const int idx = rowidx[im] + r * PW + s;
            const unsigned char* ap = pbuf + idx * 32 + ((h ^ ((idx >> 3) & 1)) << 4);
#pragma unroll
            for (int q = 0; q < 3; ++q) a[im][q] = *reinterpret_cast<const bf16x8*>(ap + q * PLP);
          }
#pragma unroll
          for (int in = 0; in < TN; ++in) {
            const int row = wn + in * 32 + (lane & 31);
            const unsigned char* bp = bs + (tp * 3 * BN + row) * 32 + ((h ^ ((row >> 3) & 1)) << 4);
#pragma unroll
            for (int q = 0; q < 3; ++q) bb[in][q] = *reinterpret_cast<const bf16x8*>(bp + q * BN * 32);
          }
          mfma_step3<G>(a, bb, acc);
          if (TPS > 1) __builtin_amdgcn_sched_barrier(0);
        }
        ++gs;
      }
      ++gc;
    }
    PSTAMP(t_comp);
    if (cur.part) {
#pragma unroll
      for (int im = 0; im < TM; ++im)
#pragma unroll
        for (int in = 0; in < TN; ++in)
#pragma unroll
          for (int rr = 0; rr < 16; ++rr)
            cur.part[(wm + im * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h) * BN + wn + in * 32 + (lane & 31)] = acc[im][in][rr];
    } else {
      const bool inner = cur.m0 + 256 <= M && cur.n0 + BN <= K;
#pragma unroll
      for (int in = 0; in < TN; ++in) {
        const int col = cur.n0 + wn + in * 32 + (lane & 31);
        const bool cok = col < K;
        const float bv = (p.bias && cok) ? p.bias[col] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int im = 0; im < TM; ++im) {
          const long o0 = (long)(cur.m0 + wm + im * 32 + 4 * h) * K + col;
          float v[16];
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) { v[rr] = acc[im][in][rr] + bv; if (p.relu) v[rr] = fmaxf(v[rr], 0.f); }
          if (inner) {
            if (p.residual) {
#pragma unroll
              for (int rr = 0; rr < 16; ++rr) v[rr] += p.residual[o0 + (long)((rr & 3) + 8 * (rr >> 2)) * K];
            }
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
              p.y[o0 + (long)((rr & 3) + 8 * (rr >> 2)) * K] = v[rr];
              s1 += v[rr]; s2 = fmaf(v[rr], v[rr], s2);
            }
          } else {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
              const int row = cur.m0 + wm + im * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
              if (row < M && cok) {
                const long o = o0 + (long)((rr & 3) + 8 * (rr >> 2)) * K;
                if (p.residual) v[rr] += p.residual[o];
                p.y[o] = v[rr];
                s1 += v[rr]; s2 = fmaf(v[rr], v[rr], s2);
              }
            }
          }
        }
        if (p.stats) {
          s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
          if (h == 0 && cok) {
            const long prow = (long)cur.tile_m * G::WAVES_M + wave / G::WAVES_N, P = p.stats_rows;
            p.stats[(long)col * P + prow] = s1;
            p.stats[((long)K + col) * P + prow] = s2;
          }
        }
      }
    }
    PSTAMP(t_epi);
  }
#if EMBNET_PLANES_STAMPS
  if (g_pstamps && lane == 0) {
    unsigned long long* d = g_pstamps + ((size_t)blockIdx.x * 8 + wave) * 8;
    d[0] = t_wait; d[1] = 0; d[2] = t_comp; d[3] = t_epi; d[4] = t_steps; d[5] = __builtin_amdgcn_s_memtime() - t_begin;
  }
#endif
}

// fp32 NHWC [pixels][C] -> chunk-major planes [3][C/16][pixels][16] bf16; one thread per (pixel, 4 channels)
__global__ __launch_bounds__(256) void split_planes_cm_kernel(const float* __restrict__ x, long pixels, int C,
                                                              unsigned short* __restrict__ planes) {
  const long total4 = pixels * C / 4, plane = pixels * C;
  const int c4 = C / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    const long pix = i / c4; const int c = (int)(i % c4) * 4;
    const Split4 s = split4(reinterpret_cast<const float4*>(x)[i]);
    const long o = ((long)(c >> 4) * pixels + pix) * 16 + (c & 15);
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<uint2*>(planes + q * plane + o) = s.p[q];
  }
}

// kernel [R,S,C,K] fp32 -> step-major planes [3][R][Cin/16][S][rows][16] bf16.
// flip = 0: rows = K, reduction channels = C:  out[r][cc][s][k][j] = w[r, s, 16 cc + j, k]
// flip = 1: rows = C, reduction channels = K:  out[r][cc][s][c][j] = w[R-1-r, S-1-s, c, 16 cc + j]   (stride-1 data gradient)
__global__ __launch_bounds__(256) void prep_weight_planes2_kernel(const float* __restrict__ w, int R, int S, int C, int K, int flip,
                                                                  unsigned short* __restrict__ out) {
  const int rows = flip ? C : K, red = flip ? K : C, ncc = red / 16;
  const long total4 = (long)R * S * C * K / 4, plane = (long)R * S * C * K;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    long e = 4 * i;                                   // output element index [r][cc][s][row][j]
    const int j = (int)(e % 16); e /= 16;
    const int row = (int)(e % rows); e /= rows;
    const int s = (int)(e % S); e /= S;
    const int cc = (int)(e % ncc); const int r = (int)(e / ncc);
    float v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int ch = cc * 16 + j + t;
      v[t] = flip ? w[((long)((R - 1 - r) * S + (S - 1 - s)) * C + row) * K + ch] : w[((long)(r * S + s) * C + ch) * K + row];
    }
    const Split4 sp = split4(make_float4(v[0], v[1], v[2], v[3]));
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<uint2*>(out + q * plane + 4 * i) = sp.p[q];
  }
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

// ---- patch convolution: host side ------------------------------------------------------------------------------------
static int patch_rows(int n, int oh, int ow, int r, int s) {                 // LDS patch rows for 256-pixel tiles
  const long M = (long)n * oh * ow;
  const int PH = oh + r - 1, PW = ow + s - 1;
  auto base = [&](long m) { const long img = m / ((long)oh * ow), rem = m % ((long)oh * ow); return img * PH * PW + (rem / ow) * PW + rem % ow; };
  long worst = 0;
  for (long m0 = 0; m0 < M; m0 += 256) {
    const long m1 = (m0 + 256 < M ? m0 + 256 : M) - 1;
    const long L = base(m1) - base(m0) + (long)(r - 1) * PW + s;
    if (L > worst) worst = L;
  }
  return (int)((worst + 31) / 32 * 32);
}

struct PatchPlan { int bn, LR, tiles, n_full, parts, cc_part, n_pieces, grid; size_t lds, ws_bytes; };

static bool patch_plan(int n, int c, int r, int s, int k, int stride, int oh, int ow, PatchPlan& pl) {
  if (stride != 1 || !((r == 3 && s == 3) || (r == 1 && s == 1)) || (c & 15) || (k & 3)) return false;
  pl.bn = k >= 128 ? 128 : 64;
  pl.LR = patch_rows(n, oh, ow, r, s);
  if (pl.LR > 512) return false;
  pl.lds = 2 * (size_t)3 * pl.LR * 32 + 2 * (size_t)s * 3 * pl.bn * 32;
  if (pl.lds > 160 * 1024) return false;
  const long M = (long)n * oh * ow;
  pl.tiles = cdiv(M, 256) * cdiv(k, pl.bn);
  pl.grid = 256;
  static const int grid_knob = (int)env_long("EMBNET_PATCH_GRID", 256);
  pl.grid = grid_knob;
  const int ncc = c / 16;
  pl.n_full = pl.tiles / pl.grid * pl.grid;
  const int rem = pl.tiles - pl.n_full;
  pl.parts = 1; pl.cc_part = ncc; pl.n_pieces = 0; pl.ws_bytes = 0;
  if (rem > 0) {
    int parts = pl.grid / rem; if (parts > ncc) parts = ncc; if (parts < 1) parts = 1;
    pl.cc_part = cdiv(ncc, parts); pl.parts = cdiv(ncc, pl.cc_part);
    if (pl.parts == 1) { pl.n_full = pl.tiles; }                            // whole tiles: nothing to fix up
    else { pl.n_pieces = rem * pl.parts; pl.ws_bytes = (size_t)pl.n_pieces * 256 * pl.bn * 4; }
  }
  return true;
}

extern "C" int embnet_conv2d_patch_supported(int n, int c, int r, int s, int k, int stride, int oh, int ow) {
  PatchPlan pl; return patch_plan(n, c, r, s, k, stride, oh, ow, pl) ? 1 : 0;
}
extern "C" size_t embnet_conv2d_patch_workspace_bytes(int n, int c, int r, int s, int k, int stride, int oh, int ow) {
  PatchPlan pl; return patch_plan(n, c, r, s, k, stride, oh, ow, pl) ? pl.ws_bytes : 0;
}
extern "C" int embnet_conv2d_patch_stats_rows(int n, int oh, int ow) { return cdiv((long)n * oh * ow, 256) * 4; }

// planes[3][rows][R*S*inner] from a Keras kernel w[r,s,c,k]: flip 0 for the forward pass (rows = k), 1 for the stride-1 data
// gradient (rows = c, taps flipped)
extern "C" int embnet_prep_weight_planes(const float* w, int r, int s, int c, int k, int flip, void* planes, void* stream) {
  EMBNET_CHECK_ARG(w && planes && r > 0 && s > 0 && c > 0 && k > 0, "prep_weight_planes: bad argument");
  EMBNET_CHECK_ARG(((flip ? k : c) & 3) == 0, "prep_weight_planes: inner channel count %% 4");
  const long total4 = (long)r * s * c * k / 4;
  EMBNET_TRACE("embnet::prep_weight_planes_kernel", TRACE_BYTES, 10.0 * r * s * c * k, stream);
  prep_weight_planes_kernel<<<(int)(total4 / 256 + 1 > 2048 ? 2048 : total4 / 256 + 1), 256, 0, (hipStream_t)stream>>>(
      w, r, s, c, k, flip, (unsigned short*)planes);
  return check_launch("prep_weight_planes");
}

template <int BN, int R, int S>
static void launch_patch(const ConvPatchParams& p, size_t lds, hipStream_t st) {
  static bool once = false;
  if (!once) { (void)hipFuncSetAttribute((const void*)conv_patch_kernel<BN, R, S>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
  conv_patch_kernel<BN, R, S><<<p.grid, 512, lds, st>>>(p);
}

// y[n,oh,ow,k] = conv(x planes, w planes) (+ bias, relu, residual, statistics as embnet_conv2d_fwd_f32), stride 1.
// The same entry point computes a stride-1 data gradient: x planes = dy, w planes prepared with flip = 1, the roles of c and
// k swapped, pad = kernel - 1 - pad, residual = the gradient of the tensor's other consumer.
extern "C" int embnet_conv2d_patch_planes(const void* xp, const void* wp, const float* bias, float* y, int n, int h, int wd, int c,
                                          int r, int s, int k, int pad_t, int pad_l, int oh, int ow, int relu,
                                          const float* residual, float* stats, void* workspace, size_t workspace_bytes,
                                          void* stream) {
  EMBNET_CHECK_ARG(xp && wp && y, "conv2d_patch_planes: null pointer");
  PatchPlan pl;
  EMBNET_CHECK_ARG(patch_plan(n, c, r, s, k, 1, oh, ow, pl), "conv2d_patch_planes: unsupported geometry (see conv2d_patch_supported)");
  ConvPatchParams p{(const unsigned short*)xp, (const unsigned short*)wp, y, bias, residual, stats, 0, relu};
  if (int rc = make_geom(p.g, n, h, wd, c, r, s, k, 1, pad_t, pad_l, oh, ow, "conv2d_patch_planes")) return rc;
  const long M = (long)n * oh * ow;
  p.x_plane_bytes = (unsigned)((size_t)n * h * wd * c * 2);
  p.w_plane_bytes = (unsigned)((size_t)r * s * c * k * 2);
  p.PH = oh + r - 1; p.PW = ow + s - 1;
  p.dPHW = FastDiv::make(p.PH * p.PW); p.dPW = FastDiv::make(p.PW);
  p.LR = pl.LR;
  p.stats_rows = cdiv(M, 256) * 4;
  p.grid = pl.grid;
  if (pl.n_pieces > 0 && (pl.ws_bytes > workspace_bytes || !workspace)) { pl.n_full = pl.tiles; pl.n_pieces = 0; pl.parts = 1; }
  p.n_full = pl.n_full; p.parts = pl.parts; p.cc_part = pl.cc_part; p.n_pieces = pl.n_pieces; p.ws = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  {
    EMBNET_TRACE_FLOP(pl.bn == 128 ? (r == 3 ? "conv_patch<128,3x3>" : "conv_patch<128,1x1>") : (r == 3 ? "conv_patch<64,3x3>" : "conv_patch<64,1x1>"),
                      2.0 * M * k * r * s * c,
                      6.0 * ((double)n * h * wd * c + (double)r * s * c * k) + 4.0 * (double)M * k * (residual ? 2 : 1), st);
    if (pl.bn == 128) { if (r == 3) launch_patch<128, 3, 3>(p, pl.lds, st); else launch_patch<128, 1, 1>(p, pl.lds, st); }
    else { if (r == 3) launch_patch<64, 3, 3>(p, pl.lds, st); else launch_patch<64, 1, 1>(p, pl.lds, st); }
  }
  if (p.n_pieces > 0)
    launch_tail_fixup(p.ws, p.parts, 256, pl.bn, 64, p.n_full, pl.tiles - p.n_full, cdiv(k, pl.bn), M, k, bias, relu, residual, y,
                      stats, p.stats_rows, st);
  return check_launch("conv2d_patch_planes");
}

// ---- patch convolution, second version: host side ---------------------------------------------------------------------
extern "C" int embnet_split_planes_cm_f32(const float* x, long pixels, int c, void* planes, void* stream) {
  EMBNET_CHECK_ARG(x && planes && pixels > 0 && c > 0 && (c & 15) == 0, "split_planes_cm: need c %% 16 == 0");
  const long n4 = pixels * c / 4;
  EMBNET_TRACE("embnet::split_planes_cm_kernel", TRACE_BYTES, 10.0 * pixels * c, stream);
  split_planes_cm_kernel<<<(int)(n4 / 256 + 1 > 4096 ? 4096 : n4 / 256 + 1), 256, 0, (hipStream_t)stream>>>(x, pixels, c, (unsigned short*)planes);
  return check_launch("split_planes_cm");
}
extern "C" int embnet_prep_weight_planes2(const float* w, int r, int s, int c, int k, int flip, void* planes, void* stream) {
  EMBNET_CHECK_ARG(w && planes && r > 0 && s > 0 && c > 0 && k > 0, "prep_weight_planes2: bad argument");
  EMBNET_CHECK_ARG(((flip ? k : c) & 15) == 0, "prep_weight_planes2: reduction channel count %% 16");
  const long total4 = (long)r * s * c * k / 4;
  prep_weight_planes2_kernel<<<(int)(total4 / 256 + 1 > 2048 ? 2048 : total4 / 256 + 1), 256, 0, (hipStream_t)stream>>>(
      w, r, s, c, k, flip, (unsigned short*)planes);
  return check_launch("prep_weight_planes2");
}

struct Patch2Plan { int bn, tps, nbs, LR, tiles, n_full, parts, cc_part, n_pieces, grid; size_t lds, ws_bytes; };
static bool patch2_plan(int n, int c, int r, int s, int k, int oh, int ow, Patch2Plan& pl) {
  if (!(r == 3 && s == 3) || (c & 15) || (k & 3)) return false;
  pl.bn = k >= 128 ? 128 : 64;
  pl.tps = pl.bn == 64 ? 3 : 1;
  pl.nbs = pl.bn == 64 ? 3 : 6;
  static const int nbs_knob = (int)env_long("EMBNET_PATCH_NBS", 0);
  if (nbs_knob > 0) pl.nbs = nbs_knob;
  pl.LR = patch_rows(n, oh, ow, r, s);
  if (pl.LR > 512) return false;
  pl.lds = 2 * (size_t)3 * pl.LR * 32 + (size_t)pl.nbs * pl.tps * 3 * pl.bn * 32;
  if (pl.lds > 160 * 1024) return false;
  const long M = (long)n * oh * ow;
  pl.tiles = cdiv(M, 256) * cdiv(k, pl.bn);
  static const int grid_knob = (int)env_long("EMBNET_PATCH_GRID", 256);
  pl.grid = grid_knob;
  const int ncc = c / 16;
  pl.n_full = pl.tiles / pl.grid * pl.grid;
  const int rem = pl.tiles - pl.n_full;
  pl.parts = 1; pl.cc_part = ncc; pl.n_pieces = 0; pl.ws_bytes = 0;
  if (rem > 0) {
    int parts = pl.grid / rem; if (parts > ncc) parts = ncc; if (parts < 1) parts = 1;
    pl.cc_part = cdiv(ncc, parts); pl.parts = cdiv(ncc, pl.cc_part);
    if (pl.parts == 1) { pl.n_full = pl.tiles; }
    else { pl.n_pieces = rem * pl.parts; pl.ws_bytes = (size_t)pl.n_pieces * 256 * pl.bn * 4; }
  }
  return true;
}
extern "C" int embnet_conv2d_patch2_supported(int n, int c, int r, int s, int k, int oh, int ow) {
  Patch2Plan pl; return patch2_plan(n, c, r, s, k, oh, ow, pl) ? 1 : 0;
}
extern "C" size_t embnet_conv2d_patch2_workspace_bytes(int n, int c, int r, int s, int k, int oh, int ow) {
  Patch2Plan pl; return patch2_plan(n, c, r, s, k, oh, ow, pl) ? pl.ws_bytes : 0;
}

template <int BN, int TPS, int NBS>
static void launch_patch2(const ConvPatch2Params& p, size_t lds, hipStream_t st) {
  static bool once = false;
  if (!once) { (void)hipFuncSetAttribute((const void*)conv_patch2_kernel<BN, 3, 3, TPS, NBS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
  conv_patch2_kernel<BN, 3, 3, TPS, NBS><<<p.grid, 640, lds, st>>>(p);
}

extern "C" int embnet_conv2d_patch2_planes(const void* xp, const void* wp, const float* bias, float* y, int n, int h, int wd, int c,
                                           int r, int s, int k, int pad_t, int pad_l, int oh, int ow, int relu,
                                           const float* residual, float* stats, void* workspace, size_t workspace_bytes,
                                           void* stream) {
  EMBNET_CHECK_ARG(xp && wp && y, "conv2d_patch2_planes: null pointer");
  Patch2Plan pl;
  EMBNET_CHECK_ARG(patch2_plan(n, c, r, s, k, oh, ow, pl), "conv2d_patch2_planes: unsupported geometry");
  ConvPatch2Params p{(const unsigned short*)xp, (const unsigned short*)wp, y, bias, residual, stats, 0, relu};
  if (int rc = make_geom(p.g, n, h, wd, c, r, s, k, 1, pad_t, pad_l, oh, ow, "conv2d_patch2_planes")) return rc;
  const long M = (long)n * oh * ow;
  p.x_plane_bytes = (unsigned)((size_t)n * h * wd * c * 2);
  p.w_plane_bytes = (unsigned)((size_t)r * s * c * k * 2);
  p.PH = oh + r - 1; p.PW = ow + s - 1;
  p.dPHW = FastDiv::make(p.PH * p.PW); p.dPW = FastDiv::make(p.PW);
  p.LR = pl.LR;
  p.stats_rows = cdiv(M, 256) * 4;
  p.grid = pl.grid;
  if (pl.n_pieces > 0 && (pl.ws_bytes > workspace_bytes || !workspace)) { pl.n_full = pl.tiles; pl.n_pieces = 0; pl.parts = 1; }
  p.n_full = pl.n_full; p.parts = pl.parts; p.cc_part = pl.cc_part; p.n_pieces = pl.n_pieces; p.ws = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  {
    EMBNET_TRACE_FLOP(pl.bn == 128 ? "conv_patch2<128>" : "conv_patch2<64>", 2.0 * M * k * r * s * c,
                      6.0 * ((double)n * h * wd * c + (double)r * s * c * k) + 4.0 * (double)M * k * (residual ? 2 : 1), st);
    if (pl.bn == 128) {
      if (pl.nbs == 6) launch_patch2<128, 1, 6>(p, pl.lds, st); else launch_patch2<128, 1, 4>(p, pl.lds, st);
    } else {
      if (pl.nbs == 3) launch_patch2<64, 3, 3>(p, pl.lds, st); else launch_patch2<64, 3, 4>(p, pl.lds, st);
    }
  }
  if (p.n_pieces > 0)
    launch_tail_fixup(p.ws, p.parts, 256, pl.bn, 64, p.n_full, pl.tiles - p.n_full, cdiv(k, pl.bn), M, k, bias, relu, residual, y,
                      stats, p.stats_rows, st);
  return check_launch("conv2d_patch2_planes");
}
