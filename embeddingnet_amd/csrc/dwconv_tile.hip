// Stride-1 depthwise convolution (forward, and the data gradient as a correlation with the flipped kernel) on SMALL maps from an
// LDS tile: EfficientNet's 28x28 ... 7x7 MBConv stages (the `efficientnet` zoo package instantiated at
// /root/reference/embedding_net/backbones.py:84-98), NHWC fp32.
//
// Why (profiles/r05_c5_conv_launches.txt, profiles/r05_dw_pmc_sq_summary.txt): mbconv_kernels.hip's row kernels keep a thread's
// input window in registers and fetch it with per-thread 16-byte loads — every input quad is requested (KS + 1) / 2 x 1.4 times,
// by workgroups on different CUs.  On the 112x112 / 56x56 maps they stream at 4.1-4.9 TB/s; from 28x28 down the launches are short
// and wait on memory (SQ_WAIT_ANY 50-67 % of the wave cycles): 1.7-3.0 TB/s.
//
// Form.  A workgroup owns a CHUNK of CQ channel quads (8 = 128 B of every pixel; 4 where C is not a multiple of 32) and walks
// over UNITS: a unit is a group of G whole images (G x H x W positions; G = 4 for 7x7 maps) or, where an image does not fit, a
// band of BH output rows of one image with its KS - 1 halo rows.
//  * the unit's input tile is fetched ONCE by LDS-DMA (buffer_load_dwordx4 ... lds: lane l of an instruction = float4
//    f = 256 k + tid of the tile, f = position * CQ + quad, so the LDS image is position-major and a pixel's chunk is one
//    contiguous global segment); rows outside the image (bands) and images past the batch are out-of-range offsets = zeros;
//  * two LDS buffers: the DMA of unit i + 1 is issued at the top of unit i, behind the barrier that ends unit i - 1;  what
//    orders the reads behind the DMA is s_waitcnt vmcnt(stores of the previous unit) + that barrier (vector-memory operations
//    retire in issue order, so the stores just issued may stay in flight);
//  * a thread computes TW = 7 output columns of one row for its quad (all EfficientNet maps at 224 are multiples of 7 wide;
//    two units next to each other in a 16-lane group are then an ODD number of positions apart: with 128-byte pixels their
//    ds_read_b128 cover all 64 banks); taps outside the image read a zero region of LDS (one select per read, no branches);
//    the weights of the chunk sit in LDS (flipped for the data gradient);
//  * outputs go out as buffer stores (masked columns / images: out-of-range offsets, so every wave issues the same number of
//    vector-memory operations per unit — the counted wait depends on it);
//  * STATS = 1: per-channel sum / sum of squares of the outputs (the following BatchNormalization's statistics partials),
//    STATS = 2: the BatchNorm-backward sums of the layer in front (dz = dx * act'(BN(e)), e read at the thread's own outputs
//    with ordinary loads issued in front of the taps): accumulated per thread over the workgroup's units, added over the
//    workgroup in a fixed order, row (image-group index) of [2][C][P].  Every (channel, row) is written.
// Bytes: 4 (in + out) per element (+ 4 for e), each fetched once (bands: the halo rows twice).
#include "dw_geom.h"
#include "../../include/embnet.h"

namespace embnet {
namespace dwt {

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned OOB = 0x80000000u;
constexpr int TW = 7, NPMAX = 2;

// lane l's 16 bytes at (rsrc base + voff) -> LDS byte lds + 16 l (M0 = LDS base).  Inline asm: behind the builtin form the
// compiler orders every later ds_read of the wave behind s_waitcnt vmcnt(0) (conv_wgrad_planes.hip); M0 is reserved to the
// compiler and cannot be named as a clobber — nothing else in this kernel uses it.
__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned lds, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(lds), "v"(voff), "s"(rsrc) : "memory");
}

struct Params {
  const float* x; const float* w; float* y; float* stats;
  const float* dy;             // weight gradient: the output gradient (y = the partial slabs [PG][KS*KS*C])
  DwBn bn;
  int N, H, W, C, pad_t, pad_l;
  int G, BH, bands, IR;        // whole images: bands = 1, BH = H, IR = G * H;  bands: G = 1, IR = BH + KS - 1
  int tile_f4, buf_bytes;      // float4s of a tile (IR * W * CQ); bytes of an LDS buffer (whole 1 KB wave blocks)
  int CB, UT, npass;           // column blocks per row, (row, column block) units of a tile, passes of 256 / CQ units
  int units_total, upw, PG, chunks;
};

// where unit u starts in the input and the output tensor (bytes; the chunk's offset not included), the byte window of the tile
// rows that exist (lo_b, hi_b: rows outside the image / images past the batch are fetched as zeros) and of the outputs (ohi_b)
struct UnitGeom { long in_origin, out_origin; unsigned lo_b, hi_b, ohi_b; };
__device__ __forceinline__ UnitGeom unit_geom(const Params& p, int u, unsigned pix_b) {
  UnitGeom ug;
  if (p.bands == 1) {
    const int n0 = u * p.G;
    ug.in_origin = ug.out_origin = (long)n0 * p.H * p.W * pix_b;
    ug.lo_b = 0; ug.hi_b = ug.ohi_b = (unsigned)(min(p.G, p.N - n0) * p.H * p.W) * pix_b;
  } else {
    const int n = u / p.bands, b = u - n * p.bands, row0 = b * p.BH - p.pad_t;
    ug.in_origin = ((long)n * p.H + row0) * p.W * pix_b;
    ug.lo_b = (unsigned)(max(0, -row0) * p.W) * pix_b; ug.hi_b = (unsigned)(min(p.IR, p.H - row0) * p.W) * pix_b;
    ug.out_origin = ((long)n * p.H + (long)b * p.BH) * p.W * pix_b; ug.ohi_b = (unsigned)(p.BH * p.W) * pix_b;
  }
  return ug;
}

// the LDS-DMA requests of one tile: lane l of request k = float4 256 k + tid of the tile (doff: its byte offset from the unit's
// first row, 0xFFFFFFFF past the tile)
template <int CQ, int NDX>
__device__ __forceinline__ void issue_tile(const Params& p, const unsigned (&doff)[NDX], const UnitGeom& ug, int chunk, int wave, unsigned dst) {
  const uint64_t base = (uint64_t)p.x + (uint64_t)(ug.in_origin + (long)chunk * CQ * 16);
  const i32x4 rs = {__builtin_amdgcn_readfirstlane((int)(uint32_t)base),
                    __builtin_amdgcn_readfirstlane((int)((base >> 32) & 0xffffu)), (int)OOB, 0x00020000};
#pragma unroll
  for (int k = 0; k < NDX; ++k)
    if (k * 256 + wave * 64 < p.tile_f4) {                    // wave-uniform: the tile ends inside some wave's 1 KB block
      const unsigned vo = (doff[k] >= ug.lo_b && doff[k] < ug.hi_b) ? doff[k] : OOB;
      dma16(rs, __builtin_amdgcn_readfirstlane(dst + (unsigned)wave * 1024u + (unsigned)k * 4096u), vo);
    }
}

// one thread's (row, column block) unit of a tile: LDS byte offset of its window's first float4 (xb), output byte offset (ooff),
// masks (bits 0..KS-1 window rows inside the image, 5..5+NX-1 window columns, 16..22 output columns)
template <int KS, int CQ>
__device__ __forceinline__ void thread_unit(const Params& p, int u, int q, unsigned pix_b, int& xb, unsigned& ooff, unsigned& msk) {
  constexpr int NX = TW + KS - 1;
  const bool whole = p.bands == 1, uv = u < p.UT;
  const int uu = uv ? u : 0;
  const int orow = uu / p.CB, cb = uu - orow * p.CB;
  const int oh = whole ? orow % p.H : orow;
  const int iw0 = cb * TW - p.pad_l;
  unsigned m = 0;
#pragma unroll
  for (int r = 0; r < KS; ++r) m |= (!whole || (unsigned)(oh + r - p.pad_t) < (unsigned)p.H) ? 1u << r : 0u;
#pragma unroll
  for (int j = 0; j < NX; ++j) m |= ((unsigned)(iw0 + j) < (unsigned)p.W) ? 1u << (5 + j) : 0u;
#pragma unroll
  for (int t = 0; t < TW; ++t) m |= (uv && cb * TW + t < p.W) ? 1u << (16 + t) : 0u;
  msk = m;
  xb = (((whole ? orow - p.pad_t : orow) * p.W + iw0) * CQ + q) * 16;
  ooff = (unsigned)(orow * p.W + cb * TW) * pix_b + (unsigned)q * 16u;
}

template <int KS, bool FLIP, int STATS, int CQ>
__global__ __launch_bounds__(256) void dw_tile_kernel(const Params p) {
  constexpr int NX = TW + KS - 1, SLOTS = 256 / CQ, NDX = CQ == 8 ? 7 : 13;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // blocks b, b + 8, ... share an XCD (observed placement; speed only): neighbouring chunks — the two 64-byte halves of a
  // 128-byte line where CQ = 4 — run next to each other on one L2
  const int per_xcd = gridDim.x >> 3;
  const int id = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (id >= p.chunks * p.PG) return;
  const int chunk = id % p.chunks, pg = id / p.chunks;
  const int q = tid % CQ, slot = tid / CQ;
  const int c4 = p.C >> 2;
  const unsigned pix_b = (unsigned)p.C * 4u;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem;
  const unsigned wl_off = 2u * (unsigned)p.buf_bytes, z_off = wl_off + KS * KS * CQ * 16;
  const int u0 = pg * p.upw;
  const int nu = max(0, min(p.upw, p.units_total - u0));

  // ---- the chunk's weights (flipped for the data gradient) and the zero region -----------------------------------------------
  {
    float4* wl = reinterpret_cast<float4*>(smem + wl_off);
    for (int i = tid; i < KS * KS * CQ; i += 256) {
      const int tap = i / CQ, qq = i % CQ;
      wl[i] = reinterpret_cast<const float4*>(p.w)[(FLIP ? KS * KS - 1 - tap : tap) * c4 + chunk * CQ + qq];
    }
    float4* zl = reinterpret_cast<float4*>(smem + z_off);
    for (int i = tid; i < NX * CQ; i += 256) zl[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // ---- load stream: this thread's float4s of a tile, relative to the unit's first row -----------------------------------------
  unsigned doff[NDX];
#pragma unroll
  for (int k = 0; k < NDX; ++k) {
    const int f = k * 256 + tid, pos = f / CQ, qq = f % CQ;
    doff[k] = f < p.tile_f4 ? (unsigned)pos * pix_b + (unsigned)qq * 16u : 0xFFFFFFFFu;
  }
  auto issue = [&](int u, int buf) {
    issue_tile<CQ, NDX>(p, doff, unit_geom(p, u, pix_b), chunk, wave, lds0 + (unsigned)buf * (unsigned)p.buf_bytes);
  };

  // ---- this thread's units of a tile (the same in every unit) ----------------------------------------------------------------------
  int xb[NPMAX]; unsigned ooff[NPMAX], msk[NPMAX];             // msk: bits 0..KS-1 rows, 5..5+NX-1 input columns, 16..22 output columns
#pragma unroll
  for (int ps = 0; ps < NPMAX; ++ps) thread_unit<KS, CQ>(p, ps * SLOTS + slot, q, pix_b, xb[ps], ooff[ps], msk[ps]);
  float4 bsc, bsh, bmu, brs;
  if (STATS == 2) {
    bsc = reinterpret_cast<const float4*>(p.bn.scale)[chunk * CQ + q]; bsh = reinterpret_cast<const float4*>(p.bn.shift)[chunk * CQ + q];
    bmu = reinterpret_cast<const float4*>(p.bn.mean)[chunk * CQ + q]; brs = reinterpret_cast<const float4*>(p.bn.rstd)[chunk * CQ + q];
  }
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  const float4* const wlq = reinterpret_cast<const float4*>(smem + wl_off) + q;
  const unsigned zq = z_off + (unsigned)q * 16u;
  __syncthreads();                                             // weights and zeros are in place (no DMA is in flight yet)

  if (nu > 0) issue(u0, 0);
  for (int i = 0; i < nu; ++i) {
    // unit i has landed (this wave's part): everything but the stores of unit i - 1 has retired
    if (i == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (p.npass == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(TW) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * TW) : "memory");
    __builtin_amdgcn_s_barrier();                              // ... everybody's; unit i - 1 is computed everywhere
    if (i + 1 < nu) issue(u0 + i + 1, (i + 1) & 1);

    const int u = u0 + i;
    const UnitGeom ug = unit_geom(p, u, pix_b);
    const unsigned ohi_b = ug.ohi_b;
    const long out_origin = ug.out_origin + (long)chunk * CQ * 16;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(p.y) + out_origin, 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t ers = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(STATS == 2 ? p.bn.e : p.x)) + out_origin, 0, (int)OOB, 0x00020000);
    const unsigned bufo = (unsigned)(i & 1) * (unsigned)p.buf_bytes;

#pragma unroll 1
    for (int ps = 0; ps < p.npass; ++ps) {
      const unsigned m = ps ? msk[1] : msk[0];
      const unsigned oo = ps ? ooff[1] : ooff[0];
      const int xbp = ps ? xb[1] : xb[0];
      unsigned vo[TW];
#pragma unroll
      for (int t = 0; t < TW; ++t) {
        const unsigned o = oo + (unsigned)t * pix_b;
        vo[t] = ((m >> (16 + t)) & 1u) && o < ohi_b ? o : OOB;
      }
      float4 ev[TW];
      if (STATS == 2) {
#pragma unroll
        for (int t = 0; t < TW; ++t) ev[t] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ers, (int)vo[t], 0, 0));
      }
      f32x2 acc[TW][2];
#pragma unroll
      for (int t = 0; t < TW; ++t) { acc[t][0] = f32x2{0.f, 0.f}; acc[t][1] = f32x2{0.f, 0.f}; }
      // one kernel row of the window in registers, the next one requested in front of this row's multiplies (220 registers with all KS
      // rows' reads hoisted, as hipcc schedules the plain loop)
      float4 xr[2][NX], wr[2][KS];
      auto load_row = [&](int r, float4 (&dst)[NX], float4 (&wdst)[KS]) {
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) wdst[s_] = wlq[(r * KS + s_) * CQ];
        const bool rok = (m >> r) & 1u;
        const unsigned rowa = bufo + (unsigned)(xbp + r * p.W * CQ * 16);
#pragma unroll
        for (int j = 0; j < NX; ++j) {
          const bool ok = rok && ((m >> (5 + j)) & 1u);
          dst[j] = *reinterpret_cast<const float4*>(smem + ((ok ? rowa : zq) + (unsigned)(j * CQ * 16)));
        }
      };
      load_row(0, xr[0], wr[0]);
#pragma unroll
      for (int r = 0; r < KS; ++r) {
        if (r + 1 < KS) load_row(r + 1, xr[(r + 1) & 1], wr[(r + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);                       // ... and the next row's reads in FRONT of this row's multiplies
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
          const float4 wv = wr[r & 1][s_];
          const f32x2 w01 = {wv.x, wv.y}, w23 = {wv.z, wv.w};
#pragma unroll
          for (int t = 0; t < TW; ++t) {
            const float4 v = xr[r & 1][t + s_];
            acc[t][0] = __builtin_elementwise_fma(f32x2{v.x, v.y}, w01, acc[t][0]);
            acc[t][1] = __builtin_elementwise_fma(f32x2{v.z, v.w}, w23, acc[t][1]);
          }
        }
        // pins this row's multiplies HERE (hipcc otherwise sinks all KS rows' multiplies behind all their reads) and keeps
        // the reads of row r + 2 behind them
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]),
                          "+v"(acc[3][0]), "+v"(acc[3][1]), "+v"(acc[4][0]), "+v"(acc[4][1]), "+v"(acc[5][0]), "+v"(acc[5][1]),
                          "+v"(acc[6][0]), "+v"(acc[6][1]) :: "memory");
      }
#pragma unroll
      for (int t = 0; t < TW; ++t) {
        const float4 v = make_float4(acc[t][0].x, acc[t][0].y, acc[t][1].x, acc[t][1].y);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, (int)vo[t], 0, 0);
        if (STATS && vo[t] != OOB) {
          if (STATS == 1) {
            s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
            s2.x = fmaf(v.x, v.x, s2.x); s2.y = fmaf(v.y, v.y, s2.y); s2.z = fmaf(v.z, v.z, s2.z); s2.w = fmaf(v.w, v.w, s2.w);
          } else {
            dw_bn_sums_add(p.bn, v, ev[t], bsc, bsh, bmu, brs, s1, s2);
          }
        }
      }
    }
  }

  // ---- statistics: the workgroup's sums per channel, slots added in order ---------------------------------------------------------
  if (STATS) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                           // the last unit's taps are read everywhere
    float4* red = reinterpret_cast<float4*>(smem);
    red[tid] = s1; red[256 + tid] = s2;
    __syncthreads();
    if (tid < CQ) {
      float4 a1 = red[tid], a2 = red[256 + tid];
      for (int sl = 1; sl < SLOTS; ++sl) {
        const float4 o1 = red[sl * CQ + tid], o2 = red[256 + sl * CQ + tid];
        a1.x += o1.x; a1.y += o1.y; a1.z += o1.z; a1.w += o1.w;
        a2.x += o2.x; a2.y += o2.y; a2.z += o2.z; a2.w += o2.w;
      }
      const long P = p.PG;
      float* d1 = p.stats + (long)(4 * (chunk * CQ + tid)) * P + pg;
      float* d2 = d1 + (long)p.C * P;
      d1[0] = a1.x; d1[P] = a1.y; d1[2 * P] = a1.z; d1[3 * P] = a1.w;
      d2[0] = a2.x; d2[P] = a2.y; d2[2 * P] = a2.z; d2[3 * P] = a2.w;
    }
  }
}

// ---- weight gradient --------------------------------------------------------------------------------------------------------------
// dW[r,s,c] = sum over outputs o of x[o + (r,s) - pad] * dy[o]: the same walk — the x tile by LDS-DMA, each thread's window rows
// from LDS — with the thread's seven dy quads in registers (ordinary loads, requested one unit AHEAD: hipcc waits vmcnt(0) at
// their first use, which would otherwise drain the DMA just issued) and KS x KS float4 sums per thread kept over all its units.
// The workgroup's sums (slots added in order, one kernel row at a time through LDS) are slab pg of [PG][KS*KS*C]; mbconv_kernels'
// dw_slab_sum_kernel adds the slabs.  One pass per tile only (npass == 1).
template <int KS, int CQ>
__global__ __launch_bounds__(256) void dw_tile_wgrad_kernel(const Params p) {
  constexpr int NX = TW + KS - 1, SLOTS = 256 / CQ, NDX = CQ == 8 ? 7 : 13;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per_xcd = gridDim.x >> 3;
  const int id = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (id >= p.chunks * p.PG) return;
  const int chunk = id % p.chunks, pg = id / p.chunks;
  const int q = tid % CQ, slot = tid / CQ;
  const unsigned pix_b = (unsigned)p.C * 4u;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem;
  const unsigned z_off = 2u * (unsigned)p.buf_bytes;
  const int u0 = pg * p.upw;
  const int nu = max(0, min(p.upw, p.units_total - u0));
  {
    float4* zl = reinterpret_cast<float4*>(smem + z_off);
    for (int i = tid; i < NX * CQ; i += 256) zl[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  unsigned doff[NDX];
#pragma unroll
  for (int k = 0; k < NDX; ++k) {
    const int f = k * 256 + tid, pos = f / CQ, qq = f % CQ;
    doff[k] = f < p.tile_f4 ? (unsigned)pos * pix_b + (unsigned)qq * 16u : 0xFFFFFFFFu;
  }
  int xb; unsigned ooff, m;
  thread_unit<KS, CQ>(p, slot, q, pix_b, xb, ooff, m);
  const unsigned zq = z_off + (unsigned)q * 16u;
  f32x2 dwa[KS][KS][2];
#pragma unroll
  for (int r = 0; r < KS; ++r)
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_) { dwa[r][s_][0] = f32x2{0.f, 0.f}; dwa[r][s_][1] = f32x2{0.f, 0.f}; }
  __syncthreads();

  auto load_dy = [&](int u, float4 (&d)[TW]) {
    const UnitGeom ug = unit_geom(p, u, pix_b);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.dy)) + ug.out_origin + (long)chunk * CQ * 16, 0, (int)OOB, 0x00020000);
#pragma unroll
    for (int t = 0; t < TW; ++t) {
      const unsigned o = ooff + (unsigned)t * pix_b;
      d[t] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(drs, (int)(((m >> (16 + t)) & 1u) && o < ug.ohi_b ? o : OOB), 0, 0));
    }
  };
  float4 dyn[TW];
  if (nu > 0) {
    issue_tile<CQ, NDX>(p, doff, unit_geom(p, u0, pix_b), chunk, wave, lds0);
    load_dy(u0, dyn);
  }
  for (int i = 0; i < nu; ++i) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // tile i (this wave's part) and the dy quads of unit i
    __builtin_amdgcn_s_barrier();
    float4 dyc[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) dyc[t] = dyn[t];
    if (i + 1 < nu) {
      issue_tile<CQ, NDX>(p, doff, unit_geom(p, u0 + i + 1, pix_b), chunk, wave, lds0 + (unsigned)((i + 1) & 1) * (unsigned)p.buf_bytes);
      load_dy(u0 + i + 1, dyn);
    }
    const unsigned bufo = (unsigned)(i & 1) * (unsigned)p.buf_bytes;
#pragma unroll
    for (int r = 0; r < KS; ++r) {
      const bool rok = (m >> r) & 1u;
      const unsigned rowa = bufo + (unsigned)(xb + r * p.W * CQ * 16);
      float4 xr[NX];
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        const bool ok = rok && ((m >> (5 + j)) & 1u);
        xr[j] = *reinterpret_cast<const float4*>(smem + ((ok ? rowa : zq) + (unsigned)(j * CQ * 16)));
      }
#pragma unroll
      for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
        for (int t = 0; t < TW; ++t) {
          const float4 v = xr[t + s_], d = dyc[t];
          dwa[r][s_][0] = __builtin_elementwise_fma(f32x2{v.x, v.y}, f32x2{d.x, d.y}, dwa[r][s_][0]);
          dwa[r][s_][1] = __builtin_elementwise_fma(f32x2{v.z, v.w}, f32x2{d.z, d.w}, dwa[r][s_][1]);
        }
      // pins this row's multiplies here and the next row's reads behind them (registers)
      if (KS == 5)
        asm volatile("" : "+v"(dwa[r][0][0]), "+v"(dwa[r][0][1]), "+v"(dwa[r][1][0]), "+v"(dwa[r][1][1]), "+v"(dwa[r][2][0]), "+v"(dwa[r][2][1]),
                          "+v"(dwa[r][3][0]), "+v"(dwa[r][3][1]), "+v"(dwa[r][KS - 1][0]), "+v"(dwa[r][KS - 1][1]) :: "memory");
      else
        asm volatile("" : "+v"(dwa[r][0][0]), "+v"(dwa[r][0][1]), "+v"(dwa[r][1][0]), "+v"(dwa[r][1][1]), "+v"(dwa[r][2][0]), "+v"(dwa[r][2][1]) :: "memory");
    }
  }

  // ---- the workgroup's sums: one kernel row at a time through LDS, slots added in order ----------------------------------------------
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float4* red = reinterpret_cast<float4*>(smem);
  float* out = p.y + (long)pg * KS * KS * p.C;
#pragma unroll
  for (int r = 0; r < KS; ++r) {
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_) red[s_ * 256 + tid] = make_float4(dwa[r][s_][0].x, dwa[r][s_][0].y, dwa[r][s_][1].x, dwa[r][s_][1].y);
    __syncthreads();
    if (tid < KS * CQ) {
      const int s_ = tid / CQ, qq = tid % CQ;
      float4 a = red[s_ * 256 + qq];
      for (int sl = 1; sl < SLOTS; ++sl) {
        const float4 o = red[s_ * 256 + sl * CQ + qq];
        a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
      }
      reinterpret_cast<float4*>(out + (long)(r * KS + s_) * p.C)[chunk * CQ + qq] = a;
    }
    __syncthreads();
  }
}

// ---- host ---------------------------------------------------------------------------------------------------------------------------
struct Plan { bool ok; int CQ, G, BH, bands, IR, tile_f4, buf_bytes, CB, UT, npass, units_total, upw, PG, chunks; size_t lds; };

static Plan make_plan(const DwGeom& g, int stats_kind) {
  Plan pl{};
  static const long on = env_long("EMBNET_DW_TILE", 1);
  static const long max_pix = env_long("EMBNET_DW_TILE_MAX_PIXELS", 1024);     // maps up to 32 x 32
  static const long band_rows = env_long("EMBNET_DW_TILE_BAND", 0);            // A/B: force bands of this many output rows
  static const long blocks_target = env_long("EMBNET_DW_TILE_BLOCKS", 0);
  static const long min_units = env_long("EMBNET_DW_TILE_MIN_UNITS", 2);
  static const long prefer_bands = env_long("EMBNET_DW_TILE_PREFER_BANDS", 1);
  const int ks = g.R;
  if (!on || g.stride != 1 || g.R != g.S || (ks != 3 && ks != 5) || g.H != g.OH || g.W != g.OW || (g.C & 15)) return pl;
  if (g.pad_t < 0 || g.pad_t >= ks || g.pad_l < 0 || g.pad_l >= ks || g.W < 4 || (long)g.H * g.W > max_pix) return pl;
  if ((long)g.N * g.H * g.W * g.C * 4 >= 0x7FFFFFFFL) return pl;              // 32-bit byte offsets inside a unit only, but keep the margins simple
  // 8 quads (128 B of every pixel) where C allows it and G whole images fit, else 4 quads; bands of rows where no image fits
  const int hw = g.H * g.W;
  pl.CQ = (!(g.C & 31) && hw <= 224 && band_rows <= 0) ? 8 : 4;
  const int posmax = pl.CQ == 8 ? 224 : 832;                                    // NDX DMAs of 256 float4s per thread
  // bands: the tallest divisor of H whose tile fits (under `cap` bytes of LDS for the two buffers)
  auto band_height = [&](long cap) {
    for (int d = g.H; d >= 2 * ks; --d)                                          // (shorter bands: the halo rows would dominate)
      if (g.H % d == 0 && (d + ks - 1) * g.W <= posmax && (band_rows <= 0 || d <= band_rows) &&
          2L * (((long)(d + ks - 1) * g.W * pl.CQ + 63) / 64) * 1024 <= cap) return d;
    return 0;
  };
  int bh = 0;
  if (band_rows > 0 || hw > posmax) {
    bh = band_height(150 * 1024);
    if (!bh) return pl;
  } else if (2L * (((long)hw * pl.CQ + 63) / 64) * 1024 > 80 * 1024 && prefer_bands) {
    // one image = one workgroup per CU: two workgroups on half images run faster although the halo rows are fetched twice
    // (28x28x240, 5x5: 126 -> 108 us forward, 204 -> 171 us data gradient; profiles/r05_exp_dw_tile_knobs.txt)
    bh = band_height(76 * 1024);
    if (bh == g.H) bh = 0;
  }
  if (!bh) {
    pl.bands = 1; pl.BH = g.H;
    pl.G = posmax / hw; if (pl.G > g.N) pl.G = g.N;
    const int slots = 256 / pl.CQ, per_img = g.H * ((g.W + TW - 1) / TW);        // one pass per tile where one image allows it
    if (pl.G > 1 && pl.G * per_img > slots) pl.G = slots / per_img > 1 ? slots / per_img : 1;
    pl.IR = pl.G * g.H;
    pl.units_total = (g.N + pl.G - 1) / pl.G;
  } else {
    pl.bands = g.H / bh; pl.BH = bh; pl.G = 1; pl.IR = bh + ks - 1;
    pl.units_total = g.N * pl.bands;
  }
  pl.tile_f4 = pl.IR * g.W * pl.CQ;
  pl.buf_bytes = ((pl.tile_f4 + 63) / 64) * 1024;
  pl.CB = (g.W + TW - 1) / TW;
  pl.UT = (pl.bands == 1 ? pl.G * g.H : pl.BH) * pl.CB;
  pl.npass = (pl.UT + 256 / pl.CQ - 1) / (256 / pl.CQ);
  if (pl.npass > NPMAX) return pl;
  pl.chunks = g.C / (4 * pl.CQ);
  const size_t tail = (size_t)(ks * ks + TW + ks - 1) * pl.CQ * 16;
  pl.lds = 2 * (size_t)pl.buf_bytes + tail;
  if (pl.lds < 2 * 256 * 16) pl.lds = 2 * 256 * 16;                            // the statistics reduction's scratch
  if (pl.lds > 160 * 1024) return pl;
  // workgroups: every CU filled as far as LDS allows, two rounds of them, at least min_units units each
  const long per_cu = (long)(160 * 1024 / pl.lds) < 8 ? (long)(160 * 1024 / pl.lds) : 8;
  const long target = blocks_target > 0 ? blocks_target : 2 * 256 * per_cu;
  long pgs = target / pl.chunks; if (pgs < 1) pgs = 1;
  long upw = (pl.units_total + pgs - 1) / pgs;
  if (upw < min_units) upw = min_units;
  if (upw > pl.units_total) upw = pl.units_total;
  pl.upw = (int)upw;
  pl.PG = (int)((pl.units_total + upw - 1) / upw);
  (void)stats_kind;
  pl.ok = true;
  return pl;
}

bool tile_applies(const DwGeom& g, int stats_kind) { return make_plan(g, stats_kind).ok; }
static Plan wgrad_plan(const DwGeom& g) {
  static const long on = env_long("EMBNET_DW_TILE_WGRAD", 1);
  Plan pl = make_plan(g, 0);
  if (!on || !pl.ok || pl.npass != 1) { pl.ok = false; return pl; }
  const size_t need = 5 * 256 * 16;                                            // the reduction's scratch: KS float4 per thread
  if (pl.lds < need) pl.lds = need;
  return pl;
}
int tile_wgrad_slabs(const DwGeom& g) { const Plan pl = wgrad_plan(g); return pl.ok ? pl.PG : 0; }
int tile_stats_rows(const DwGeom& g, int stats_kind) { const Plan pl = make_plan(g, stats_kind); return pl.ok ? pl.PG : 0; }

template <int KS, bool FLIP, int STATS, int CQ>
static void launch_one(const Params& p, const Plan& pl, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dw_tile_kernel<KS, FLIP, STATS, CQ>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  dw_tile_kernel<KS, FLIP, STATS, CQ><<<(pl.chunks * pl.PG + 7) / 8 * 8, 256, pl.lds, st>>>(p);
}
template <int KS, bool FLIP, int STATS>
static void launch_cq(const Params& p, const Plan& pl, hipStream_t st) {
  if (pl.CQ == 8) launch_one<KS, FLIP, STATS, 8>(p, pl, st);
  else launch_one<KS, FLIP, STATS, 4>(p, pl, st);
}
template <int KS>
static void launch_ks(const Params& p, const Plan& pl, bool flip, int stats_kind, hipStream_t st) {
  if (!flip) { if (stats_kind == 1) launch_cq<KS, false, 1>(p, pl, st); else launch_cq<KS, false, 0>(p, pl, st); }
  else { if (stats_kind == 2) launch_cq<KS, true, 2>(p, pl, st); else launch_cq<KS, true, 0>(p, pl, st); }
}

void launch_tile(const float* x, const float* w, const DwGeom& g, bool flip, float* y, float* stats, const DwBn* bn, hipStream_t st) {
  const int stats_kind = !stats ? 0 : (flip ? 2 : 1);
  const Plan pl = make_plan(g, stats_kind);
  Params p{};
  p.x = x; p.w = w; p.y = y; p.stats = stats;
  if (bn) p.bn = *bn;
  p.N = g.N; p.H = g.H; p.W = g.W; p.C = g.C; p.pad_t = g.pad_t; p.pad_l = g.pad_l;
  p.G = pl.G; p.BH = pl.BH; p.bands = pl.bands; p.IR = pl.IR; p.tile_f4 = pl.tile_f4; p.buf_bytes = pl.buf_bytes;
  p.CB = pl.CB; p.UT = pl.UT; p.npass = pl.npass; p.units_total = pl.units_total; p.upw = pl.upw; p.PG = pl.PG; p.chunks = pl.chunks;
  if (g.R == 3) launch_ks<3>(p, pl, flip, stats_kind, st);
  else launch_ks<5>(p, pl, flip, stats_kind, st);
}

template <int KS, int CQ>
static void launch_wgrad_one(const Params& p, const Plan& pl, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dw_tile_wgrad_kernel<KS, CQ>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  dw_tile_wgrad_kernel<KS, CQ><<<(pl.chunks * pl.PG + 7) / 8 * 8, 256, pl.lds, st>>>(p);
}

void launch_tile_wgrad(const float* x, const float* dy, const DwGeom& g, float* slabs, hipStream_t st) {
  const Plan pl = wgrad_plan(g);
  Params p{};
  p.x = x; p.dy = dy; p.y = slabs;
  p.N = g.N; p.H = g.H; p.W = g.W; p.C = g.C; p.pad_t = g.pad_t; p.pad_l = g.pad_l;
  p.G = pl.G; p.BH = pl.BH; p.bands = pl.bands; p.IR = pl.IR; p.tile_f4 = pl.tile_f4; p.buf_bytes = pl.buf_bytes;
  p.CB = pl.CB; p.UT = pl.UT; p.npass = pl.npass; p.units_total = pl.units_total; p.upw = pl.upw; p.PG = pl.PG; p.chunks = pl.chunks;
  if (g.R == 3) { if (pl.CQ == 8) launch_wgrad_one<3, 8>(p, pl, st); else launch_wgrad_one<3, 4>(p, pl, st); }
  else { if (pl.CQ == 8) launch_wgrad_one<5, 8>(p, pl, st); else launch_wgrad_one<5, 4>(p, pl, st); }
}

}  // namespace dwt
}  // namespace embnet
