// Stride-1 convolution on PRE-SPLIT operands ("patch" kernel): forward and data gradient of the 3x3 layers that dominate
// the zoo ResNets (image-classifiers residual units, /root/reference/embedding_net/backbones.py:99-104).
//
// Why another conv kernel (round-3 measurements): the implicit-GEMM loop of conv.hip re-gathers every
// input pixel once per tap and splits it into its three bf16 pieces each time; its K-tile period turned out to be the
// round trip of the gather (L2 / Infinity-Cache rate per CU, ~2 us under load), not the matrix pipe.  Here
//  * the three bf16 pieces of every fp32 value ("planes", gemm_engine.h split4) are written ONCE by the tensor's producer
//    (embnet_affine_act_planes, embnet_bn_bwd's dx_planes) in a CHUNK-MAJOR layout [plane][C/16][pixels][16], and the
//    kernels' weights once per optimizer step (embnet_conv_weight_planes) STEP-MAJOR [plane][r][C/16][s][K][16];
//  * a workgroup (256 output pixels x BN channels) keeps a PATCH of the zero-padded input in LDS — the contiguous run of
//    padded-image positions its pixels' R x S windows cover, 16 channels at a time — and every tap reads its A fragments
//    from the patch at a constant row offset r*PW + s:  position(n, oh, ow) = n*PH*PW + oh*PW + ow,  PH = OH+R-1, PW = OW+S-1.
//    Each input value is fetched once per tile instead of R*S times; padding positions are written as zeros by the DMA
//    (out-of-range buffer offsets), so borders and image-to-image seams need no special case;
//  * operand tiles reach LDS by LDS-DMA (buffer_load ... lds) issued by two LOADER waves (wave 8: weights, wave 9:
//    patches) beside the eight MFMA waves: no register staging, no split arithmetic, no ds_write and no vmcnt wait in the
//    MFMA waves; with the layouts above a 1 KiB DMA covers 8 whole cache lines;
//  * weight ring of NBS slots requested NBS-1 steps ahead, patch double buffer requested a chunk ahead (a DMA takes ~2 us
//    to land under load); one s_barrier per step (TPS taps of one 16-channel chunk);
//  * the workgroup is persistent: it walks its output tiles with the loaders running ahead across tile boundaries;
//    left-over tiles (tiles mod grid) are cut along the channel chunks into equal pieces, one per workgroup
//    (raw partial tiles + conv.hip's tail_fixup_kernel).
// Same pieces and six-term products as gemm_mainloop3 (fp32-exact split, include/embnet.h); the summation order over k is
// (chunk, r, s, channel) instead of (r, s, channel).  LDS rows are 32 bytes per plane; the two 16-byte halves of a row are
// swapped when (row >> 3) & 1, which makes any 16 consecutive rows conflict-free for ds_read_b128 whatever the tap shift.
// Measured (profiles/r03_exp_patch2_ab.txt): 192 / 183 / 202 TFLOP/s fp32-equivalent on the 56x56x64, 28x28x128,
// 14x14x256 ResNet18 layers where conv.hip's forward kernel reaches 155 / 156 / 183.
#include "gemm_engine.h"
#include "conv_geom.h"
#include "../../include/embnet.h"

namespace embnet {
namespace patch {

typedef __attribute__((address_space(3))) void* lds_ptr;
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned char* lds, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)lds, 16, (int)voff, (int)soff, 0, 0);
}

template <int BN_>
struct GeomP {      // 8 MFMA waves as 4 (rows) x 2 (columns); 256 output pixels x BN channels
  static constexpr int BM = 256, BN = BN_, WAVES_M = 4, WAVES_N = 2, WTM = 64, WTN = BN_ / 2, TM = 2, TN = WTN / 32;
};

struct PatchParams {
  const unsigned short* xp;      // [3][C/16][N*H*W][16] bf16 pieces of the input
  const unsigned short* wp;      // [3][R][C/16][S][K][16] bf16 pieces of the kernel
  float* y; const float* bias; const float* residual; float* stats; int stats_rows; int relu;
  ConvGeom g; unsigned x_plane_bytes, w_plane_bytes;
  int PH, PW; FastDiv dPHW, dPW;
  int LR;                        // LDS patch rows (multiple of 32)
  int n_full, parts, cc_part, n_pieces, grid; float* ws;
  int xcd_rows;                  // work_item: the column tiles of a tile row on one XCD (EMBNET_PATCH_XCD_ROWS=1; off by default)
  BnSums bn;                     // data-gradient use: the BatchNorm-backward sums of the layer in front (bn.x == NULL: off)
  // conv1x1_a32_kernel: the activation operand is the fp32 tensor itself (xp points at floats [pixels][C]), split in the matrix
  // waves; its scale comes from its range slot
  const uint32_t* a_range; unsigned x32_bytes;
};

typedef const PatchParams __attribute__((address_space(4)))* kargp;
// The kernel's only argument, re-read from the kernel-argument segment where it is needed (scalar loads) instead of being
// kept in registers across the main loops: per-tile set-up uses a dozen dividers and geometry words.
__device__ __forceinline__ kargp kargs() {
  kargp pp = (kargp)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(pp));
  return pp;
}

struct Item { int m0, n0, cc_b, cc_e, tile_m; float* part; };

template <int BN, int CCH = 16>      // CCH: channels per reduction unit `cc` (a 16-channel chunk; the 1x1 kernel walks groups of chunks)
__device__ __forceinline__ Item work_item(int item, int n_mine) {
  kargp pp = kargs();
  const int b = blockIdx.x, NCC = pp->g.C / CCH, tiles_n = (pp->g.K + BN - 1) / BN;
  Item t; int id;
  if (item < n_mine) {
    // whole rounds of `grid` tiles.  Workgroups b, b + 8, b + 16 ... share an XCD (observed placement; speed only): where a row
    // of tiles has T = 2, 4, ... column tiles (256 / 512 filters) they go to T consecutive workgroups OF ONE XCD, so the input
    // patch they all read is fetched into one L2 once instead of into T of them (profiles/r05_pmc_traffic_c2.txt: 1.58 x the
    // algorithmic bytes per launch).  A bijection of the round's positions when T divides grid / 8; other T keep the plain order.
    int pos = b;
    const int per_xcd = pp->grid >> 3;
    if (pp->xcd_rows && tiles_n > 1 && (pp->grid & 7) == 0 && per_xcd % tiles_n == 0 && (item + 1) * pp->grid <= pp->n_full) {   // (whole rounds only)
      const int xcd = b & 7, slot = b >> 3;
      pos = ((slot / tiles_n) * 8 + xcd) * tiles_n + slot % tiles_n;
    }
    id = pos + item * pp->grid; t.cc_b = 0; t.cc_e = NCC; t.part = nullptr;
  }
  else {
    id = pp->n_full + b / pp->parts;
    t.cc_b = (b % pp->parts) * pp->cc_part; t.cc_e = min(NCC, t.cc_b + pp->cc_part);
    t.part = pp->ws + (long)b * (256 * BN);
  }
  t.tile_m = id / tiles_n; t.m0 = t.tile_m * 256; t.n0 = (id % tiles_n) * BN;
  return t;
}

__device__ __forceinline__ int padded_pos(int m) {            // padded-image position of output pixel m
  kargp pp = kargs();
  FastDiv dOHW, dOW;
  dOHW.mul = pp->g.dOHW.mul; dOHW.shift = pp->g.dOHW.shift; dOHW.d = pp->g.dOHW.d;
  dOW.mul = pp->g.dOW.mul; dOW.shift = pp->g.dOW.shift; dOW.d = pp->g.dOW.d;
  uint32_t n, rem, oh, ow;
  dOHW.divmod((uint32_t)m, n, rem); dOW.divmod(rem, oh, ow);
  return (int)n * (pp->PH * pp->PW) + (int)oh * pp->PW + (int)ow;
}

// one step of the main loop in the planes' format: six bf16 terms, or three fp16 terms (the fragments are bit patterns either way)
template <class G, bool F16, int NP>
__device__ __forceinline__ void step_planes(const bf16x8 (&a)[G::TM][NP], const bf16x8 (&b)[G::TN][NP], f32x16 (&acc)[G::TM][G::TN]) {
  if constexpr (F16) {
    f16x8 ah[G::TM][2], bh[G::TN][2];
#pragma unroll
    for (int i = 0; i < G::TM; ++i) { ah[i][0] = __builtin_bit_cast(f16x8, a[i][0]); ah[i][1] = __builtin_bit_cast(f16x8, a[i][1]); }
#pragma unroll
    for (int i = 0; i < G::TN; ++i) { bh[i][0] = __builtin_bit_cast(f16x8, b[i][0]); bh[i][1] = __builtin_bit_cast(f16x8, b[i][1]); }
    mfma_step_h<G::TM, G::TN>(ah, bh, acc);
  } else {
    mfma_step3<G>(a, b, acc);
  }
}

// BNS: the data-gradient form that also emits the BatchNorm-backward sums (PatchParams::bn) — an instantiation of its own: the
// 64 extra epilogue registers and scalar spills cost the plain kernel ~10 % when both forms share one body
// F16: the planes hold two fp16 pieces + a scale (gemm_engine.h, EMBNET_PLANES_F16): NP = 2 planes are fetched, held in LDS and read,
// three matrix products per step instead of six, the accumulators x 1 / (s_x s_w) in the epilogue
// The epilogue of one output tile, straight from the accumulators (shared by the 3x3 patch kernel and the 1x1 planes kernel):
// 1 / (s_x s_w) (F16), then a raw partial tile (K-split left-overs) or bias / ReLU / residual / BatchNorm statistics / backward sums.
template <class G, int BN, bool BNS, bool F16>
__device__ __forceinline__ void patch_epilogue(const PatchParams& p, const Item& cur, f32x16 (&acc)[G::TM][G::TN], int M, int K,
                                               int wave, int lane, int h, int wm, int wn) {
  constexpr int TM = G::TM, TN = G::TN;
  // epilogue straight from the accumulators: register rr of a 32x32 block holds row (rr&3) + 8*(rr>>2) + 4*h, column
  // lane & 31, so a store instruction writes two 128-byte row segments
  if (F16) {                                           // 1 / (s_x s_w): powers of two, exact
    // (one after the other: each factor is a normal number, their product need not be — gemm_engine.h scale_exponent_of)
    const float osx = p.a_range ? scale_pair(scale_exponent_of(__uint_as_float(*p.a_range))).y
                                : planes_scale_slot(p.xp, (long)(p.x_plane_bytes >> 1))[1];
    const float osw = planes_scale_slot(p.wp, (long)(p.w_plane_bytes >> 1))[1];
#pragma unroll
    for (int im = 0; im < TM; ++im)
#pragma unroll
      for (int in = 0; in < TN; ++in)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) acc[im][in][rr] = (acc[im][in][rr] * osw) * osx;
  }
  if (cur.part) {
    float* part = cur.part + (wm + 4 * h) * BN + wn + (lane & 31);
#pragma unroll
    for (int im = 0; im < TM; ++im)
#pragma unroll
      for (int in = 0; in < TN; ++in)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr)
          part[(im * 32 + (rr & 3) + 8 * (rr >> 2)) * BN + in * 32] = acc[im][in][rr];
  } else {
    const bool inner = cur.m0 + 256 <= M && cur.n0 + BN <= K;      // wave-uniform: no edge tests on interior tiles
#pragma unroll
    for (int in = 0; in < TN; ++in) {
      const int col = cur.n0 + wn + in * 32 + (lane & 31);
      const bool cok = col < K;
      const float bv = (p.bias && cok) ? p.bias[col] : 0.f;
      float s1 = 0.f, s2 = 0.f;
      // BNS — data gradient: the outputs are d(act(BN(e))): the BatchNorm-backward sums  sum dz, sum dz * ehat  of this row band
      // (dz = v * act'(scale e + shift)), e read here at 4 bytes per element (conv.hip's BnSums; the arithmetic of bn_sums_add).
      // All of the column block's e values are requested BEFORE the first store (the stores may alias them as far as hipcc
      // knows, and one exposed memory latency per 16 values costs 10 us per tile)
      constexpr bool bnm = BNS;
      float bsc = 0.f, bsh = 0.f, bmu = 0.f, brs = 0.f;
      float ev[BNS ? TM : 1][16];
      if (BNS) {
        if (cok) { bsc = p.bn.scale[col]; bsh = p.bn.shift[col]; bmu = p.bn.mean[col]; brs = p.bn.rstd[col]; }
#pragma unroll
        for (int im = 0; im < TM; ++im) {
          const long o0 = (long)(cur.m0 + wm + im * 32 + 4 * h) * K + col;
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) {
            const int row = cur.m0 + wm + im * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
            ev[BNS ? im : 0][rr] = (inner || (row < M && cok)) ? p.bn.x[o0 + (long)((rr & 3) + 8 * (rr >> 2)) * K] : 0.f;
          }
        }
      }
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
            if (BNS) {
              const float e = ev[BNS ? im : 0][rr], dz = act_grad(p.bn.act, fmaf(e, bsc, bsh), v[rr]);
              s1 += dz; s2 = fmaf(dz, (e - bmu) * brs, s2);
            } else { s1 += v[rr]; s2 = fmaf(v[rr], v[rr], s2); }
          }
        } else {
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) {
            const int row = cur.m0 + wm + im * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
            if (row < M && cok) {
              const long o = o0 + (long)((rr & 3) + 8 * (rr >> 2)) * K;
              if (p.residual) v[rr] += p.residual[o];
              p.y[o] = v[rr];
              if (BNS) {
                const float e = ev[BNS ? im : 0][rr], dz = act_grad(p.bn.act, fmaf(e, bsc, bsh), v[rr]);
                s1 += dz; s2 = fmaf(dz, (e - bmu) * brs, s2);
              } else { s1 += v[rr]; s2 = fmaf(v[rr], v[rr], s2); }
            }
          }
        }
      }
      if (p.stats || bnm) {             // BatchNorm statistics of the layer that follows (as conv.hip's epilogue) / backward sums
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (h == 0 && cok) {
          float* const dst = bnm ? p.bn.partial : p.stats;
          const long prow = (long)cur.tile_m * G::WAVES_M + wave / G::WAVES_N, P = bnm ? p.bn.rows : p.stats_rows;
          dst[(long)col * P + prow] = s1;
          dst[((long)K + col) * P + prow] = s2;
        }
      }
    }
  }

}

// PIPE (F16 only): the MFMA waves read the fragments of tap t + 1 while the matrix pipe works on tap t (two fragment sets in
// registers, reads and MFMAs interleaved one to one by sched_group_barrier), across the step barriers too for the patch fragments
// (the patch of a chunk stays put for its nine taps; only the weights of a step become valid at its barrier).  The plain loop reads
// a tap's eight fragments and then multiplies: hipcc serialises it into read - wait - 2 MFMAs - wait - 2 MFMAs ... with three to
// four exposed LDS round trips per 12 MFMAs (r05: 43 % MFMA-busy at 2.3 GHz).  DESIGN 3.14.
template <int BN, int R, int S, int TPS, int NBS, bool BNS = false, bool F16 = false, bool PIPE = false>
__global__ __launch_bounds__(640) void conv_patch_kernel(const PatchParams p) {
  using G = GeomP<BN>;
  constexpr int NP = F16 ? 2 : 3;
  constexpr int TM = G::TM, TN = G::TN, SPC = R * S / TPS, D = NBS - 1;
  constexpr int SBY = TPS * NP * BN * 32;                // one weight slot: TPS taps x NP planes x BN rows x 32 bytes
  constexpr int NBI = TPS * NP * (BN / 32);              // DMA instructions per weight slot
  static_assert((R * S) % TPS == 0 && SPC >= 2, "steps per chunk");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int LR = p.LR, PLP = LR * 32, PB = NP * PLP;     // patch plane / patch buffer bytes
  unsigned char* const bslot0 = smem + 2 * PB;
  const int b = blockIdx.x;
  const int n_mine = b < p.n_full ? (p.n_full - b + p.grid - 1) / p.grid : 0;
  const int n_items = n_mine + (b < p.n_pieces ? 1 : 0);
  if (n_items == 0) return;
  const int dhalf = (lane & 1) ^ ((lane >> 4) & 1);      // logical 16-byte half this lane's DMA piece holds

  if (wave == 8) {
    // ---- weight loader: slot (gs % NBS) <- weights of step gs, D steps ahead of the MFMA waves ----------------------
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.wp), 0, 3u * p.w_plane_bytes, 0x00020000);
    const int K = p.g.K, NCC = p.g.C / 16;
    const unsigned wpb = p.w_plane_bytes;
    int it = 0; Item t = work_item<BN>(0, n_mine);               // the step being REQUESTED: (it, cc, st)
    int cc = t.cc_b, st = 0; bool live = true;
    unsigned rowoff[BN / 32];                                    // per lane, per 32-row group of the tile: constant per tile
    auto tile_rows = [&]() {
#pragma unroll
      for (int gb = 0; gb < BN / 32; ++gb) {
        const int row = t.n0 + gb * 32 + (lane >> 1);
        rowoff[gb] = row < K ? 32u * (unsigned)row + 16u * dhalf : OOB;
      }
    };
    tile_rows();
    auto issue = [&](int gs) {
      unsigned char* slot = bslot0 + (gs % NBS) * SBY;
#pragma unroll
      for (int tp = 0; tp < TPS; ++tp) {
        const int tap = st * TPS + tp, r = tap / S, s = tap % S;
        const unsigned so = 32u * (unsigned)(((r * NCC + cc) * S + s) * K);
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
          for (int gb = 0; gb < BN / 32; ++gb)
            dma16(wr, slot + ((tp * NP + q) * BN + gb * 32) * 32, live ? rowoff[gb] : OOB, q * wpb + so);
      }
      if (live && ++st == SPC) {
        st = 0;
        if (++cc == t.cc_e) {
          if (++it < n_items) { t = work_item<BN>(it, n_mine); cc = t.cc_b; tile_rows(); } else live = false;
        }
      }
    };
    int total = 0;                                               // steps of this workgroup
    for (int i = 0; i < n_items; ++i) { const Item q = work_item<BN>(i, n_mine); total += (q.cc_e - q.cc_b) * SPC; }
    for (int gs = 0; gs < D; ++gs) issue(gs);
    for (int gs = 0; gs < total; ++gs) {
      // the D - 1 youngest requests may still be in flight: the weights of step gs have landed.  (Past the last step the
      // loader keeps issuing out-of-range requests — zeros into slots nobody reads — so the count stays a constant.)
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
    auto tile_offsets = [&](const Item& t) {
      kargp pp = kargs();
      FastDiv dPHW, dPW;
      dPHW.mul = pp->dPHW.mul; dPHW.shift = pp->dPHW.shift; dPHW.d = pp->dPHW.d;
      dPW.mul = pp->dPW.mul; dPW.shift = pp->dPW.shift; dPW.d = pp->dPW.d;
      const int P0 = padded_pos(t.m0), N = pp->g.N, H = pp->g.H, W = pp->g.W, pt = pp->g.pad_t, pl = pp->g.pad_l;
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
      for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int gI = 0; gI < 16; ++gI)
          if (gI < NG) dma16(xr, buf + q * PLP + gI * 1024, poff[gI], q * xpb + (unsigned)cc * chunk_bytes);
    };
    int it = 0; Item t = work_item<BN>(0, n_mine);               // the chunk being REQUESTED
    int cc = t.cc_b; bool live = true;
    tile_offsets(t);
    auto advance = [&]() {
      if (++cc == t.cc_e) {
        if (++it < n_items) { t = work_item<BN>(it, n_mine); cc = t.cc_b; tile_offsets(t); } else live = false;
      }
    };
    issue(cc, 0); advance();
    int total_chunks = 0;
    for (int i = 0; i < n_items; ++i) { const Item q = work_item<BN>(i, n_mine); total_chunks += q.cc_e - q.cc_b; }
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
  for (int item = 0; item < n_items; ++item) {
    const Item cur = work_item<BN>(item, n_mine);
    int rowidx[TM];                                      // patch row of this lane's output pixel(s), tap (0, 0)
    {
      const int P0 = padded_pos(cur.m0);
#pragma unroll
      for (int im = 0; im < TM; ++im) {
        const int m = cur.m0 + wm + im * 32 + (lane & 31);
        rowidx[im] = m < M ? padded_pos(m) - P0 : 0;
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
      if constexpr (F16 && PIPE) {
        // all R * S taps of the chunk unrolled; fragment set (t & 1) holds tap t.  A fragments (patch) of tap t + 1 are requested
        // during tap t whatever step t + 1 belongs to; B fragments (weights) of the first tap of a step right behind its barrier.
        f16x8 fa[2][TM][2], fb[2][TN][2];
        auto load_a = [&](int tap, f16x8 (&a)[TM][2]) {
          const int r = tap / S, s2 = tap % S;
#pragma unroll
          for (int im = 0; im < TM; ++im) {
            const int idx = rowidx[im] + r * PW + s2;
            const unsigned char* ap = pbuf + idx * 32 + ((h ^ ((idx >> 3) & 1)) << 4);
#pragma unroll
            for (int q = 0; q < 2; ++q) a[im][q] = *reinterpret_cast<const f16x8*>(ap + q * PLP);
          }
        };
        auto load_b = [&](const unsigned char* bs, int tp, f16x8 (&b)[TN][2]) {
#pragma unroll
          for (int in = 0; in < TN; ++in) {
            const int row = wn + in * 32 + (lane & 31);
            const unsigned char* bp = bs + (tp * 2 * BN + row) * 32 + ((h ^ ((row >> 3) & 1)) << 4);
#pragma unroll
            for (int q = 0; q < 2; ++q) b[in][q] = *reinterpret_cast<const f16x8*>(bp + q * BN * 32);
          }
        };
        constexpr int NT = R * S, NRD = 2 * (TM + TN), NMF = 3 * TM * TN;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int tp = t % TPS;
          if (tp == 0) {
            __syncthreads();           // barrier #gs: this step's weights (and, at t = 0, this chunk's patch) are in LDS
            if (t == 0) load_a(0, fa[0]);
            load_b(bslot0 + (gs % NBS) * SBY, 0, fb[t & 1]);
            __builtin_amdgcn_sched_barrier(0);
          }
          const unsigned char* bs = bslot0 + (gs % NBS) * SBY;
          const bool nxt_a = t + 1 < NT, nxt_b = tp + 1 < TPS;
          if (nxt_a) load_a(t + 1, fa[(t + 1) & 1]);
          if (nxt_b) load_b(bs, tp + 1, fb[(t + 1) & 1]);
          mfma_step_h<TM, TN>(fa[t & 1], fb[t & 1], acc);
          // one fragment read per matrix instruction while there are reads, then the remaining matrix instructions
          constexpr int NR_AB = NRD, NR_A = 2 * TM;
          const int nr = (nxt_a ? NR_A : 0) + (nxt_b ? NR_AB - NR_A : 0);
#pragma unroll
          for (int i = 0; i < NMF; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < nr) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (tp == TPS - 1) ++gs;
        }
        ++gc;
        continue;
      }
#pragma unroll 1
      for (int st = 0; st < SPC; ++st) {
        __syncthreads();             // barrier #gs: this step's weights (and, at st = 0, this chunk's patch) are in LDS
        const unsigned char* bs = bslot0 + (gs % NBS) * SBY;
#pragma unroll
        for (int tp = 0; tp < TPS; ++tp) {
          const int tap = st * TPS + tp, r = tap / S, s = tap % S;
          bf16x8 a[TM][NP], bb[TN][NP];
#pragma unroll
          for (int im = 0; im < TM; ++im) {
            const int idx = rowidx[im] + r * PW + s;
            const unsigned char* ap = pbuf + idx * 32 + ((h ^ ((idx >> 3) & 1)) << 4);
#pragma unroll
            for (int q = 0; q < NP; ++q) a[im][q] = *reinterpret_cast<const bf16x8*>(ap + q * PLP);
          }
#pragma unroll
          for (int in = 0; in < TN; ++in) {
            const int row = wn + in * 32 + (lane & 31);
            const unsigned char* bp = bs + (tp * NP * BN + row) * 32 + ((h ^ ((row >> 3) & 1)) << 4);
#pragma unroll
            for (int q = 0; q < NP; ++q) bb[in][q] = *reinterpret_cast<const bf16x8*>(bp + q * BN * 32);
          }
          step_planes<G, F16>(a, bb, acc);
          if (TPS > 1) __builtin_amdgcn_sched_barrier(0);     // one tap's fragments at a time (all taps' reads hoisted: spills)
        }
        ++gs;
      }
      ++gc;
    }
    patch_epilogue<G, BN, BNS, F16>(p, cur, acc, M, K, wave, lane, h, wm, wn);
  }
}

// ---- 1x1 convolutions on the planes ("planes GEMM"; VERDICT r03-r05 #2a) ---------------------------------------------------------
// y[m][k] = sum_c a[pix(m)][c] w[c][k] for the ResNets' 1x1 convs (reference backbones.py:99-104: the bottleneck's conv1 / conv3 and the
// projection shortcuts): the operands the patch kernel reads — activation planes [2][C/16][pixels][16], kernel planes
// [2][C/16][K][16] (embnet_conv_weight_planes with r = s = 1: flip 0 forward, flip 1 data gradient) — by LDS-DMA, no register
// staging and no split arithmetic in the loop.  A 1x1 conv has one ninth of the 3x3's matrix work per operand byte, so what shapes
// the kernel is the LOAD stream, not the matrix pipe:
//  * a STEP is G = 2 chunks (32 channels) of both operands: A 256 pixels x 32 B x 2 planes x G = 32 KB, B BN x 32 B x 2 x G; a ring
//    of NS = 3 stages (144 KB at BN = 128), requested two steps ahead, one s_barrier per step (24 MFMAs per wave at BN = 128);
//  * THREE loader waves: wave 8 the kernel planes (16 DMA instructions per step), waves 9 and 10 one chunk of the pixels each (16
//    each) — one wave issuing all 48 would need ~4 000 cycles per step against ~800 of matrix work;
//  * output pixel m reads input pixel (n, oh * stride, ow * stride): the per-lane DMA offset does the strided gather (shortcuts);
//  * the workgroup walks its tiles persistently, left-over tiles are cut along the channel groups (work_item<BN, 16 * G>), the
//    epilogue is the patch kernel's (patch_epilogue: bias / ReLU / residual / statistics / partial tiles).
template <int BN, int G, int NS, bool F16>
__global__ __launch_bounds__(704) void conv1x1_planes_kernel(const PatchParams p) {
  using GP = GeomP<BN>;
  static_assert(F16, "the 1x1 planes kernel is built for the two-piece fp16 format");
  constexpr int NP = 2, TM = GP::TM, TN = GP::TN, D = NS - 1;
  constexpr int A_BYTES = G * NP * 256 * 32, B_BYTES = G * NP * BN * 32, STAGE = A_BYTES + B_BYTES;
  constexpr int NBI_B = G * NP * (BN / 32), NBI_A = NP * 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  const int n_mine = b < p.n_full ? (p.n_full - b + p.grid - 1) / p.grid : 0;
  const int n_items = n_mine + (b < p.n_pieces ? 1 : 0);
  if (n_items == 0) return;
  const int dhalf = (lane & 1) ^ ((lane >> 4) & 1);      // logical 16-byte half this lane's DMA piece holds
  int total = 0;                                         // steps (channel groups) of this workgroup
  for (int i = 0; i < n_items; ++i) { const Item q = work_item<BN, 16 * G>(i, n_mine); total += q.cc_e - q.cc_b; }

  if (wave >= 8) {
    // ---- loaders: stage (gs % NS) <- operands of step gs, D steps ahead of the MFMA waves ----------------------------------
    const bool is_b = wave == 8;
    const int tp = wave - 9;                                      // an A loader's chunk of the group
    const __amdgpu_buffer_rsrc_t rs = is_b
        ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.wp), 0, 3u * p.w_plane_bytes, 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.xp), 0, 3u * p.x_plane_bytes, 0x00020000);
    const int K = p.g.K;
    const unsigned wpb = p.w_plane_bytes, xpb = p.x_plane_bytes, chunk_bytes = 32u * (unsigned)(p.g.N * p.g.H * p.g.W);
    int it = 0; Item t = work_item<BN, 16 * G>(0, n_mine);        // the step being REQUESTED: (it, cg)
    int cg = t.cc_b; bool live = true;
    unsigned off[8];                                              // per lane: B rows (BN / 32 groups) or A pixels (8 groups): constant per tile
    auto tile_offsets = [&]() {
      if (is_b) {
#pragma unroll
        for (int gb = 0; gb < BN / 32; ++gb) {
          const int row = t.n0 + gb * 32 + (lane >> 1);
          off[gb] = row < K ? 32u * (unsigned)row + 16u * dhalf : OOB;
        }
      } else {
        kargp pp = kargs();
        FastDiv dOHW, dOW;
        dOHW.mul = pp->g.dOHW.mul; dOHW.shift = pp->g.dOHW.shift; dOHW.d = pp->g.dOHW.d;
        dOW.mul = pp->g.dOW.mul; dOW.shift = pp->g.dOW.shift; dOW.d = pp->g.dOW.d;
        const int M = pp->g.N * pp->g.OH * pp->g.OW, st = pp->g.stride, H = pp->g.H, W = pp->g.W;
#pragma unroll
        for (int gI = 0; gI < 8; ++gI) {
          const int m = t.m0 + 32 * gI + (lane >> 1);
          uint32_t n, rem, oh, ow;
          dOHW.divmod((uint32_t)min(m, M - 1), n, rem); dOW.divmod(rem, oh, ow);
          off[gI] = m < M ? 32u * (unsigned)(((int)n * H + (int)oh * st) * W + (int)ow * st) + 16u * dhalf : OOB;
        }
      }
    };
    tile_offsets();
    auto issue = [&](int gs) {
      unsigned char* stage = smem + (gs % NS) * STAGE;
      if (is_b) {
#pragma unroll
        for (int c2 = 0; c2 < G; ++c2)
#pragma unroll
          for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int gb = 0; gb < BN / 32; ++gb)
              dma16(rs, stage + A_BYTES + ((c2 * NP + q) * BN + gb * 32) * 32, live ? off[gb] : OOB,
                    q * wpb + 32u * (unsigned)((cg * G + c2) * K));
      } else {
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
          for (int gI = 0; gI < 8; ++gI)
            dma16(rs, stage + ((tp * NP + q) * 256 + gI * 32) * 32, live ? off[gI] : OOB, q * xpb + (unsigned)(cg * G + tp) * chunk_bytes);
      }
      if (live && ++cg == t.cc_e) {
        if (++it < n_items) { t = work_item<BN, 16 * G>(it, n_mine); cg = t.cc_b; tile_offsets(); } else live = false;
      }
    };
    for (int gs = 0; gs < D; ++gs) issue(gs);
    for (int gs = 0; gs < total; ++gs) {
      // the D - 1 youngest requests may still be in flight: step gs has landed.  (Past the last step the loaders keep issuing
      // out-of-range requests — zeros into stages nobody reads — so the count stays a constant.)
      if (is_b) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D - 1) * NBI_B) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D - 1) * NBI_A) : "memory");
      __builtin_amdgcn_s_barrier();                              // #gs: step gs - 1 is done everywhere -> its stage is free
      issue(gs + D);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ---- MFMA waves ---------------------------------------------------------------------------------------------------
  const int wm = (wave / GP::WAVES_N) * GP::WTM, wn = (wave % GP::WAVES_N) * GP::WTN;
  const int K = p.g.K, M = p.g.N * p.g.OH * p.g.OW;
  int aoff[TM], boff[TN];                                // byte offsets of this lane's fragment rows inside a (chunk, plane) image
#pragma unroll
  for (int im = 0; im < TM; ++im) { const int row = wm + im * 32 + (lane & 31); aoff[im] = row * 32 + ((h ^ ((row >> 3) & 1)) << 4); }
#pragma unroll
  for (int in = 0; in < TN; ++in) { const int row = wn + in * 32 + (lane & 31); boff[in] = row * 32 + ((h ^ ((row >> 3) & 1)) << 4); }
  int gs = 0;
  f32x16 acc[TM][TN];
  for (int item = 0; item < n_items; ++item) {
    const Item cur = work_item<BN, 16 * G>(item, n_mine);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll 1
    for (int cg = cur.cc_b; cg < cur.cc_e; ++cg) {
      __syncthreads();                 // barrier #gs: this step's operands are in LDS
      const unsigned char* sa = smem + (gs % NS) * STAGE;
      const unsigned char* sb = sa + A_BYTES;
      f16x8 fa[2][TM][2], fb[2][TN][2];
      auto load = [&](int c2, f16x8 (&a)[TM][2], f16x8 (&bb)[TN][2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
          for (int im = 0; im < TM; ++im) a[im][q] = *reinterpret_cast<const f16x8*>(sa + (c2 * NP + q) * 256 * 32 + aoff[im]);
#pragma unroll
          for (int in = 0; in < TN; ++in) bb[in][q] = *reinterpret_cast<const f16x8*>(sb + (c2 * NP + q) * BN * 32 + boff[in]);
        }
      };
      load(0, fa[0], fb[0]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c2 = 0; c2 < G; ++c2) {
        if (c2 + 1 < G) load(c2 + 1, fa[(c2 + 1) & 1], fb[(c2 + 1) & 1]);
        mfma_step_h<TM, TN>(fa[c2 & 1], fb[c2 & 1], acc);
        constexpr int NMF = 3 * TM * TN, NRD = 2 * (TM + TN);
#pragma unroll
        for (int i = 0; i < NMF; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (c2 + 1 < G && i < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      ++gs;
    }
    patch_epilogue<GP, BN, false, F16>(p, cur, acc, M, K, wave, lane, h, wm, wn);
  }
}

// ---- 1x1 convolutions with the ACTIVATION operand read as fp32 ("fp32 by DMA"; VERDICT r05 weak #9 / missing #2) -------------------
// The same product as conv1x1_planes_kernel — kernel planes [2][C/16][K][16] by LDS-DMA, three fp16 products per fp32 product — but the
// activations are the fp32 NHWC tensor itself: no planes of x have to exist (in the step a planes copy BESIDE the fp32 tensor costs
// more than the faster product saves).  What it replaces is the gather loop's fp32 load -> register split -> LDS store
// with ONE 32-deep tile in flight per workgroup (conv.hip; 13-30 % of the matrix pipe).
//  * a STEP is 32 channels: A 256 pixels x 128 B (one cache line per pixel) = 32 KB fetched in 16-byte DMA pieces, eight lanes per
//    pixel row; a row's eight quads land at  row * 128 + ((quad ^ f(row)) << 4),  f(row) = (row >> 1) & 7 — the lane picks the quad it
//    FETCHES so that the DMA's fixed lane -> LDS order produces the swizzle — which makes the two 16-byte reads of a fragment
//    conflict-free over any sixteen consecutive rows; B as in the planes kernel (16 KB at BN = 128).  Three stages, two steps ahead.
//  * the planes kernel's three loader waves (wave 8: kernel planes; waves 9, 10: 128 pixel rows each, 16 pieces per step); the eight
//    matrix waves split the eight floats of a fragment into the two fp16 pieces of x s themselves — 2 to 3 VALU operations per element
//    (v_pk_mul_f32, v_cvt_pk_f16_f32, v_fma_mix / cvt + fma) — reading and splitting k-step c2 + 1 behind the matrix instructions of c2.
//  * tiles, K-split left-overs and the epilogue (bias / ReLU / residual / statistics) are the planes kernel's (work_item<BN, 32>,
//    patch_epilogue).  (Its data-gradient form with the BatchNorm-backward sums was instantiated and dropped: 85 spilled registers
//    in the epilogue, 394 us per launch in the C3 step against 222 for the gather kernel.)
// Tried and not kept (tools/exp/conv1x1_a32_8waves.diff, profiles/r06_exp_conv1x1_dma.txt): the same kernel WITHOUT loader waves —
// eight waves that issue their share of the DMA pieces themselves, 256 registers each, the loop rotated so that no read or split is
// exposed behind a barrier.  (a) hipcc puts s_waitcnt vmcnt(0) in front of every ds_read that follows a buffer_load ... lds builtin in
// the same wave — one exposed DMA round trip per step, 2.5 us; (b) with the DMA as inline assembly that wait is gone and the kernel
// is still 1.35x slower than this one (2048 -> 512 at 7x7: 151 vs 112 us): a wave that stalls at VMEM issue stalls its matrix
// instructions too.  With loader waves the workgroup is three waves per SIMD = 168 registers, which the rotated loop does not fit
// (95 - 120 spilled registers inside the loop: 5x slower) — hence the plain two-k-step loop.
template <int BN>
__global__ __launch_bounds__(704) void conv1x1_a32_kernel(const PatchParams p) {
  using GP = GeomP<BN>;
  constexpr int G = 2, NS = 3, NP = 2, TM = GP::TM, TN = GP::TN, D = NS - 1;
  constexpr int A_BYTES = 256 * 128, B_BYTES = G * NP * BN * 32, STAGE = A_BYTES + B_BYTES;
  constexpr int NBI_B = G * NP * (BN / 32), NBI_A = 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  const int n_mine = b < p.n_full ? (p.n_full - b + p.grid - 1) / p.grid : 0;
  const int n_items = n_mine + (b < p.n_pieces ? 1 : 0);
  if (n_items == 0) return;
  int total = 0;                                         // steps (32-channel groups) of this workgroup
  for (int i = 0; i < n_items; ++i) { const Item q = work_item<BN, 16 * G>(i, n_mine); total += q.cc_e - q.cc_b; }

  if (wave >= 8) {
    // ---- loaders: stage (gs % NS) <- operands of step gs, D steps ahead of the matrix waves ----------------------------------
    const bool is_b = wave == 8;
    const int tp = wave - 9;                                      // an A loader's half of the tile's rows
    const __amdgpu_buffer_rsrc_t rs = is_b
        ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.wp), 0, 3u * p.w_plane_bytes, 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.xp), 0, p.x32_bytes, 0x00020000);
    const int K = p.g.K;
    const unsigned wpb = p.w_plane_bytes;
    const int dhalf = (lane & 1) ^ ((lane >> 4) & 1);             // kernel planes: the logical 16-byte half this lane's piece holds
    int it = 0; Item t = work_item<BN, 16 * G>(0, n_mine);        // the step being REQUESTED: (it, cg)
    int cg = t.cc_b; bool live = true;
    unsigned off[16];                                             // per lane: B rows (BN / 32 groups) or A pixels (16 pieces): constant per tile
    auto tile_offsets = [&]() {
      if (is_b) {
#pragma unroll
        for (int gb = 0; gb < BN / 32; ++gb) {
          const int row = t.n0 + gb * 32 + (lane >> 1);
          off[gb] = row < K ? 32u * (unsigned)row + 16u * dhalf : OOB;
        }
      } else {
        kargp pp = kargs();
        FastDiv dOHW, dOW;
        dOHW.mul = pp->g.dOHW.mul; dOHW.shift = pp->g.dOHW.shift; dOHW.d = pp->g.dOHW.d;
        dOW.mul = pp->g.dOW.mul; dOW.shift = pp->g.dOW.shift; dOW.d = pp->g.dOW.d;
        const int M = pp->g.N * pp->g.OH * pp->g.OW, st = pp->g.stride, H = pp->g.H, W = pp->g.W;
        const unsigned rowbytes = 4u * (unsigned)pp->g.C;
#pragma unroll
        for (int gI = 0; gI < 16; ++gI) {
          // piece gI: rows 128 tp + 8 gI .. + 7; lane: row lane >> 3, LDS quad slot lane & 7, fetched quad = slot ^ f(row),
          // f(row) = (row >> 1) & 7 = 4 (gI & 1) + (lane >> 4)   (tile rows start at multiples of 256)
          const int m = t.m0 + 128 * tp + 8 * gI + (lane >> 3);
          uint32_t n, rem, oh, ow;
          dOHW.divmod((uint32_t)min(m, M - 1), n, rem); dOW.divmod(rem, oh, ow);
          const unsigned quad = (unsigned)((lane & 7) ^ (4 * (gI & 1) + (lane >> 4)));
          off[gI] = m < M ? rowbytes * (unsigned)(((int)n * H + (int)oh * st) * W + (int)ow * st) + 16u * quad : OOB;
        }
      }
    };
    tile_offsets();
    auto issue = [&](int gs) {
      unsigned char* stage = smem + (gs % NS) * STAGE;
      if (is_b) {
#pragma unroll
        for (int c2 = 0; c2 < G; ++c2)
#pragma unroll
          for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int gb = 0; gb < BN / 32; ++gb)
              dma16(rs, stage + A_BYTES + ((c2 * NP + q) * BN + gb * 32) * 32, live ? off[gb] : OOB,
                    q * wpb + 32u * (unsigned)((cg * G + c2) * K));
      } else {
#pragma unroll
        for (int gI = 0; gI < 16; ++gI)
          dma16(rs, stage + (128 * tp + 8 * gI) * 128, live ? off[gI] : OOB, 128u * (unsigned)cg);
      }
      if (live && ++cg == t.cc_e) {
        if (++it < n_items) { t = work_item<BN, 16 * G>(it, n_mine); cg = t.cc_b; tile_offsets(); } else live = false;
      }
    };
    for (int gs = 0; gs < D; ++gs) issue(gs);
    for (int gs = 0; gs < total; ++gs) {
      if (is_b) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D - 1) * NBI_B) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D - 1) * NBI_A) : "memory");
      __builtin_amdgcn_s_barrier();                              // #gs: step gs - 1 is done everywhere -> its stage is free
      issue(gs + D);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ---- matrix waves -----------------------------------------------------------------------------------------------------------
  const int wm = (wave / GP::WAVES_N) * GP::WTM, wn = (wave % GP::WAVES_N) * GP::WTN;
  const int K = p.g.K, M = p.g.N * p.g.OH * p.g.OW;
  int arow[TM], afr[TM], boff[TN];
#pragma unroll
  for (int im = 0; im < TM; ++im) { const int row = wm + im * 32 + (lane & 31); arow[im] = row * 128; afr[im] = (row >> 1) & 7; }
#pragma unroll
  for (int in = 0; in < TN; ++in) { const int row = wn + in * 32 + (lane & 31); boff[in] = row * 32 + ((h ^ ((row >> 3) & 1)) << 4); }
  const float a_scale = scale_pair(scale_exponent_of(__uint_as_float(*p.a_range))).x;
  int gs = 0;
  f32x16 acc[TM][TN];
  for (int item = 0; item < n_items; ++item) {
    const Item cur = work_item<BN, 16 * G>(item, n_mine);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll 1
    for (int cg = cur.cc_b; cg < cur.cc_e; ++cg) {
      __syncthreads();                 // barrier #gs: this step's operands are in LDS
      const unsigned char* sa = smem + (gs % NS) * STAGE;
      const unsigned char* sb = sa + A_BYTES;
      float4 raw[TM][2];
      f16x8 fa[2][TM][2], fb[2][TN][2];
      auto load_raw = [&](int c2) {    // this lane's eight channels of k-step c2: quads 4 c2 + 2 h and + 1 of its row
#pragma unroll
        for (int im = 0; im < TM; ++im)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            raw[im][j] = *reinterpret_cast<const float4*>(sa + arow[im] + (((4 * c2 + 2 * h + j) ^ afr[im]) << 4));
      };
      auto load_b = [&](int c2, f16x8 (&bb)[TN][2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int in = 0; in < TN; ++in) bb[in][q] = *reinterpret_cast<const f16x8*>(sb + (c2 * NP + q) * BN * 32 + boff[in]);
      };
      auto split = [&](f16x8 (&a)[TM][2]) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int im = 0; im < TM; ++im) {
          const Split4H s0 = split4h(raw[im][0], a_scale), s1 = split4h(raw[im][1], a_scale);
          a[im][0] = __builtin_bit_cast(f16x8, u32x4{s0.p[0].x, s0.p[0].y, s1.p[0].x, s1.p[0].y});
          a[im][1] = __builtin_bit_cast(f16x8, u32x4{s0.p[1].x, s0.p[1].y, s1.p[1].x, s1.p[1].y});
        }
      };
      load_raw(0); load_b(0, fb[0]);
      split(fa[0]);
#pragma unroll
      for (int c2 = 0; c2 < G; ++c2) {
        if (c2 + 1 < G) { load_raw(c2 + 1); load_b(c2 + 1, fb[(c2 + 1) & 1]); }
        mfma_step_h<TM, TN>(fa[c2 & 1], fb[c2 & 1], acc);
        if (c2 + 1 < G) split(fa[(c2 + 1) & 1]);
      }
      ++gs;
    }
    patch_epilogue<GP, BN, false, true>(p, cur, acc, M, K, wave, lane, h, wm, wn);
  }
}

// kernel [R,S,C,K] fp32 -> step-major planes [3][R][red/16][S][rows][16] bf16, one thread per four consecutive elements.
// flip = 0: rows = K, reduction channels = C:  out[r][cc][s][k][j] = w[r, s, 16 cc + j, k]
// flip = 1: rows = C, reduction channels = K:  out[r][cc][s][c][j] = w[R-1-r, S-1-s, c, 16 cc + j]   (stride-1 data gradient:
//           the correlation of dy with the flipped kernel, channel roles swapped)
struct WPlanesTensor { const float* w; unsigned short* out; int R, S, C, K, flip, pad; };
static_assert(sizeof(WPlanesTensor) == 40, "descriptor layout is part of the ABI (include/embnet.h)");
// f16 = 1: two fp16 pieces of w s, s from the slot (gemm_engine.h) — weight_absmax_kernel / weight_scale_kernel below put it there
__global__ __launch_bounds__(256) void weight_planes_kernel(const WPlanesTensor* __restrict__ table, const int* __restrict__ chunks, int f16) {
  const WPlanesTensor t = table[chunks[2 * blockIdx.x]];
  const int rows = t.flip ? t.C : t.K, red = t.flip ? t.K : t.C, ncc = red / 16;
  const long plane = (long)t.R * t.S * t.C * t.K, total4 = plane / 4;
  const long i = (long)chunks[2 * blockIdx.x + 1] * 1024 + threadIdx.x;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long i4 = i + u * 256;
    if (i4 >= total4) return;
    long e = 4 * i4;                                    // output element index [r][cc][s][row][j]
    const int j = (int)(e % 16); e /= 16;
    const int row = (int)(e % rows); e /= rows;
    const int s = (int)(e % t.S); e /= t.S;
    const int cc = (int)(e % ncc); const int r = (int)(e / ncc);
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = cc * 16 + j + q;
      v[q] = t.flip ? t.w[((long)((t.R - 1 - r) * t.S + (t.S - 1 - s)) * t.C + row) * t.K + ch]
                    : t.w[((long)(r * t.S + s) * t.C + ch) * t.K + row];
    }
    if (f16) {
      const Split4H sp = split4h(make_float4(v[0], v[1], v[2], v[3]), planes_scale_slot(t.out, plane)[0]);
#pragma unroll
      for (int q = 0; q < 2; ++q) *reinterpret_cast<uint2*>(t.out + q * plane + 4 * i4) = sp.p[q];
    } else {
      const Split4 sp = split4(make_float4(v[0], v[1], v[2], v[3]));
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<uint2*>(t.out + q * plane + 4 * i4) = sp.p[q];
    }
  }
}

// the largest |w| of each kernel: a workgroup per chunk of 4096 elements (the chunk list of the planes pass), one atomic per workgroup
// — an unsigned maximum of bit patterns: independent of the order, so reproducible
__global__ __launch_bounds__(256) void weight_absmax_kernel(const WPlanesTensor* __restrict__ table, const int* __restrict__ chunks) {
  const WPlanesTensor t = table[chunks[2 * blockIdx.x]];
  const long plane = (long)t.R * t.S * t.C * t.K, total4 = plane / 4;
  const long i = (long)chunks[2 * blockIdx.x + 1] * 1024 + threadIdx.x;
  float m = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long i4 = i + u * 256;
    if (i4 < total4) {
      const float4 v = reinterpret_cast<const float4*>(t.w)[i4];
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
  }
  m = wave_max(m);
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicMax(reinterpret_cast<unsigned*>(planes_scale_slot(t.out, plane)) + 2, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
}
// phase 0: zero the max word of every kernel's slot; phase 1: max word -> (s, 1 / s)
__global__ __launch_bounds__(256) void weight_scale_kernel(const WPlanesTensor* __restrict__ table, int n, int phase) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const WPlanesTensor t = table[i];
  float* sl = planes_scale_slot(t.out, (long)t.R * t.S * t.C * t.K);
  if (phase == 0) { reinterpret_cast<unsigned*>(sl)[2] = 0u; return; }
  const float2 sp = scale_pair(scale_exponent_of(sl[2]));
  sl[0] = sp.x; sl[1] = sp.y;
}

// fp32 NHWC [pixels][C] -> chunk-major planes [3][C/16][pixels][16] bf16; one thread per (pixel, 4 channels)
__global__ __launch_bounds__(256) void planes_from_f32_kernel(const float* __restrict__ x, long pixels, int C,
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

// the same in the two-piece fp16 format: MODE 2 leaves the workgroups' max |x| behind the scale slot, planes_scale_of_kernel turns
// them into (s, 1 / s), MODE 1 writes the pieces of x s
template <int MODE>
__global__ __launch_bounds__(256) void planes16_from_f32_kernel(const float* __restrict__ x, long pixels, int C,
                                                                unsigned short* __restrict__ planes) {
  const long total4 = pixels * C / 4, plane = pixels * C;
  const int c4 = C / 4;
  float amax = 0.f;
  const float sc = MODE == 1 ? planes_scale_slot(planes, plane)[0] : 1.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    if (MODE == 2) { amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)))); continue; }
    const long pix = i / c4; const int c = (int)(i % c4) * 4;
    const Split4H s = split4h(v, sc);
    const long o = ((long)(c >> 4) * pixels + pix) * 16 + (c & 15);
#pragma unroll
    for (int q = 0; q < 2; ++q) *reinterpret_cast<uint2*>(planes + q * plane + o) = s.p[q];
  }
  if (MODE == 2) {
    amax = wave_max(amax);
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) planes_scale_slot(planes, plane)[2 + blockIdx.x] = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
  }
}
__global__ __launch_bounds__(256) void planes_scale_of_kernel(float* __restrict__ slot, int blocks) {   // nn_kernels.hip's planes_scale_kernel
  float m = 0.f;
  for (int i = threadIdx.x; i < blocks; i += 256) m = fmaxf(m, slot[2 + i]);
  m = wave_max(m);
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    const float2 sp = scale_pair(scale_exponent_of(m));
    slot[0] = sp.x; slot[1] = sp.y;
  }
}

}  // namespace patch
}  // namespace embnet

using namespace embnet;
using namespace embnet::patch;

// ---- host side ----------------------------------------------------------------------------------------------------------
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

struct Plan { int bn, tps, nbs, LR, tiles, n_full, parts, cc_part, n_pieces, grid; size_t lds, ws_bytes; };

// the 1x1 planes GEMM (conv1x1_planes_kernel): G = 2 chunks per step, a ring of three stages; reduction units are 32-channel groups
constexpr int ONE_G = 2, ONE_NS = 3;
static bool make_plan1(int n, int c, int k, int stride, int oh, int ow, Plan& pl) {
  pl = Plan{};
  if (!planes_f16() || (stride != 1 && stride != 2) || (c % (16 * ONE_G)) || (k & 3) || n <= 0 || oh <= 0 || ow <= 0) return false;
  const size_t in_pix = (size_t)n * ((size_t)(oh - 1) * stride + 1) * ((size_t)(ow - 1) * stride + 1);
  if (in_pix * 32 >= 0x7FFFFFF0ull || (size_t)n * oh * ow * k * 4 >= 0x7FFFFFF0ull) return false;
  pl.bn = k >= 128 ? 128 : 64;
  pl.tps = ONE_G; pl.nbs = ONE_NS; pl.LR = 256;
  pl.lds = (size_t)ONE_NS * ONE_G * 2 * 32 * (256 + pl.bn);
  const long M = (long)n * oh * ow;
  pl.tiles = cdiv(M, 256) * cdiv(k, pl.bn);
  pl.grid = 256;
  const int ncc = c / (16 * ONE_G);
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

static bool make_plan(int n, int c, int r, int s, int k, int stride, int oh, int ow, Plan& pl) {
  if (r == 1 && s == 1) return make_plan1(n, c, k, stride, oh, ow, pl);
  static const int enabled = (int)env_long("EMBNET_CONV_PATCH", 1);
  if (!enabled || stride != 1 || r != 3 || s != 3 || (c & 15) || (k & 3) || n <= 0 || oh <= 0 || ow <= 0) return false;
  if ((size_t)n * (oh + 2) * (ow + 2) * 32 >= 0x7FFFFFF0ull || (size_t)n * oh * ow * c * 2 >= 0x7FFFFFF0ull / 3) return false;
  // the geometry's plan is a pure function of its arguments: remember the last few (patch_rows walks every tile)
  struct Key { int n, c, k, oh, ow; };
  static thread_local Key keys[8]; static thread_local Plan plans[8]; static thread_local int used = 0, next = 0;
  for (int i = 0; i < used; ++i)
    if (keys[i].n == n && keys[i].c == c && keys[i].k == k && keys[i].oh == oh && keys[i].ow == ow) { pl = plans[i]; return pl.bn != 0; }
  pl = Plan{};
  const size_t np = planes_f16() ? 2 : 3;                   // planes held in LDS
  pl.bn = k >= 128 ? 128 : 64;
  pl.tps = pl.bn == 64 ? 3 : 1;
  pl.nbs = pl.bn == 64 ? 3 : 6;
  pl.LR = patch_rows(n, oh, ow, r, s);
  // three taps per barrier at BN = 128 too (two 36 KB slots): level with one tap per barrier while a step held six products per
  // fragment pair (10.39 vs 10.39 ms), + 1 % with three (C2 8.14 -> 8.06 ms, three alternating pairs): the default in that format
  static const int tps3 = (int)env_long("EMBNET_PATCH_TPS3", planes_f16() ? 1 : 0);
  if (pl.bn == 128 && tps3 && 2 * np * pl.LR * 32 + (size_t)2 * 3 * np * 128 * 32 <= 160 * 1024) {
    pl.tps = 3; pl.nbs = 2;
    if (np == 2 && 2 * np * pl.LR * 32 + (size_t)3 * 3 * np * 128 * 32 <= 160 * 1024) pl.nbs = 3;      // two planes: room for a third slot
  }
  pl.lds = 2 * np * pl.LR * 32 + (size_t)pl.nbs * pl.tps * np * pl.bn * 32;
  if (pl.bn == 128 && pl.tps == 1 && pl.lds > 160 * 1024) {             // a long patch (small maps: many image seams per tile): shorter weight ring
    pl.nbs = 4;
    pl.lds = 2 * np * pl.LR * 32 + (size_t)pl.nbs * pl.tps * np * pl.bn * 32;
  }
  bool ok = pl.LR <= 512 && pl.lds <= 160 * 1024;
  if (ok) {
    const long M = (long)n * oh * ow;
    pl.tiles = cdiv(M, 256) * cdiv(k, pl.bn);
    pl.grid = 256;
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
  } else {
    pl.bn = 0;
  }
  keys[next] = Key{n, c, k, oh, ow}; plans[next] = pl; next = (next + 1) % 8; if (used < 8) ++used;
  return ok;
}

extern "C" int embnet_conv2d_patch_supported(int n, int c, int r, int s, int k, int stride, int oh, int ow) {
  Plan pl; return make_plan(n, c, r, s, k, stride, oh, ow, pl) ? 1 : 0;
}
extern "C" size_t embnet_conv2d_patch_workspace_bytes(int n, int c, int r, int s, int k, int oh, int ow) {
  Plan pl; return make_plan(n, c, r, s, k, 1, oh, ow, pl) ? pl.ws_bytes : 0;
}
extern "C" int embnet_conv2d_patch_stats_rows(int n, int oh, int ow) { return cdiv((long)n * oh * ow, 256) * 4; }

extern "C" int embnet_planes_from_f32(const float* x, long pixels, int c, void* planes, void* stream) {
  EMBNET_CHECK_ARG(x && planes && pixels > 0 && c > 0 && (c & 15) == 0, "planes_from_f32: need c %% 16 == 0");
  EMBNET_CHECK_ARG((size_t)pixels * c * 2 < 0x7FFFFFF0ull / 3, "planes_from_f32: tensor too large");
  const long n4 = pixels * c / 4;
  EMBNET_TRACE("embnet::patch::planes_from_f32_kernel", TRACE_BYTES, 10.0 * pixels * c, stream);
  const int blocks = (int)(n4 / 256 + 1 > 4096 ? 4096 : n4 / 256 + 1);
  if (planes_f16()) {                                       // the tensor's range first (its largest |x| -> [2^14, 2^15)), then the pieces
    hipStream_t st = (hipStream_t)stream;
    float* slot = planes_scale_slot(planes, pixels * c);
    // (the third plane's space — 2 * n4 floats, >= 8 — holds the slot and the dry run's workgroup maxima: fewer workgroups for a tiny tensor)
    const int dry = (int)(2 * n4 - 2 < blocks ? 2 * n4 - 2 : blocks);
    planes16_from_f32_kernel<2><<<dry, 256, 0, st>>>(x, pixels, c, (unsigned short*)planes);
    planes_scale_of_kernel<<<1, 256, 0, st>>>(slot, dry);
    planes16_from_f32_kernel<1><<<blocks, 256, 0, st>>>(x, pixels, c, (unsigned short*)planes);
    return check_launch("planes_from_f32");
  }
  planes_from_f32_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(x, pixels, c, (unsigned short*)planes);
  return check_launch("planes_from_f32");
}

extern "C" int embnet_conv_weight_planes_chunk_elems(void) { return 4096; }
extern "C" int embnet_conv_weight_planes(const void* table, int n_tensors, const int32_t* chunks, int n_chunks, void* stream) {
  EMBNET_CHECK_ARG(table && chunks && n_tensors > 0 && n_chunks > 0, "conv_weight_planes: bad argument");
  EMBNET_TRACE("embnet::patch::weight_planes_kernel", TRACE_BYTES, 10.0 * 4096 * n_chunks, stream);
  if (planes_f16()) {                                       // each kernel's own range first (both layouts of a kernel find the same)
    hipStream_t st = (hipStream_t)stream;
    weight_scale_kernel<<<cdiv(n_tensors, 256), 256, 0, st>>>((const WPlanesTensor*)table, n_tensors, 0);
    weight_absmax_kernel<<<n_chunks, 256, 0, st>>>((const WPlanesTensor*)table, chunks);
    weight_scale_kernel<<<cdiv(n_tensors, 256), 256, 0, st>>>((const WPlanesTensor*)table, n_tensors, 1);
    weight_planes_kernel<<<n_chunks, 256, 0, st>>>((const WPlanesTensor*)table, chunks, 1);
    return check_launch("conv_weight_planes");
  }
  weight_planes_kernel<<<n_chunks, 256, 0, (hipStream_t)stream>>>((const WPlanesTensor*)table, chunks, 0);
  return check_launch("conv_weight_planes");
}

// (off: measured on C2, three alternating pairs — 85.5 us either way, FETCH + WRITE 117.0 -> 109.2 MB per launch: the layers with several
// column tiles per row have fewer tiles than the grid, i.e. no whole round; profiles/r06_exp_xcd_rows.txt)
static int patch_xcd_rows() { static const int v = env_long("EMBNET_PATCH_XCD_ROWS", 0) != 0; return v; }
static bool patch_pipe() { static const bool v = env_long("EMBNET_PATCH_PIPE", 1) != 0; return v; }   // 0: the plain loop (A/B)
template <int BN, int TPS, int NBS>
static void launch_patch(const PatchParams& p, size_t lds, hipStream_t st) {
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void*)conv_patch_kernel<BN, 3, 3, TPS, NBS, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_patch_kernel<BN, 3, 3, TPS, NBS, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_patch_kernel<BN, 3, 3, TPS, NBS, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_patch_kernel<BN, 3, 3, TPS, NBS, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_patch_kernel<BN, 3, 3, TPS, NBS, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    once = true;
  }
  if (planes_f16()) {
    if (p.bn.x) conv_patch_kernel<BN, 3, 3, TPS, NBS, true, true><<<p.grid, 640, lds, st>>>(p);
    else if (patch_pipe()) conv_patch_kernel<BN, 3, 3, TPS, NBS, false, true, true><<<p.grid, 640, lds, st>>>(p);
    else conv_patch_kernel<BN, 3, 3, TPS, NBS, false, true><<<p.grid, 640, lds, st>>>(p);
    return;
  }
  if (p.bn.x) conv_patch_kernel<BN, 3, 3, TPS, NBS, true, false><<<p.grid, 640, lds, st>>>(p);
  else conv_patch_kernel<BN, 3, 3, TPS, NBS, false, false><<<p.grid, 640, lds, st>>>(p);
}

static int conv2d_patch_impl(const void* xp, const void* wp, const float* bias, float* y, int n, int h, int wd, int c,
                             int r, int s, int k, int pad_t, int pad_l, int oh, int ow, int relu,
                             const float* residual, float* stats, const BnSums& bsum, void* workspace, size_t workspace_bytes,
                             void* stream) {
  EMBNET_CHECK_ARG(xp && wp && y, "conv2d_patch: null pointer");
  EMBNET_CHECK_ARG(r == 3 && s == 3, "conv2d_patch: 3x3 kernels (1x1: embnet_conv2d_planes1x1_f32)");
  Plan pl;
  EMBNET_CHECK_ARG(make_plan(n, c, r, s, k, 1, oh, ow, pl), "conv2d_patch: unsupported geometry (see embnet_conv2d_patch_supported)");
  PatchParams p{(const unsigned short*)xp, (const unsigned short*)wp, y, bias, residual, stats, 0, relu};
  if (int rc = make_geom(p.g, n, h, wd, c, r, s, k, 1, pad_t, pad_l, oh, ow, "conv2d_patch")) return rc;
  const long M = (long)n * oh * ow;
  p.x_plane_bytes = (unsigned)((size_t)n * h * wd * c * 2);
  p.w_plane_bytes = (unsigned)((size_t)r * s * c * k * 2);
  p.PH = oh + r - 1; p.PW = ow + s - 1;
  p.dPHW = FastDiv::make(p.PH * p.PW); p.dPW = FastDiv::make(p.PW);
  p.LR = pl.LR;
  p.stats_rows = cdiv(M, 256) * 4;
  p.bn = bsum;
  p.grid = pl.grid;
  if (pl.n_pieces > 0 && (pl.ws_bytes > workspace_bytes || !workspace)) { pl.n_full = pl.tiles; pl.n_pieces = 0; pl.parts = 1; }
  p.n_full = pl.n_full; p.parts = pl.parts; p.cc_part = pl.cc_part; p.n_pieces = pl.n_pieces; p.ws = (float*)workspace;
  p.xcd_rows = patch_xcd_rows();
  hipStream_t st = (hipStream_t)stream;
  {
    // (the names rocprofv3 prints: bench.py looks the kernel's measured HBM traffic up by them)
    static thread_local char kname[160];
    snprintf(kname, sizeof kname, "void embnet::patch::conv_patch_kernel<%d, 3, 3, %d, %d, %s%s>(embnet::patch::PatchParams)", pl.bn, pl.tps, pl.nbs,
             p.bn.x ? "true" : "false", planes_f16() ? ((patch_pipe() && !p.bn.x) ? ", true, true" : ", true") : "");
    EMBNET_TRACE_FLOP(kname,
                      2.0 * M * k * r * s * c,
                      (planes_f16() ? 4.0 : 6.0) * ((double)n * h * wd * c + (double)r * s * c * k) + 4.0 * (double)M * k * (residual ? 2 : 1), st);
    if (pl.bn == 128 && pl.tps == 3) { if (pl.nbs == 3) launch_patch<128, 3, 3>(p, pl.lds, st); else launch_patch<128, 3, 2>(p, pl.lds, st); }
    else if (pl.bn == 128) { if (pl.nbs == 6) launch_patch<128, 1, 6>(p, pl.lds, st); else launch_patch<128, 1, 4>(p, pl.lds, st); }
    else launch_patch<64, 3, 3>(p, pl.lds, st);
  }
  if (p.n_pieces > 0)
    launch_tail_fixup(p.ws, p.parts, 256, pl.bn, 64, p.n_full, pl.tiles - p.n_full, cdiv(k, pl.bn), M, k, bias, relu, residual, y,
                      stats, p.stats_rows, bsum, st);
  return check_launch("conv2d_patch");
}

extern "C" int embnet_conv2d_patch_f32(const void* xp, const void* wp, const float* bias, float* y, int n, int h, int wd, int c,
                                       int r, int s, int k, int pad_t, int pad_l, int oh, int ow, int relu,
                                       const float* residual, float* stats, void* workspace, size_t workspace_bytes,
                                       void* stream) {
  return conv2d_patch_impl(xp, wp, bias, y, n, h, wd, c, r, s, k, pad_t, pad_l, oh, ow, relu, residual, stats,
                           BnSums{nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0}, workspace, workspace_bytes, stream);
}

// 1x1 convolution on the planes (conv1x1_planes_kernel): y[n,oh,ow,k] = sum_c x[n, oh*stride, ow*stride, c] w[c,k]; x planes
// [.][c/16][n*h*wd][16], kernel planes of a [1,1,c,k] kernel (flip 0) — or, with dy planes and flip-1 planes, c and k swapped, the
// stride-1 data gradient.  Epilogue options as embnet_conv2d_patch_f32.
template <int BN>
static void launch_1x1(const PatchParams& p, size_t lds, hipStream_t st) {
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void*)conv1x1_planes_kernel<BN, ONE_G, ONE_NS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    once = true;
  }
  conv1x1_planes_kernel<BN, ONE_G, ONE_NS, true><<<p.grid, 704, lds, st>>>(p);
}
extern "C" int embnet_conv2d_planes1x1_f32(const void* xp, const void* wp, const float* bias, float* y, int n, int h, int wd, int c,
                                           int k, int stride, int oh, int ow, int relu, const float* residual, float* stats,
                                           void* workspace, size_t workspace_bytes, void* stream) {
  EMBNET_CHECK_ARG(xp && wp && y, "conv2d_planes1x1: null pointer");
  Plan pl;
  EMBNET_CHECK_ARG(make_plan1(n, c, k, stride, oh, ow, pl), "conv2d_planes1x1: unsupported geometry (embnet_conv2d_patch_supported with r = s = 1)");
  PatchParams p{(const unsigned short*)xp, (const unsigned short*)wp, y, bias, residual, stats, 0, relu};
  if (int rc = make_geom(p.g, n, h, wd, c, 1, 1, k, stride, 0, 0, oh, ow, "conv2d_planes1x1")) return rc;
  const long M = (long)n * oh * ow;
  p.x_plane_bytes = (unsigned)((size_t)n * h * wd * c * 2);
  p.w_plane_bytes = (unsigned)((size_t)c * k * 2);
  p.PH = oh; p.PW = ow; p.dPHW = FastDiv::make(oh * ow); p.dPW = FastDiv::make(ow);
  p.LR = 256;
  p.stats_rows = cdiv(M, 256) * 4;
  p.bn = BnSums{nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0};
  p.grid = pl.grid;
  if (pl.n_pieces > 0 && (pl.ws_bytes > workspace_bytes || !workspace)) { pl.n_full = pl.tiles; pl.n_pieces = 0; pl.parts = 1; }
  p.n_full = pl.n_full; p.parts = pl.parts; p.cc_part = pl.cc_part; p.n_pieces = pl.n_pieces; p.ws = (float*)workspace;
  p.xcd_rows = patch_xcd_rows();
  hipStream_t st = (hipStream_t)stream;
  {
    static thread_local char kname[160];
    snprintf(kname, sizeof kname, "void embnet::patch::conv1x1_planes_kernel<%d, %d, %d, true>(embnet::patch::PatchParams)", pl.bn, ONE_G, ONE_NS);
    EMBNET_TRACE_FLOP(kname, 2.0 * M * k * c, 4.0 * ((double)M * c + (double)c * k) + 4.0 * (double)M * k * (residual ? 2 : 1), st);
    if (pl.bn == 128) launch_1x1<128>(p, pl.lds, st); else launch_1x1<64>(p, pl.lds, st);
  }
  if (p.n_pieces > 0)
    launch_tail_fixup(p.ws, p.parts, 256, pl.bn, 64, p.n_full, pl.tiles - p.n_full, cdiv(k, pl.bn), M, k, bias, relu, residual, y,
                      stats, p.stats_rows, p.bn, st);
  return check_launch("conv2d_planes1x1");
}

// 1x1 convolution with the ACTIVATION operand read as fp32 by LDS-DMA (conv1x1_a32_kernel): x fp32 [n,h,wd,c] with its RANGE SLOT
// x_range (include/embnet.h), kernel planes as embnet_conv2d_planes1x1_f32.  With dy, its range, the flip-1 planes and c, k swapped
// (stride 1): the data gradient (residual: a gradient to add).
static bool dma1x1_ok(int n, int h, int wd, int c, int k, int stride, int oh, int ow, Plan& pl) {
  if (!make_plan1(n, c, k, stride, oh, ow, pl)) return false;
  return (size_t)n * h * wd * c * 4 < 0xFFFFFFF0ull;            // (32-bit buffer offsets over the fp32 tensor)
}
template <int BN>
static void launch_dma1x1(const PatchParams& p, size_t lds, hipStream_t st) {
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void*)conv1x1_a32_kernel<BN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    once = true;
  }
  conv1x1_a32_kernel<BN><<<p.grid, 704, lds, st>>>(p);
}
extern "C" int embnet_conv2d_dma1x1_supported(int n, int h, int wd, int c, int k, int stride, int oh, int ow) {
  Plan pl;
  return dma1x1_ok(n, h, wd, c, k, stride, oh, ow, pl) ? 1 : 0;
}
extern "C" int embnet_conv2d_dma1x1_f32(const float* x, const void* wp, const float* bias, float* y, int n, int h, int wd, int c, int k,
                                        int stride, int oh, int ow, int relu, const float* residual, float* stats,
                                        const uint32_t* x_range, void* workspace, size_t workspace_bytes, void* stream) {
  EMBNET_CHECK_ARG(x && wp && y && x_range, "conv2d_dma1x1: null pointer (the activation's range slot is required)");
  EMBNET_CHECK_ARG(!(reinterpret_cast<uintptr_t>(x) & 15) && !(reinterpret_cast<uintptr_t>(x_range) & 3), "conv2d_dma1x1: alignment");
  Plan pl;
  EMBNET_CHECK_ARG(dma1x1_ok(n, h, wd, c, k, stride, oh, ow, pl), "conv2d_dma1x1: unsupported geometry (embnet_conv2d_dma1x1_supported)");
  PatchParams p{(const unsigned short*)x, (const unsigned short*)wp, y, bias, residual, stats, 0, relu};
  if (int rc = make_geom(p.g, n, h, wd, c, 1, 1, k, stride, 0, 0, oh, ow, "conv2d_dma1x1")) return rc;
  const long M = (long)n * oh * ow;
  p.x_plane_bytes = 0; p.x32_bytes = (unsigned)((size_t)n * h * wd * c * 4); p.a_range = x_range;
  p.w_plane_bytes = (unsigned)((size_t)c * k * 2);
  p.PH = oh; p.PW = ow; p.dPHW = FastDiv::make(oh * ow); p.dPW = FastDiv::make(ow);
  p.LR = 256;
  p.stats_rows = cdiv(M, 256) * 4;
  p.bn = BnSums{nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0};
  p.grid = pl.grid;
  if (pl.n_pieces > 0 && (pl.ws_bytes > workspace_bytes || !workspace)) { pl.n_full = pl.tiles; pl.n_pieces = 0; pl.parts = 1; }
  p.n_full = pl.n_full; p.parts = pl.parts; p.cc_part = pl.cc_part; p.n_pieces = pl.n_pieces; p.ws = (float*)workspace;
  p.xcd_rows = patch_xcd_rows();
  hipStream_t st = (hipStream_t)stream;
  {
    static thread_local char kname[160];
    snprintf(kname, sizeof kname, "void embnet::patch::conv1x1_a32_kernel<%d>(embnet::patch::PatchParams)", pl.bn);
    EMBNET_TRACE_FLOP(kname, 2.0 * M * k * c, 4.0 * ((double)M * c + (double)c * k) + 4.0 * (double)M * k * (residual ? 2 : 1), st);
    if (pl.bn == 128) launch_dma1x1<128>(p, pl.lds, st); else launch_dma1x1<64>(p, pl.lds, st);
  }
  if (p.n_pieces > 0)
    launch_tail_fixup(p.ws, p.parts, 256, pl.bn, 64, p.n_full, pl.tiles - p.n_full, cdiv(k, pl.bn), M, k, bias, relu, residual, y,
                      stats, p.stats_rows, p.bn, st);
  return check_launch("conv2d_dma1x1");
}

// The patch kernel as a stride-1 DATA GRADIENT (xp: planes of dy [n,h,wd,c], wp: the flipped kernel planes, y: the gradient
// [n,oh,ow,k] of the conv's input a = act(BN(bn_x))) that also emits the BatchNorm-backward sums of that BatchNormalization —
// embnet_conv2d_dgrad_bnsums_f32's contract, partial [2][k][bn_rows], bn_rows = embnet_conv2d_patch_stats_rows(n, oh, ow).
extern "C" int embnet_conv2d_patch_bnsums_f32(const void* xp, const void* wp, float* y, int n, int h, int wd, int c, int r, int s, int k,
                                              int pad_t, int pad_l, int oh, int ow, const float* bn_x, const float* bn_scale,
                                              const float* bn_shift, const float* bn_mean, const float* bn_rstd, int bn_act,
                                              float* bn_partial, int bn_rows, void* workspace, size_t workspace_bytes, void* stream) {
  EMBNET_CHECK_ARG(bn_x && bn_scale && bn_shift && bn_mean && bn_rstd && bn_partial, "conv2d_patch_bnsums: null pointer");
  EMBNET_CHECK_ARG(bn_act >= 0 && bn_act <= 2, "conv2d_patch_bnsums: activation code %d", bn_act);
  EMBNET_CHECK_ARG(bn_rows > 0 && bn_rows == embnet_conv2d_patch_stats_rows(n, oh, ow),
                   "conv2d_patch_bnsums: rows %d for this geometry (see embnet_conv2d_patch_stats_rows)", bn_rows);
  return conv2d_patch_impl(xp, wp, nullptr, y, n, h, wd, c, r, s, k, pad_t, pad_l, oh, ow, 0, nullptr, nullptr,
                           BnSums{bn_x, bn_scale, bn_shift, bn_mean, bn_rstd, bn_act, bn_partial, bn_rows}, workspace, workspace_bytes, stream);
}
