// fp32 MFMA GEMM engine for gfx950 (v_mfma_f32_32x32x2_f32, exact f32).
//
// One 256-thread workgroup (4 wave64) computes a BM x BN output tile; each wave
// owns a (BM/WAVES_M) x (BN/WAVES_N) sub-tile as TM x TN accumulators of 32x32.
// The K loop runs in tiles of BK = 32: every thread prefetches its float4 pieces
// of the NEXT A/B tiles from HBM/L2 into registers while the MFMAs consume the
// current tiles from LDS (two LDS stages, one barrier per tile, fragments read one
// k-step ahead — see gemm_mainloop; f32 MFMA issues once per 64 cycles per SIMD,
// so operand delivery is cheap next to the matrix pipe).
//
// Global loads are BRANCH-FREE buffer loads: every operand is addressed through a
// buffer descriptor (base in SGPRs, 32-bit byte offset per lane) and anything that
// must read as zero — rows past the matrix edge, k past K, convolution taps in the
// padding — is given an out-of-range offset, for which the hardware returns 0.
// (A `cond ? load : 0` in the source makes hipcc branch around each load and wait
// vmcnt(0) per element, serialising the whole prefetch; measured 2x slower.)
// Consequence: one operand tensor must be smaller than 2 GiB.
//
// Operands come in two LDS layouts, chosen per operand by the op:
//   TileKC<ROWS>: [ROWS][BK+4]  k contiguous  (activations gathered NHWC, rows of X)
//   TileKM<ROWS>: [BK][ROWS]    row contiguous (weights [k][cout], dY for wgrad)
// MFMA 32x32x2 wants lane (i = lane&31, h = lane>>5) to supply A[i][k] for the two
// k of a step.  We fix ONE k schedule for both operands: at step (j,t), j in 0..3,
// t in 0..3, lane-half h multiplies k = 8j + 4h + t.  A KC tile then feeds four
// steps from a single ds_read_b128 (conflict-free with the +4 pad), a KM tile
// from four ds_read_b32 of 32 consecutive floats.  Summation order inside a
// tile differs from plain k order; results are f32-rounding-equivalent.
//
// Diagnostic / ablation builds (tools/build_variant.sh -DEMBNET_DIAG_ENGINE=1 ...): the same engine with the round-1/2
// experiment switches threaded through its main loops (EMBNET_ABLATE, _INTERLEAVE, _PIN, _SETPRIO, _SPLIT_DIST,
// _SPLIT_ABLATE, _LDS_STAGES, in-loop stamps) lives in tools/exp/gemm_engine_diag.h; this file is the product.
#pragma once
#if defined(EMBNET_DIAG_ENGINE)
// The diagnostic copy of this header (round-1/2 experiment switches) is frozen at round 3: it lacks the K32 main loop and the
// k-major swizzle bit conv.hip needs since round 4.  To rebuild those experiment variants check out a round-3 tree (f78adeb).
#error "tools/exp/gemm_engine_diag.h is the round-3 engine; build the diagnostic variants from a round-3 checkout"
#else
#include "common.h"
#include <type_traits>

namespace embnet {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int NTHREADS = 256;

// In-kernel time stamps (diagnostic build only: -DEMBNET_STAMPS=1, tools/exp/conv_timeline.py).  The product
// build compiles every stamp() to nothing.  Slots per workgroup: 0 entry (s_memtime), 1 entry (s_memrealtime,
// 100 MHz, chip-wide), 2 loaders initialised, 3 first K tile in LDS (first MFMA can issue), 4 main loop done,
// 5 exit (s_memtime), 6 exit (s_memrealtime), 7 HW_ID | XCC_ID << 32, 8 first loads issued, 9 first tile written to
// LDS (loads landed), 10..15 free.  16 x uint64 per workgroup, in a buffer of their own.
#ifndef EMBNET_STAMPS
#define EMBNET_STAMPS 0
#endif
#if EMBNET_STAMPS
static __device__ unsigned long long* g_stamps = nullptr;
__device__ __forceinline__ void stamp(int slot) {
  if (threadIdx.x != 0 || !g_stamps) return;
  unsigned long long* d = g_stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16;
  d[slot] = __builtin_amdgcn_s_memtime();
  if (slot == 0 || slot == 5) d[slot + 1] = __builtin_amdgcn_s_memrealtime();
  if (slot == 0) d[7] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) |
                        ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
}
#else
__device__ __forceinline__ void stamp(int) {}
#endif
constexpr unsigned OOB = 0x80000000u;          // byte offset no operand (< 2 GiB) reaches
constexpr size_t MAX_OPERAND_BYTES = 0x7FFFFFF0ull;

// (opaque(): hook used by the diagnostic engine to pin address arithmetic; identity here)
__device__ __forceinline__ unsigned opaque(unsigned v) { return v; }

// Raw buffer view of one operand tensor.
struct Buf {
  __amdgpu_buffer_rsrc_t r;
  __device__ __forceinline__ void init(const void* p, size_t bytes) {
    r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (unsigned)bytes, 0x00020000);
  }
  __device__ __forceinline__ float4 ld4(unsigned off) const {
    // NB: cast the WHOLE vector.  Per-component __builtin_bit_cast(float, v.x) makes hipcc (ROCm 7.2)
    // narrow the load to buffer_load_dword while still consuming v[0:3] — three garbage lanes.
    const f32x4 f = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
    return make_float4(f.x, f.y, f.z, f.w);
  }
  __device__ __forceinline__ float ld1(unsigned off) const {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
  }
};

template <int ROWS>
struct TileKC {
  static constexpr int LD = BK + 4;
  static constexpr int FLOATS = ROWS * LD;
  static constexpr int PASSES = ROWS * (BK / 4) / NTHREADS;
  static_assert(PASSES >= 1, "tile too small for 256 threads");
  // pass p of thread tid covers row row_of(tid,p), k = k_of(tid) .. +3
  __device__ static __forceinline__ int row_of(int tid, int p) { return (p * NTHREADS + tid) >> 3; }
  __device__ static __forceinline__ int k_of(int tid) { return (tid & 7) * 4; }
  __device__ static __forceinline__ void store(float* s, const float4 (&r)[PASSES], int tid) {
#pragma unroll
    for (int p = 0; p < PASSES; ++p)
      *reinterpret_cast<float4*>(&s[row_of(tid, p) * LD + k_of(tid)]) = r[p];
  }
  __device__ static __forceinline__ void frag(const float* s, int r0, int j, int lane, float (&v)[4]) {
    const float4 q = *reinterpret_cast<const float4*>(&s[(r0 + (lane & 31)) * LD + 8 * j + 4 * (lane >> 5)]);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
  }
};

// Row pitch of the k-major tile: ROWS + 8 floats.  With a pitch of ROWS (a multiple of 64 floats) the two lane halves
// of a fragment read (k rows 4 apart) land on the same LDS banks: a plain GEMM with a k-major B operand ran 101 TFLOP/s
// at n = 4096 against 116 with the pad (both operands k-major: 109 -> 134); the conv kernels do not care (+-1 %).
#ifndef EMBNET_KM_PAD
#define EMBNET_KM_PAD 8
#endif
template <int ROWS>
struct TileKM {
  static constexpr int LD = ROWS + EMBNET_KM_PAD;
  static constexpr int FLOATS = BK * LD;
  static constexpr int PASSES = BK * (ROWS / 4) / NTHREADS;
  static_assert(PASSES >= 1, "tile too small for 256 threads");
  // pass p of thread tid covers k = k_of(tid,p), rows row_of(tid,p) .. +3
  __device__ static __forceinline__ int k_of(int tid, int p) { return (p * NTHREADS + tid) / (ROWS / 4); }
  __device__ static __forceinline__ int row_of(int tid, int p) { return ((p * NTHREADS + tid) % (ROWS / 4)) * 4; }
  __device__ static __forceinline__ void store(float* s, const float4 (&r)[PASSES], int tid) {
#pragma unroll
    for (int p = 0; p < PASSES; ++p)
      *reinterpret_cast<float4*>(&s[k_of(tid, p) * LD + row_of(tid, p)]) = r[p];
  }
  __device__ static __forceinline__ void frag(const float* s, int r0, int j, int lane, float (&v)[4]) {
    const float* b = &s[(8 * j + 4 * (lane >> 5)) * LD + r0 + (lane & 31)];
    v[0] = b[0]; v[1] = b[LD]; v[2] = b[2 * LD]; v[3] = b[3 * LD];
  }
};

// ---- generic dense-matrix loaders ----------------------------------------
// Matrix whose k index is contiguous: elem(row,k) = base[row*ld + k].
// VEC needs ld % 4 == 0, K % 4 == 0 and a 16-byte aligned base (whole float4 chunks in or out).
template <int ROWS, bool VEC>
struct LoadRowsKC {
  using Tile = TileKC<ROWS>;
  static constexpr bool CAN_INTERLEAVE = VEC;
  Buf buf; int K, tid;
  unsigned row_off[Tile::PASSES];
  __device__ void init(const float* b, long ld, int rows, int K_, int row0, int tid_) {
    buf.init(b, (size_t)rows * ld * 4); K = K_; tid = tid_;
#pragma unroll
    for (int p = 0; p < Tile::PASSES; ++p) {
      const int row = row0 + Tile::row_of(tid, p);
      row_off[p] = row < rows ? (unsigned)((long)row * ld * 4) : OOB;
    }
  }
  __device__ __forceinline__ void fix(float4 (&)[Tile::PASSES]) const {}
  // one pass (one 16-byte piece per thread) of K tile kt: the main loop issues the passes one at a time between its MFMAs
  __device__ __forceinline__ void load_pass(int kt, int p, float4& r) const {
    const int k = kt * BK + Tile::k_of(tid);
    const unsigned off = row_off[p] + 4u * k;              // stays out of range when the row is
    if (VEC) {
      r = buf.ld4(k < K ? off : OOB);
    } else {
      r = make_float4(buf.ld1(k < K ? off : OOB), buf.ld1(k + 1 < K ? off + 4 : OOB),
                      buf.ld1(k + 2 < K ? off + 8 : OOB), buf.ld1(k + 3 < K ? off + 12 : OOB));
    }
  }
  __device__ __forceinline__ void load(int kt, float4 (&r)[Tile::PASSES]) const {
#pragma unroll
    for (int p = 0; p < Tile::PASSES; ++p) load_pass(kt, p, r[p]);
  }
};

// Matrix whose row index is contiguous: elem(k,row) = base[k*ld + row].
// VEC needs ld % 4 == 0, rows % 4 == 0 and a 16-byte aligned base.
template <int ROWS, bool VEC>
struct LoadRowsKM {
  using Tile = TileKM<ROWS>;
  static constexpr bool CAN_INTERLEAVE = VEC;
  Buf buf; int K, rows, tid; unsigned ldb;
  unsigned col_off[Tile::PASSES];
  __device__ void init(const float* b, long ld, int rows_, int K_, int row0, int tid_) {
    buf.init(b, (size_t)K_ * ld * 4); K = K_; rows = rows_; tid = tid_; ldb = (unsigned)(ld * 4);
#pragma unroll
    for (int p = 0; p < Tile::PASSES; ++p) col_off[p] = 4u * (unsigned)(row0 + Tile::row_of(tid, p));
  }
  __device__ __forceinline__ void fix(float4 (&)[Tile::PASSES]) const {}
  __device__ __forceinline__ void load_pass(int kt, int p, float4& r) const {
    const int k = kt * BK + Tile::k_of(tid, p);
    const unsigned off = (unsigned)k * ldb + col_off[p];
    const int row = (int)(col_off[p] >> 2);
    if (VEC) {
      r = buf.ld4((k < K && row < rows) ? off : OOB);
    } else {
      const bool kin = k < K;
      r = make_float4(buf.ld1(kin && row < rows ? off : OOB), buf.ld1(kin && row + 1 < rows ? off + 4 : OOB),
                      buf.ld1(kin && row + 2 < rows ? off + 8 : OOB), buf.ld1(kin && row + 3 < rows ? off + 12 : OOB));
    }
  }
  __device__ __forceinline__ void load(int kt, float4 (&r)[Tile::PASSES]) const {
#pragma unroll
    for (int p = 0; p < Tile::PASSES; ++p) load_pass(kt, p, r[p]);
  }
};

// ---- tile geometry ----------------------------------------------------------
template <int BM_, int BN_, int WAVES_M_, int WAVES_N_>
struct Geom {
  static constexpr int BM = BM_, BN = BN_, WAVES_M = WAVES_M_, WAVES_N = WAVES_N_;
  static_assert(WAVES_M * WAVES_N == 4, "four waves per workgroup");
  static constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  static constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile below one MFMA");
};

template <class TA, class TB>
constexpr int MAIN_FLOATS = TA::FLOATS + TB::FLOATS;

template <class G, class TA, class TB>
__device__ __forceinline__ void load_frags(const float* st, int wm, int wn, int j, int lane,
                                           float (&a)[G::TM][4], float (&b)[G::TN][4]) {
#pragma unroll
  for (int i = 0; i < G::TM; ++i) TA::frag(st, wm + 32 * i, j, lane, a[i]);
#pragma unroll
  for (int i = 0; i < G::TN; ++i) TB::frag(st + TA::FLOATS, wn + 32 * i, j, lane, b[i]);
}

// Issue priority by phase.  A workgroup that starts on a CU whose other slots are in their main loops is the
// youngest wave on each SIMD and gets the left-over issue slots: its prologue (tile decode, first loads) ran 3x
// longer than on an idle CU (16-22k cycles vs 5k, in-kernel stamps: tools/exp/conv_timeline.py) and its epilogue
// likewise, during which its slot does no matrix work.  Prologue and epilogue therefore run at priority 3, the
// main loop at 0: the few hundred vector instructions they hold barely touch the older waves' MFMA stream.
__device__ __forceinline__ void prio_hi() { __builtin_amdgcn_s_setprio(3); }
__device__ __forceinline__ void prio_lo() { __builtin_amdgcn_s_setprio(0); }
// `fair` main loops: priority 3, 2, 1, 0 by quarter of the K range done (see gemm_mainloop)
__device__ __forceinline__ void prio_by_progress(bool fair, int kt, int kt_begin, int kt_end) {
  if (!fair) { prio_lo(); return; }
  const int done = 4 * (kt - kt_begin), span = kt_end - kt_begin;
  if (done >= 3 * span) __builtin_amdgcn_s_setprio(0);
  else if (done >= 2 * span) __builtin_amdgcn_s_setprio(1);
  else if (done >= span) __builtin_amdgcn_s_setprio(2);
}

template <class G>
__device__ __forceinline__ void mfma_step(const float (&a)[G::TM][4], const float (&b)[G::TN][4],
                                          f32x16 (&acc)[G::TM][G::TN]) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int im = 0; im < G::TM; ++im)
#pragma unroll
      for (int in = 0; in < G::TN; ++in)
        acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[im][t], b[in][t], acc[im][in], 0, 0, 0);
}

// Main loop (fp32 MFMA).  LA/LB: loaders with .load(kt, regs) and .fix(regs) — fix() runs on the fetched registers just
// before they go to LDS (one load outstanding per loader, so it may use state load() left); TA/TB: their LDS tile
// types.  smem holds MAIN_FLOATS<TA,TB> floats: one LDS buffer, the next K tile prefetched into registers under the MFMAs
// of the current one, two barriers per tile.  (A two-stage LDS, software-pipelined variant measured 3-5 % slower: it
// costs a resident workgroup per CU; tools/exp/gemm_engine_diag.h keeps it.)
// `fair` (wave-uniform): lower this wave's issue priority as it progresses through its K range (3, 2, 1, 0 by
// quarter) instead of running the loop at 0.  The SIMD arbitrates oldest-first, so workgroups that start together
// finish one after the other (4 co-resident 128x64 tiles: 68, 78, 91, 105 us) — fine while new workgroups keep
// arriving, but in the LAST round of a launch the CU ends up with 3, 2, 1 waves per SIMD and an under-fed matrix
// pipe.  Progress-ordered priority lets the waves that are behind catch up, so the last round ends together.
template <class G, class TA, class TB, class LA, class LB>
__device__ __forceinline__ void gemm_mainloop(const LA& la, const LB& lb, int kt_begin, int kt_end,
                                              float* smem, f32x16 (&acc)[G::TM][G::TN], bool fair = false,
                                              bool zero_acc = true) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = (wave / G::WAVES_N) * G::WTM, wn = (wave % G::WAVES_N) * G::WTN;
  constexpr int PAST = 1 << 24;            // k tile index past any K: every offset out of range, loads return 0

  if (zero_acc) {                            // false: continue accumulating (a K range visited in two pieces)
#pragma unroll
    for (int i = 0; i < G::TM; ++i)
#pragma unroll
      for (int j = 0; j < G::TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }

  float4 ra[TA::PASSES], rb[TB::PASSES];
  float* sA = smem;
  if (kt_begin < kt_end) { la.load(kt_begin, ra); lb.load(kt_begin, rb); la.fix(ra); lb.fix(rb); }
  stamp(8);
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    __syncthreads();                       // everyone finished reading the previous tile
    TA::store(sA, ra, tid);
    TB::store(sA + TA::FLOATS, rb, tid);
    __syncthreads();
    prio_by_progress(fair, kt, kt_begin, kt_end);
    // prefetch of the next tile, in flight under the MFMAs; past the end the k bound makes every
    // offset out of range, so the loads return zeros and need no branch
    la.load(kt + 1 < kt_end ? kt + 1 : PAST, ra);
    lb.load(kt + 1 < kt_end ? kt + 1 : PAST, rb);
#pragma unroll
    for (int j = 0; j < BK / 8; ++j) {
      float a[G::TM][4], b[G::TN][4];
      load_frags<G, TA, TB>(sA, wm, wn, j, lane, a, b);
      mfma_step<G>(a, b, acc);
    }
    // loader's register-side transform of the tile just fetched (usually none): here, behind the MFMAs it
    // overlaps with and outside the barrier pair, not between the barrier and the LDS store
    la.fix(ra); lb.fix(rb);
  }
  prio_hi();                                 // epilogue
}

// ---- fp32 products from bf16 matrix instructions: exact three-way split, six terms ("bf16x6") -----------------
// gfx950 has no xf32 and its fp32 MFMA runs at 1/16 of the bf16 rate.  Every fp32 operand x is split EXACTLY into
// three bf16 pieces x = x1 + x2 + x3 (8 significant bits each, see split4), and a product x*y is accumulated as the
// six terms
//   x1y3 + x3y1 + x2y2 + x1y2 + x2y1 + x1y1        (each exact in the fp32 accumulator's multiplier)
// the dropped x2y3 + x3y2 + x3y3 are <= 2^-20 |x||y| (typically 2^-22).  Measured against an
// fp64 reference (tools/exp/bf16x6_gemm.hip, K = 576..4096): max error 2.8-3.7e-7 of sum|a b|, the
// k-ordered fp32 fma chain of v_mfma_f32_32x32x2_f32 3.4-4.8e-7 — the split path is not a reduced-precision path.  Six
// v_mfma_f32_32x32x16_bf16 (32 cycles each, K = 16) replace eight fp32 MFMAs (64 cycles each, K = 2): 192 vs 512
// matrix-pipe cycles for the same 32x32x16 block.
// LDS holds the three pieces as separate bf16 planes.  The thread -> (row, k) mapping of the loaders is the fp32
// tiles' (TileKC / TileKM), so every loader works unchanged.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct Split4 { uint2 p[3]; };                 // four consecutive elements: 8 bytes per plane

// ---- the two-piece fp16 format of the planes (EMBNET_PLANES_F16=1) ---------------------------------------------
// x = (h1 + h2) / s with s a power of two per tensor: h1 = fp16(x s) (round to nearest), h2 = fp16(x s - h1); a product keeps
// three of the four piece products (h1 h1', h1 h2', h2 h1') on v_mfma_f32_32x32x16_f16 and is multiplied by 1 / (s s') in the
// epilogue (exact).  The planes buffers keep their three-plane size: planes 0 and 1 hold h1 and h2, and the first two floats of
// the third plane's space hold (s, 1 / s) — written by whoever writes the planes, read by whoever multiplies.
// PRECISION — what the format keeps, stated for |x s| (s puts a bound B >= max |x| of the tensor into [2^14, 2^15)):
//   * |x s| in [2^-3, 65504]: h2 is a normal fp16 -> x is kept to 2^-22 relative (22 mantissa bits);
//   * |x s| < 2^-3: h2 is an fp16 SUBNORMAL (spacing 2^-24) -> an ABSOLUTE error <= 2^-25, i.e. <= 2^-39 B: elements more than 2^17
//     below the tensor's bound lose relative precision, bit by bit;
//   * |x s| > 65504: fp16 overflows to +-inf and the products to inf / NaN — LOUD, never clamped.  It cannot happen with a sound
//     bound: every s in the library comes from an exact maximum (kernels, gradients) or from an upper bound (activations).
// There is NO fixed scale: round 5 ran activations at s = 1, which put every tensor of amplitude << 2^-3 on the subnormal floor
// (VERDICT r05 weak #2).  Every operand now carries a range; a conv without both ranges runs the six-term bf16 kernels.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
struct Split4H { uint2 p[2]; };
__device__ __forceinline__ Split4H split4h(const float4 v, float s) {
  typedef _Float16 h2v __attribute__((ext_vector_type(2)));
  const float x[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
  _Float16 hi[4], lo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hi[i] = (_Float16)x[i];
    lo[i] = (_Float16)(x[i] - (float)hi[i]);
  }
  Split4H r;
  r.p[0] = make_uint2(__builtin_bit_cast(uint32_t, h2v{hi[0], hi[1]}), __builtin_bit_cast(uint32_t, h2v{hi[2], hi[3]}));
  r.p[1] = make_uint2(__builtin_bit_cast(uint32_t, h2v{lo[0], lo[1]}), __builtin_bit_cast(uint32_t, h2v{lo[2], lo[3]}));
  return r;
}
// (s, 1 / s) for a tensor whose largest |element| is at most `bound`: bound * s lands in [2^14, 2^15).  bound = 0 (an all-zero
// tensor), infinite or NaN: s = 1 (nothing to scale / the values are poisoned anyway and stay so).  |k| <= 126 keeps s and 1 / s
// normal numbers; consumers multiply their sums by the two operands' 1 / s ONE AFTER THE OTHER (the product of two of them may not
// be representable although the result is).
__device__ __forceinline__ int scale_exponent_of(float bound) {
  const uint32_t bits = __float_as_uint(bound) & 0x7fffffffu;
  const int e = (int)(bits >> 23);                       // bound = f 2^(e - 127), f in [1, 2)  (e = 0: subnormal)
  if (bits == 0u || e >= 255) return 0;
  const int k = 141 - (e > 1 ? e : 1);
  return k > 126 ? 126 : (k < -126 ? -126 : k);
}
__device__ __forceinline__ float2 scale_pair(int k) {
  return make_float2(__uint_as_float((uint32_t)(127 + k) << 23), __uint_as_float((uint32_t)(127 - k) << 23));
}
// where a planes buffer of `plane_elems` 16-bit elements per plane keeps (s, 1 / s)
__host__ __device__ __forceinline__ float* planes_scale_slot(void* planes, long plane_elems) {
  return reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(planes) + 2 * plane_elems);
}
__host__ __device__ __forceinline__ const float* planes_scale_slot(const void* planes, long plane_elems) {
  return reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(planes) + 2 * plane_elems);
}
// three-term step of the fp16 format (the counterpart of mfma_step3 below)
template <int TM_, int TN_>
__device__ __forceinline__ void mfma_step_h(const f16x8 (&a)[TM_][2], const f16x8 (&b)[TN_][2], f32x16 (&acc)[TM_][TN_]) {
  constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};      // smallest first
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int im = 0; im < TM_; ++im)
#pragma unroll
      for (int in = 0; in < TN_; ++in)
        acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[im][PA[t]], b[in][PB[t]], acc[im][in], 0, 0, 0);
}

__device__ __forceinline__ uint32_t hi16_pair(uint32_t a, uint32_t b) {      // {top half of a, top half of b}
  return __builtin_amdgcn_perm(b, a, 0x07060302u);
}

// x = x1 + x2 + x3 by truncation: x1 = top 16 bits of x, x2 = top 16 bits of x - x1, x3 = x - x1 - x2 (both
// differences are exact and the last has at most 8 significant bits left, so it is a bf16).
// Round-to-nearest pieces (v_cvt_pk_bf16_f32) halve the dropped terms and remove their bias, but measured the same
// error against fp64 (3.4e-7 vs 3.7e-7 of sum|a b| at K = 4096: the fp32 accumulation dominates either way) and ran the
// conv kernels 6-7 % slower (tools/exp/bf16x6_gemm.hip, profiles/r02_exp_ab_split_rne.txt) -> truncation.
__device__ __forceinline__ Split4 split4(const float4 v) {
  const float x[4] = {v.x, v.y, v.z, v.w};
  uint32_t a[4], b[4], c[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = __float_as_uint(x[i]);
    const float r1 = x[i] - __uint_as_float(a[i] & 0xffff0000u);
    b[i] = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(b[i] & 0xffff0000u);
    c[i] = __float_as_uint(r2);
  }
  Split4 s;
  s.p[0] = make_uint2(hi16_pair(a[0], a[1]), hi16_pair(a[2], a[3]));
  s.p[1] = make_uint2(hi16_pair(b[0], b[1]), hi16_pair(b[2], b[3]));
  s.p[2] = make_uint2(hi16_pair(c[0], c[1]), hi16_pair(c[2], c[3]));
  return s;
}

// k-contiguous operand: three planes [ROWS][32 k] of bf16, 64-byte rows, the four 16-byte chunks of a row XOR-swizzled
// with (row >> 2) & 3.  Lane (i = lane&31, h = lane>>5) of k16-step st reads k = 16 st + 8 h .. +7 of row i with one
// ds_read_b128 per plane — the A/B operand map of v_mfma_f32_32x32x16_bf16; the swizzle puts the 16 lanes of each
// ds_read_b128 lane group ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) on 16 distinct 16-byte slots of the 256-byte
// bank window, and a 16-lane ds_write_b64 group covers two whole rows = 128 contiguous bytes.  (First version: 80-byte
// pitch, no swizzle — reads conflict-free, but the row pairs of a store overlapped mod 128 B: SQ_LDS_BANK_CONFLICT =
// 20-33 % of SQ_LDS_IDX_ACTIVE in the forward / data-gradient kernels, 0-1 % in the k-major weight-gradient kernels.)
template <int ROWS>
struct TileKC3 {
  using Map = TileKC<ROWS>;
  static constexpr int PASSES = Map::PASSES, PITCH = 64, PLANE = ROWS * PITCH, BYTES = 3 * PLANE;
  __device__ static __forceinline__ int off(int row, int chunk) { return row * PITCH + ((chunk ^ ((row >> 2) & 3)) << 4); }
  // (SP: Split4 — three bf16 planes — or Split4H — the two fp16 planes of the three-product format, which use planes 0 and 1)
  template <class SP>
  __device__ static __forceinline__ void store(unsigned char* s, const SP (&r)[PASSES], int tid) {
    constexpr int NP = sizeof(SP) / sizeof(uint2);
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      unsigned char* d = s + off(Map::row_of(tid, p), (tid & 7) >> 1) + (tid & 1) * 8;      // k_of(tid) = 4 * (tid & 7)
#pragma unroll
      for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(d + q * PLANE) = r[p].p[q];
    }
  }
  template <class V, int NP>
  __device__ static __forceinline__ void frag(const unsigned char* s, int r0, int st, int lane, V (&v)[NP]) {
    const unsigned char* a = s + off(r0 + (lane & 31), 2 * st + (lane >> 5));
#pragma unroll
    for (int q = 0; q < NP; ++q) v[q] = *reinterpret_cast<const V*>(a + q * PLANE);
  }
};

// row-contiguous (k-major) operand: three planes [32 k][ROWS] of bf16; the fragment (8 consecutive k of one row per
// lane) comes from two ds_read_b64_tr_b16 per plane — each 16-lane group reads a 4 k x 16 rows block and receives it
// transposed (lane 4q+p supplies the address of k-row q, rows 4p..4p+3; lane i receives row i, k-row q in element q).
// A 32-lane half reads 64 contiguous bytes of four consecutive k-rows, which must fall on different banks (256-byte
// window): ROWS = 64 / 128 (128- / 256-byte rows) swizzle the 64-byte slots of a k-row with its k (see off), other sizes
// pad the pitch to 64 mod 128 bytes.  (Unpadded: 12 / 24 KB per operand instead of 18 / 30.)
template <int ROWS>
struct TileKM3 {
  using Map = TileKM<ROWS>;
  static constexpr int PASSES = Map::PASSES;
  static constexpr bool SWZ = ROWS == 64 || ROWS == 128;
  static constexpr int PITCH = SWZ ? 2 * ROWS : 2 * ROWS + ((2 * ROWS) % 128 == 64 ? 0 : 64);
  static constexpr int PLANE = BK * PITCH, BYTES = 3 * PLANE;
  // byte offset of row `row` (a multiple of 4) of k-row k.  128-byte rows: k-rows 0..3 -> slots (0|1) of the windows
  // [0,128) and [128,256): flip the 64-byte half with bit 1 of k; 256-byte rows: XOR the 64-byte slot with k & 3.
  // (bit 3 of k also flips the 32-byte half: the K = 32 fragments of frag16 read k-rows 8 apart in one 32-lane pass; the
  // K = 16 fragments never mix the two values of bit 3 in a pass, so they do not notice)
  __device__ static __forceinline__ int off(int k, int row) {
    const int b = (row * 2) ^ (((k >> 3) & 1) << 5);
    if (!SWZ) return k * PITCH + b;
    return k * PITCH + (ROWS == 64 ? (b ^ (((k >> 1) & 1) << 6)) : (b ^ ((k & 3) << 6)));
  }
  template <class SP>
  __device__ static __forceinline__ void store(unsigned char* s, const SP (&r)[PASSES], int tid) {
    constexpr int NP = sizeof(SP) / sizeof(uint2);
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      unsigned char* d = s + off(Map::k_of(tid, p), Map::row_of(tid, p));
#pragma unroll
      for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(d + q * PLANE) = r[p].p[q];
    }
  }
  template <class V, int NP>
  __device__ static __forceinline__ void frag(const unsigned char* s, int r0, int st, int lane, V (&v)[NP]) {
    const int k = 16 * st + 8 * (lane >> 5) + ((lane & 15) >> 2);
    const int row = r0 + ((lane >> 4) & 1) * 16 + 4 * (lane & 3);
    const unsigned char* a = s + off(k, row);
    const unsigned char* a4 = s + off(k + 4, row);
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a + q * PLANE));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a4 + q * PLANE));
      const s16x8 w = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      v[q] = __builtin_bit_cast(V, w);
    }
  }
  // the A/B operand of v_mfma_f32_16x16x32_bf16 for rows r0 .. r0+15 over ALL 32 k of the tile: lane (i = lane & 15,
  // g = lane >> 4) receives k = 8g .. 8g+7 of row r0 + i, again from two transposed reads per plane (k-rows 8g+0..3, 8g+4..7)
  __device__ static __forceinline__ void frag16(const unsigned char* s, int r0, int lane, bf16x8 (&v)[3]) {
    const int k = 8 * (lane >> 4) + ((lane & 15) >> 2);
    const int row = r0 + 4 * (lane & 3);
    const unsigned char* a = s + off(k, row);
    const unsigned char* a4 = s + off(k + 4, row);
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a + q * PLANE));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a4 + q * PLANE));
      const s16x8 w = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      v[q] = __builtin_bit_cast(bf16x8, w);
    }
  }
};

template <class T> struct SplitTile;
template <int R> struct SplitTile<TileKC<R>> { using type = TileKC3<R>; };
template <int R> struct SplitTile<TileKM<R>> { using type = TileKM3<R>; };

template <class TA, class TB>
constexpr int MAIN3_BYTES = SplitTile<TA>::type::BYTES + SplitTile<TB>::type::BYTES;

// EMBNET_EXP_TERMS (build-time experiment, tools/exp/run_r05_terms3.sh): 3 executes only the three largest of the six terms —
// NOT the product's arithmetic (16-bit products) — to measure what a three-term scheme (an fp16 x 2 split) would
// run at.  The product is built with 6.
#ifndef EMBNET_EXP_TERMS
#define EMBNET_EXP_TERMS 6
#endif
// one k16-step: six terms per accumulator, smallest first; the accumulators of a wave alternate inside a term so
// consecutive MFMAs are independent
template <class G>
__device__ __forceinline__ void mfma_step3(const bf16x8 (&a)[G::TM][3], const bf16x8 (&b)[G::TN][3],
                                           f32x16 (&acc)[G::TM][G::TN]) {
  constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
  for (int t = 6 - EMBNET_EXP_TERMS; t < 6; ++t)
#pragma unroll
    for (int im = 0; im < G::TM; ++im)
#pragma unroll
      for (int in = 0; in < G::TN; ++in)
        acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[im][PA[t]], b[in][PB[t]], acc[im][in], 0, 0, 0);
}

// Main loop of the split path: one LDS buffer, two barriers per K tile, the split arithmetic (11 VALU ops per pair of
// elements) on prefetched registers BEHIND the MFMAs and outside the barrier pair, so between the barriers there are
// only the 8-byte LDS stores.  TA/TB are the fp32 tile types the loaders were written for.
// The registers are free as soon as they are split, so tile kt+2 is requested right behind the split of tile kt+1
// (before the barrier pair and the LDS stores of the next iteration), not after them (+1.5 %).
// Measured and not adopted (tools/exp/gemm_engine_diag.h, profiles/r02_exp_ab_split_*.txt): two register sets in flight
// (EMBNET_SPLIT_DIST=2: +48..130 registers, a workgroup per CU fewer, 5-30 % slower), rounded instead of truncated
// pieces (same error, 6-7 % slower).  Round 3: what bounds this loop is the per-CU gather rate / latency.
// H = true: the operands are split into the TWO fp16 pieces of x * s (split4h; sa / sb: each operand's power-of-two scale, chosen by
// the caller so that the tensor's largest element lands in [2^14, 2^15)) and a product keeps three piece products — the planes
// kernels' format, here made on the fly from fp32 operands.  The accumulators then hold sa * sb times the result:
// the caller multiplies by 1 / (sa sb) — exact, a power of two — before its epilogue.
template <class G, class TA, class TB, class LA, class LB, bool H = false>
__device__ __forceinline__ void gemm_mainloop3(const LA& la, const LB& lb, int kt_begin, int kt_end,
                                               unsigned char* smem, f32x16 (&acc)[G::TM][G::TN], bool fair = false,
                                               bool zero_acc = true, float sa = 1.f, float sb = 1.f) {
  using SA = typename SplitTile<TA>::type;
  using SB = typename SplitTile<TB>::type;
  using SP = typename std::conditional<H, Split4H, Split4>::type;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = (wave / G::WAVES_N) * G::WTM, wn = (wave % G::WAVES_N) * G::WTN;
  constexpr int PAST = 1 << 24;
  if (zero_acc) {
#pragma unroll
    for (int i = 0; i < G::TM; ++i)
#pragma unroll
      for (int j = 0; j < G::TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }
  if (kt_begin >= kt_end) { prio_hi(); return; }
  SP pa[TA::PASSES], pb[TB::PASSES];
  unsigned char* sA = smem;
  unsigned char* sB = smem + SA::BYTES;
  float4 ra[TA::PASSES], rb[TB::PASSES];
  auto split_all = [&]() {
    if constexpr (H) {
#pragma unroll
      for (int p = 0; p < TA::PASSES; ++p) pa[p] = split4h(ra[p], sa);
#pragma unroll
      for (int p = 0; p < TB::PASSES; ++p) pb[p] = split4h(rb[p], sb);
    } else {
#pragma unroll
      for (int p = 0; p < TA::PASSES; ++p) pa[p] = split4(ra[p]);
#pragma unroll
      for (int p = 0; p < TB::PASSES; ++p) pb[p] = split4(rb[p]);
    }
  };
  la.load(kt_begin, ra); lb.load(kt_begin, rb); la.fix(ra); lb.fix(rb);
  split_all();
  la.load(kt_begin + 1 < kt_end ? kt_begin + 1 : PAST, ra);
  lb.load(kt_begin + 1 < kt_end ? kt_begin + 1 : PAST, rb);
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    __syncthreads();                         // everyone finished reading the previous tile
    SA::store(sA, pa, tid);
    SB::store(sB, pb, tid);
    __syncthreads();
    prio_by_progress(fair, kt, kt_begin, kt_end);
#pragma unroll
    for (int st = 0; st < BK / 16; ++st) {
      if constexpr (H) {
        f16x8 a[G::TM][2], b[G::TN][2];
#pragma unroll
        for (int i = 0; i < G::TM; ++i) SA::frag(sA, wm + 32 * i, st, lane, a[i]);
#pragma unroll
        for (int i = 0; i < G::TN; ++i) SB::frag(sB, wn + 32 * i, st, lane, b[i]);
        mfma_step_h<G::TM, G::TN>(a, b, acc);
      } else {
        bf16x8 a[G::TM][3], b[G::TN][3];
#pragma unroll
        for (int i = 0; i < G::TM; ++i) SA::frag(sA, wm + 32 * i, st, lane, a[i]);
#pragma unroll
        for (int i = 0; i < G::TN; ++i) SB::frag(sB, wn + 32 * i, st, lane, b[i]);
        mfma_step3<G>(a, b, acc);
      }
    }
    la.fix(ra); lb.fix(rb);
    if (kt + 1 < kt_end) split_all();        // tile kt+1, requested one iteration ago
    la.load(kt + 2 < kt_end ? kt + 2 : PAST, ra);
    lb.load(kt + 2 < kt_end ? kt + 2 : PAST, rb);
  }
  prio_hi();                                 // epilogue
}

// ---- the same six-term products on v_mfma_f32_16x16x32_bf16 ("K32") ------------------------------------------------------
// Same matrix-pipe cycles per FLOP as the 32x32x16 form, same fragment-read count when BOTH operands are k-major (two
// transposed reads per 16 rows x 32 k and plane, against two per 32 rows x 16 k), but the chip holds a higher clock on this
// shape under load (HIP guide rule 28 / microarchitecture guide 'DVFS give-back' item 7: 1.12-1.15 x on random data in bare
// loops).  A wave's WTM x WTN tile is (2 TM) x (2 TN) blocks of 16 x 16, four accumulator registers each:
// acc[im][in][sm * 2 + sn] is the block at rows 32 im + 16 sm, columns 32 in + 16 sn; C/D map: col = lane & 15,
// row = 4 * (lane >> 4) + r.  One K tile (32 k) is one step.
template <class G>
using Acc16 = f32x4[G::TM][G::TN][4];

template <class G, class SA, class SB>
__device__ __forceinline__ void mfma_tile3_k32(const unsigned char* sA, const unsigned char* sB, int wm, int wn, int lane,
                                               f32x4 (&acc)[G::TM][G::TN][4]) {
  constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
  bf16x8 a[2 * G::TM][3];
#pragma unroll
  for (int i = 0; i < 2 * G::TM; ++i) SA::frag16(sA, wm + 16 * i, lane, a[i]);
#pragma unroll
  for (int j = 0; j < 2 * G::TN; ++j) {
    bf16x8 b[3];
    SB::frag16(sB, wn + 16 * j, lane, b);
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < 2 * G::TM; ++i)
        acc[i >> 1][j >> 1][(i & 1) * 2 + (j & 1)] =
            __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][PA[t]], b[PB[t]], acc[i >> 1][j >> 1][(i & 1) * 2 + (j & 1)], 0, 0, 0);
  }
}

// gemm_mainloop3 on the K32 form; both tiles k-major (the weight-gradient GEMM).
template <class G, class TA, class TB, class LA, class LB>
__device__ __forceinline__ void gemm_mainloop3_k32(const LA& la, const LB& lb, int kt_begin, int kt_end,
                                                   unsigned char* smem, f32x4 (&acc)[G::TM][G::TN][4], bool fair = false,
                                                   bool zero_acc = true) {
  using SA = typename SplitTile<TA>::type;
  using SB = typename SplitTile<TB>::type;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = (wave / G::WAVES_N) * G::WTM, wn = (wave % G::WAVES_N) * G::WTN;
  constexpr int PAST = 1 << 24;
  if (zero_acc) {
#pragma unroll
    for (int i = 0; i < G::TM; ++i)
#pragma unroll
      for (int j = 0; j < G::TN; ++j)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][b][r] = 0.f;
  }
  if (kt_begin >= kt_end) { prio_hi(); return; }
  Split4 pa[TA::PASSES], pb[TB::PASSES];
  unsigned char* sA = smem;
  unsigned char* sB = smem + SA::BYTES;
  float4 ra[TA::PASSES], rb[TB::PASSES];
  auto split_all = [&]() {
#pragma unroll
    for (int p = 0; p < TA::PASSES; ++p) pa[p] = split4(ra[p]);
#pragma unroll
    for (int p = 0; p < TB::PASSES; ++p) pb[p] = split4(rb[p]);
  };
  la.load(kt_begin, ra); lb.load(kt_begin, rb); la.fix(ra); lb.fix(rb);
  split_all();
  la.load(kt_begin + 1 < kt_end ? kt_begin + 1 : PAST, ra);
  lb.load(kt_begin + 1 < kt_end ? kt_begin + 1 : PAST, rb);
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    __syncthreads();
    SA::store(sA, pa, tid);
    SB::store(sB, pb, tid);
    __syncthreads();
    prio_by_progress(fair, kt, kt_begin, kt_end);
    mfma_tile3_k32<G, SA, SB>(sA, sB, wm, wn, lane, acc);
    la.fix(ra); lb.fix(rb);
    if (kt + 1 < kt_end) split_all();
    la.load(kt + 2 < kt_end ? kt + 2 : PAST, ra);
    lb.load(kt + 2 < kt_end ? kt + 2 : PAST, rb);
  }
  prio_hi();
}

// for_each_acc_row4 for the K32 accumulators (same staging, same callback)
template <class G, class F>
__device__ __forceinline__ void for_each_acc16_row4(const f32x4 (&acc)[G::TM][G::TN][4], float* smem, F&& f) {
  constexpr int LDW = G::WTN + 4, LPR = G::WTN / 4, RPI = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave / G::WAVES_N) * G::WTM, wn = (wave % G::WAVES_N) * G::WTN;
  float* s = smem + wave * 32 * LDW;
  __syncthreads();
#pragma unroll
  for (int im = 0; im < G::TM; ++im) {
#pragma unroll
    for (int in = 0; in < G::TN; ++in)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          s[((b >> 1) * 16 + 4 * (lane >> 4) + r) * LDW + in * 32 + (b & 1) * 16 + (lane & 15)] = acc[im][in][b][r];
#pragma unroll
    for (int rr = 0; rr < 32; rr += RPI) {
      const int row = rr + lane / LPR, c4 = (lane % LPR) * 4;
      const float4 v = *reinterpret_cast<const float4*>(&s[row * LDW + c4]);
      f(wm + 32 * im + row, wn + c4, v);
    }
  }
}

template <class G, class F>
__device__ __forceinline__ void for_each_acc16(const f32x4 (&acc)[G::TM][G::TN][4], F&& f) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave / G::WAVES_N) * G::WTM, wn = (wave % G::WAVES_N) * G::WTN;
#pragma unroll
  for (int im = 0; im < G::TM; ++im)
#pragma unroll
    for (int in = 0; in < G::TN; ++in)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          f(wm + 32 * im + (b >> 1) * 16 + 4 * (lane >> 4) + r, wn + 32 * in + (b & 1) * 16 + (lane & 15), acc[im][in][b][r]);
}

// Walk the accumulators: f(row_in_tile, col_in_tile, value) for this lane's elements.
// C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
template <class G, class F>
__device__ __forceinline__ void for_each_acc(const f32x16 (&acc)[G::TM][G::TN], F&& f) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave / G::WAVES_N) * G::WTM, wn = (wave % G::WAVES_N) * G::WTN;
#pragma unroll
  for (int im = 0; im < G::TM; ++im)
#pragma unroll
    for (int in = 0; in < G::TN; ++in)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        f(wm + 32 * im + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), wn + 32 * in + (lane & 31), acc[im][in][r]);
}

// Same walk with the block indices: f(im, in, row, col, value) — im / in are constants after unrolling, so values the
// epilogue adds per column block (bias) can sit in a register array loaded before the walk.
template <class G, class F>
__device__ __forceinline__ void for_each_acc_idx(const f32x16 (&acc)[G::TM][G::TN], F&& f) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave / G::WAVES_N) * G::WTM, wn = (wave % G::WAVES_N) * G::WTN;
#pragma unroll
  for (int im = 0; im < G::TM; ++im)
#pragma unroll
    for (int in = 0; in < G::TN; ++in)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        f(im, in, wm + 32 * im + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), wn + 32 * in + (lane & 31), acc[im][in][r]);
}

// Same walk, but as 16-byte row pieces: each wave stages one 32-row band of its sub-tile through a
// private LDS region ([32][WTN+4]) and reads it back row-major, so the caller can issue
// global_store_dwordx4 (256 contiguous bytes per 16 lanes) instead of one dword per lane — 4x fewer
// store instructions; a dword-per-lane epilogue is store-ISSUE-bound (~7 B/clk/CU), and because all
// workgroups of a round reach their epilogue together the matrix pipe idles meanwhile.
// f(row_in_tile, col_in_tile (multiple of 4), float4).  smem must hold EPI_FLOATS<G> floats and may be
// the operand tile buffer (a barrier is taken first).  LDS ops of one wave execute in order, so the
// band's ds_writes are visible to the same wave's ds_reads without a barrier.
template <class G>
constexpr int EPI_FLOATS = 4 * 32 * (G::WTN + 4);

// The lane's share of the tile in that epilogue: NJ row quads per 32-row block, all in ONE column quad —
//   row(im, j) = epi_row<G>(im, j),  column = epi_col<G>()      (tile-relative; im < G::TM, j < EpiIdx<G>::NJ).
// An epilogue that ADDS something read from memory (bias, the residual tensor, the other gradient of a skip connection,
// row / column norms) loads it for all its (im, j) BEFORE the staging loop: written inside the per-row callback, under
// the edge test, each load is followed by s_waitcnt vmcnt(0) — 16 serial memory round trips per tile (the generated code
// of the round-1..3 epilogues: 30 us of a 1x1 convolution's tile whose main loop is 4..16 K tiles).
// settle(v): a use + redefinition of v the compiler cannot see through.  Placed after the pre-loop loads (in straight-line
// code) it takes their s_waitcnt ONCE; without it the wait is re-inserted at every use inside the staging loop — the uses
// sit in divergent edge-test branches, so the "load still pending" state survives each merge — as s_waitcnt vmcnt(0),
// which on gfx9 also waits for the previous iteration's STORE: 16 serial store round trips per tile.
__device__ __forceinline__ void settle(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void settle(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
template <class G> struct EpiIdx { static constexpr int LPR = G::WTN / 4, RPI = 64 / LPR, NJ = 32 / RPI; };
template <class G> __device__ __forceinline__ int epi_row(int im, int j) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  return (wave / G::WAVES_N) * G::WTM + 32 * im + j * EpiIdx<G>::RPI + lane / EpiIdx<G>::LPR;
}
template <class G> __device__ __forceinline__ int epi_col() {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  return (wave % G::WAVES_N) * G::WTN + (lane % EpiIdx<G>::LPR) * 4;
}

template <class G, class F>     // f(im, j, tile row, tile column, float4): im, j are compile-time constants after unrolling
__device__ __forceinline__ void for_each_acc_row4_idx(const f32x16 (&acc)[G::TM][G::TN], float* smem, F&& f) {
  constexpr int LDW = G::WTN + 4, LPR = G::WTN / 4, RPI = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave / G::WAVES_N) * G::WTM, wn = (wave % G::WAVES_N) * G::WTN;
  float* s = smem + wave * 32 * LDW;
  __syncthreads();
#pragma unroll
  for (int im = 0; im < G::TM; ++im) {
#pragma unroll
    for (int in = 0; in < G::TN; ++in)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        s[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDW + in * 32 + (lane & 31)] = acc[im][in][r];
#pragma unroll
    for (int j = 0; j < 32 / RPI; ++j) {
      const int row = j * RPI + lane / LPR, c4 = (lane % LPR) * 4;
      const float4 v = *reinterpret_cast<const float4*>(&s[row * LDW + c4]);
      f(im, j, wm + 32 * im + row, wn + c4, v);
    }
  }
}

template <class G, class F>
__device__ __forceinline__ void for_each_acc_row4(const f32x16 (&acc)[G::TM][G::TN], float* smem, F&& f) {
  for_each_acc_row4_idx<G>(acc, smem, [&](int, int, int r, int c, float4 v) { f(r, c, v); });
}

// Sum a per-lane float4 over the lanes that share a column quad after for_each_acc_row4 (lane % LPR equal).
template <int LPR>
__device__ __forceinline__ float4 colquad_sum(float4 v) {
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) {
    v.x += __shfl_xor(v.x, o, 64); v.y += __shfl_xor(v.y, o, 64);
    v.z += __shfl_xor(v.z, o, 64); v.w += __shfl_xor(v.w, o, 64);
  }
  return v;
}

template <int LPR>
__device__ __forceinline__ float4 colquad_max(float4 v) {
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) {
    v.x = fmaxf(v.x, __shfl_xor(v.x, o, 64)); v.y = fmaxf(v.y, __shfl_xor(v.y, o, 64));
    v.z = fmaxf(v.z, __shfl_xor(v.z, o, 64)); v.w = fmaxf(v.w, __shfl_xor(v.w, o, 64));
  }
  return v;
}

// blockIdx.x -> (tile_m, tile_n) keeping the workgroups that share an XCD (ids equal mod 8)
// on neighbouring tiles so they reuse operand panels in that XCD's L2.  Bijective for any count.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

}  // namespace embnet
#endif  // EMBNET_DIAG_ENGINE
