// Experiment (standalone, not in the library): 3x3 / stride 1 / pad 1 forward convolution that loads every input
// element ONCE per workgroup instead of once per tap.  The library's implicit-GEMM kernels gather the im2col operand per
// K tile (one tap x 32 channels), so a 3x3 layer issues nine global loads and nine bf16 splits per input element and
// workgroup; measured their throughput follows those loads per MFMA.  Here a workgroup owns an 8 x 16
// block of output pixels of one image: the 10 x 18 halo patch of a 32-channel chunk is split into the three bf16 planes
// and stored in LDS once, and the nine taps read it with a shifted fragment address; only the weights stream per tap.
//   global loads per MFMA (128 x 64 tile): (6 patch + 9 x 2 weight) / 216 = 0.11   vs   9 x (4 + 2) / 216 = 0.25
// Cost: edge blocks (56 = 3.5 x 16) do MFMA work on pixels outside the image (12.5 % on 56x56, 27 % on 28x28).
// Same arithmetic as the library (exact three-way bf16 split, six terms, fp32 accumulation).
//   hipcc -O3 --offload-arch=gfx950 -o tools/exp/patch3x3_conv tools/exp/patch3x3_conv.hip && tools/exp/patch3x3_conv
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int TH = 8, TW = 16, PH = TH + 2, PW = TW + 2, NPIX = PH * PW;        // 180 patch pixels
constexpr int BN = 64, CK = 32;                                                   // output channels per workgroup, channels per chunk
constexpr int A_PLANE = NPIX * 64, B_PITCH = 2 * BN, B_PLANE = CK * B_PITCH;      // bytes
constexpr int A_PASSES = (NPIX * 8 + 255) / 256;                                  // float4 pieces of a patch chunk per thread: 6
#ifndef TG
#define TG 3                                                                      // taps per LDS stage (per barrier pair): 1 or 3
#endif
constexpr int B_PASSES = 2 * TG;

struct Split4 { uint2 p[3]; };
__device__ __forceinline__ uint32_t hi16_pair(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
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

// patch image: [pixel][32 ch] bf16 = 64-byte rows, 16-byte chunks swizzled with (pixel >> 2) & 3
__device__ __forceinline__ int a_off(int pix, int chunk) { return pix * 64 + ((chunk ^ ((pix >> 2) & 3)) << 4); }
// weight image: [32 c][64 k] bf16 = 128-byte rows, the 64-byte half flipped with bit 1 of c (transposed reads)
__device__ __forceinline__ int b_off(int c, int kbyte) { return c * B_PITCH + (kbyte ^ (((c >> 1) & 1) << 6)); }

#ifndef OCC
#define OCC 3
#endif
__global__ __launch_bounds__(256, OCC) void conv3x3_patch_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               float* __restrict__ y, int N, int H, int W, int C, int K) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * A_PLANE + 3 * TG * B_PLANE];
  unsigned char* sA = lds;
  unsigned char* sB = lds + 3 * A_PLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH, tiles_k = K / BN;
  int t = blockIdx.x;
  const int kt = t % tiles_k; t /= tiles_k;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int n = t / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW, k0 = kt * BN;

  // patch pieces of this thread: piece idx = pass * 256 + tid -> pixel idx >> 3, channel quad idx & 7 (fixed per thread)
  int a_src[A_PASSES], a_dst[A_PASSES]; bool a_ok[A_PASSES], a_in[A_PASSES];      // element offsets: tensors < 2^31 elements
#pragma unroll
  for (int p = 0; p < A_PASSES; ++p) {
    const int idx = p * 256 + tid, pix = idx >> 3, kq = idx & 7;
    const int py = pix / PW, px = pix - py * PW;
    const int ih = y0 + py - 1, iw = x0 + px - 1;
    a_in[p] = idx < NPIX * 8;
    a_ok[p] = a_in[p] && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    a_src[p] = a_ok[p] ? ((n * H + ih) * W + iw) * C + kq * 4 : 0;
    a_dst[p] = a_off(pix, kq >> 1) + (kq & 1) * 8;
  }
  // weight pieces: 32 c x 64 k floats per (tap, chunk) = 512 float4: two per thread and tap
  int b_c[2], b_k[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) { const int idx = p * 256 + tid; b_c[p] = idx >> 4; b_k[p] = (idx & 15) * 4; }

  // fragment bases: rows of this wave = 64 pixels = 4 rows of the 8 x 16 block; lane i of block b -> pixel row 2b + (i >> 4)
  int a_base[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) a_base[b] = (wm * 4 + 2 * b + ((lane & 31) >> 4)) * PW + (lane & 15);
  const int b_row = wn * 32 + ((lane >> 4) & 1) * 16 + 4 * (lane & 3);      // k (output channel) of the transposed read
  const int b_cq = 8 * (lane >> 5) + ((lane & 15) >> 2);

  f32x16 acc[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;

  const int nchunk = C / CK;
  float4 ra[A_PASSES], rb[B_PASSES];
  Split4 pb[B_PASSES];
  auto load_a = [&](int cc) {
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {                   // unconditional load (offset 0 for halo pixels outside the image), zeroed after
      ra[p] = *reinterpret_cast<const float4*>(x + a_src[p] + cc * CK);
      if (!a_ok[p]) ra[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto load_b = [&](int cc, int tap0) {
#pragma unroll
    for (int g = 0; g < TG; ++g)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        rb[2 * g + p] = *reinterpret_cast<const float4*>(w + ((long)(tap0 + g) * C + cc * CK + b_c[p]) * K + k0 + b_k[p]);
  };
  load_a(0); load_b(0, 0);
#pragma unroll
  for (int p = 0; p < B_PASSES; ++p) pb[p] = split4(rb[p]);

  constexpr int NSTAGE = 9 / TG;
  for (int cc = 0; cc < nchunk; ++cc) {
    for (int sg = 0; sg < NSTAGE; ++sg) {
      __syncthreads();
      if (sg == 0) {                                       // the patch chunk is split here, once per nine taps (no split registers
#pragma unroll                                             // held across the taps: one more workgroup per CU)
        for (int p = 0; p < A_PASSES; ++p)
          if (a_in[p]) {
            const Split4 t4 = split4(ra[p]);
#pragma unroll
            for (int q = 0; q < 3; ++q) *reinterpret_cast<uint2*>(sA + q * A_PLANE + a_dst[p]) = t4.p[q];
          }
      }
#pragma unroll
      for (int g = 0; g < TG; ++g)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int q = 0; q < 3; ++q)
            *reinterpret_cast<uint2*>(sB + (q * TG + g) * B_PLANE + b_off(b_c[p], b_k[p] * 2)) = pb[2 * g + p].p[q];
      __syncthreads();
      // next weights (and, at the first stage, the next patch chunk: it has the nine taps of this chunk to arrive)
      const int nsg = sg == NSTAGE - 1 ? 0 : sg + 1, ncc = sg == NSTAGE - 1 ? cc + 1 : cc;
      if (ncc < nchunk) load_b(ncc, nsg * TG);
      if (sg == 0 && cc + 1 < nchunk) load_a(cc + 1);
#pragma unroll
      for (int g = 0; g < TG; ++g) {
        const int tap = sg * TG + g;
        const int tapoff = (tap / 3) * PW + (tap % 3);
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          bf16x8 a[2][3], b[3];
#pragma unroll
          for (int bl = 0; bl < 2; ++bl) {
            const unsigned char* ap = sA + a_off(a_base[bl] + tapoff, 2 * st + (lane >> 5));
#pragma unroll
            for (int q = 0; q < 3; ++q) a[bl][q] = *reinterpret_cast<const bf16x8*>(ap + q * A_PLANE);
          }
          {
            typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
            const unsigned char* bp = sB + g * B_PLANE + b_off(16 * st + b_cq, b_row * 2);
            const unsigned char* bp4 = sB + g * B_PLANE + b_off(16 * st + b_cq + 4, b_row * 2);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(bp + q * TG * B_PLANE));
              const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(bp4 + q * TG * B_PLANE));
              const s16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
              b[q] = __builtin_bit_cast(bf16x8, v);
            }
          }
          constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
          for (int tt = 0; tt < 6; ++tt)
#pragma unroll
            for (int bl = 0; bl < 2; ++bl)
              acc[bl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[bl][PA[tt]], b[PB[tt]], acc[bl], 0, 0, 0);
        }
      }
      if (ncc < nchunk) {
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p) pb[p] = split4(rb[p]);
      }
    }
  }
  // epilogue (plain): C/D map of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int bl = 0; bl < 2; ++bl)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = wm * 64 + bl * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      const int oy = y0 + (row >> 4), ox = x0 + (row & 15), col = k0 + wn * 32 + (lane & 31);
      if (oy < H && ox < W) y[(((long)n * H + oy) * W + ox) * K + col] = acc[bl][e];
    }
}

#define CK_(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static void run(int N, int H, int W, int C, int K) {
  std::vector<float> hx((size_t)N * H * W * C), hw((size_t)9 * C * K);
  srand(2);
  for (auto& v : hx) v = (float)rand() / (float)RAND_MAX * 2.f - 1.f;
  for (auto& v : hw) v = ((float)rand() / (float)RAND_MAX * 2.f - 1.f) * 0.1f;
  float *dx, *dw, *dy;
  CK_(hipMalloc(&dx, hx.size() * 4)); CK_(hipMalloc(&dw, hw.size() * 4)); CK_(hipMalloc(&dy, (size_t)N * H * W * K * 4));
  CK_(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK_(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  const int grid = N * ((H + TH - 1) / TH) * ((W + TW - 1) / TW) * (K / BN);
  conv3x3_patch_kernel<<<grid, 256>>>(dx, dw, dy, N, H, W, C, K);
  CK_(hipDeviceSynchronize());
  std::vector<float> hy((size_t)N * H * W * K);
  CK_(hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int t = 0; t < 3000; ++t) {
    const int n = t % N, oy = (t * 7) % H, ox = (t * 13) % W, k = (t * 29) % K;
    double ref = 0, mag = 0;
    for (int r = 0; r < 3; ++r)
      for (int s = 0; s < 3; ++s) {
        const int ih = oy + r - 1, iw = ox + s - 1;
        if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
        for (int c = 0; c < C; ++c) {
          const double a = hx[(((size_t)n * H + ih) * W + iw) * C + c], b = hw[((size_t)(r * 3 + s) * C + c) * K + k];
          ref += a * b; mag += fabs(a * b);
        }
      }
    worst = fmax(worst, fabs(hy[(((size_t)n * H + oy) * W + ox) * K + k] - ref) / fmax(mag, 1e-30));
  }
  hipEvent_t e0, e1;
  CK_(hipEventCreate(&e0)); CK_(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) conv3x3_patch_kernel<<<grid, 256>>>(dx, dw, dy, N, H, W, C, K);
  const int it = 30;
  CK_(hipEventRecord(e0));
  for (int i = 0; i < it; ++i) conv3x3_patch_kernel<<<grid, 256>>>(dx, dw, dy, N, H, W, C, K);
  CK_(hipEventRecord(e1));
  CK_(hipEventSynchronize(e1));
  float ms; CK_(hipEventElapsedTime(&ms, e0, e1));
  const double us = 1e3 * ms / it, flop = 2.0 * N * H * W * K * 9.0 * C;
  const double waste = (double)((H + TH - 1) / TH * TH) * ((W + TW - 1) / TW * TW) / ((double)H * W);
  printf("n%d %dx%dx%d 3x3 -> %d: %8.1f us  %6.1f TFLOP/s fp32-equivalent (algorithmic; the edge blocks compute %.1f %% more)  "
         "max |err| / sum|a b| = %.2e   grid %d\n", N, H, W, C, K, us, flop / us / 1e6, 100.0 * (waste - 1.0), worst, grid);
  CK_(hipFree(dx)); CK_(hipFree(dw)); CK_(hipFree(dy));
}

int main() {
  run(128, 56, 56, 64, 64);
  run(128, 28, 28, 128, 128);
  run(128, 14, 14, 256, 256);
  run(128, 64, 64, 64, 64);        // no edge waste: 64 = 4 x 16 = 8 x 8
  return 0;
}
