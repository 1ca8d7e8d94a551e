// uint8 images -> the float32 NHWC batch the backbones take: the last step of the input pipeline.
// The reference decodes with cv2 (BGR, uint8), resizes, and divides by 255 on the host for every batch
// (/root/reference/embedding_net/datagenerators.py:145-156; utils.py:13-21), then Keras copies float32 to the device: 4 bytes
// per value over PCIe.  Here the decoded images stay uint8 — either a whole dataset resident in HBM (DeviceImageStore) or a
// pinned staging batch (BatchPrefetcher) — and this kernel gathers the batch's images by index, converts and divides:
// 1 byte per value crosses PCIe (or none), the division is the reference's float32 `x / 255.` bit for bit.
#include "common.h"
#include "../../include/embnet.h"

namespace embnet {

// dst[i][p][j] = j < c_in ? src[idx(i)][p][j] / 255 : 0   (idx = index[i] or i); elems = pixels * c_in bytes per image
template <bool VEC>
__global__ __launch_bounds__(256) void u8_to_f32_same_kernel(const unsigned char* __restrict__ src, const int* __restrict__ index,
                                                             long elems, float* __restrict__ dst, float denom) {
  const long img = blockIdx.y;
  const unsigned char* s = src + (index ? (long)index[img] : img) * elems;
  float* d = dst + img * elems;
  if (VEC) {                         // elems % 4 == 0 and 4-byte aligned images: one dword in, one float4 out
    const long n4 = elems >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
      const unsigned v = reinterpret_cast<const unsigned*>(s)[i];
      reinterpret_cast<float4*>(d)[i] = make_float4((float)(v & 255u) / denom, (float)((v >> 8) & 255u) / denom,
                                                    (float)((v >> 16) & 255u) / denom, (float)(v >> 24) / denom);
    }
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < elems; i += (long)gridDim.x * 256) d[i] = (float)s[i] / denom;
  }
}

// channel padding (c_out > c_in, zeros): one thread per pixel
__global__ __launch_bounds__(256) void u8_to_f32_pad_kernel(const unsigned char* __restrict__ src, const int* __restrict__ index,
                                                            long pixels, int c_in, int c_out, float* __restrict__ dst, float denom) {
  const long img = blockIdx.y;
  const unsigned char* s = src + (index ? (long)index[img] : img) * pixels * c_in;
  float* d = dst + img * pixels * c_out;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (long)gridDim.x * 256)
    for (int j = 0; j < c_out; ++j) d[p * c_out + j] = j < c_in ? (float)s[p * c_in + j] / denom : 0.f;
}

}  // namespace embnet

using namespace embnet;

extern "C" int embnet_u8_to_f32(const void* src, const int32_t* index, int n, long pixels, int c_in, int c_out, float denom,
                                float* dst, void* stream) {
  EMBNET_CHECK_ARG(src && dst, "u8_to_f32: null pointer");
  EMBNET_CHECK_ARG(n > 0 && n <= 65535 && pixels > 0 && c_in > 0 && c_out >= c_in && c_out <= 16 && denom > 0.f,
                   "u8_to_f32: n=%d pixels=%ld c_in=%d c_out=%d denom=%f (n <= 65535, c_in <= c_out <= 16)", n, pixels, c_in, c_out, denom);
  hipStream_t st = (hipStream_t)stream;
  EMBNET_TRACE("embnet::u8_to_f32_kernel", TRACE_BYTES, (double)n * pixels * (c_in + 4.0 * c_out), st);
  if (c_in == c_out) {
    const long elems = pixels * c_in;
    const bool vec = (elems & 3) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
    const long work = vec ? elems / 4 : elems;
    const dim3 grid((unsigned)(work / 256 + 1 > 1024 ? 1024 : work / 256 + 1), (unsigned)n);
    if (vec) u8_to_f32_same_kernel<true><<<grid, 256, 0, st>>>((const unsigned char*)src, index, elems, dst, denom);
    else u8_to_f32_same_kernel<false><<<grid, 256, 0, st>>>((const unsigned char*)src, index, elems, dst, denom);
  } else {
    const dim3 grid((unsigned)(pixels / 256 + 1 > 1024 ? 1024 : pixels / 256 + 1), (unsigned)n);
    u8_to_f32_pad_kernel<<<grid, 256, 0, st>>>((const unsigned char*)src, index, pixels, c_in, c_out, dst, denom);
  }
  return check_launch("u8_to_f32");
}
