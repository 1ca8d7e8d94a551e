// Shared host/device helpers for libembnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>

namespace embnet {

// ---- error reporting (embnet_last_error) ---------------------------------
char* last_error_buf();                      // thread-local, 512 bytes
int fail(int code, const char* fmt, ...);    // formats into last_error_buf, returns code

enum : int {
  EMBNET_OK = 0,
  EMBNET_EINVAL = -1,      // bad argument (null pointer, size <= 0, unsupported shape)
  EMBNET_ELAUNCH = -2,     // hipLaunch / hipGetLastError failure
  EMBNET_EWORKSPACE = -3,  // caller's workspace too small
};

#define EMBNET_CHECK_ARG(cond, ...) \
  do { if (!(cond)) return ::embnet::fail(::embnet::EMBNET_EINVAL, __VA_ARGS__); } while (0)

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(EMBNET_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return EMBNET_OK;
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- optional per-kernel timing (embnet_trace_*, include/embnet.h) --------
// A TraceScope around ONE kernel launch records a HIP event before and after it on the launch stream together with
// the kernel's name and its ALGORITHMIC work (FLOP for the MFMA-bound kernels, bytes for the HBM-bound ones).  Off
// (the default) it costs one relaxed load per launch.  bench.py turns it on for a sample of the timed steps.
enum : int { TRACE_FLOP = 0, TRACE_BYTES = 1 };
bool trace_on();
struct TraceScope {
  int idx;
  hipStream_t s;
  TraceScope(const char* name, int unit, double work, void* stream, double bytes = -1.0);
  ~TraceScope();
};
#define EMBNET_CAT2(a, b) a##b
#define EMBNET_CAT(a, b) EMBNET_CAT2(a, b)
#define EMBNET_TRACE(name, unit, work, stream) \
  ::embnet::TraceScope EMBNET_CAT(trace_scope_, __LINE__)(name, unit, (double)(work), (void*)(stream))
// MFMA-bound kernel: FLOP plus the bytes its operands and result occupy (what it must move once)
#define EMBNET_TRACE_FLOP(name, flop, bytes, stream) \
  ::embnet::TraceScope EMBNET_CAT(trace_scope_, __LINE__)(name, ::embnet::TRACE_FLOP, (double)(flop), (void*)(stream), (double)(bytes))

// tuning knobs: read once per process (each call site keeps its own `static const`)
inline long env_long(const char* name, long dflt) { const char* e = getenv(name); return e ? atol(e) : dflt; }
// The format of the pre-split planes of this process (gemm_engine.h): two fp16 pieces + a power-of-two scale per tensor, three matrix
// products per fp32 product (default), or EMBNET_PLANES_F16=0: three bf16 pieces, six products (rounds 2-5).  Read once: every
// producer and consumer of planes in a process uses the same format.
inline bool planes_f16() { static const bool on = env_long("EMBNET_PLANES_F16", 1) != 0; return on; }

// ---- device helpers ------------------------------------------------------
constexpr int WAVE = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Range emission (conv.hip "Ranges"): the workgroup's max |value| joins a range word — the bit pattern of a non-negative float
// under an unsigned maximum, independent of the order, so reproducible.  One atomic per workgroup (256 threads, every thread
// calls; amax >= 0, threads without an element pass 0).  A kernel of thousands of workgroups must NOT aim them all at one word
// (same-address atomics serialise at the memory side: a BatchNorm-backward pass ran 4x longer): the BatchNorm passes spread
// theirs over the RANGE_PARTIALS words behind the slot's first (word 1 + blockIdx % RANGE_PARTIALS, zeroed by the finalize kernel
// in front) and a one-workgroup kernel folds them into word 0, which is what the consumers read.
constexpr int RANGE_PARTIALS = 1024;
__device__ __forceinline__ void range_emit_block(uint32_t* word, float amax) {
  __shared__ float range_wm[4];
  amax = wave_max(amax);
  if ((threadIdx.x & 63) == 0) range_wm[(threadIdx.x >> 6) & 3] = amax;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicMax(word, __float_as_uint(fmaxf(fmaxf(range_wm[0], range_wm[1]), fmaxf(range_wm[2], range_wm[3]))));
}
__device__ __forceinline__ float amax4(float m, float4 v) {
  return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}

// Activation fused behind the BN affine: 0 none, 1 ReLU, 2 swish (z * sigmoid(z)).  Backward recomputes
// z = x*scale + shift from the saved input, so no activation tensor or mask is stored.
__device__ __forceinline__ float act_apply(int act, float z) {
  return act == 1 ? fmaxf(z, 0.f) : (act == 2 ? z / (1.f + __expf(-z)) : z);
}
__device__ __forceinline__ float act_grad(int act, float z, float dy) {
  if (act == 1) return z <= 0.f ? 0.f : dy;
  if (act == 2) { const float sg = 1.f / (1.f + __expf(-z)); return dy * (sg + z * sg * (1.f - sg)); }
  return dy;
}

// Counter-based RNG: one 64-bit mix (splitmix64 finaliser) of (seed, a, b).
// Used for the two random negative-selection rules and for dropout masks.
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint32_t rng_u32(uint64_t seed, uint64_t a, uint64_t b) {
  return (uint32_t)(mix64(mix64(seed ^ (a * 0xD6E8FEB86659FD93ull)) + b) >> 32);
}

}  // namespace embnet
