// One launch updates every trainable tensor of a model with the update rule of the Keras optimizer the
// reference builds in /root/reference/embedding_net/utils.py:143-153 (SGD / Adam / RMSprop / keras_radam RAdam,
// library defaults, only `lr` passed).  Multi-tensor: a device table of {w, g, slot1, slot2, n} descriptors and a
// device list of (tensor, first element) chunks, so the step is one kernel instead of ~10 framework launches per
// rule (ResNet18: 62 tensors, 11.3 M floats).  HBM-bound: 4 B/element x (w r+w, g r, slots r+w) = 12..28 B/element.
// The scalar coefficients (bias corrections, RAdam's rectifier) are computed on the host in double and passed in;
// the arithmetic per element is the rule's own order in fp32.
// l2x2 = 2*lambda of the tensor's kernel_regularizer=l2(lambda) (backbones.py:22-36; 0 for none): the regulariser's
// gradient 2*lambda*w is added to g here, which is what Keras' loss + regulariser differentiates to, so no separate
// 2*lambda*w tensors or accumulation launches exist.
#include "common.h"
#include "../../include/embnet.h"

namespace embnet {

struct OptTensor { float* w; const float* g; float* s1; float* s2; long n; float l2x2; int pad; };
static_assert(sizeof(OptTensor) == 48, "descriptor layout is part of the ABI (include/embnet.h)");

constexpr int OPT_CHUNK = 4096;          // elements per workgroup: 256 threads x 4 float4

struct OptCoef { float lr, b1, b2, eps, c1, c2; };

template <int RULE>
__device__ __forceinline__ void opt_update(float& w, float g, float& s1, float& s2, const OptCoef& k) {
  if (RULE == EMBNET_OPT_SGD) {
    w -= k.lr * g;
  } else if (RULE == EMBNET_OPT_RMSPROP) {                 // rms <- rho rms + (1-rho) g^2; w -= lr g / (sqrt(rms) + eps)
    s1 = k.b1 * s1 + (1.f - k.b1) * g * g;
    w -= k.lr * g / (sqrtf(s1) + k.eps);
  } else {
    s1 = k.b1 * s1 + (1.f - k.b1) * g;                     // m
    s2 = k.b2 * s2 + (1.f - k.b2) * g * g;                 // v
    if (RULE == EMBNET_OPT_ADAM) w -= k.c1 * s1 / (sqrtf(s2) + k.eps);                    // c1 = lr sqrt(1-b2^t)/(1-b1^t)
    else if (RULE == EMBNET_OPT_RADAM) w -= k.c1 * s1 / (sqrtf(s2 * k.c2) + k.eps);       // c1 = lr r_t/(1-b1^t), c2 = 1/(1-b2^t)
    else w -= k.c1 * s1;                                                                   // RAdam before rectification: c1 = lr/(1-b1^t)
  }
}

template <int RULE>
__global__ __launch_bounds__(256) void opt_step_kernel(const OptTensor* __restrict__ table, const int* __restrict__ chunks,
                                                       OptCoef k, const float* __restrict__ coef_dev) {
  if (coef_dev) {                                          // step-dependent scalars from device memory: a captured graph replays
    k.lr = coef_dev[0]; k.b1 = coef_dev[1]; k.b2 = coef_dev[2]; k.eps = coef_dev[3]; k.c1 = coef_dev[4]; k.c2 = coef_dev[5];
  }
  const int ti = chunks[2 * blockIdx.x];
  const long first = (long)chunks[2 * blockIdx.x + 1] * OPT_CHUNK;
  const OptTensor t = table[ti];
  if (!t.g) return;                                        // no gradient this step: Keras skips the variable
  constexpr bool S1 = RULE != EMBNET_OPT_SGD, S2 = RULE >= EMBNET_OPT_ADAM;
  const long end = min(first + OPT_CHUNK, t.n);
  const bool vec = (((uintptr_t)t.w | (uintptr_t)t.g | (uintptr_t)t.s1 | (uintptr_t)t.s2) & 15) == 0;
  if (vec && end - first == OPT_CHUNK) {
#pragma unroll
    for (int it = 0; it < OPT_CHUNK / 1024; ++it) {
      const long i = first + it * 1024 + threadIdx.x * 4;
      float4 w = *reinterpret_cast<const float4*>(t.w + i);
      float4 g = *reinterpret_cast<const float4*>(t.g + i);
      g.x = fmaf(t.l2x2, w.x, g.x); g.y = fmaf(t.l2x2, w.y, g.y); g.z = fmaf(t.l2x2, w.z, g.z); g.w = fmaf(t.l2x2, w.w, g.w);
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
      if (S1) a = *reinterpret_cast<const float4*>(t.s1 + i);
      if (S2) b = *reinterpret_cast<const float4*>(t.s2 + i);
      opt_update<RULE>(w.x, g.x, a.x, b.x, k); opt_update<RULE>(w.y, g.y, a.y, b.y, k);
      opt_update<RULE>(w.z, g.z, a.z, b.z, k); opt_update<RULE>(w.w, g.w, a.w, b.w, k);
      *reinterpret_cast<float4*>(t.w + i) = w;
      if (S1) *reinterpret_cast<float4*>(t.s1 + i) = a;
      if (S2) *reinterpret_cast<float4*>(t.s2 + i) = b;
    }
    return;
  }
  for (long i = first + threadIdx.x; i < end; i += 256) {
    float w = t.w[i], a = S1 ? t.s1[i] : 0.f, b = S2 ? t.s2[i] : 0.f;
    opt_update<RULE>(w, fmaf(t.l2x2, w, t.g[i]), a, b, k);
    t.w[i] = w;
    if (S1) t.s1[i] = a;
    if (S2) t.s2[i] = b;
  }
}

}  // namespace embnet

using namespace embnet;

extern "C" int embnet_optimizer_chunk_elems(void) { return OPT_CHUNK; }

extern "C" int embnet_optimizer_step(int rule, const void* table, int n_tensors, const int32_t* chunks, int n_chunks,
                                     float lr, float b1, float b2, float eps, float c1, float c2, const float* coef_dev,
                                     void* stream) {
  EMBNET_CHECK_ARG(table && chunks, "optimizer_step: null pointer");
  EMBNET_CHECK_ARG(n_tensors > 0 && n_chunks > 0, "optimizer_step: empty table");
  EMBNET_CHECK_ARG(rule >= EMBNET_OPT_SGD && rule <= EMBNET_OPT_RADAM_WARM, "optimizer_step: unknown rule %d", rule);
  const OptCoef k{lr, b1, b2, eps, c1, c2};
  const OptTensor* t = (const OptTensor*)table;
  hipStream_t s = (hipStream_t)stream;
  // algorithmic bytes: chunk count x chunk size x 4 B x (w read+write, g read, each slot read+write)
  EMBNET_TRACE("embnet::opt_step_kernel", TRACE_BYTES,
               4.0 * n_chunks * OPT_CHUNK * (rule == EMBNET_OPT_SGD ? 3 : (rule == EMBNET_OPT_RMSPROP ? 5 : 7)), s);
  switch (rule) {
    case EMBNET_OPT_SGD: opt_step_kernel<EMBNET_OPT_SGD><<<n_chunks, 256, 0, s>>>(t, chunks, k, coef_dev); break;
    case EMBNET_OPT_RMSPROP: opt_step_kernel<EMBNET_OPT_RMSPROP><<<n_chunks, 256, 0, s>>>(t, chunks, k, coef_dev); break;
    case EMBNET_OPT_ADAM: opt_step_kernel<EMBNET_OPT_ADAM><<<n_chunks, 256, 0, s>>>(t, chunks, k, coef_dev); break;
    case EMBNET_OPT_RADAM: opt_step_kernel<EMBNET_OPT_RADAM><<<n_chunks, 256, 0, s>>>(t, chunks, k, coef_dev); break;
    default: opt_step_kernel<EMBNET_OPT_RADAM_WARM><<<n_chunks, 256, 0, s>>>(t, chunks, k, coef_dev); break;
  }
  return check_launch("optimizer_step");
}
