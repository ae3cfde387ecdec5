// K17: the optimizer step of train.py:1553-1559 -- clip_grad_norm_(params, 5.0) followed by
// torch.optim.AdamW(betas=(0.7596, 0.9203), weight_decay=0).step() -- as two passes over the gradients and one over
// the parameters and moments:
//   1. unet_grad_sqnorm      per-block partial sums of g^2 over every gradient of the group        (reads g)
//   2. unet_clip_adamw_prepare   one block: ||g||, clip = min(1, max_norm / (||g|| + 1e-6))
//   3. unet_clip_adamw_apply     p, m, v updated from clip * g                                     (reads g, p, m, v; writes p, m, v)
// The clipped gradient is never written back (nothing reads it after the step: train.py:1560 drops the gradients), which
// is the pass torch's clip_grad_norm_ spends on `g *= clip`.  Everything stays on the device: no host synchronisation,
// capturable in a HIP graph.
//
// Layout: the moments of a group live in two flat fp32 buffers; tensor t owns elements [4*start4[t], 4*start4[t] +
// numel[t]) (starts padded to 4 elements so that float4 accesses of p / g / m / v share their alignment).  Parameter
// pointers, starts and sizes are a device-resident table built once; gradient pointers change from step to step
// (autograd allocates them) and travel in the kernel arguments.
#include "common.h"

namespace {
constexpr int OPT_THREADS = 256;
constexpr int OPT_CH4 = 2048;          // float4 slots per block (8192 elements)

struct GradPtrs { const float* g[UNET_OPT_MAX_TENSORS]; };

// tensor that holds flat float4 slot f4: largest t with start4[t] <= f4 (start4[n] = end)
__device__ __forceinline__ int find_tensor(const uint32_t* start4, int n, uint32_t f4) {
  int lo = 0, hi = n;                 // invariant: start4[lo] <= f4 < start4[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (start4[mid] <= f4) lo = mid; else hi = mid;
  }
  return lo;
}

// Also advances the step counter of every tensor that has a gradient (torch keeps one counter per parameter: a
// parameter without a gradient in some step falls behind, and so do its bias corrections).
__global__ __launch_bounds__(OPT_THREADS) void grad_sqnorm_kernel(const GradPtrs gp, const uint32_t* start4,
                                                                   const uint32_t* numel, int n, float* partial,
                                                                   float* step_t) {
  static_assert(UNET_OPT_MAX_TENSORS <= OPT_THREADS, "one thread per tensor advances its step counter");
  if (blockIdx.x == 0 && (int)threadIdx.x < n && gp.g[threadIdx.x]) step_t[threadIdx.x] += 1.f;
  const uint32_t base = start4[0];
  const uint32_t lo = base + blockIdx.x * OPT_CH4;
  const uint32_t end = start4[n];
  const uint32_t hi = lo + OPT_CH4 < end ? lo + OPT_CH4 : end;
  float s = 0.f;
  if (lo < hi) {
    int t = find_tensor(start4, n, lo);
    const float* g0 = gp.g[t];
    // the usual case -- the block's whole slice inside one tensor, whole float4s: all eight loads of a thread in flight
    if (hi <= start4[t + 1] && (hi - start4[t]) * 4 <= numel[t] && !(reinterpret_cast<uintptr_t>(g0) & 15)) {
      if (g0) {
        const f32x4* src = reinterpret_cast<const f32x4*>(g0) + (lo - start4[t]);
        f32x4 v[OPT_CH4 / OPT_THREADS];
#pragma unroll
        for (int u = 0; u < OPT_CH4 / OPT_THREADS; ++u) {
          const uint32_t k = threadIdx.x + u * OPT_THREADS;
          v[u] = lo + k < hi ? src[k] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < OPT_CH4 / OPT_THREADS; ++u) s += v[u][0] * v[u][0] + v[u][1] * v[u][1] + v[u][2] * v[u][2] + v[u][3] * v[u][3];
      }
    } else
    for (uint32_t f4 = lo + threadIdx.x; f4 < hi; f4 += OPT_THREADS) {
      while (f4 >= start4[t + 1]) ++t;
      const float* g = gp.g[t];
      if (!g) continue;
      const uint32_t i = (f4 - start4[t]) * 4, ne = numel[t];
      if (i + 4 <= ne && !(reinterpret_cast<uintptr_t>(g) & 15)) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(g + i);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
      } else {
        for (uint32_t e = i; e < ne && e < i + 4; ++e) s += g[e] * g[e];
      }
    }
  }
  s = wave_sum(s);
  __shared__ float red[OPT_THREADS / 64];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f;
    for (int w = 0; w < OPT_THREADS / 64; ++w) a += red[w];
    partial[blockIdx.x] = a;
  }
}

// coef[0] = clip, coef[3] = ||g||, coef[4] = lr
__global__ __launch_bounds__(OPT_THREADS) void clip_adamw_prepare_kernel(const float* partial, int blocks, float max_norm,
                                                                          const float* lr, float* coef) {
  __shared__ double red[OPT_THREADS];
  double a = 0.0;
  for (int i = threadIdx.x; i < blocks; i += OPT_THREADS) a += (double)partial[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int s = OPT_THREADS / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]);
    float clip = 1.f;
    if (max_norm > 0.f) {                         // torch.nn.utils.clip_grad_norm_: max_norm / (total + 1e-6), at most 1
      clip = max_norm / (norm + 1e-6f);
      clip = clip < 1.f ? clip : 1.f;
    }
    coef[0] = clip;
    coef[3] = norm;
    coef[4] = *lr;
  }
}

__device__ __forceinline__ void adamw_elem(float& p, float g, float& m, float& v, float clip, float step_size,
                                           float rsqrt_bc2, float decay, float beta1, float beta2, float eps) {
  g *= clip;
  p *= decay;                                       // p *= 1 - lr * weight_decay
  m = beta1 * m + (1.f - beta1) * g;                // torch: exp_avg.lerp_(grad, 1 - beta1)
  v = beta2 * v + (1.f - beta2) * g * g;
  const float denom = sqrtf(v) * rsqrt_bc2 + eps;
  p -= step_size * (m / denom);
}

__global__ __launch_bounds__(OPT_THREADS) void clip_adamw_apply_kernel(const GradPtrs gp, float* const* p_ptrs,
                                                                        const uint32_t* start4, const uint32_t* numel,
                                                                        int n, float* m_flat, float* v_flat,
                                                                        const float* coef, const float* step_t,
                                                                        float beta1, float beta2, float eps,
                                                                        float weight_decay) {
  const uint32_t base = start4[0];
  const uint32_t lo = base + blockIdx.x * OPT_CH4;
  const uint32_t end = start4[n];
  const uint32_t hi = lo + OPT_CH4 < end ? lo + OPT_CH4 : end;
  if (lo >= hi) return;
  const float clip = coef[0], lr = coef[4];
  const float decay = 1.f - lr * weight_decay;
  int t = find_tensor(start4, n, lo), t_bc = -1;
  float step_size = 0.f, rsqrt_bc2 = 0.f;
  for (uint32_t f4 = lo + threadIdx.x; f4 < hi; f4 += OPT_THREADS) {
    while (f4 >= start4[t + 1]) ++t;
    const float* g = gp.g[t];
    if (!g) continue;                               // no gradient this step: torch skips the parameter too
    if (t != t_bc) {                                // bias corrections of this tensor's own step count
      const double st = (double)step_t[t];
      step_size = (float)((double)lr / (1.0 - pow((double)beta1, st)));
      rsqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)beta2, st)));
      t_bc = t;
    }
    float* p = p_ptrs[t];
    const uint32_t i = (f4 - start4[t]) * 4, ne = numel[t];
    float* m = m_flat + (size_t)f4 * 4;
    float* v = v_flat + (size_t)f4 * 4;
    if (i + 4 <= ne && !((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(p)) & 15)) {
      const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i);
      f32x4 pv = *reinterpret_cast<f32x4*>(p + i);
      f32x4 mv = *reinterpret_cast<f32x4*>(m);
      f32x4 vv = *reinterpret_cast<f32x4*>(v);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pe = pv[e], me = mv[e], ve = vv[e];
        adamw_elem(pe, gv[e], me, ve, clip, step_size, rsqrt_bc2, decay, beta1, beta2, eps);
        pv[e] = pe; mv[e] = me; vv[e] = ve;
      }
      *reinterpret_cast<f32x4*>(p + i) = pv;
      *reinterpret_cast<f32x4*>(m) = mv;
      *reinterpret_cast<f32x4*>(v) = vv;
    } else {
      for (uint32_t e = i; e < ne && e < i + 4; ++e)
        adamw_elem(p[e], g[e], m[e - i], v[e - i], clip, step_size, rsqrt_bc2, decay, beta1, beta2, eps);
    }
  }
}

int check_table(const char* what, const float* const* grads_host, int32_t n, const void* a, const void* b) {
  UNET_REQUIRE(grads_host && a && b, "%s: null pointer", what);
  UNET_REQUIRE(n >= 1 && n <= UNET_OPT_MAX_TENSORS, "%s: 1..%d tensors per call, got %d", what, UNET_OPT_MAX_TENSORS, n);
  return 0;
}
}  // namespace

extern "C" int32_t unet_opt_blocks(int64_t flat_elements) {
  return (int32_t)((flat_elements / 4 + OPT_CH4 - 1) / OPT_CH4);
}

extern "C" int unet_grad_sqnorm(const float* const* grads_host, int32_t n_tensors, const uint32_t* start4,
                                const uint32_t* numel, int64_t flat_elements, float* partial, float* step_t,
                                void* stream) {
  if (check_table("grad_sqnorm", grads_host, n_tensors, start4, numel)) return -1;
  UNET_REQUIRE(partial && step_t && flat_elements > 0 && flat_elements % 4 == 0, "grad_sqnorm: null pointer or bad flat length");
  GradPtrs gp;
  for (int i = 0; i < UNET_OPT_MAX_TENSORS; ++i) gp.g[i] = i < n_tensors ? grads_host[i] : nullptr;
  const int blocks = unet_opt_blocks(flat_elements);
  hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(blocks), dim3(OPT_THREADS), 0, static_cast<hipStream_t>(stream), gp, start4,
                     numel, n_tensors, partial, step_t);
  UNET_LAUNCH_CHECK("grad_sqnorm");
  return 0;
}

extern "C" int unet_clip_adamw_prepare(const float* partial, int32_t blocks, float max_norm, const float* lr, float* coef,
                                       void* stream) {
  UNET_REQUIRE(partial && lr && coef && blocks > 0, "clip_adamw_prepare: null pointer or no partials");
  hipLaunchKernelGGL(clip_adamw_prepare_kernel, dim3(1), dim3(OPT_THREADS), 0, static_cast<hipStream_t>(stream), partial,
                     blocks, max_norm, lr, coef);
  UNET_LAUNCH_CHECK("clip_adamw_prepare");
  return 0;
}

extern "C" int unet_clip_adamw_apply(const float* const* grads_host, int32_t n_tensors, float* const* p_ptrs,
                                     const uint32_t* start4, const uint32_t* numel, int64_t flat_elements, float* m_flat,
                                     float* v_flat, const float* coef, const float* step_t, float beta1, float beta2,
                                     float eps, float weight_decay, void* stream) {
  if (check_table("clip_adamw_apply", grads_host, n_tensors, start4, numel)) return -1;
  UNET_REQUIRE(p_ptrs && m_flat && v_flat && coef && step_t && flat_elements > 0 && flat_elements % 4 == 0,
               "clip_adamw_apply: null pointer or bad flat length");
  UNET_REQUIRE(aligned16(m_flat) && aligned16(v_flat), "clip_adamw_apply: moment buffers must be 16-byte aligned");
  GradPtrs gp;
  for (int i = 0; i < UNET_OPT_MAX_TENSORS; ++i) gp.g[i] = i < n_tensors ? grads_host[i] : nullptr;
  const int blocks = unet_opt_blocks(flat_elements);
  hipLaunchKernelGGL(clip_adamw_apply_kernel, dim3(blocks), dim3(OPT_THREADS), 0, static_cast<hipStream_t>(stream), gp,
                     p_ptrs, start4, numel, n_tensors, m_flat, v_flat, coef, step_t, beta1, beta2, eps, weight_decay);
  UNET_LAUNCH_CHECK("clip_adamw_apply");
  return 0;
}
