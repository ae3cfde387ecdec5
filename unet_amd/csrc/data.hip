// Data path in front of the model, HBM-bound:
//   * window gather from device-resident recordings with the optional per-window z-score
//     (data.py:4586-4612 ECoGDataset.__getitem__, data.py:3120-3124): the reference slices and normalises one
//     window at a time on the host and ships each to the device; here the recordings live in HBM
//     (288 GB holds whole datasets) and a batch is ONE kernel: one workgroup per (window, channel) row,
//     row staged in LDS, mean / biased std in fp64, written once.
//   * NoiseAugmentor.forward (models.py:4897-4946) as ONE pass: the random draws stay with the caller
//     (torch's generator, so the stream of numbers is the one the reference would draw); the arithmetic
//       y = (x + g*sg + pink*sp) * keep_channel * keep_time + amp*sin(2*pi*f*t + phase)
//     that the reference spreads over ~12 elementwise passes reads x, g, pink once and writes y once.
#include "common.h"

namespace {

constexpr int THREADS = 256;

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum, result broadcast to every thread
__device__ __forceinline__ double block_sum_bcast(double v, double* red /* [THREADS/64] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_sum_d(v);
  __syncthreads();                 // protect red[] against the previous use
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < THREADS / 64; ++w) s += red[w];
  return s;
}

template <typename TI, typename TO>
__global__ __launch_bounds__(THREADS) void gather_windows_kernel(const TI* src, long long src_ld, const long long* starts,
                                                                 int c, int w, int zscore, float eps, TO* out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* buf = reinterpret_cast<float*>(smem_raw);     // w floats (only when zscore)
  __shared__ double red[THREADS / 64];
  const int ch = blockIdx.x % c;
  const long long b = blockIdx.x / c;
  const TI* p = src + (long long)ch * src_ld + starts[b];
  TO* o = out + ((long long)b * c + ch) * w;
  if (!zscore) {
    for (int i = threadIdx.x; i < w; i += THREADS) o[i] = from_f32<TO>(to_f32(p[i]));
    return;
  }
  double s = 0.0;
  for (int i = threadIdx.x; i < w; i += THREADS) { const float v = to_f32(p[i]); buf[i] = v; s += v; }
  const float m = (float)(block_sum_bcast(s, red) / w);
  double q = 0.0;
  for (int i = threadIdx.x; i < w; i += THREADS) { const float d = buf[i] - m; q += (double)d * d; }
  const float sd = (float)sqrt(block_sum_bcast(q, red) / w);     // numpy std: biased
  const float r = 1.f / (sd + eps);
  for (int i = threadIdx.x; i < w; i += THREADS) o[i] = from_f32<TO>((buf[i] - m) * r);
}

struct NoiseArgs {
  const float* scales;      // device: [0] gaussian_std * data_std, [1] pink_noise_std * data_std / (pink_std + 1e-6)
  const float* ch_u;        // [B*C] uniforms or null
  const float* t_u;         // [B*T] uniforms or null
  const float* amp;         // [B] line-noise amplitude (already scaled) or null
  const float* phase;       // [B]
  float ch_p, t_p;          // dropout probabilities: keep iff u > p
  float omega;              // (float)(2*pi*line_noise_hz)
  float t_end, step;        // torch.linspace(0, T/fs, T) in fp32: step = t_end/(T-1); first half start + step*i,
                            // second half t_end - step*(T-1-i) (ATen's symmetric evaluation)
};

template <typename T>
__global__ __launch_bounds__(THREADS) void noise_augment_kernel(const T* x, const T* gauss, const T* pink, T* y, int c, int t,
                                                                NoiseArgs a) {
  const long long row = blockIdx.y;              // b*c + ch
  const long long b = row / c;
  const int i0 = blockIdx.x * THREADS * 4 + threadIdx.x;   // 4 elements per thread, THREADS apart: coalesced lines
  const float sg = gauss ? a.scales[0] : 0.f, sp = pink ? a.scales[1] : 0.f;
  const float keep_c = a.ch_u ? (a.ch_u[row] > a.ch_p ? 1.f : 0.f) : 1.f;
  const float amp = a.amp ? a.amp[b] : 0.f, ph = a.amp ? a.phase[b] : 0.f;
  const long long base = row * t;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = i0 + k * THREADS;
    if (i >= t) break;
    float v = to_f32(x[base + i]);
    if (gauss) v += to_f32(gauss[base + i]) * sg;
    if (pink) v += to_f32(pink[base + i]) * sp;
    v *= keep_c;
    if (a.t_u) v *= a.t_u[b * t + i] > a.t_p ? 1.f : 0.f;
    if (a.amp) {   // same fp32 roundings as the reference's tensor expression (no fused multiply-add)
      const float ti = i < t / 2 ? __fmul_rn(a.step, (float)i) : __fsub_rn(a.t_end, __fmul_rn(a.step, (float)(t - 1 - i)));
      v = __fadd_rn(v, __fmul_rn(amp, sinf(__fadd_rn(__fmul_rn(a.omega, ti), ph))));
    }
    y[base + i] = from_f32<T>(v);
  }
}

}  // namespace

extern "C" int unet_gather_windows(int32_t in_dtype, int32_t out_dtype, const void* src, int64_t src_ld, int64_t src_len,
                                   const int64_t* starts_dev, int32_t n_windows, int32_t c, int32_t w, int32_t zscore,
                                   float eps, void* out, void* stream) {
  UNET_REQUIRE(src && starts_dev && out, "gather_windows: null pointer");
  UNET_REQUIRE(n_windows >= 0 && c > 0 && w > 0 && src_ld >= src_len && src_len >= w,
               "gather_windows: bad geometry (n=%d c=%d w=%d ld=%lld len=%lld)", n_windows, c, w, (long long)src_ld,
               (long long)src_len);
  UNET_REQUIRE((long long)n_windows * c < (1ll << 31), "gather_windows: %lld rows", (long long)n_windows * c);
  UNET_REQUIRE(!zscore || (size_t)w * sizeof(float) <= 150 * 1024, "gather_windows: window of %d samples exceeds the LDS row buffer", w);
  if (n_windows == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)((long long)n_windows * c));
  const size_t lds = zscore ? (size_t)w * sizeof(float) : 0;
#define UNET_GATHER_LAUNCH(TI, TO)                                                                                  \
  do {                                                                                                               \
    if (lds > 64 * 1024) {                                                                                           \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_windows_kernel<TI, TO>),             \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                    \
      UNET_REQUIRE(e == hipSuccess, "gather_windows: hipFuncSetAttribute failed: %s", hipGetErrorString(e));        \
    }                                                                                                                \
    hipLaunchKernelGGL((gather_windows_kernel<TI, TO>), grid, dim3(THREADS), lds, st, (const TI*)src, (long long)src_ld, \
                       (const long long*)starts_dev, c, w, zscore, eps, (TO*)out);                                   \
  } while (0)
  if (in_dtype == UNET_F32 && out_dtype == UNET_F32) UNET_GATHER_LAUNCH(float, float);
  else if (in_dtype == UNET_F32 && out_dtype == UNET_BF16) UNET_GATHER_LAUNCH(float, __bf16);
  else if (in_dtype == UNET_BF16 && out_dtype == UNET_BF16) UNET_GATHER_LAUNCH(__bf16, __bf16);
  else if (in_dtype == UNET_BF16 && out_dtype == UNET_F32) UNET_GATHER_LAUNCH(__bf16, float);
  else { unet_set_error("gather_windows: bad dtypes %d -> %d", in_dtype, out_dtype); return -1; }
#undef UNET_GATHER_LAUNCH
  UNET_LAUNCH_CHECK("gather_windows");
  return 0;
}

extern "C" int unet_noise_augment(int32_t dtype, const void* x, const void* gauss, const void* pink, const float* scales_dev,
                                  const float* channel_u, float channel_p, const float* time_u, float time_p,
                                  const float* line_amp, const float* line_phase, float line_hz, float sample_rate,
                                  int32_t n, int32_t c, int32_t t, void* y, void* stream) {
  UNET_REQUIRE(x && y, "noise_augment: null pointer");
  UNET_REQUIRE((!gauss && !pink) || scales_dev, "noise_augment: noise given without its scales");
  UNET_REQUIRE(!line_amp || line_phase, "noise_augment: line amplitude without phase");
  UNET_REQUIRE(n > 0 && c > 0 && t > 0, "noise_augment: bad shape n=%d c=%d t=%d", n, c, t);
  UNET_REQUIRE((long long)n * c <= 65535, "noise_augment: %lld rows exceed the grid's y extent", (long long)n * c);
  NoiseArgs a;
  a.scales = scales_dev; a.ch_u = channel_u; a.t_u = time_u; a.amp = line_amp; a.phase = line_phase;
  a.ch_p = channel_p; a.t_p = time_p;
  a.omega = (float)(6.283185307179586 * (double)line_hz);
  a.t_end = (float)((double)t / (double)sample_rate);
  a.step = t > 1 ? a.t_end / (float)(t - 1) : 0.f;
  const dim3 grid(cdiv(t, THREADS * 4), (unsigned)((long long)n * c));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == UNET_F32)
    hipLaunchKernelGGL(noise_augment_kernel<float>, grid, dim3(THREADS), 0, st, (const float*)x, (const float*)gauss,
                       (const float*)pink, (float*)y, c, t, a);
  else if (dtype == UNET_BF16)
    hipLaunchKernelGGL(noise_augment_kernel<__bf16>, grid, dim3(THREADS), 0, st, (const __bf16*)x, (const __bf16*)gauss,
                       (const __bf16*)pink, (__bf16*)y, c, t, a);
  else { unet_set_error("noise_augment: bad dtype %d", dtype); return -1; }
  UNET_LAUNCH_CHECK("noise_augment");
  return 0;
}
