// Loss / evaluation reductions of the hot path (HBM-bound, forward-only except the L1 gradient):
//   * pair moments over the cropped window  -> pooled R^2 / Pearson / MSE / MAE (train.py:931-952,
//     1098-1135), per-row Pearson (models.py:4262-4287), explained variance (models.py:4331-4362)
//   * L1-on-crop gradient (train.py:1481-1486)
//   * per-row z-score of NCL tensors (train.py:476-495 per_channel_normalize; data.py:3121 window z-score)
//   * Welch PSD (models.py:4512-4550; scipy.signal.welch as used by utils/dsp_metrics.py:163),
//     dB error / bias and band-power reductions (models.py:4596-4629, 4798-4816; dsp_metrics.py:168-177)
//   * overlap-average reconstruction of sliding-window predictions (phase_five/sliding_window.py:238-251)
// All inputs are NCL rows (contiguous time axis): one workgroup (or one wave) per (window, channel) row,
// lanes stride over time so every load is a full coalesced line.  Sums are carried in fp64 and written
// per row; cross-row totals are added by a second, deterministic kernel (no float atomics).
#include "common.h"

namespace {

constexpr int THREADS = 256;

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum of K doubles per thread; result valid in thread 0
template <int K>
__device__ __forceinline__ void block_sum_d(double (&v)[K], double* red /* [K][THREADS/64] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    v[k] = wave_sum_d(v[k]);
    if (lane == 0) red[k * (THREADS / 64) + wave] = v[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      double s = 0.0;
      for (int w = 0; w < THREADS / 64; ++w) s += red[k * (THREADS / 64) + w];
      v[k] = s;
    }
  }
}

template <typename T> struct Vec4 {};
template <> struct Vec4<float> {
  __device__ static __forceinline__ void ld(const float* p, float (&f)[4]) {
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
    f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
  }
  static constexpr int ALIGN = 16;
};
template <> struct Vec4<__bf16> {
  __device__ static __forceinline__ void ld(const __bf16* p, float (&f)[4]) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    f[0] = (float)v[0]; f[1] = (float)v[1]; f[2] = (float)v[2]; f[3] = (float)v[3];
  }
  static constexpr int ALIGN = 8;
};

// ---------------------------------------------------------------------------------------------
// pair moments: out[row] = {n, St, Stt, Sp, Spp, Spt, S(p-t)^2, S|p-t|} over t in [lo, hi)
// (the order of train.py:1076-1085's stats tensor)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(THREADS) void pair_moments_kernel(const T* pred, const T* target, long long ld_p,
                                                                long long ld_t, int lo, int hi, double* out) {
  __shared__ double red[7 * (THREADS / 64)];
  const long long row = blockIdx.x;
  const T* p = pred + row * ld_p;
  const T* t = target + row * ld_t;
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  auto add = [&](float pf, float tf) {
    const float d = pf - tf;                 // fp32 difference, as the reference's fp32 tensors
    acc[0] += tf; acc[1] += (double)tf * tf; acc[2] += pf; acc[3] += (double)pf * pf;
    acc[4] += (double)pf * tf; acc[5] += (double)d * d; acc[6] += fabsf(d);
  };
  const bool vec = (reinterpret_cast<uintptr_t>(p + lo) % Vec4<T>::ALIGN) == 0 &&
                   (reinterpret_cast<uintptr_t>(t + lo) % Vec4<T>::ALIGN) == 0;
  const int n = hi - lo;
  int done = 0;
  if (vec) {
    const int nv = n >> 2;
    for (int i = threadIdx.x; i < nv; i += THREADS) {
      float pf[4], tf[4];
      Vec4<T>::ld(p + lo + 4 * i, pf);
      Vec4<T>::ld(t + lo + 4 * i, tf);
#pragma unroll
      for (int k = 0; k < 4; ++k) add(pf[k], tf[k]);
    }
    done = nv << 2;
  }
  for (int i = done + threadIdx.x; i < n; i += THREADS) add(to_f32(p[lo + i]), to_f32(t[lo + i]));
  block_sum_d<7>(acc, red);
  if (threadIdx.x == 0) {
    double* o = out + row * 8;
    o[0] = (double)n;
#pragma unroll
    for (int k = 0; k < 7; ++k) o[1 + k] = acc[k];
  }
}

// out[k] = sum_rows in[row][k]   (k < width <= 64)
__global__ __launch_bounds__(THREADS) void sum_rows_d_kernel(const double* in, long long rows, int width, double* out) {
  __shared__ double red[THREADS];
  const int k = threadIdx.x % width, seg = threadIdx.x / width, nseg = THREADS / width;
  double a = 0.0;
  if (seg < nseg)
    for (long long r = seg; r < rows; r += nseg) a += in[r * width + k];
  red[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x < width) {
    double s = 0.0;
    for (int j = 0; j < nseg; ++j) s += red[j * width + threadIdx.x];
    out[threadIdx.x] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// d/dpred of  weight * mean_{crop} |pred - target|  :  dpred = gscale[0]*sign(pred-target) inside
// [lo,hi), 0 outside (the crop's backward is a zero-padded scatter).  gscale is a device scalar so the
// call needs no host synchronisation.
// ---------------------------------------------------------------------------------------------
template <typename T, typename TG>
__global__ __launch_bounds__(THREADS) void l1_crop_bwd_kernel(const T* pred, const T* target, long long ld_p,
                                                               long long ld_t, int t_len, int lo, int hi,
                                                               const float* gscale, float mul, TG* dpred) {
  const long long row = blockIdx.y;
  const int i = blockIdx.x * THREADS + threadIdx.x;
  if (i >= t_len) return;
  float g = 0.f;
  if (i >= lo && i < hi) {
    const float d = to_f32(pred[row * ld_p + i]) - to_f32(target[row * ld_t + i]);
    const float s = gscale[0] * mul;
    g = d > 0.f ? s : (d < 0.f ? -s : 0.f);
  }
  dpred[row * (long long)t_len + i] = from_f32<TG>(g);
}

// ---------------------------------------------------------------------------------------------
// per-row z-score of an NCL tensor in one pass over HBM (row staged in LDS):
//   mode 0: (x-mean)/(std+eps)       models.py:3235 (unbiased), data.py:3121 (biased)
//   mode 1: (x-mean)/max(std,eps)    train.py:476-495 per_channel_normalize
// ---------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ __launch_bounds__(THREADS) void row_zscore_kernel(const TI* x, TO* y, int t, float eps, int mode, int unbiased) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* buf = reinterpret_cast<float*>(smem_raw);     // t floats
  __shared__ double red[2 * (THREADS / 64)];
  __shared__ float stat[2];
  const long long row = blockIdx.x;
  const TI* p = x + row * (long long)t;
  double s[1] = {0.0};
  for (int i = threadIdx.x; i < t; i += THREADS) { const float v = to_f32(p[i]); buf[i] = v; s[0] += v; }
  block_sum_d<1>(s, red);
  if (threadIdx.x == 0) stat[0] = (float)(s[0] / t);
  __syncthreads();
  const float m = stat[0];
  double q[1] = {0.0};
  for (int i = threadIdx.x; i < t; i += THREADS) { const float d = buf[i] - m; q[0] += (double)d * d; }
  block_sum_d<1>(q, red + (THREADS / 64));
  if (threadIdx.x == 0) {
    const double den = unbiased ? (t > 1 ? t - 1 : 1) : t;
    const float sd = (float)sqrt(q[0] / den);
    stat[1] = mode == 0 ? 1.f / (sd + eps) : 1.f / fmaxf(sd, eps);
  }
  __syncthreads();
  const float r = stat[1];
  TO* o = y + row * (long long)t;
  for (int i = threadIdx.x; i < t; i += THREADS) o[i] = from_f32<TO>((buf[i] - m) * r);
}

// ---------------------------------------------------------------------------------------------
// Welch PSD: one workgroup of nperseg/2 threads per signal row.  Segments x[lo + s*hop + i], i<nperseg,
// are windowed (periodic Hann), transformed by an in-LDS radix-2 FFT, |X_k|^2 is averaged over segments
// and scaled: psd[k] = scale_k * mean_s |X_s[k]|^2, scale_k = edge_scale for k in {0, nperseg/2}, else scale.
//   models.py:4512-4550 : no detrend, scale = edge_scale = 2/(fs*S2)
//   scipy.signal.welch   : detrend='constant', scale = 2/(fs*S2), edge_scale = 1/(fs*S2)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void welch_psd_kernel(const T* x, long long ld, int lo, int n_seg, int nperseg, int log2n, int detrend,
                                 float scale, float edge_scale, float* psd) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float2* buf = reinterpret_cast<float2*>(smem_raw);           // nperseg complex
  float2* tw = buf + nperseg;                                    // nperseg/2 twiddles exp(-2 pi i k / n)
  float* win = reinterpret_cast<float*>(tw + nperseg / 2);       // nperseg window values
  float* redf = win + nperseg;                                   // up to 16 partial sums
  const int tid = threadIdx.x, half = nperseg >> 1, hop = nperseg >> 1;
  const long long row = blockIdx.x;
  const T* p = x + row * ld + lo;
  {
    float s, c;
    sincospif(-2.0f * (float)tid / (float)nperseg, &s, &c);
    tw[tid] = make_float2(c, s);
    win[tid] = 0.5f - 0.5f * cospif(2.0f * (float)tid / (float)nperseg);
    win[tid + half] = 0.5f - 0.5f * cospif(2.0f * (float)(tid + half) / (float)nperseg);
  }
  float acc = 0.f, acc_ny = 0.f;   // bin tid; thread 0 also carries the Nyquist bin
  for (int sidx = 0; sidx < n_seg; ++sidx) {
    __syncthreads();
    const T* sp = p + (long long)sidx * hop;
    float v0 = to_f32(sp[tid]), v1 = to_f32(sp[tid + half]);
    if (detrend) {
      float s = wave_sum(v0 + v1);
      if ((tid & 63) == 0) redf[tid >> 6] = s;
      __syncthreads();
      float tot = 0.f;
      for (int w = 0; w < (half + 63) / 64; ++w) tot += redf[w];
      const float mean = tot / (float)nperseg;
      v0 -= mean; v1 -= mean;
      __syncthreads();
    }
    // bit-reversed placement, then decimation-in-time butterflies
    const unsigned r0 = __brev((unsigned)tid) >> (32 - log2n);
    const unsigned r1 = __brev((unsigned)(tid + half)) >> (32 - log2n);
    buf[r0] = make_float2(v0 * win[tid], 0.f);
    buf[r1] = make_float2(v1 * win[tid + half], 0.f);
    for (int st = 0; st < log2n; ++st) {
      __syncthreads();
      const int m = 1 << st;                       // half-size of the butterfly group
      const int j = tid & (m - 1);
      const int i0 = ((tid >> st) << (st + 1)) + j;
      const float2 w = tw[j << (log2n - 1 - st)];
      const float2 a = buf[i0], b = buf[i0 + m];
      const float2 wb = make_float2(w.x * b.x - w.y * b.y, w.x * b.y + w.y * b.x);
      buf[i0] = make_float2(a.x + wb.x, a.y + wb.y);
      buf[i0 + m] = make_float2(a.x - wb.x, a.y - wb.y);
    }
    __syncthreads();
    const float2 z = buf[tid];
    acc += z.x * z.x + z.y * z.y;
    if (tid == 0) { const float2 zn = buf[half]; acc_ny += zn.x * zn.x + zn.y * zn.y; }
  }
  float* o = psd + row * (long long)(half + 1);
  const float inv = 1.f / (float)n_seg;
  o[tid] = acc * inv * (tid == 0 ? edge_scale : scale);
  if (tid == 0) o[half] = acc_ny * inv * edge_scale;
}

// one wave per row: out[row] = { sum_k |dB(p)-dB(t)|, sum_k (dB(p)-dB(t)) } over bins [k_lo, k_hi]
// (models.py:4611-4627, 4686-4695); normalize: divide each PSD by its clamped band total first.
__global__ __launch_bounds__(THREADS) void psd_db_stats_kernel(const float* pp, const float* pt, long long rows, int nfreq,
                                                                int k_lo, int k_hi, int normalize, double* out) {
  const long long row = (long long)blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* a = pp + row * nfreq;
  const float* b = pt + row * nfreq;
  float na = 1.f, nb = 1.f;
  if (normalize) {
    float sa = 0.f, sb = 0.f;
    for (int k = k_lo + lane; k <= k_hi; k += 64) { sa += a[k]; sb += b[k]; }
    na = fmaxf(wave_sum(sa), 1e-12f); nb = fmaxf(wave_sum(sb), 1e-12f);
  }
  double s_abs = 0.0, s_sgn = 0.0;
  for (int k = k_lo + lane; k <= k_hi; k += 64) {
    const float da = 10.f * log10f(a[k] / na + 1e-12f), db = 10.f * log10f(b[k] / nb + 1e-12f);
    const float d = da - db;
    s_abs += fabsf(d); s_sgn += d;
  }
  s_abs = wave_sum_d(s_abs); s_sgn = wave_sum_d(s_sgn);
  if (lane == 0) { out[row * 2] = s_abs; out[row * 2 + 1] = s_sgn; }
}

// one wave per row: band power of each of nb bands over bins [blo[b], bhi[b]] (inclusive; empty if hi<lo)
//   mode 0: plain sum (models.py:4811-4812)    mode 1: trapezoid * df (np.trapz, dsp_metrics.py:172)
struct Bands { int lo[16]; int hi[16]; };
__global__ __launch_bounds__(THREADS) void band_power_kernel(const float* psd, long long rows, int nfreq, Bands bands,
                                                              int nb, int mode, float df, float* out) {
  const long long row = (long long)blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* a = psd + row * nfreq;
  for (int b = 0; b < nb; ++b) {
    const int lo = bands.lo[b], hi = bands.hi[b];
    double s = 0.0;
    if (mode == 0) {
      for (int k = lo + lane; k <= hi; k += 64) s += a[k];
    } else {
      for (int k = lo + lane; k < hi; k += 64) s += 0.5 * ((double)a[k] + (double)a[k + 1]) * df;
    }
    s = wave_sum_d(s);
    if (lane == 0) out[row * nb + b] = (float)s;
  }
}

// ---------------------------------------------------------------------------------------------
// overlap-average of window predictions pred[w][c][W] (window w starts at (w0+w)*stride) into
// out[c][n_samples] (+)= ..., gather form: each output sample adds the windows covering it in window
// order (deterministic), divides by max(count,1) when `finalize` (sliding_window.py:238-251).
// ---------------------------------------------------------------------------------------------
// log band power of a one-sided spectrum (SpectroTemporalEncoder._spectral_forward, models.py:1294-1301):
// out[row][k] = log( sum_f |X[row][f]|^2 * mask[k][f] + eps ).  One workgroup per row, lanes stride over f (the
// complex spectrum is read once, 8 bytes per bin, coalesced); K <= 16 running sums per lane in fp32, summed
// across the workgroup in fp64.
constexpr int BLP_MAXK = 16;
__global__ __launch_bounds__(THREADS) void band_log_power_kernel(const float2* spec, const float* masks, float* out,
                                                                 int nfreq, int nbands, float eps) {
  __shared__ double red[BLP_MAXK * (THREADS / 64)];
  const long long row = blockIdx.x;
  const float2* x = spec + row * nfreq;
  float acc[BLP_MAXK];
#pragma unroll
  for (int k = 0; k < BLP_MAXK; ++k) acc[k] = 0.f;
  for (int f = threadIdx.x; f < nfreq; f += THREADS) {
    const float2 v = x[f];
    const float pw = v.x * v.x + v.y * v.y;
#pragma unroll
    for (int k = 0; k < BLP_MAXK; ++k)
      if (k < nbands) acc[k] = fmaf(pw, masks[(long long)k * nfreq + f], acc[k]);
  }
  double d[BLP_MAXK];
#pragma unroll
  for (int k = 0; k < BLP_MAXK; ++k) d[k] = (double)acc[k];
  block_sum_d<BLP_MAXK>(d, red);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < BLP_MAXK; ++k)
      if (k < nbands) out[row * nbands + k] = logf((float)d[k] + eps);
  }
}


template <typename T>
__global__ __launch_bounds__(THREADS) void overlap_add_kernel(const T* pred, int n_win, int w0, int total_win, int c,
                                                               int wsize, int stride, long long n_samples, float* out,
                                                               float* counts, int accumulate, int finalize, int mode) {
  const long long s = (long long)blockIdx.x * THREADS + threadIdx.x;
  const int ch = blockIdx.y;
  if (s >= n_samples) return;
  // windows g of the WHOLE recording with g*stride <= s < g*stride + wsize:  [c_lo, t_hi]
  const long long c_hi = s / stride;
  const long long c_lo = s - wsize + 1 <= 0 ? 0 : (s - wsize + stride) / stride;   // ceil((s-wsize+1)/stride)
  const long long t_hi = c_hi > total_win - 1 ? total_win - 1 : c_hi;
  const long long w1 = (long long)w0 + n_win - 1;
  float* o = out + (long long)ch * n_samples + s;
  auto px = [&](long long g) { return to_f32(pred[((g - w0) * c + ch) * (long long)wsize + (s - g * stride)]); };
  if (mode == 1) {         // "last": the last covering window owns the sample
    if (t_hi >= c_lo && t_hi >= w0 && t_hi <= w1) *o = px(t_hi);
    return;
  }
  if (mode == 2) {         // "first": the first covering window owns it
    if (t_hi >= c_lo && c_lo >= w0 && c_lo <= w1) *o = px(c_lo);
    return;
  }
  const long long g_lo = c_lo < w0 ? w0 : c_lo;
  const long long g_hi = t_hi > w1 ? w1 : t_hi;
  float acc = accumulate ? *o : 0.f;
  for (long long g = g_lo; g <= g_hi; ++g) acc += px(g);
  if (finalize) {
    const float cnt = t_hi >= c_lo ? (float)(t_hi - c_lo + 1) : 0.f;
    acc = acc / fmaxf(cnt, 1.f);
    if (counts && ch == 0) counts[s] = cnt;
  }
  *o = acc;
}

}  // namespace

#define METRIC_DTYPE_SWITCH(dtype, what, ...)                                  \
  if ((dtype) == UNET_F32) { typedef float T; __VA_ARGS__ }                    \
  else if ((dtype) == UNET_BF16) { typedef __bf16 T; __VA_ARGS__ }             \
  else { unet_set_error("%s: bad dtype %d", what, (int)(dtype)); return -1; }

extern "C" int unet_pair_moments(int32_t dtype, const void* pred, const void* target, int64_t rows, int64_t ld_pred,
                                 int64_t ld_target, int32_t lo, int32_t hi, double* out_rows, void* stream) {
  UNET_REQUIRE(pred && target && out_rows, "pair_moments: null pointer");
  UNET_REQUIRE(rows > 0 && rows < (1ll << 31) && lo >= 0 && hi > lo && ld_pred >= hi && ld_target >= hi,
               "pair_moments: bad geometry rows=%lld lo=%d hi=%d ld=%lld/%lld", (long long)rows, lo, hi,
               (long long)ld_pred, (long long)ld_target);
  hipStream_t st = static_cast<hipStream_t>(stream);
  METRIC_DTYPE_SWITCH(dtype, "pair_moments", {
    hipLaunchKernelGGL(pair_moments_kernel<T>, dim3((unsigned)rows), dim3(THREADS), 0, st, (const T*)pred,
                       (const T*)target, (long long)ld_pred, (long long)ld_target, lo, hi, out_rows);
  })
  UNET_LAUNCH_CHECK("pair_moments");
  return 0;
}

extern "C" int unet_sum_rows_f64(const double* in, int64_t rows, int32_t width, double* out, void* stream) {
  UNET_REQUIRE(in && out && rows > 0 && width > 0 && width <= 64, "sum_rows_f64: bad arguments");
  hipLaunchKernelGGL(sum_rows_d_kernel, dim3(1), dim3(THREADS), 0, static_cast<hipStream_t>(stream), in,
                     (long long)rows, width, out);
  UNET_LAUNCH_CHECK("sum_rows_f64");
  return 0;
}

extern "C" int unet_l1_crop_bwd(int32_t dtype, int32_t grad_dtype, const void* pred, const void* target, int64_t rows,
                                int64_t ld_pred, int64_t ld_target, int32_t t, int32_t lo, int32_t hi,
                                const float* gscale, float mul, void* dpred, void* stream) {
  UNET_REQUIRE(pred && target && gscale && dpred, "l1_crop_bwd: null pointer");
  UNET_REQUIRE(rows > 0 && rows <= 65535 && t > 0 && lo >= 0 && hi >= lo && hi <= t && ld_pred >= t && ld_target >= t,
               "l1_crop_bwd: bad geometry (rows=%lld must be <= 65535)", (long long)rows);
  hipStream_t st = static_cast<hipStream_t>(stream);
  dim3 grid(cdiv(t, THREADS), (unsigned)rows);
  METRIC_DTYPE_SWITCH(dtype, "l1_crop_bwd", {
    if (grad_dtype == UNET_F32)
      hipLaunchKernelGGL((l1_crop_bwd_kernel<T, float>), grid, dim3(THREADS), 0, st, (const T*)pred, (const T*)target,
                         (long long)ld_pred, (long long)ld_target, t, lo, hi, gscale, mul, (float*)dpred);
    else if (grad_dtype == UNET_BF16)
      hipLaunchKernelGGL((l1_crop_bwd_kernel<T, __bf16>), grid, dim3(THREADS), 0, st, (const T*)pred, (const T*)target,
                         (long long)ld_pred, (long long)ld_target, t, lo, hi, gscale, mul, (__bf16*)dpred);
    else { unet_set_error("l1_crop_bwd: bad grad dtype %d", grad_dtype); return -1; }
  })
  UNET_LAUNCH_CHECK("l1_crop_bwd");
  return 0;
}

extern "C" int unet_row_zscore(int32_t in_dtype, int32_t out_dtype, const void* x, void* y, int64_t rows, int32_t t,
                               float eps, int32_t mode, int32_t unbiased, void* stream) {
  UNET_REQUIRE(x && y && rows > 0 && rows < (1ll << 31) && t > 0, "row_zscore: bad arguments");
  UNET_REQUIRE((size_t)t * 4 <= 150 * 1024, "row_zscore: rows of %d samples do not fit LDS (max 38400)", t);
  UNET_REQUIRE(mode == 0 || mode == 1, "row_zscore: mode must be 0 or 1");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t lds = (size_t)t * 4;
#define RZ_LAUNCH(TI, TO)                                                                                         \
  do {                                                                                                            \
    static bool attr = false;                                                                                     \
    if (!attr) {                                                                                                  \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&row_zscore_kernel<TI, TO>),               \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);                 \
      UNET_REQUIRE(e == hipSuccess, "row_zscore: hipFuncSetAttribute failed: %s", hipGetErrorString(e));          \
      attr = true;                                                                                                \
    }                                                                                                             \
    hipLaunchKernelGGL((row_zscore_kernel<TI, TO>), dim3((unsigned)rows), dim3(THREADS), lds, st, (const TI*)x,   \
                       (TO*)y, t, eps, mode, unbiased);                                                           \
  } while (0)
  if (in_dtype == UNET_F32 && out_dtype == UNET_F32) RZ_LAUNCH(float, float);
  else if (in_dtype == UNET_F32 && out_dtype == UNET_BF16) RZ_LAUNCH(float, __bf16);
  else if (in_dtype == UNET_BF16 && out_dtype == UNET_F32) RZ_LAUNCH(__bf16, float);
  else if (in_dtype == UNET_BF16 && out_dtype == UNET_BF16) RZ_LAUNCH(__bf16, __bf16);
  else { unet_set_error("row_zscore: bad dtypes %d/%d", in_dtype, out_dtype); return -1; }
#undef RZ_LAUNCH
  UNET_LAUNCH_CHECK("row_zscore");
  return 0;
}

extern "C" int unet_welch_psd(int32_t dtype, const void* x, int64_t rows, int64_t ld, int32_t lo, int32_t hi,
                              int32_t nperseg, int32_t detrend, float scale, float edge_scale, float* psd,
                              void* stream) {
  UNET_REQUIRE(x && psd && rows > 0 && rows < (1ll << 31), "welch_psd: bad arguments");
  int log2n = 0;
  while ((1 << log2n) < nperseg) ++log2n;
  UNET_REQUIRE(nperseg >= 128 && nperseg <= 2048 && (1 << log2n) == nperseg,
               "welch_psd: nperseg must be a power of two in [128, 2048] (got %d)", nperseg);
  UNET_REQUIRE(lo >= 0 && hi - lo >= nperseg && ld >= hi, "welch_psd: signal [%d,%d) shorter than nperseg=%d", lo, hi, nperseg);
  const int hop = nperseg / 2;
  const int n_seg = (hi - lo - nperseg) / hop + 1;      // torch.unfold(size=nperseg, step=hop)
  const size_t lds = (size_t)nperseg * 8 + (size_t)(nperseg / 2) * 8 + (size_t)nperseg * 4 + 64;
  hipStream_t st = static_cast<hipStream_t>(stream);
  METRIC_DTYPE_SWITCH(dtype, "welch_psd", {
    hipLaunchKernelGGL(welch_psd_kernel<T>, dim3((unsigned)rows), dim3(nperseg / 2), lds, st, (const T*)x, (long long)ld,
                       lo, n_seg, nperseg, log2n, detrend, scale, edge_scale, psd);
  })
  UNET_LAUNCH_CHECK("welch_psd");
  return 0;
}

extern "C" int unet_psd_db_stats(const float* psd_pred, const float* psd_target, int64_t rows, int32_t nfreq,
                                 int32_t k_lo, int32_t k_hi, int32_t normalize, double* out_rows, void* stream) {
  UNET_REQUIRE(psd_pred && psd_target && out_rows && rows > 0 && nfreq > 0, "psd_db_stats: bad arguments");
  UNET_REQUIRE(k_lo >= 0 && k_hi >= k_lo && k_hi < nfreq, "psd_db_stats: bin range [%d,%d] outside [0,%d)", k_lo, k_hi, nfreq);
  const long long blocks = (rows + THREADS / 64 - 1) / (THREADS / 64);
  hipLaunchKernelGGL(psd_db_stats_kernel, dim3((unsigned)blocks), dim3(THREADS), 0, static_cast<hipStream_t>(stream),
                     psd_pred, psd_target, (long long)rows, nfreq, k_lo, k_hi, normalize, out_rows);
  UNET_LAUNCH_CHECK("psd_db_stats");
  return 0;
}

extern "C" int unet_band_power(const float* psd, int64_t rows, int32_t nfreq, int32_t n_bands, const int32_t* lo_host,
                               const int32_t* hi_host, int32_t mode, float df, float* out, void* stream) {
  UNET_REQUIRE(psd && out && lo_host && hi_host && rows > 0 && nfreq > 0, "band_power: bad arguments");
  UNET_REQUIRE(n_bands > 0 && n_bands <= 16, "band_power: 1..16 bands (got %d)", n_bands);
  Bands b;
  for (int i = 0; i < 16; ++i) { b.lo[i] = 0; b.hi[i] = -1; }
  for (int i = 0; i < n_bands; ++i) {
    UNET_REQUIRE(lo_host[i] >= 0 && hi_host[i] < nfreq, "band_power: band %d bins [%d,%d] outside [0,%d)", i, lo_host[i],
                 hi_host[i], nfreq);
    b.lo[i] = lo_host[i]; b.hi[i] = hi_host[i];
  }
  const long long blocks = (rows + THREADS / 64 - 1) / (THREADS / 64);
  hipLaunchKernelGGL(band_power_kernel, dim3((unsigned)blocks), dim3(THREADS), 0, static_cast<hipStream_t>(stream), psd,
                     (long long)rows, nfreq, b, n_bands, mode, df, out);
  UNET_LAUNCH_CHECK("band_power");
  return 0;
}

extern "C" int unet_overlap_add(int32_t dtype, const void* pred, int32_t n_windows, int32_t first_window,
                                int32_t total_windows, int32_t c, int32_t window_size, int32_t stride,
                                int64_t n_samples, float* out, float* counts, int32_t accumulate, int32_t finalize,
                                int32_t mode, void* stream) {
  UNET_REQUIRE(out && (pred || n_windows == 0), "overlap_add: null pointer");
  UNET_REQUIRE(n_windows >= 0 && first_window >= 0 && c > 0 && c <= 65535 && window_size > 0 && stride > 0 && n_samples > 0,
               "overlap_add: bad geometry");
  UNET_REQUIRE(mode >= 0 && mode <= 2, "overlap_add: mode must be 0 (average), 1 (last) or 2 (first)");
  UNET_REQUIRE(first_window + n_windows <= total_windows, "overlap_add: windows [%d,%d) exceed total %d", first_window,
               first_window + n_windows, total_windows);
  UNET_REQUIRE(total_windows == 0 || ((long long)total_windows - 1) * stride + window_size <= n_samples,
               "overlap_add: last window ends beyond n_samples=%lld", (long long)n_samples);
  const long long bx = (n_samples + THREADS - 1) / THREADS;
  UNET_REQUIRE(bx < (1ll << 31), "overlap_add: recording too long");
  hipStream_t st = static_cast<hipStream_t>(stream);
  METRIC_DTYPE_SWITCH(dtype, "overlap_add", {
    hipLaunchKernelGGL(overlap_add_kernel<T>, dim3((unsigned)bx, (unsigned)c), dim3(THREADS), 0, st, (const T*)pred,
                       n_windows, first_window, total_windows, c, window_size, stride, (long long)n_samples, out, counts,
                       accumulate, finalize, mode);
  })
  UNET_LAUNCH_CHECK("overlap_add");
  return 0;
}

extern "C" int unet_band_log_power(const float* spectrum, const float* masks, int64_t rows, int32_t nfreq, int32_t n_bands,
                                   float eps, float* out, void* stream) {
  UNET_REQUIRE(spectrum && masks && out, "band_log_power: null pointer");
  UNET_REQUIRE(rows >= 0 && rows < (1ll << 31) && nfreq > 0, "band_log_power: bad shape rows=%lld nfreq=%d", (long long)rows, nfreq);
  UNET_REQUIRE(n_bands >= 1 && n_bands <= BLP_MAXK, "band_log_power: n_bands %d outside 1..%d", n_bands, BLP_MAXK);
  UNET_REQUIRE((reinterpret_cast<uintptr_t>(spectrum) & 7) == 0, "band_log_power: spectrum must be 8-byte aligned");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(band_log_power_kernel, dim3((unsigned)rows), dim3(THREADS), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float2*>(spectrum), masks, out, nfreq, n_bands, eps);
  UNET_LAUNCH_CHECK("band_log_power");
  return 0;
}
