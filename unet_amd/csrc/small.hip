// Per-window vector algebra of the blocks: the Linear layers that act on [windows, channels] vectors --
// FiLM's k_proj / v_proj / gate (models.py:493-494, 497-500, 506-509) and the SqueezeExcitation MLP
// (models.py:650-655, 660-661).  Each is a few MFLOP; issued through a BLAS library and elementwise tensor ops they
// were ~45 launches per block and step (~1.5 ms of a 27 ms config-B step, launch-latency-bound).  Here a group of up
// to UNET_SMALL_MAX linear maps that share their input rows is ONE launch each way:
//   forward   y_p = act_p(scale_p * (x_p W_p^T + b_p))                        (+ the pre-activation, kept for GELU')
//   backward  dpre = dy * act'(.) * scale ;  dW_p = dpre^T x ;  db_p = sum_n dpre ;  dx partials per output tile
// fp32 throughout (VALU FMAs; the sizes do not feed a matrix core), deterministic (no atomics: the input-gradient
// partials of the output tiles are added by unet_reduce_rows).
#include "common.h"

namespace {

constexpr int SL_T = 64;          // windows per tile and width of a reduction chunk
constexpr int SL_TO = 16;         // outputs per block: the maps are small (<= 2048 outputs), so narrow tiles are what
                                  // gives a launch enough blocks (64-output tiles ran 4..48 blocks: 80-340 us per launch)
constexpr int SL_THREADS = 256;
constexpr int SL_LD = SL_T + 1;   // LDS row pitch (floats): conflict-free column reads

struct SmallBatch {
  UnetSmallLinear p[UNET_SMALL_MAX];
  int np, n, max_splits;
};

__device__ __forceinline__ float sl_act(int act, float v) {
  if (act == UNET_SMALL_GELU) { float c, e; gelu_cdf(v, c, e); return v * c; }
  if (act == UNET_SMALL_SIGMOID) return 1.f / (1.f + __expf(-v));
  return v;
}

// block = (16-output tile, problem, 64-window tile x K split); thread = (window t & 63, output group t >> 6 of 4 outputs).
// Maps with more than SL_KS inputs are split along the reduction (the SE squeeze has up to 2048): each split writes
// its partial sums to q.part [split][n][n_out] and small_finish_kernel adds them and applies bias / scale / activation.
constexpr int SL_KS = 128;
__host__ __device__ inline int sl_splits(int n_in) { return (n_in + SL_KS - 1) / SL_KS; }

__global__ __launch_bounds__(SL_THREADS) void small_linear_fwd_kernel(const SmallBatch b) {
  __shared__ float xs[SL_T * SL_LD], ws[SL_TO * SL_LD];
  const UnetSmallLinear& q = b.p[blockIdx.y];
  const int splits = sl_splits(q.n_in);
  const int kz = blockIdx.z % b.max_splits, n0 = (blockIdx.z / b.max_splits) * SL_T;
  const int o0 = blockIdx.x * SL_TO;
  if (o0 >= q.n_out || kz >= splits) return;
  const int t = threadIdx.x, nl = t & 63, og = t >> 6;
  const int k_end = min(q.n_in, (kz + 1) * SL_KS);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = kz * SL_KS; k0 < k_end; k0 += SL_T) {
    // all 20 loads of the chunk issued back to back (clamped addresses, masked afterwards): as conditional loads in
    // a loop they ran one after the other, ~1 us each, and every launch took 19 us whatever its size
    float xr[SL_T * SL_T / SL_THREADS], wr[SL_TO * SL_T / SL_THREADS];
#pragma unroll
    for (int j = 0; j < SL_T * SL_T / SL_THREADS; ++j) {
      const int i = t + j * SL_THREADS, r = i >> 6, k = i & 63;
      const bool ok = n0 + r < b.n && k0 + k < k_end;
      xr[j] = q.x[ok ? (long long)(n0 + r) * q.n_in + k0 + k : 0];
      xr[j] = ok ? xr[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < SL_TO * SL_T / SL_THREADS; ++j) {
      const int i = t + j * SL_THREADS, r = i >> 6, k = i & 63;
      const bool ok = o0 + r < q.n_out && k0 + k < k_end;
      wr[j] = q.w[ok ? (long long)(o0 + r) * q.n_in + k0 + k : 0];
      wr[j] = ok ? wr[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < SL_T * SL_T / SL_THREADS; ++j) {
      const int i = t + j * SL_THREADS;
      xs[(i >> 6) * SL_LD + (i & 63)] = xr[j];
    }
#pragma unroll
    for (int j = 0; j < SL_TO * SL_T / SL_THREADS; ++j) {
      const int i = t + j * SL_THREADS;
      ws[(i >> 6) * SL_LD + (i & 63)] = wr[j];
    }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < SL_T; ++k) {
      const float xv = xs[nl * SL_LD + k];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] += xv * ws[(og * 4 + u) * SL_LD + k];
    }
    __syncthreads();
  }
  const int n = n0 + nl;
  if (n >= b.n) return;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int o = o0 + og * 4 + u;
    if (o >= q.n_out) continue;
    if (splits > 1) {
      q.part[((long long)kz * b.n + n) * q.n_out + o] = acc[u];
    } else {
      const float pre = q.scale * (acc[u] + (q.b ? q.b[o] : 0.f));
      if (q.pre) q.pre[(long long)n * q.n_out + o] = pre;
      q.y[(long long)n * q.n_out + o] = sl_act(q.act, pre);
    }
  }
}

__global__ __launch_bounds__(SL_THREADS) void small_finish_kernel(const SmallBatch b) {
  const UnetSmallLinear& q = b.p[blockIdx.y];
  const int splits = sl_splits(q.n_in);
  if (splits <= 1) return;
  const long long total = (long long)b.n * q.n_out;
  for (long long e = (long long)blockIdx.x * SL_THREADS + threadIdx.x; e < total; e += (long long)gridDim.x * SL_THREADS) {
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += q.part[z * total + e];
    const int o = (int)(e % q.n_out);
    const float pre = q.scale * (s + (q.b ? q.b[o] : 0.f));
    if (q.pre) q.pre[e] = pre;
    q.y[e] = sl_act(q.act, pre);
  }
}

// block = (16-output tile, problem, 64-wide chunk of the inputs): dW[tile][chunk] (reduction over all windows,
// accumulated in registers across the window tiles) and, per window tile, the input-gradient partial of this output
// tile.  dx_part is [16-output tiles][n][n_in]; db from the blocks of chunk 0.
__global__ __launch_bounds__(SL_THREADS) void small_linear_bwd_kernel(const SmallBatch b) {
  __shared__ float ds[SL_T * (SL_TO + 1)], xs[SL_T * SL_LD], ws[SL_TO * SL_LD];
  constexpr int DLD = SL_TO + 1;
  const UnetSmallLinear& q = b.p[blockIdx.y];
  const int o0 = blockIdx.x * SL_TO;
  if (o0 >= q.n_out) return;
  const int t = threadIdx.x, il = t & 63, g4 = t >> 6;
  float dbias = 0.f;                       // threads 0..15: output o0 + t
  const int k0 = blockIdx.z * SL_T;
  if (k0 >= q.n_in) return;
  {
    float dw[4] = {0.f, 0.f, 0.f, 0.f};    // dW[o0 + g4*4 + u][k0 + il]
    {
      float wr[SL_TO * SL_T / SL_THREADS];
#pragma unroll
      for (int j = 0; j < SL_TO * SL_T / SL_THREADS; ++j) {
        const int i = t + j * SL_THREADS, r = i >> 6, k = i & 63;
        const bool ok = o0 + r < q.n_out && k0 + k < q.n_in;
        wr[j] = q.w[ok ? (long long)(o0 + r) * q.n_in + k0 + k : 0];
        wr[j] = ok ? wr[j] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < SL_TO * SL_T / SL_THREADS; ++j) {
        const int i = t + j * SL_THREADS;
        ws[(i >> 6) * SL_LD + (i & 63)] = wr[j];
      }
    }
    const float* aux = q.act == UNET_SMALL_GELU ? q.pre : q.y;      // what the activation's derivative is taken from
    for (int n0 = 0; n0 < b.n; n0 += SL_T) {
      // loads first (clamped addresses), arithmetic and LDS writes after: see the forward kernel
      float dr[SL_T * SL_TO / SL_THREADS], ar[SL_T * SL_TO / SL_THREADS], xr[SL_T * SL_T / SL_THREADS];
#pragma unroll
      for (int j = 0; j < SL_T * SL_TO / SL_THREADS; ++j) {
        const int i = t + j * SL_THREADS, r = i >> 4, k = i & 15;          // r: window, k: output of the tile
        const bool ok = n0 + r < b.n && o0 + k < q.n_out;
        const long long e = ok ? (long long)(n0 + r) * q.n_out + o0 + k : 0;
        dr[j] = q.dy[e];
        ar[j] = q.act != UNET_SMALL_NONE ? aux[e] : 0.f;
        dr[j] = ok ? dr[j] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < SL_T * SL_T / SL_THREADS; ++j) {
        const int i = t + j * SL_THREADS, r = i >> 6, k = i & 63;
        const bool ok = n0 + r < b.n && k0 + k < q.n_in;
        xr[j] = q.x[ok ? (long long)(n0 + r) * q.n_in + k0 + k : 0];
        xr[j] = ok ? xr[j] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < SL_T * SL_TO / SL_THREADS; ++j) {
        const int i = t + j * SL_THREADS;
        float d = dr[j] * q.scale;
        if (q.act == UNET_SMALL_GELU) d *= act_bwd(UNET_ACT_GELU, ar[j]);
        else if (q.act == UNET_SMALL_SIGMOID) d *= ar[j] * (1.f - ar[j]);
        ds[(i >> 4) * DLD + (i & 15)] = d;
      }
#pragma unroll
      for (int j = 0; j < SL_T * SL_T / SL_THREADS; ++j) {
        const int i = t + j * SL_THREADS;
        xs[(i >> 6) * SL_LD + (i & 63)] = xr[j];
      }
      __syncthreads();
      if (k0 == 0 && t < SL_TO) {
        float s = 0.f;
        for (int r = 0; r < SL_T; ++r) s += ds[r * DLD + t];
        dbias += s;
      }
      // dW += dpre^T x  (reduction over the 64 windows of this tile)
#pragma unroll 8
      for (int r = 0; r < SL_T; ++r) {
        const float xv = xs[r * SL_LD + il];
#pragma unroll
        for (int u = 0; u < 4; ++u) dw[u] += ds[r * DLD + g4 * 4 + u] * xv;
      }
      // dx partial of this output tile: dxp[n][k0 + il] = sum_{o in tile} dpre[n][o] W[o][k0 + il]
      if (q.dx_part) {
        float wv[SL_TO];
#pragma unroll
        for (int o = 0; o < SL_TO; ++o) wv[o] = ws[o * SL_LD + il];
        float* dst = q.dx_part + ((long long)(q.dx_slot0 + blockIdx.x) * b.n) * q.n_in;
#pragma unroll 4
        for (int u = 0; u < 16; ++u) {
          const int r = g4 * 16 + u;
          float dx = 0.f;
#pragma unroll
          for (int o = 0; o < SL_TO; ++o) dx += ds[r * DLD + o] * wv[o];
          if (n0 + r < b.n && k0 + il < q.n_in) dst[(long long)(n0 + r) * q.n_in + k0 + il] = dx;
        }
      }
      __syncthreads();
    }
    if (q.dw && k0 + il < q.n_in) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int o = o0 + g4 * 4 + u;
        if (o < q.n_out) q.dw[(long long)o * q.n_in + k0 + il] = dw[u];
      }
    }
  }
  if (q.db && k0 == 0 && t < SL_TO && o0 + t < q.n_out) q.db[o0 + t] = dbias;
}

int fill_batch(SmallBatch& b, const UnetSmallLinear* probs, int np, int n, bool bwd, int* max_tiles, int* max_in = nullptr) {
  UNET_REQUIRE(probs && np > 0 && np <= UNET_SMALL_MAX, "small_linear: %d problems (1..%d)", np, UNET_SMALL_MAX);
  UNET_REQUIRE(n > 0, "small_linear: no rows");
  b.np = np; b.n = n; b.max_splits = 1;
  *max_tiles = 0;
  if (max_in) *max_in = 0;
  for (int i = 0; i < np; ++i) {
    const UnetSmallLinear& q = probs[i];
    UNET_REQUIRE(q.n_in > 0 && q.n_out > 0 && q.x && q.w && q.y, "small_linear: problem %d: bad sizes or null pointer", i);
    UNET_REQUIRE(q.act >= UNET_SMALL_NONE && q.act <= UNET_SMALL_SIGMOID, "small_linear: problem %d: bad activation %d", i, q.act);
    if (bwd) {
      UNET_REQUIRE(q.dy, "small_linear_bwd: problem %d: null dy", i);
      UNET_REQUIRE(q.act != UNET_SMALL_GELU || q.pre, "small_linear_bwd: problem %d: GELU needs the saved pre-activation", i);
    }
    b.p[i] = q;
    if (!bwd) {
      const int splits = cdiv(q.n_in, SL_KS);
      UNET_REQUIRE(splits == 1 || q.part, "small_linear_fwd: problem %d: %d inputs need the partial-sum buffer", i, q.n_in);
      b.max_splits = splits > b.max_splits ? splits : b.max_splits;
    }
    if (max_in && q.n_in > *max_in) *max_in = q.n_in;
    const int tiles = cdiv(q.n_out, SL_TO);
    *max_tiles = tiles > *max_tiles ? tiles : *max_tiles;
  }
  return 0;
}

}  // namespace

extern "C" int unet_small_linear_fwd(const UnetSmallLinear* problems, int32_t n_problems, int32_t n, void* stream) {
  SmallBatch b;
  int tiles = 0;
  if (fill_batch(b, problems, n_problems, n, false, &tiles)) return -1;
  UNET_REQUIRE((long long)cdiv(n, SL_T) * b.max_splits <= 65535, "small_linear_fwd: too many rows");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(small_linear_fwd_kernel, dim3(tiles, n_problems, cdiv(n, SL_T) * b.max_splits), dim3(SL_THREADS), 0, st, b);
  UNET_LAUNCH_CHECK("small_linear_fwd");
  if (b.max_splits > 1) {
    hipLaunchKernelGGL(small_finish_kernel, dim3(64, n_problems), dim3(SL_THREADS), 0, st, b);
    UNET_LAUNCH_CHECK("small_finish");
  }
  return 0;
}

extern "C" int unet_small_linear_bwd(const UnetSmallLinear* problems, int32_t n_problems, int32_t n, void* stream) {
  SmallBatch b;
  int tiles = 0, max_in = 0;
  if (fill_batch(b, problems, n_problems, n, true, &tiles, &max_in)) return -1;
  hipLaunchKernelGGL(small_linear_bwd_kernel, dim3(tiles, n_problems, cdiv(max_in, SL_T)), dim3(SL_THREADS), 0,
                     static_cast<hipStream_t>(stream), b);
  UNET_LAUNCH_CHECK("small_linear_bwd");
  return 0;
}

extern "C" int32_t unet_small_linear_tiles(int32_t n_out) { return cdiv(n_out, SL_TO); }
extern "C" int32_t unet_small_linear_splits(int32_t n_in) { return cdiv(n_in, SL_KS); }
