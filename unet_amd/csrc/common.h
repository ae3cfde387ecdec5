// Shared device/host helpers for libunet_hip (gfx950 only).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/unet_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// ---- fp32 band-pair attention (attn_f32.hip), reached through the extern "C" entry points of attn.hip ---------
int unet_cfc_attn_fwd_f32(const void* q, const void* k, const void* v, const float* coupling, int n, int l, int c, int n_heads,
                          int n_bands, float scale, void* out, void* o_pairs, float* lse, hipStream_t st);
int unet_cfc_attn_bwd_f32(const void* q, const void* k, const void* v, const float* coupling, const void* o_pairs,
                          const float* lse, const void* dout, int n, int l, int c, int n_heads, int n_bands, float scale,
                          void* dq, void* dk, void* dv, float* dcoupling_part, float* delta, hipStream_t st);

// ---- error plumbing (error.cpp) -------------------------------------------------------------
void unet_set_error(const char* fmt, ...);
#define UNET_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      unet_set_error(__VA_ARGS__);         \
      return -1;                           \
    }                                      \
  } while (0)
#define UNET_LAUNCH_CHECK(what)                                              \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      unet_set_error("%s: launch failed: %s", what, hipGetErrorString(e__)); \
      return -2;                                                             \
    }                                                                        \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- element access by dtype ----------------------------------------------------------------
template <bool F32> struct Elem;
template <> struct Elem<false> {
  typedef __bf16 T;
  static constexpr int SIZE = 2;
  static constexpr int EPC = 8;  // elements per 16-byte chunk
};
template <> struct Elem<true> {
  typedef float T;
  static constexpr int SIZE = 4;
  static constexpr int EPC = 4;
};

__device__ __forceinline__ float to_f32(__bf16 v) { return (float)v; }
__device__ __forceinline__ float to_f32(float v) { return v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ __bf16 from_f32<__bf16>(float v) { return (__bf16)v; }
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }

// vector of V elements <-> float[V]   (V*sizeof(T) in {2,4,8,16})
template <typename T, int V> struct VecIO {
  __device__ static __forceinline__ void load(const T* p, float (&f)[V]) {
#pragma unroll
    for (int i = 0; i < V; ++i) f[i] = to_f32(p[i]);
  }
  __device__ static __forceinline__ void store(T* p, const float (&f)[V]) {
#pragma unroll
    for (int i = 0; i < V; ++i) p[i] = from_f32<T>(f[i]);
  }
};
template <> struct VecIO<__bf16, 8> {
  __device__ static __forceinline__ void load(const __bf16* p, float (&f)[8]) {
    bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
  }
  __device__ static __forceinline__ void store(__bf16* p, const float (&f)[8]) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (__bf16)f[i];
    *reinterpret_cast<bf16x8*>(p) = v;
  }
};
template <> struct VecIO<__bf16, 4> {
  __device__ static __forceinline__ void load(const __bf16* p, float (&f)[4]) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = (float)v[i];
  }
  __device__ static __forceinline__ void store(__bf16* p, const float (&f)[4]) {
    bf16x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (__bf16)f[i];
    *reinterpret_cast<bf16x4*>(p) = v;
  }
};
template <> struct VecIO<float, 4> {
  __device__ static __forceinline__ void load(const float* p, float (&f)[4]) {
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = v[i];
  }
  __device__ static __forceinline__ void store(float* p, const float (&f)[4]) {
    f32x4 v = {f[0], f[1], f[2], f[3]};
    *reinterpret_cast<f32x4*>(p) = v;
  }
};

// ---- activations (models.py:59-65) and their derivatives ------------------------------------
// Exact-form GELU (nn.GELU(): 0.5 x (1 + erf(x/sqrt 2))) without libm's erff: the streaming kernels are
// HBM-bound only if the per-element math stays ~15 VALU ops.  Phi(x) via the Abramowitz-Stegun 7.1.26
// rational form of erfc (|error| <= 1.5e-7, evaluated on |x| so there is no cancellation in the negative
// tail); e = exp(-x^2/2) is shared with the derivative's density term.
__device__ __forceinline__ void gelu_cdf(float x, float& cdf, float& e) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
  e = __expf(-z * z);
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float hq = 0.5f * p * t * e;          // 0.5 * erfc(|x|/sqrt 2)
  cdf = x >= 0.f ? 1.f - hq : hq;
}
__device__ __forceinline__ float act_fwd(int act, float x) {
  switch (act) {
    case UNET_ACT_RELU: return x > 0.f ? x : 0.f;
    case UNET_ACT_LEAKY: return x > 0.f ? x : 0.2f * x;
    case UNET_ACT_GELU: { float c, e; gelu_cdf(x, c, e); return x * c; }
    case UNET_ACT_SILU: return x / (1.f + __expf(-x));
    case UNET_ACT_MISH: {
      // x * tanh(softplus(x)); with e = exp(x): tanh(log(1 + e)) = e(e + 2) / (e(e + 2) + 2)  (one exp, one divide)
      if (x > 20.f) return x;
      const float e = __expf(x), n = e * (e + 2.f);
      return x * (n / (n + 2.f));
    }
    default: return x;
  }
}
__device__ __forceinline__ float act_bwd(int act, float x) {
  switch (act) {
    case UNET_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case UNET_ACT_LEAKY: return x > 0.f ? 1.f : 0.2f;
    case UNET_ACT_GELU: { float c, e; gelu_cdf(x, c, e); return fmaf(x * 0.3989422804014327f, e, c); }
    case UNET_ACT_SILU: {
      float s = 1.f / (1.f + __expf(-x));
      return s * (1.f + x * (1.f - s));
    }
    case UNET_ACT_MISH: {
      if (x > 20.f) return 1.f;
      const float e = __expf(x), n = e * (e + 2.f);
      const float t = n / (n + 2.f);             // tanh(softplus(x))
      const float s = e / (1.f + e);             // sigmoid(x)
      return t + x * (1.f - t * t) * s;
    }
    default: return 1.f;
  }
}

// act(x) and act'(x) together (the GELU pair shares its erf / exp)
__device__ __forceinline__ void act_both(int act, float x, float& f, float& d) {
  if (act == UNET_ACT_GELU) {
    float c, e;
    gelu_cdf(x, c, e);
    f = x * c;
    d = fmaf(x * 0.3989422804014327f, e, c);
  } else {
    f = act_fwd(act, x);
    d = act_bwd(act, x);
  }
}

// The same with the activation known at compile time (ACT >= 0) or at run time (ACT < 0).  The streaming kernels
// evaluate it per element inside their row loop: with a run-time switch there the compiler keeps one branchy basic
// block per element (49 VALU instructions per GELU on the ISA, no packed math); specialised, the eight elements of a
// vector are straight-line code it can interleave and pack.
template <int ACT> __device__ __forceinline__ float act_fwd_t(int act, float x) {
  if constexpr (ACT == UNET_ACT_GELU) { float c, e; gelu_cdf(x, c, e); return x * c; }
  else return act_fwd(act, x);
}
template <int ACT> __device__ __forceinline__ float act_bwd_t(int act, float x) {
  if constexpr (ACT == UNET_ACT_GELU) { float c, e; gelu_cdf(x, c, e); return fmaf(x * 0.3989422804014327f, e, c); }
  else return act_bwd(act, x);
}
template <int ACT> __device__ __forceinline__ void act_both_t(int act, float x, float& f, float& d) {
  if constexpr (ACT == UNET_ACT_GELU) {
    float c, e;
    gelu_cdf(x, c, e);
    f = x * c;
    d = fmaf(x * 0.3989422804014327f, e, c);
  } else act_both(act, x, f, d);
}

// sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15), result in every lane of the row: four v_add_f32 with DPP
// operands (quad swaps, half-row mirror, row mirror), no LDS traffic -- __shfl_xor would be a ds_bpermute per step
__device__ __forceinline__ float row16_sum(float v) {
  auto dpp = [](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
  v += dpp(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
  v += dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror
  return v;
}

__device__ __forceinline__ float row16_max(float v) {
  auto dpp = [](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  v = fmaxf(v, dpp(v, std::integral_constant<int, 0xB1>{}));
  v = fmaxf(v, dpp(v, std::integral_constant<int, 0x4E>{}));
  v = fmaxf(v, dpp(v, std::integral_constant<int, 0x141>{}));
  v = fmaxf(v, dpp(v, std::integral_constant<int, 0x140>{}));
  return v;
}

// all 64 lanes: four DPP steps inside the rows of 16, two cross-row exchanges (ds_bpermute)
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

// bijective XCD-aware remap of a 1-D grid: blocks that share an XCD (id % 8) get a
// contiguous range of work ids, so neighbouring tiles hit the same L2.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}
