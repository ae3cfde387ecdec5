// Channels-last streaming kernels (HBM-bound): per-(window,channel) coefficient broadcast,
// activations, and per-(window,channel) reductions over rows.
//
// Thread mapping: channels are the contiguous axis, so a thread owns ONE vector column
// (V = 16 bytes of channels) and walks rows; consecutive lanes read consecutive 16-byte vectors
// (full-line coalescing), per-channel partial sums stay in registers for the whole walk and are
// combined across the block's row lanes through LDS once.  Partial sums go to
// [n][part][c] fp32 (deterministic; no float atomics) and the host adds the parts.
#include "common.h"

namespace {

// Rows per workgroup: 1/32 of the window (at most 128, at least 16, a multiple of 8), so that short, wide tensors
// (1024 rows x 1024 channels at the bottleneck) still launch 32 x N workgroups -- with a fixed 128 rows they launched
// 8 x N and ran at 3.4 TB/s instead of 5+.  Host (partial-buffer sizes, unet_parts) and kernels use the same rule.
__host__ __device__ inline int rows_per_part(int rows) {
  int r = ((rows + 31) / 32 + 7) & ~7;
  return r < 16 ? 16 : (r > 128 ? 128 : r);
}
constexpr int THREADS = 256;

template <class OP>
__global__ __launch_bounds__(THREADS) void rowstream_kernel(OP op, int rows, int c, int parts) {
  constexpr int V = OP::V;
  constexpr int NR = OP::NRED;
  __shared__ float red_s[(NR > 0 ? NR : 1) * V * (THREADS + 8)];
  const int n = blockIdx.y, part = blockIdx.x, tid = threadIdx.x;
  const int cv = c / V;
  const int cvb = cv < THREADS ? cv : THREADS;
  const int RL = THREADS / cvb;
  const int rl = tid / cvb, cgl = tid - rl * cvb;
  const int rpp = rows_per_part(rows);
  const int r_begin = part * rpp;
  const int r_end = rows < r_begin + rpp ? rows : r_begin + rpp;
  for (int cb = 0; cb < cv; cb += cvb) {
    const int cg = cb + cgl;
    const bool active = rl < RL && cg < cv;
    float acc[NR > 0 ? NR : 1][V];
#pragma unroll
    for (int k = 0; k < (NR > 0 ? NR : 1); ++k)
#pragma unroll
      for (int v = 0; v < V; ++v) acc[k][v] = 0.f;
    if (active) {
      typename OP::State st;
      op.init(st, n, cg * V, c);
      // U rows per iteration, all of their loads issued before the first use: a thread walks only rows / (parts * RL)
      // rows (16 at C = 256), and with one row in flight per thread the pass is latency-bound (the two pure reductions
      // ran at 3.5 TB/s); with four it is bandwidth-bound
      constexpr int U = OP::UNROLL, NIN = OP::NIN;
      const long long base = ((long long)n * rows) * c + cg * V;
      int r = r_begin + rl;
      for (; r + (U - 1) * RL < r_end; r += U * RL) {
        float in[U][NIN][V];
#pragma unroll
        for (int u = 0; u < U; ++u) op.ld(base + (long long)(r + u * RL) * c, in[u]);
        // keep every load of the iteration above the arithmetic (left alone, the scheduler sank some of them next to
        // their first use behind a vmcnt(0): measured +20 % on the norm-backward apply pass)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) op.row(st, base + (long long)(r + u * RL) * c, in[u], acc);
      }
      for (; r < r_end; r += RL) {
        float in[NIN][V];
        op.ld(base + (long long)r * c, in);
        op.row(st, base + (long long)r * c, in, acc);
      }
    }
    if constexpr (NR > 0) {
      // block-level sum over the RL row lanes: every thread owns a few (sum, channel) outputs (with one thread per
      // column group doing all NR * V of them the tail of a five-sum pass was 320 serial LDS reads on 32 lanes)
      constexpr int LDP = THREADS + 8;      // row pitch: 64 consecutive outputs (8 channels x 8 column groups) read 64 banks
#pragma unroll
      for (int k = 0; k < NR; ++k)
#pragma unroll
        for (int v = 0; v < V; ++v) red_s[(k * V + v) * LDP + tid] = acc[k][v];
      __syncthreads();
      const long long obase = ((long long)n * parts + part) * c;
      for (int idx = tid; idx < NR * V * cvb; idx += THREADS) {
        const int v = idx % V, g2 = (idx / V) % cvb, k = idx / (V * cvb);
        if (cb + g2 < cv) {
          const float* src = red_s + (k * V + v) * LDP + g2;
          float sum = 0.f;
          for (int j = 0; j < RL; ++j) sum += src[j * cvb];
          op.part(k)[obase + (long long)(cb + g2) * V + v] = sum;
        }
      }
      __syncthreads();
    }
  }
}

template <typename T, int V_> struct OpBase {
  static constexpr int V = V_;
  static constexpr int UNROLL = 4;      // rows in flight per thread (an op with heavy per-element math may lower it)
  __device__ static __forceinline__ void ldc(const float* p, float (&f)[V_]) {
#pragma unroll
    for (int i = 0; i < V_; ++i) f[i] = p[i];
  }
  __device__ static __forceinline__ void stc(float* p, const float (&f)[V_]) {
#pragma unroll
    for (int i = 0; i < V_; ++i) p[i] = f[i];
  }
};

// ---- s1 = sum y, s2 = sum y^2 ------------------------------------------------------------------
template <typename T, int V_> struct RowStatsOp : OpBase<T, V_> {
  static constexpr int NRED = 2, NIN = 1;
  const T* y; float* s1; float* s2;
  struct State {};
  __device__ void init(State&, int, int, int) const {}
  __device__ void ld(long long e, float (&in)[1][V_]) const { VecIO<T, V_>::load(y + e, in[0]); }
  __device__ void row(State&, long long, const float (&in)[1][V_], float (&acc)[2][V_]) const {
#pragma unroll
    for (int i = 0; i < V_; ++i) { acc[0][i] += in[0][i]; acc[1][i] += in[0][i] * in[0][i]; }
  }
  __device__ float* part(int k) const { return k ? s2 : s1; }
};

// ---- column sum --------------------------------------------------------------------------------
template <typename T, int V_> struct ColSumOp : OpBase<T, V_> {
  static constexpr int NRED = 1, NIN = 1;
  const T* x; float* out;
  struct State {};
  __device__ void init(State&, int, int, int) const {}
  __device__ void ld(long long e, float (&in)[1][V_]) const { VecIO<T, V_>::load(x + e, in[0]); }
  __device__ void row(State&, long long, const float (&in)[1][V_], float (&acc)[1][V_]) const {
#pragma unroll
    for (int i = 0; i < V_; ++i) acc[0][i] += in[0][i];
  }
  __device__ float* part(int) const { return out; }
};

// ---- z = act(a*y+b), pool = sum z -----------------------------------------------------------------
template <typename T, int V_, bool POOL, int ACT = -1> struct AffineActFwdOp : OpBase<T, V_> {
  static constexpr int NRED = POOL ? 1 : 0, NIN = 1;
  const T* y; T* z; const float* a; const float* b; float* pool; int act;
  struct State { float a[V_], b[V_]; };
  __device__ void init(State& s, int n, int c0, int c) const {
    this->ldc(a + (long long)n * c + c0, s.a); this->ldc(b + (long long)n * c + c0, s.b);
  }
  __device__ void ld(long long e, float (&in)[1][V_]) const { VecIO<T, V_>::load(y + e, in[0]); }
  __device__ void row(State& s, long long e, const float (&in)[1][V_], float (&acc)[POOL ? 1 : 1][V_]) const {
    float f[V_];
#pragma unroll
    for (int i = 0; i < V_; ++i) {
      f[i] = act_fwd_t<ACT>(act, s.a[i] * in[0][i] + s.b[i]);
      if (POOL) acc[0][i] += f[i];
    }
    VecIO<T, V_>::store(z + e, f);
  }
  __device__ float* part(int) const { return pool; }
};

// ---- backward of the above ---------------------------------------------------------------------
// SC: dz is the gradient of z = act(a*y+b)*sc (the squeeze-excite gate folded in): g = (dz*sc + dpool) * act'
template <typename T, int V_, bool APPLY, bool BSUM = false, bool SC = false, int ACT = -1> struct AffineActBwdOp : OpBase<T, V_> {
  static constexpr int NRED = APPLY ? (BSUM ? 1 : 0) : 2, NIN = 2;
  const float* sc = nullptr;
  static constexpr int UNROLL = APPLY ? 4 : 2;   // the reduce pass at 4 rows crosses 128 VGPRs and loses a wave per SIMD (measured +10 %)
  const T* y; const T* dz; const float* dpool; const float* a; const float* b; int act;
  float* dysum;                               // apply with BSUM: per-part column sums of dy (the producing conv's bias gradient)
  float* g1; float* g2;                       // reduce
  const float* ca; const float* cb; const float* cc; T* dy;  // apply
  struct State { float a[V_], b[V_], dp[V_], ca[V_], cb[V_], cc[V_], sc[SC ? V_ : 1]; };
  __device__ void init(State& s, int n, int c0, int c) const {
    const long long o = (long long)n * c + c0;
    this->ldc(a + o, s.a); this->ldc(b + o, s.b);
    if constexpr (SC) this->ldc(sc + o, s.sc);
    if (dpool) this->ldc(dpool + o, s.dp);
    else {
#pragma unroll
      for (int i = 0; i < V_; ++i) s.dp[i] = 0.f;
    }
    if (APPLY) { this->ldc(ca + o, s.ca); this->ldc(cb + o, s.cb); this->ldc(cc + o, s.cc); }
  }
  __device__ void ld(long long e, float (&in)[2][V_]) const {
    VecIO<T, V_>::load(y + e, in[0]);
    VecIO<T, V_>::load(dz + e, in[1]);
  }
  __device__ void row(State& s, long long e, const float (&in)[2][V_], float (&acc)[APPLY ? 1 : 2][V_]) const {
    float out[V_];
#pragma unroll
    for (int i = 0; i < V_; ++i) {
      const float fy = in[0][i];
      const float dzi = SC ? fmaf(in[1][i], s.sc[SC ? i : 0], s.dp[i]) : in[1][i] + s.dp[i];
      const float g = dzi * act_bwd_t<ACT>(act, s.a[i] * fy + s.b[i]);
      if (APPLY) {
        out[i] = s.ca[i] * g + s.cb[i] * fy + s.cc[i];
        if (BSUM) acc[0][i] += out[i];
      } else { acc[0][i] += g; acc[1][i] += g * fy; }
    }
    if (APPLY) VecIO<T, V_>::store(dy + e, out);
  }
  __device__ float* part(int k) const { return APPLY ? dysum : (k ? g2 : g1); }
};

// ---- norm + activation + squeeze-excite as one unit (models.py:801-839: norm, GELU, SE) -------------------------
// z = act(a*y+b) * s[n][c] with s = SE-MLP(mean_l act(a*y+b)): the gate needs the whole window first, so forward is a
// pool-only pass (1 read, nothing stored) and a fused pass (1 read, 1 write) -- the unscaled activation never exists
// in HBM (norm+act+pool R/W followed by scale R/W: 4 touches -> 3).
template <typename T, int V_, int ACT = -1> struct AffinePoolOp : OpBase<T, V_> {
  static constexpr int NRED = 1, NIN = 1;
  const T* y; const float* a; const float* b; float* pool; int act;
  struct State { float a[V_], b[V_]; };
  __device__ void init(State& s, int n, int c0, int c) const {
    this->ldc(a + (long long)n * c + c0, s.a); this->ldc(b + (long long)n * c + c0, s.b);
  }
  __device__ void ld(long long e, float (&in)[1][V_]) const { VecIO<T, V_>::load(y + e, in[0]); }
  __device__ void row(State& s, long long, const float (&in)[1][V_], float (&acc)[1][V_]) const {
#pragma unroll
    for (int i = 0; i < V_; ++i) acc[0][i] += act_fwd_t<ACT>(act, s.a[i] * in[0][i] + s.b[i]);
  }
  __device__ float* part(int) const { return pool; }
};
template <typename T, int V_, int ACT = -1> struct AffineActScaleFwdOp : OpBase<T, V_> {
  static constexpr int NRED = 0, NIN = 1;
  const T* y; T* z; const float* a; const float* b; const float* sc; int act;
  struct State { float a[V_], b[V_], sc[V_]; };
  __device__ void init(State& s, int n, int c0, int c) const {
    const long long o = (long long)n * c + c0;
    this->ldc(a + o, s.a); this->ldc(b + o, s.b); this->ldc(sc + o, s.sc);
  }
  __device__ void ld(long long e, float (&in)[1][V_]) const { VecIO<T, V_>::load(y + e, in[0]); }
  __device__ void row(State& s, long long e, const float (&in)[1][V_], float (&)[1][V_]) const {
    float f[V_];
#pragma unroll
    for (int i = 0; i < V_; ++i) f[i] = act_fwd_t<ACT>(act, s.a[i] * in[0][i] + s.b[i]) * s.sc[i];
    VecIO<T, V_>::store(z + e, f);
  }
  __device__ float* part(int) const { return nullptr; }
};
// Backward, given dz = d loss / d z.  With u = a*y+b:  d s = sum_l dz*act(u), and the two sums the norm backward
// needs, sum_l g and sum_l g*y with g = (dz*s + dpool)*act'(u), are LINEAR in the per-(window, channel) constants s
// and dpool (dpool = the pool's gradient, known only after d s went through the SE MLP backward).  So ONE pass over
// (y, dz) accumulates five sums per (window, channel),
//   P0 = sum dz*act(u)   P1 = sum dz*act'(u)   P2 = sum dz*act'(u)*y   P3 = sum act'(u)   P4 = sum act'(u)*y,
// and  sum g = s*P1 + dpool*P3,  sum g*y = s*P2 + dpool*P4  are formed on [N, C] afterwards (se_combine_kernel).
// Replaces the SE-excite backward pass (3 touches) and the two-sum reduction pass (2 touches) by 2 touches.
template <typename T, int V_, int ACT = -1> struct AffineActSEBwdReduceOp : OpBase<T, V_> {
  static constexpr int NRED = 5, NIN = 2;
  static constexpr int UNROLL = 2;
  const T* y; const T* dz; const float* a; const float* b; float* p; long long pstride; int act;
  struct State { float a[V_], b[V_]; };
  __device__ void init(State& s, int n, int c0, int c) const {
    this->ldc(a + (long long)n * c + c0, s.a); this->ldc(b + (long long)n * c + c0, s.b);
  }
  __device__ void ld(long long e, float (&in)[2][V_]) const {
    VecIO<T, V_>::load(y + e, in[0]);
    VecIO<T, V_>::load(dz + e, in[1]);
  }
  __device__ void row(State& s, long long, const float (&in)[2][V_], float (&acc)[5][V_]) const {
#pragma unroll
    for (int i = 0; i < V_; ++i) {
      const float fy = in[0][i], d = in[1][i];
      float f, fp;
      act_both_t<ACT>(act, s.a[i] * fy + s.b[i], f, fp);
      const float dfp = d * fp;
      acc[0][i] += d * f;
      acc[1][i] += dfp;
      acc[2][i] += dfp * fy;
      acc[3][i] += fp;
      acc[4][i] += fp * fy;
    }
  }
  __device__ float* part(int k) const { return p + k * pstride; }
};

__global__ void se_combine_kernel(const float* __restrict__ p, long long nc, const float* __restrict__ s,
                                  const float* __restrict__ dpool, float* __restrict__ g1, float* __restrict__ g2) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nc) return;
  const float sv = s[i], dp = dpool ? dpool[i] : 0.f;
  g1[i] = fmaf(sv, p[nc + i], dp * p[3 * nc + i]);
  g2[i] = fmaf(sv, p[2 * nc + i], dp * p[4 * nc + i]);
}

// ---- out = x*s ;  dx = dout*s, ds = sum dout*x ----------------------------------------------------
template <typename T, int V_> struct ScaleFwdOp : OpBase<T, V_> {
  static constexpr int NRED = 0, NIN = 1;
  const T* x; T* out; const float* s;
  struct State { float s[V_]; };
  __device__ void init(State& st, int n, int c0, int c) const { this->ldc(s + (long long)n * c + c0, st.s); }
  __device__ void ld(long long e, float (&in)[1][V_]) const { VecIO<T, V_>::load(x + e, in[0]); }
  __device__ void row(State& st, long long e, const float (&in)[1][V_], float (&)[1][V_]) const {
    float f[V_];
#pragma unroll
    for (int i = 0; i < V_; ++i) f[i] = in[0][i] * st.s[i];
    VecIO<T, V_>::store(out + e, f);
  }
  __device__ float* part(int) const { return nullptr; }
};
template <typename T, int V_> struct ScaleBwdOp : OpBase<T, V_> {
  static constexpr int NRED = 1, NIN = 2;
  const T* x; const T* dout; T* dx; const float* s; float* ds;
  struct State { float s[V_]; };
  __device__ void init(State& st, int n, int c0, int c) const { this->ldc(s + (long long)n * c + c0, st.s); }
  __device__ void ld(long long e, float (&in)[2][V_]) const {
    VecIO<T, V_>::load(x + e, in[0]);
    VecIO<T, V_>::load(dout + e, in[1]);
  }
  __device__ void row(State& st, long long e, const float (&in)[2][V_], float (&acc)[1][V_]) const {
    float fd[V_];
#pragma unroll
    for (int i = 0; i < V_; ++i) { acc[0][i] += in[1][i] * in[0][i]; fd[i] = in[1][i] * st.s[i]; }
    VecIO<T, V_>::store(dx + e, fd);
  }
  __device__ float* part(int) const { return ds; }
};

// ---- out = res + g*o ;  d_o = g*dout, dg = sum dout*o ----------------------------------------------
template <typename T, int V_> struct GatedAddFwdOp : OpBase<T, V_> {
  static constexpr int NRED = 0, NIN = 2;
  const T* res; const T* o; T* out; const float* g;
  struct State { float g[V_]; };
  __device__ void init(State& st, int n, int c0, int c) const { this->ldc(g + (long long)n * c + c0, st.g); }
  __device__ void ld(long long e, float (&in)[2][V_]) const {
    VecIO<T, V_>::load(res + e, in[0]);
    VecIO<T, V_>::load(o + e, in[1]);
  }
  __device__ void row(State& st, long long e, const float (&in)[2][V_], float (&)[1][V_]) const {
    float fr[V_];
#pragma unroll
    for (int i = 0; i < V_; ++i) fr[i] = in[0][i] + st.g[i] * in[1][i];
    VecIO<T, V_>::store(out + e, fr);
  }
  __device__ float* part(int) const { return nullptr; }
};
// SUM: also the per-part column sums of d_o (the bias gradient of the conv that produced o), saving a pass over d_o
template <typename T, int V_, bool SUM> struct GatedAddBwdOp : OpBase<T, V_> {
  static constexpr int NRED = SUM ? 2 : 1, NIN = 2;
  const T* o; const T* dout; T* d_o; const float* g; float* dg; float* dsum;
  struct State { float g[V_]; };
  __device__ void init(State& st, int n, int c0, int c) const { this->ldc(g + (long long)n * c + c0, st.g); }
  __device__ void ld(long long e, float (&in)[2][V_]) const {
    VecIO<T, V_>::load(o + e, in[0]);
    VecIO<T, V_>::load(dout + e, in[1]);
  }
  __device__ void row(State& st, long long e, const float (&in)[2][V_], float (&acc)[SUM ? 2 : 1][V_]) const {
    float fd[V_];
#pragma unroll
    for (int i = 0; i < V_; ++i) {
      acc[0][i] += in[1][i] * in[0][i];
      fd[i] = in[1][i] * st.g[i];
      if (SUM) acc[SUM ? 1 : 0][i] += fd[i];
    }
    VecIO<T, V_>::store(d_o + e, fd);
  }
  __device__ float* part(int k) const { return k ? dsum : dg; }
};

template <class OP>
int launch_rows(const OP& op, int n, int rows, int c, hipStream_t st, const char* what) {
  const int parts = cdiv(rows, rows_per_part(rows));
  hipLaunchKernelGGL(rowstream_kernel<OP>, dim3(parts, n), dim3(THREADS), 0, st, op, rows, c, parts);
  UNET_LAUNCH_CHECK(what);
  return 0;
}

// vector width usable for a set of pointers: 16 bytes when channels and bases allow, else scalar
template <typename T> bool can_vec(int c, std::initializer_list<const void*> ptrs) {
  constexpr int V = 16 / (int)sizeof(T);
  if (c % V) return false;
  for (const void* p : ptrs)
    if (p && !aligned16(p)) return false;
  return true;
}

#define DISPATCH_ROWS(dtype, c, ptrs, ...)                                        \
  do {                                                                              \
    if ((dtype) == UNET_F32) {                                                      \
      typedef float T;                                                              \
      if (can_vec<T>(c, ptrs)) { constexpr int V = 4; __VA_ARGS__ } else { constexpr int V = 1; __VA_ARGS__ } \
    } else if ((dtype) == UNET_BF16) {                                              \
      typedef __bf16 T;                                                             \
      if (can_vec<T>(c, ptrs)) { constexpr int V = 8; __VA_ARGS__ } else { constexpr int V = 1; __VA_ARGS__ } \
    } else {                                                                        \
      unet_set_error("bad dtype %d", (int)(dtype));                                 \
      return -1;                                                                    \
    }                                                                               \
  } while (0)

#define PL(...) std::initializer_list<const void*>{__VA_ARGS__}

// GELU (every production config) gets kernels with the activation compiled in; the rest keep the run-time switch
#define DISPATCH_ACT(act, ...)                                                      \
  do {                                                                              \
    if ((act) == UNET_ACT_GELU) { constexpr int ACT = UNET_ACT_GELU; __VA_ARGS__ }  \
    else { constexpr int ACT = -1; __VA_ARGS__ }                                    \
  } while (0)

int check_dims(int n, int rows, int c, const char* what) {
  UNET_REQUIRE(n > 0 && rows > 0 && c > 0, "%s: non-positive dims n=%d rows=%d c=%d", what, n, rows, c);
  UNET_REQUIRE(n <= 65535, "%s: n=%d exceeds grid.y limit", what, n);
  return 0;
}

// ---------------------------------------------------------------------------------------------
// FiLM channel softmax: one wave per row, lanes stride over channel vectors of 4.
// ---------------------------------------------------------------------------------------------
// rows a wave keeps in flight: the row is a dependent chain (load -> max -> exp -> sum -> scale -> store).  Measured
// (scripts/ab_rows.py): four rows at C = 256 (one vector per lane) -6 %; two rows at C = 512 +2 %, so wider rows keep one;
// the backward kernel with two rows at C = 256 ran 35 % slower (registers) and keeps one everywhere.
template <int NV> struct FilmRows { static constexpr int U = NV <= 1 ? 4 : 1; };

template <typename T, int V, int NV>
__global__ __launch_bounds__(THREADS) void film_softmax_fwd_kernel(const T* q, const float* kk, const float* vv,
                                                                    T* h, int rows, int c) {
  constexpr int U = FilmRows<NV>::U;
  const int n = blockIdx.y, part = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cv = c / V;
  float k[NV][V], v[NV][V];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int cg = lane + 64 * i;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      k[i][e] = cg < cv ? kk[(long long)n * c + cg * V + e] : 0.f;
      v[i][e] = cg < cv ? vv[(long long)n * c + cg * V + e] : 0.f;
    }
  }
  const int rpp = rows_per_part(rows);
  const int r_begin = part * rpp;
  const int r_end = rows < r_begin + rpp ? rows : r_begin + rpp;
  for (int r0 = r_begin + wave * U; r0 < r_end; r0 += (THREADS / 64) * U) {
    float s[U][NV][V];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long base = ((long long)n * rows + r0 + u) * c;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int cg = lane + 64 * i;
        if (cg < cv && r0 + u < r_end) VecIO<T, V>::load(q + base + cg * V, s[u][i]);
        else {
#pragma unroll
          for (int e = 0; e < V; ++e) s[u][i][e] = 0.f;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);      // every row's loads above the arithmetic
    float mx[U], sum[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      mx[u] = -INFINITY;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const bool in = lane + 64 * i < cv;
#pragma unroll
        for (int e = 0; e < V; ++e) { s[u][i][e] = in ? s[u][i][e] * k[i][e] : -INFINITY; mx[u] = fmaxf(mx[u], s[u][i][e]); }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) mx[u] = wave_max(mx[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      sum[u] = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < V; ++e) {
          s[u][i][e] = __expf(s[u][i][e] - mx[u]);       // channels past the end: exp(-inf) = 0
          sum[u] += s[u][i][e];
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) sum[u] = wave_sum(sum[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r0 + u >= r_end) continue;          // wave-uniform
      const float inv = 1.f / sum[u];
      const long long base = ((long long)n * rows + r0 + u) * c;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int cg = lane + 64 * i;
        if (cg < cv) {
#pragma unroll
          for (int e = 0; e < V; ++e) s[u][i][e] = s[u][i][e] * inv * v[i][e];
          VecIO<T, V>::store(h + base + cg * V, s[u][i]);
        }
      }
    }
  }
}

// SUM: also the per-part column sums of dq (the bias gradient of q_proj) -- one more accumulator per channel
template <typename T, int V, int NV, bool SUM>
__global__ __launch_bounds__(THREADS) void film_softmax_bwd_kernel(const T* q, const T* dh, const float* kk,
                                                                    const float* vv, T* dq, float* dkk_part,
                                                                    float* dv_part, float* dq_sum_part, int rows, int c,
                                                                    int parts) {
  constexpr int U = 1;
  __shared__ float red_s[SUM ? 3 : 2][THREADS / 64][64 * NV * V];
  const int n = blockIdx.y, part = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cv = c / V;
  float k[NV][V], v[NV][V], akk[NV][V], av[NV][V], adq[SUM ? NV : 1][V];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int cg = lane + 64 * i;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      k[i][e] = cg < cv ? kk[(long long)n * c + cg * V + e] : 0.f;
      v[i][e] = cg < cv ? vv[(long long)n * c + cg * V + e] : 0.f;
      akk[i][e] = 0.f; av[i][e] = 0.f;
      if (SUM) adq[SUM ? i : 0][e] = 0.f;
    }
  }
  const int rpp = rows_per_part(rows);
  const int r_begin = part * rpp;
  const int r_end = rows < r_begin + rpp ? rows : r_begin + rpp;
  for (int r0 = r_begin + wave * U; r0 < r_end; r0 += (THREADS / 64) * U) {
    float s[U][NV][V], qv[U][NV][V], g[U][NV][V];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long base = ((long long)n * rows + r0 + u) * c;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int cg = lane + 64 * i;
        if (cg < cv && r0 + u < r_end) {
          VecIO<T, V>::load(q + base + cg * V, qv[u][i]);
          VecIO<T, V>::load(dh + base + cg * V, g[u][i]);
        } else {
#pragma unroll
          for (int e = 0; e < V; ++e) { qv[u][i][e] = 0.f; g[u][i][e] = 0.f; }   // a row past the end adds zeros
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);      // every row's loads above the arithmetic
    float mx[U], sum[U], dot[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      mx[u] = -INFINITY;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const bool in = lane + 64 * i < cv;
#pragma unroll
        for (int e = 0; e < V; ++e) { s[u][i][e] = in ? qv[u][i][e] * k[i][e] : -INFINITY; mx[u] = fmaxf(mx[u], s[u][i][e]); }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) mx[u] = wave_max(mx[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      sum[u] = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < V; ++e) { s[u][i][e] = __expf(s[u][i][e] - mx[u]); sum[u] += s[u][i][e]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) sum[u] = wave_sum(sum[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float inv = 1.f / sum[u];
      dot[u] = 0.f;  // sum_c a * da,  da = dh * v
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < V; ++e) {
          s[u][i][e] *= inv;                             // a
          av[i][e] += g[u][i][e] * s[u][i][e];           // dv += dh * a
          g[u][i][e] *= v[i][e];                         // da
          dot[u] += s[u][i][e] * g[u][i][e];
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) dot[u] = wave_sum(dot[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long base = ((long long)n * rows + r0 + u) * c;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int cg = lane + 64 * i;
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float ds = s[u][i][e] * (g[u][i][e] - dot[u]);  // d(q*kk)
          akk[i][e] += ds * qv[u][i][e];
          g[u][i][e] = ds * k[i][e];
          if (SUM) adq[SUM ? i : 0][e] += g[u][i][e];
        }
        if (cg < cv && r0 + u < r_end) VecIO<T, V>::store(dq + base + cg * V, g[u][i]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < V; ++e) {
      red_s[0][wave][(i * V + e) * 64 + lane] = akk[i][e];
      red_s[1][wave][(i * V + e) * 64 + lane] = av[i][e];
      if (SUM) red_s[SUM ? 2 : 0][wave][(i * V + e) * 64 + lane] = adq[SUM ? i : 0][e];
    }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int cg = lane + 64 * i;
      if (cg < cv) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
          float a0 = 0.f, a1 = 0.f, a2 = 0.f;
          for (int w = 0; w < THREADS / 64; ++w) {
            a0 += red_s[0][w][(i * V + e) * 64 + lane];
            a1 += red_s[1][w][(i * V + e) * 64 + lane];
            if (SUM) a2 += red_s[SUM ? 2 : 0][w][(i * V + e) * 64 + lane];
          }
          const long long o = ((long long)n * parts + part) * c + cg * V + e;
          dkk_part[o] = a0; dv_part[o] = a1;
          if (SUM) dq_sum_part[o] = a2;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Column sums over partial rows.  The partial buffers are [rows = n*parts][c] fp32 with rows in the
// thousands; a block owns CS_CH = 32 channels and spreads the rows over CS_SEG = 32 row segments
// (1024 threads), each thread keeping 8 independent loads in flight, so the reduction is bandwidth-
// not latency-bound.  fp64 accumulation; result of column (ch0 + cl) returned to the threads with seg == 0.
// ---------------------------------------------------------------------------------------------
constexpr int CS_CH = 32, CS_SEG = 32, CS_THREADS = CS_CH * CS_SEG;

template <int K>
__device__ __forceinline__ void column_sums(const float* const (&src)[K], int rows, int c, int ch, double (&out)[K],
                                            double* red /* [K][CS_SEG][CS_CH] */) {
  const int cl = threadIdx.x % CS_CH, seg = threadIdx.x / CS_CH;
  double acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k] = 0.0;
  if (ch < c) {
    int r = seg;
    for (; r + 7 * CS_SEG < rows; r += 8 * CS_SEG) {
      float v[K][8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < K; ++k) v[k][u] = src[k][(long long)(r + u * CS_SEG) * c + ch];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] += v[k][u];
    }
    for (; r < rows; r += CS_SEG)
#pragma unroll
      for (int k = 0; k < K; ++k) acc[k] += src[k][(long long)r * c + ch];
  }
#pragma unroll
  for (int k = 0; k < K; ++k) red[(k * CS_SEG + seg) * CS_CH + cl] = acc[k];
  __syncthreads();
  if (seg == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      double sum = 0.0;
      for (int j = 0; j < CS_SEG; ++j) sum += red[(k * CS_SEG + j) * CS_CH + cl];
      out[k] = sum;
    }
  }
}

// BatchNorm (training) statistics finalisation: one launch replaces ~30 tiny tensor ops.
__global__ __launch_bounds__(CS_THREADS) void bn_finalize_fwd_kernel(const float* s1p, const float* s2p, int n, int parts, int c,
                                                                     double count, float eps, const float* gamma,
                                                                     const float* beta, float* run_mean, float* run_var,
                                                                     float momentum, float* mean_o, float* rstd_o, float* a_o,
                                                                     float* b_o, long long* batches_tracked) {
  __shared__ double red[2 * CS_SEG * CS_CH];
  __shared__ float coef[4][CS_CH];
  if (batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *batches_tracked += 1;
  const int cl = threadIdx.x % CS_CH, seg = threadIdx.x / CS_CH;
  const int ch = blockIdx.x * CS_CH + cl;
  const float* const src[2] = {s1p, s2p};
  double s[2];
  column_sums<2>(src, n * parts, c, ch, s, red);
  if (seg == 0 && ch < c) {
    const double mean = s[0] / count;
    double var = s[1] / count - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    if (run_mean) {
      const double unb = var * (count / (count > 1.0 ? count - 1.0 : 1.0));
      run_mean[ch] = (float)((1.0 - momentum) * run_mean[ch] + momentum * mean);
      run_var[ch] = (float)((1.0 - momentum) * run_var[ch] + momentum * unb);
    }
    const double g = gamma ? (double)gamma[ch] : 1.0, bt = beta ? (double)beta[ch] : 0.0;
    coef[0][cl] = (float)mean; coef[1][cl] = (float)rstd;
    coef[2][cl] = (float)(rstd * g); coef[3][cl] = (float)(bt - mean * rstd * g);
  }
  __syncthreads();
  if (ch >= c) return;
  const float m = coef[0][cl], r = coef[1][cl], a = coef[2][cl], b = coef[3][cl];
  for (int i = seg; i < n; i += CS_SEG) {
    const long long o = (long long)i * c + ch;
    mean_o[o] = m; rstd_o[o] = r; a_o[o] = a; b_o[o] = b;
  }
}

// backward: dgamma, dbeta and the per-(window, channel) coefficients of  dy = ca*g + cb*y + cc
__global__ __launch_bounds__(CS_THREADS) void bn_finalize_bwd_kernel(const float* g1p, const float* g2p, int n, int parts, int c,
                                                                     double count, const float* mean_i, const float* rstd_i,
                                                                     const float* gamma, float* ca_o, float* cb_o, float* cc_o,
                                                                     float* dgamma, float* dbeta) {
  __shared__ double red[2 * CS_SEG * CS_CH];
  __shared__ float coef[3][CS_CH];
  const int cl = threadIdx.x % CS_CH, seg = threadIdx.x / CS_CH;
  const int ch = blockIdx.x * CS_CH + cl;
  const float* const src[2] = {g1p, g2p};
  double s[2];
  column_sums<2>(src, n * parts, c, ch, s, red);
  if (seg == 0 && ch < c) {
    const double g1 = s[0], g2 = s[1];
    const double mean = mean_i[ch], rstd = rstd_i[ch];   // row 0 of the [n][c] arrays (identical for all windows)
    const double gm = gamma ? (double)gamma[ch] : 1.0;
    const double gy = rstd * (g2 - mean * g1);           // sum g * yhat
    if (dgamma) dgamma[ch] = (float)gy;
    if (dbeta) dbeta[ch] = (float)g1;
    const double m1 = gm * g1 / count, m2 = gm * gy / count;
    coef[0][cl] = (float)(rstd * gm); coef[1][cl] = (float)(-rstd * rstd * m2);
    coef[2][cl] = (float)(-rstd * m1 + rstd * rstd * mean * m2);
  }
  __syncthreads();
  if (ch >= c) return;
  const float ca = coef[0][cl], cb = coef[1][cl], cc = coef[2][cl];
  for (int i = seg; i < n; i += CS_SEG) {
    const long long o = (long long)i * c + ch;
    ca_o[o] = ca; cb_o[o] = cb; cc_o[o] = cc;
  }
}

// out[c] = sum over rows of in[rows][c]   (second stage of the per-part column sums)
__global__ __launch_bounds__(CS_THREADS) void reduce_rows_kernel(const float* in, int rows, int c, float* out) {
  __shared__ double red[CS_SEG * CS_CH];
  const int ch = blockIdx.x * CS_CH + threadIdx.x % CS_CH;
  const float* const src[1] = {in};
  double s[1];
  column_sums<1>(src, rows, c, ch, s, red);
  if (threadIdx.x / CS_CH == 0 && ch < c) out[ch] = (float)s[0];
}

// several reductions in ONE launch: out[u][c] = sum_r in[u][r][c] for every unit u of every job (a reduce_rows is one
// unit, a reduce_parts is one unit per window).  These sums are launch-bound (a few microseconds each, tens of them
// per training step); the backward of a FiLM block alone needs five.
struct ReduceM {
  UnetReduceJob j[UNET_REDUCE_MAX_JOBS];
  int first[UNET_REDUCE_MAX_JOBS + 1];    // first[k] = units of jobs 0..k-1
  int n;
};
__global__ __launch_bounds__(CS_THREADS) void reduce_multi_kernel(const ReduceM m) {
  __shared__ double red[CS_SEG * CS_CH];
  int k = 0;
#pragma unroll
  for (int i = 1; i < UNET_REDUCE_MAX_JOBS; ++i)
    if (i < m.n && (int)blockIdx.y >= m.first[i]) k = i;
  const UnetReduceJob& job = m.j[k];
  const int u = blockIdx.y - m.first[k];
  if ((int)blockIdx.x * CS_CH >= job.c) return;            // block-uniform: this job has fewer channels than the widest
  const int ch = blockIdx.x * CS_CH + threadIdx.x % CS_CH;
  const float* const src[1] = {job.in + (long long)u * job.rows * job.c};
  double s[1];
  column_sums<1>(src, job.rows, job.c, ch, s, red);
  if (threadIdx.x / CS_CH == 0 && ch < job.c) job.out[(long long)u * job.c + ch] = (float)s[0];
}

// out[n][c] = sum_p in[n][p][c]   (adds the per-row-chunk partials of one window; replaces a strided
// torch reduction over the middle axis)
__global__ __launch_bounds__(256) void reduce_parts_kernel(const float* in, int parts, int c, float* out) {
  const int n = blockIdx.y;
  const int ch = blockIdx.x * 256 + threadIdx.x;
  if (ch >= c) return;
  const float* p = in + (long long)n * parts * c + ch;
  float a = 0.f;
  int r = 0;
  for (; r + 7 < parts; r += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(long long)(r + u) * c];
#pragma unroll
    for (int u = 0; u < 8; ++u) a += v[u];
  }
  for (; r < parts; ++r) a += p[(long long)r * c];
  out[(long long)n * c + ch] = a;
}

template <typename T, int V>
int film_dispatch(bool bwd, const void* q, const void* dh, const float* kk, const float* v, void* out,
                  float* dkk, float* dv, float* dqs, bool* fused, int n, int rows, int c, hipStream_t st) {
  const int cv = c / V;
  const int nv = cdiv(cv, 64);
  const int parts = cdiv(rows, rows_per_part(rows));
  dim3 grid(parts, n), blk(THREADS);
#define FILM_CASE(NVV)                                                                                  \
  if (nv <= NVV) {                                                                                      \
    if (bwd && dqs && NVV <= 2) {                                                                       \
      if constexpr (NVV <= 2)                                                                           \
        hipLaunchKernelGGL((film_softmax_bwd_kernel<T, V, NVV, true>), grid, blk, 0, st, (const T*)q, (const T*)dh, kk, \
                           v, (T*)out, dkk, dv, dqs, rows, c, parts);                                   \
      *fused = true;                                                                                    \
    } else if (bwd)                                                                                       \
      hipLaunchKernelGGL((film_softmax_bwd_kernel<T, V, NVV, false>), grid, blk, 0, st, (const T*)q, (const T*)dh, kk, \
                         v, (T*)out, dkk, dv, nullptr, rows, c, parts);                                 \
    else                                                                                                \
      hipLaunchKernelGGL((film_softmax_fwd_kernel<T, V, NVV>), grid, blk, 0, st, (const T*)q, kk, v, (T*)out, rows, c); \
    UNET_LAUNCH_CHECK("film_softmax");                                                                  \
    return 0;                                                                                           \
  }
  FILM_CASE(1) FILM_CASE(2) FILM_CASE(4) FILM_CASE(8)
  if (V == 4) { FILM_CASE(16) }
#undef FILM_CASE
  unet_set_error("film_softmax: %d channels unsupported (max %d)", c, V == 4 ? 4096 : 512);
  return -1;
}

int film_entry(bool bwd, int dtype, const void* q, const void* dh, const float* kk, const float* v, void* out,
               float* dkk, float* dv, float* dqs, int n, int rows, int c, void* stream) {
  if (check_dims(n, rows, c, "film_softmax")) return -1;
  UNET_REQUIRE(q && kk && v && out && (!bwd || (dh && dkk && dv)), "film_softmax: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  bool fused = false;
  int rc = -2;
  if (dtype == UNET_F32) {
    const bool vec = c % 4 == 0 && aligned16(q) && aligned16(out) && (!dh || aligned16(dh));
    rc = vec ? film_dispatch<float, 4>(bwd, q, dh, kk, v, out, dkk, dv, dqs, &fused, n, rows, c, st)
               : film_dispatch<float, 1>(bwd, q, dh, kk, v, out, dkk, dv, dqs, &fused, n, rows, c, st);
  } else if (dtype == UNET_BF16) {
    const bool vec = c % 4 == 0 && ((uintptr_t)q % 8 == 0) && ((uintptr_t)out % 8 == 0) && (!dh || (uintptr_t)dh % 8 == 0);
    rc = vec ? film_dispatch<__bf16, 4>(bwd, q, dh, kk, v, out, dkk, dv, dqs, &fused, n, rows, c, st)
             : film_dispatch<__bf16, 1>(bwd, q, dh, kk, v, out, dkk, dv, dqs, &fused, n, rows, c, st);
  } else {
    unet_set_error("film_softmax: bad dtype %d", dtype);
    return -1;
  }
  if (rc == 0 && bwd && dqs && !fused)   // wide rows: the kernel has no registers to spare for it; separate pass over dq
    return unet_colsum(dtype, out, dqs, n, rows, c, stream);
  return rc;
}

// ---------------------------------------------------------------------------------------------
// layout change at the model boundary (tiled transpose through LDS, both sides coalesced)
//   NCL->NLC:  y[n][t][c] = (x[n][c][t] - sub[n][c]) * mul[n][c]          (z-score apply)
//   NLC->NCL:  y[n][c][t] = ga[n][c]*x[n][t][c] + gb[n][c]*x2[n][t][c] + be[n][c]
// ---------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ __launch_bounds__(THREADS) void ncl_to_nlc_kernel(const TI* x, TO* y, const float* sub, const float* mul,
                                                              int c, int t) {
  __shared__ float tile[64][65];
  const int n = blockIdx.z, t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
  for (int i = ly; i < 64; i += 4) {
    const int cc = c0 + i, tt = t0 + lx;
    float v = 0.f;
    if (cc < c && tt < t) {
      v = to_f32(x[((long long)n * c + cc) * t + tt]);
      if (sub) v = (v - sub[(long long)n * c + cc]) * mul[(long long)n * c + cc];
    }
    tile[i][lx] = v;
  }
  __syncthreads();
  for (int i = ly; i < 64; i += 4) {
    const int tt = t0 + i, cc = c0 + lx;
    if (cc < c && tt < t) y[((long long)n * t + tt) * c + cc] = from_f32<TO>(tile[lx][i]);
  }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(THREADS) void nlc_to_ncl_kernel(const TI* x, const TI* x2, TO* y, const float* ga,
                                                              const float* gb, const float* be, int c, int t) {
  __shared__ float tile[64][65];
  const int n = blockIdx.z, t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
  for (int i = ly; i < 64; i += 4) {
    const int tt = t0 + i, cc = c0 + lx;
    float v = 0.f;
    if (cc < c && tt < t) {
      const long long e = ((long long)n * t + tt) * c + cc;
      v = to_f32(x[e]);
      if (ga) v *= ga[(long long)n * c + cc];
      if (x2) v += gb[(long long)n * c + cc] * to_f32(x2[e]);
      if (be) v += be[(long long)n * c + cc];
    }
    tile[i][lx] = v;
  }
  __syncthreads();
  for (int i = ly; i < 64; i += 4) {
    const int cc = c0 + i, tt = t0 + lx;
    if (cc < c && tt < t) y[((long long)n * c + cc) * t + tt] = from_f32<TO>(tile[lx][i]);
  }
}

// per-(window,channel) mean and 1/(std_unbiased + eps) over the contiguous T axis of an NCL tensor
template <typename TI>
__global__ __launch_bounds__(THREADS) void ncl_rowmoments_kernel(const TI* x, float* mean, float* rstd, int t, float eps) {
  __shared__ float red[THREADS / 64];
  __shared__ float mean_s;
  const long long row = blockIdx.x;
  const TI* p = x + row * t;
  float s = 0.f;
  for (int i = threadIdx.x; i < t; i += THREADS) s += to_f32(p[i]);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) mean_s = (red[0] + red[1] + red[2] + red[3]) / (float)t;
  __syncthreads();
  const float m = mean_s;
  float q = 0.f;
  for (int i = threadIdx.x; i < t; i += THREADS) { const float d = to_f32(p[i]) - m; q += d * d; }
  q = wave_sum(q);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float var = (red[0] + red[1] + red[2] + red[3]) / (float)(t > 1 ? t - 1 : 1);
    mean[row] = m;
    rstd[row] = 1.f / (sqrtf(var) + eps);
  }
}

template <typename TI>
int transpose_in(int dtype, const void* x, void* y, const float* sub, const float* mul, int n, int c, int t, hipStream_t st) {
  dim3 grid(cdiv(t, 64), cdiv(c, 64), n), blk(THREADS);
  if (dtype == UNET_F32) hipLaunchKernelGGL((ncl_to_nlc_kernel<TI, float>), grid, blk, 0, st, (const TI*)x, (float*)y, sub, mul, c, t);
  else hipLaunchKernelGGL((ncl_to_nlc_kernel<TI, __bf16>), grid, blk, 0, st, (const TI*)x, (__bf16*)y, sub, mul, c, t);
  UNET_LAUNCH_CHECK("ncl_to_nlc");
  return 0;
}
template <typename TI>
int transpose_out(int out_dtype, const void* x, const void* x2, void* y, const float* ga, const float* gb,
                  const float* be, int n, int c, int t, hipStream_t st) {
  dim3 grid(cdiv(t, 64), cdiv(c, 64), n), blk(THREADS);
  if (out_dtype == UNET_F32) hipLaunchKernelGGL((nlc_to_ncl_kernel<TI, float>), grid, blk, 0, st, (const TI*)x, (const TI*)x2, (float*)y, ga, gb, be, c, t);
  else hipLaunchKernelGGL((nlc_to_ncl_kernel<TI, __bf16>), grid, blk, 0, st, (const TI*)x, (const TI*)x2, (__bf16*)y, ga, gb, be, c, t);
  UNET_LAUNCH_CHECK("nlc_to_ncl");
  return 0;
}

}  // namespace

extern "C" int32_t unet_parts(int32_t rows) { return cdiv(rows, rows_per_part(rows)); }

extern "C" int unet_rowstats(int32_t dtype, const void* y, float* s1_part, float* s2_part, int32_t n,
                             int32_t rows, int32_t c, void* stream) {
  if (check_dims(n, rows, c, "rowstats")) return -1;
  UNET_REQUIRE(y && s1_part && s2_part, "rowstats: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH_ROWS(dtype, c, PL(y), {
    RowStatsOp<T, V> op; op.y = (const T*)y; op.s1 = s1_part; op.s2 = s2_part;
    return launch_rows(op, n, rows, c, st, "rowstats");
  });
  return 0;
}

extern "C" int unet_colsum(int32_t dtype, const void* x, float* out_part, int32_t n, int32_t rows, int32_t c,
                           void* stream) {
  if (check_dims(n, rows, c, "colsum")) return -1;
  UNET_REQUIRE(x && out_part, "colsum: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH_ROWS(dtype, c, PL(x), {
    ColSumOp<T, V> op; op.x = (const T*)x; op.out = out_part;
    return launch_rows(op, n, rows, c, st, "colsum");
  });
  return 0;
}

extern "C" int unet_affine_act_fwd(int32_t dtype, const void* y, const float* a, const float* b, int32_t act,
                                   void* z, float* pool_part, int32_t n, int32_t rows, int32_t c, void* stream) {
  if (check_dims(n, rows, c, "affine_act_fwd")) return -1;
  UNET_REQUIRE(y && a && b && z, "affine_act_fwd: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (pool_part) {
    DISPATCH_ROWS(dtype, c, PL(y, z), {
      DISPATCH_ACT(act, {
        AffineActFwdOp<T, V, true, ACT> op; op.y = (const T*)y; op.z = (T*)z; op.a = a; op.b = b; op.pool = pool_part; op.act = act;
        return launch_rows(op, n, rows, c, st, "affine_act_fwd");
      });
    });
  } else {
    DISPATCH_ROWS(dtype, c, PL(y, z), {
      DISPATCH_ACT(act, {
        AffineActFwdOp<T, V, false, ACT> op; op.y = (const T*)y; op.z = (T*)z; op.a = a; op.b = b; op.pool = nullptr; op.act = act;
        return launch_rows(op, n, rows, c, st, "affine_act_fwd");
      });
    });
  }
  return 0;
}

extern "C" int unet_affine_act_bwd_reduce(int32_t dtype, const void* y, const void* dz, const float* dpool,
                                          const float* a, const float* b, int32_t act, float* g1_part,
                                          float* g2_part, int32_t n, int32_t rows, int32_t c, void* stream) {
  if (check_dims(n, rows, c, "affine_act_bwd_reduce")) return -1;
  UNET_REQUIRE(y && dz && a && b && g1_part && g2_part, "affine_act_bwd_reduce: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH_ROWS(dtype, c, PL(y, dz), {
    DISPATCH_ACT(act, {
      AffineActBwdOp<T, V, false, false, false, ACT> op; op.y = (const T*)y; op.dz = (const T*)dz; op.dpool = dpool; op.a = a; op.b = b;
      op.act = act; op.g1 = g1_part; op.g2 = g2_part; op.ca = op.cb = op.cc = nullptr; op.dy = nullptr; op.dysum = nullptr;
      return launch_rows(op, n, rows, c, st, "affine_act_bwd_reduce");
    });
  });
  return 0;
}

extern "C" int unet_affine_act_bwd_apply(int32_t dtype, const void* y, const void* dz, const float* dpool,
                                         const float* a, const float* b, int32_t act, const float* ca,
                                         const float* cb, const float* cc, void* dy, int32_t n, int32_t rows,
                                         int32_t c, void* stream) {
  if (check_dims(n, rows, c, "affine_act_bwd_apply")) return -1;
  UNET_REQUIRE(y && dz && a && b && ca && cb && cc && dy, "affine_act_bwd_apply: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH_ROWS(dtype, c, PL(y, dz, dy), {
    DISPATCH_ACT(act, {
      AffineActBwdOp<T, V, true, false, false, ACT> op; op.y = (const T*)y; op.dz = (const T*)dz; op.dpool = dpool; op.a = a; op.b = b;
      op.act = act; op.g1 = op.g2 = nullptr; op.ca = ca; op.cb = cb; op.cc = cc; op.dy = (T*)dy; op.dysum = nullptr;
      return launch_rows(op, n, rows, c, st, "affine_act_bwd_apply");
    });
  });
  return 0;
}

extern "C" int unet_affine_act_bwd_apply_sum(int32_t dtype, const void* y, const void* dz, const float* dpool,
                                             const float* a, const float* b, int32_t act, const float* ca,
                                             const float* cb, const float* cc, void* dy, float* dy_sum_part, int32_t n,
                                             int32_t rows, int32_t c, void* stream) {
  if (!dy_sum_part) return unet_affine_act_bwd_apply(dtype, y, dz, dpool, a, b, act, ca, cb, cc, dy, n, rows, c, stream);
  if (check_dims(n, rows, c, "affine_act_bwd_apply_sum")) return -1;
  UNET_REQUIRE(y && dz && a && b && ca && cb && cc && dy, "affine_act_bwd_apply_sum: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH_ROWS(dtype, c, PL(y, dz, dy), {
    DISPATCH_ACT(act, {
      AffineActBwdOp<T, V, true, true, false, ACT> op; op.y = (const T*)y; op.dz = (const T*)dz; op.dpool = dpool; op.a = a; op.b = b;
      op.act = act; op.g1 = op.g2 = nullptr; op.ca = ca; op.cb = cb; op.cc = cc; op.dy = (T*)dy; op.dysum = dy_sum_part;
      return launch_rows(op, n, rows, c, st, "affine_act_bwd_apply_sum");
    });
  });
  return 0;
}

extern "C" int unet_affine_pool(int32_t dtype, const void* y, const float* a, const float* b, int32_t act,
                                float* pool_part, int32_t n, int32_t rows, int32_t c, void* stream) {
  if (check_dims(n, rows, c, "affine_pool")) return -1;
  UNET_REQUIRE(y && a && b && pool_part, "affine_pool: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH_ROWS(dtype, c, PL(y), {
    DISPATCH_ACT(act, {
      AffinePoolOp<T, V, ACT> op; op.y = (const T*)y; op.a = a; op.b = b; op.pool = pool_part; op.act = act;
      return launch_rows(op, n, rows, c, st, "affine_pool");
    });
  });
  return 0;
}

extern "C" int unet_affine_act_scale_fwd(int32_t dtype, const void* y, const float* a, const float* b, int32_t act,
                                         const float* s, void* z, int32_t n, int32_t rows, int32_t c, void* stream) {
  if (check_dims(n, rows, c, "affine_act_scale_fwd")) return -1;
  UNET_REQUIRE(y && a && b && s && z, "affine_act_scale_fwd: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH_ROWS(dtype, c, PL(y, z), {
    DISPATCH_ACT(act, {
      AffineActScaleFwdOp<T, V, ACT> op; op.y = (const T*)y; op.z = (T*)z; op.a = a; op.b = b; op.sc = s; op.act = act;
      return launch_rows(op, n, rows, c, st, "affine_act_scale_fwd");
    });
  });
  return 0;
}

extern "C" int unet_affine_act_se_bwd_reduce(int32_t dtype, const void* y, const void* dz, const float* a, const float* b,
                                             int32_t act, float* p_part, int32_t n, int32_t rows, int32_t c, void* stream) {
  if (check_dims(n, rows, c, "affine_act_se_bwd_reduce")) return -1;
  UNET_REQUIRE(y && dz && a && b && p_part, "affine_act_se_bwd_reduce: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH_ROWS(dtype, c, PL(y, dz), {
    DISPATCH_ACT(act, {
      AffineActSEBwdReduceOp<T, V, ACT> op; op.y = (const T*)y; op.dz = (const T*)dz; op.a = a; op.b = b; op.act = act;
      op.p = p_part; op.pstride = (long long)n * cdiv(rows, rows_per_part(rows)) * c;
      return launch_rows(op, n, rows, c, st, "affine_act_se_bwd_reduce");
    });
  });
  return 0;
}

extern "C" int unet_se_combine(const float* p, const float* s, const float* dpool, float* g1, float* g2, int32_t n,
                               int32_t c, void* stream) {
  UNET_REQUIRE(p && s && g1 && g2 && n > 0 && c > 0, "se_combine: bad arguments");
  const long long nc = (long long)n * c;
  hipLaunchKernelGGL(se_combine_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), p,
                     nc, s, dpool, g1, g2);
  UNET_LAUNCH_CHECK("se_combine");
  return 0;
}

extern "C" int unet_affine_act_se_bwd_apply(int32_t dtype, const void* y, const void* dz, const float* dpool,
                                            const float* s, const float* a, const float* b, int32_t act, const float* ca,
                                            const float* cb, const float* cc, void* dy, float* dy_sum_part, int32_t n,
                                            int32_t rows, int32_t c, void* stream) {
  if (check_dims(n, rows, c, "affine_act_se_bwd_apply")) return -1;
  UNET_REQUIRE(y && dz && s && a && b && ca && cb && cc && dy && dy_sum_part, "affine_act_se_bwd_apply: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH_ROWS(dtype, c, PL(y, dz, dy), {
    DISPATCH_ACT(act, {
      AffineActBwdOp<T, V, true, true, true, ACT> op; op.y = (const T*)y; op.dz = (const T*)dz; op.dpool = dpool; op.sc = s;
      op.a = a; op.b = b; op.act = act; op.g1 = op.g2 = nullptr; op.ca = ca; op.cb = cb; op.cc = cc; op.dy = (T*)dy;
      op.dysum = dy_sum_part;
      return launch_rows(op, n, rows, c, st, "affine_act_se_bwd_apply");
    });
  });
  return 0;
}

extern "C" int unet_scale_fwd(int32_t dtype, const void* x, const float* s, void* out, int32_t n, int32_t rows,
                              int32_t c, void* stream) {
  if (check_dims(n, rows, c, "scale_fwd")) return -1;
  UNET_REQUIRE(x && s && out, "scale_fwd: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH_ROWS(dtype, c, PL(x, out), {
    ScaleFwdOp<T, V> op; op.x = (const T*)x; op.out = (T*)out; op.s = s;
    return launch_rows(op, n, rows, c, st, "scale_fwd");
  });
  return 0;
}

extern "C" int unet_scale_bwd(int32_t dtype, const void* x, const void* dout, const float* s, void* dx,
                              float* ds_part, int32_t n, int32_t rows, int32_t c, void* stream) {
  if (check_dims(n, rows, c, "scale_bwd")) return -1;
  UNET_REQUIRE(x && dout && s && dx && ds_part, "scale_bwd: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH_ROWS(dtype, c, PL(x, dout, dx), {
    ScaleBwdOp<T, V> op; op.x = (const T*)x; op.dout = (const T*)dout; op.dx = (T*)dx; op.s = s; op.ds = ds_part;
    return launch_rows(op, n, rows, c, st, "scale_bwd");
  });
  return 0;
}

extern "C" int unet_gated_add_fwd(int32_t dtype, const void* res, const void* o, const float* g, void* out,
                                  int32_t n, int32_t rows, int32_t c, void* stream) {
  if (check_dims(n, rows, c, "gated_add_fwd")) return -1;
  UNET_REQUIRE(res && o && g && out, "gated_add_fwd: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH_ROWS(dtype, c, PL(res, o, out), {
    GatedAddFwdOp<T, V> op; op.res = (const T*)res; op.o = (const T*)o; op.out = (T*)out; op.g = g;
    return launch_rows(op, n, rows, c, st, "gated_add_fwd");
  });
  return 0;
}

extern "C" int unet_gated_add_bwd(int32_t dtype, const void* o, const void* dout, const float* g, void* d_o,
                                  float* dg_part, float* do_sum_part, int32_t n, int32_t rows, int32_t c, void* stream) {
  if (check_dims(n, rows, c, "gated_add_bwd")) return -1;
  UNET_REQUIRE(o && dout && g && d_o && dg_part, "gated_add_bwd: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  DISPATCH_ROWS(dtype, c, PL(o, dout, d_o), {
    if (do_sum_part) {
      GatedAddBwdOp<T, V, true> op; op.o = (const T*)o; op.dout = (const T*)dout; op.d_o = (T*)d_o; op.g = g;
      op.dg = dg_part; op.dsum = do_sum_part;
      return launch_rows(op, n, rows, c, st, "gated_add_bwd");
    }
    GatedAddBwdOp<T, V, false> op; op.o = (const T*)o; op.dout = (const T*)dout; op.d_o = (T*)d_o; op.g = g;
    op.dg = dg_part; op.dsum = nullptr;
    return launch_rows(op, n, rows, c, st, "gated_add_bwd");
  });
  return 0;
}

extern "C" int unet_film_softmax_fwd(int32_t dtype, const void* q, const float* kk, const float* v, void* h,
                                     int32_t n, int32_t rows, int32_t c, void* stream) {
  return film_entry(false, dtype, q, nullptr, kk, v, h, nullptr, nullptr, nullptr, n, rows, c, stream);
}
extern "C" int unet_film_softmax_bwd(int32_t dtype, const void* q, const void* dh, const float* kk,
                                     const float* v, void* dq, float* dkk_part, float* dv_part, float* dq_sum_part,
                                     int32_t n, int32_t rows, int32_t c, void* stream) {
  return film_entry(true, dtype, q, dh, kk, v, dq, dkk_part, dv_part, dq_sum_part, n, rows, c, stream);
}

extern "C" int unet_ncl_moments(int32_t in_dtype, const void* x, float* mean, float* rstd, int32_t n, int32_t c,
                                int32_t t, float eps, void* stream) {
  UNET_REQUIRE(x && mean && rstd && n > 0 && c > 0 && t > 0, "ncl_moments: bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long rows = (long long)n * c;
  UNET_REQUIRE(rows < (1ll << 31), "ncl_moments: too many rows");
  if (in_dtype == UNET_F32) hipLaunchKernelGGL(ncl_rowmoments_kernel<float>, dim3((unsigned)rows), dim3(THREADS), 0, st, (const float*)x, mean, rstd, t, eps);
  else if (in_dtype == UNET_BF16) hipLaunchKernelGGL(ncl_rowmoments_kernel<__bf16>, dim3((unsigned)rows), dim3(THREADS), 0, st, (const __bf16*)x, mean, rstd, t, eps);
  else { unet_set_error("ncl_moments: bad dtype %d", in_dtype); return -1; }
  UNET_LAUNCH_CHECK("ncl_moments");
  return 0;
}

extern "C" int unet_ncl_to_nlc(int32_t in_dtype, int32_t dtype, const void* x, const float* sub, const float* mul,
                               void* y, int32_t n, int32_t c, int32_t t, void* stream) {
  UNET_REQUIRE(x && y && n > 0 && c > 0 && t > 0 && n <= 65535, "ncl_to_nlc: bad arguments");
  UNET_REQUIRE((sub == nullptr) == (mul == nullptr), "ncl_to_nlc: sub and mul must both be given or both be null");
  UNET_REQUIRE(dtype == UNET_F32 || dtype == UNET_BF16, "ncl_to_nlc: bad dtype %d", dtype);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (in_dtype == UNET_F32) return transpose_in<float>(dtype, x, y, sub, mul, n, c, t, st);
  if (in_dtype == UNET_BF16) return transpose_in<__bf16>(dtype, x, y, sub, mul, n, c, t, st);
  unet_set_error("ncl_to_nlc: bad in_dtype %d", in_dtype);
  return -1;
}

extern "C" int unet_nlc_to_ncl(int32_t dtype, int32_t out_dtype, const void* x, const void* x2, const float* ga,
                               const float* gb, const float* be, void* y, int32_t n, int32_t c, int32_t t,
                               void* stream) {
  UNET_REQUIRE(x && y && n > 0 && c > 0 && t > 0 && n <= 65535, "nlc_to_ncl: bad arguments");
  UNET_REQUIRE((x2 == nullptr) == (gb == nullptr), "nlc_to_ncl: x2 and gb must both be given or both be null");
  UNET_REQUIRE(out_dtype == UNET_F32 || out_dtype == UNET_BF16, "nlc_to_ncl: bad out_dtype %d", out_dtype);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == UNET_F32) return transpose_out<float>(out_dtype, x, x2, y, ga, gb, be, n, c, t, st);
  if (dtype == UNET_BF16) return transpose_out<__bf16>(out_dtype, x, x2, y, ga, gb, be, n, c, t, st);
  unet_set_error("nlc_to_ncl: bad dtype %d", dtype);
  return -1;
}

extern "C" int unet_bn_finalize_fwd(const float* s1_part, const float* s2_part, int32_t n, int32_t parts, int32_t c,
                                    int32_t rows, float eps, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, float momentum, float* mean,
                                    float* rstd, float* a, float* b, int64_t* batches_tracked, void* stream) {
  UNET_REQUIRE(s1_part && s2_part && mean && rstd && a && b && n > 0 && parts > 0 && c > 0 && rows > 0,
               "bn_finalize_fwd: bad arguments");
  UNET_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize_fwd: running stats must come in pairs");
  hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(cdiv(c, CS_CH)), dim3(CS_THREADS), 0, static_cast<hipStream_t>(stream), s1_part,
                     s2_part, n, parts, c, (double)n * rows, eps, gamma, beta, running_mean, running_var, momentum, mean,
                     rstd, a, b, reinterpret_cast<long long*>(batches_tracked));
  UNET_LAUNCH_CHECK("bn_finalize_fwd");
  return 0;
}

extern "C" int unet_bn_finalize_bwd(const float* g1_part, const float* g2_part, int32_t n, int32_t parts, int32_t c,
                                    int32_t rows, const float* mean, const float* rstd, const float* gamma, float* ca,
                                    float* cb, float* cc, float* dgamma, float* dbeta, void* stream) {
  UNET_REQUIRE(g1_part && g2_part && mean && rstd && ca && cb && cc && n > 0 && parts > 0 && c > 0 && rows > 0,
               "bn_finalize_bwd: bad arguments");
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(cdiv(c, CS_CH)), dim3(CS_THREADS), 0, static_cast<hipStream_t>(stream), g1_part,
                     g2_part, n, parts, c, (double)n * rows, mean, rstd, gamma, ca, cb, cc, dgamma, dbeta);
  UNET_LAUNCH_CHECK("bn_finalize_bwd");
  return 0;
}

extern "C" int unet_reduce_rows(const float* in, int32_t rows, int32_t c, float* out, void* stream) {
  UNET_REQUIRE(in && out && rows > 0 && c > 0, "reduce_rows: bad arguments");
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(cdiv(c, CS_CH)), dim3(CS_THREADS), 0, static_cast<hipStream_t>(stream), in, rows, c, out);
  UNET_LAUNCH_CHECK("reduce_rows");
  return 0;
}

extern "C" int unet_reduce_multi(const UnetReduceJob* jobs, int32_t n_jobs, void* stream) {
  UNET_REQUIRE(jobs && n_jobs >= 1 && n_jobs <= UNET_REDUCE_MAX_JOBS, "reduce_multi: 1..%d jobs per call, got %d",
               UNET_REDUCE_MAX_JOBS, n_jobs);
  ReduceM m;
  m.n = n_jobs;
  int cmax = 0;
  long long units = 0;
  for (int i = 0; i < UNET_REDUCE_MAX_JOBS; ++i) {
    m.j[i] = jobs[i < n_jobs ? i : 0];
    m.first[i] = (int)units;
    if (i < n_jobs) {
      const UnetReduceJob& j = jobs[i];
      UNET_REQUIRE(j.in && j.out && j.units > 0 && j.rows > 0 && j.c > 0, "reduce_multi: bad job %d", i);
      units += j.units;
      cmax = j.c > cmax ? j.c : cmax;
    }
  }
  m.first[UNET_REDUCE_MAX_JOBS] = (int)units;
  UNET_REQUIRE(units <= 65535, "reduce_multi: %lld units in one call (max 65535)", units);
  hipLaunchKernelGGL(reduce_multi_kernel, dim3(cdiv(cmax, CS_CH), (unsigned)units), dim3(CS_THREADS), 0,
                     static_cast<hipStream_t>(stream), m);
  UNET_LAUNCH_CHECK("reduce_multi");
  return 0;
}

extern "C" int unet_reduce_parts(const float* in, int32_t n, int32_t parts, int32_t c, float* out, void* stream) {
  UNET_REQUIRE(in && out && n > 0 && n <= 65535 && parts > 0 && c > 0, "reduce_parts: bad arguments");
  hipLaunchKernelGGL(reduce_parts_kernel, dim3(cdiv(c, 256), n), dim3(256), 0, static_cast<hipStream_t>(stream), in, parts, c, out);
  UNET_LAUNCH_CHECK("reduce_parts");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Merged-tap weights of MultiScaleDilatedConv1d (models.py:752-759).  The softmax-weighted sum of the
// parallel dilated branches is one conv whose tap t has  W_eff[t] = sum_{(b,k) -> t} sw[b] * W_b[:, :, k].
// One thread per (co, ci): reads the B*K branch weights of that pair (K contiguous floats per branch),
// writes the T merged taps (coalesced across threads).  Backward scatters dW_eff back to the branches
// and reduces d sw[b] = <dW_eff[t(b,k)], W_b[..., k]> per block (deterministic partials).
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int MT_MAXB = 8, MT_MAXK = 9;
struct MergeK {
  const float* w[MT_MAXB];
  float* dw[MT_MAXB];
  const float* bias[MT_MAXB];   // optional: branch biases [co]  (b_eff = sum_b sw[b] * bias_b)
  float* dbias[MT_MAXB];
  int tap[MT_MAXB * MT_MAXK];
  int nb, k, n_taps, co;
  int softmax;                  // sw arrives raw: the kernel applies softmax over the branches itself
  long long pairs;   // co * ci
};

// the branch weights: softmax(raw) over <= 8 values, or the values as they are
__device__ __forceinline__ void branch_weights(const MergeK& p, const float* sw, float (&s)[MT_MAXB]) {
  float m = -INFINITY;
#pragma unroll
  for (int b = 0; b < MT_MAXB; ++b) { s[b] = b < p.nb ? sw[b] : -INFINITY; m = fmaxf(m, s[b]); }
  if (p.softmax) {
    float sum = 0.f;
#pragma unroll
    for (int b = 0; b < MT_MAXB; ++b) { s[b] = b < p.nb ? expf(s[b] - m) : 0.f; sum += s[b]; }
#pragma unroll
    for (int b = 0; b < MT_MAXB; ++b) s[b] /= sum;
  }
}

// A block owns 256 (co, ci) pairs.  Branch weights are [co][ci][k]: the block's 256 * k floats of a branch are one
// contiguous run, moved to and from LDS with unit-stride accesses; the thread of a pair then works on its k values
// there (stride k floats: conflict-free for the odd kernel sizes in use).  As per-thread strided global loads in
// run-time loops plus a 32-way select chain per tap the two kernels took 37 / 35 us per conv.
__global__ __launch_bounds__(256) void merge_taps_fwd_kernel(MergeK p, const float* sw, float* w_eff, float* b_eff,
                                                             float* sw_out) {
  __shared__ float wt[256 * MT_MAXK];
  __shared__ float acc[UNET_MAX_TAPS * 256];
  const int tid = threadIdx.x;
  float sb[MT_MAXB];
  branch_weights(p, sw, sb);
  if (blockIdx.x == 0) {                   // the few per-channel / per-branch by-products ride on the first block
    if (sw_out && tid < p.nb) {
#pragma unroll
      for (int b = 0; b < MT_MAXB; ++b) if (b == tid) sw_out[b] = sb[b];
    }
    if (b_eff)
      for (int c = tid; c < p.co; c += 256) {
        float v = 0.f;
#pragma unroll
        for (int b = 0; b < MT_MAXB; ++b) if (b < p.nb) v += sb[b] * p.bias[b][c];
        b_eff[c] = v;
      }
  }
  const long long i0 = (long long)blockIdx.x * 256;
  const int cnt = (int)(p.pairs - i0 < 256 ? p.pairs - i0 : 256);     // pairs of this block
  for (int t = 0; t < p.n_taps; ++t) acc[t * 256 + tid] = 0.f;
  for (int b = 0; b < p.nb; ++b) {
    const float* wb = p.w[b] + i0 * p.k;
    float r[MT_MAXK];
#pragma unroll
    for (int j = 0; j < MT_MAXK; ++j) {
      const int e = j * 256 + tid;
      const bool ok = j < p.k && e < cnt * p.k;
      r[j] = wb[ok ? e : 0];
      r[j] = ok ? r[j] : 0.f;
    }
    __syncthreads();                       // the previous branch's reads of wt are done
#pragma unroll
    for (int j = 0; j < MT_MAXK; ++j)
      if (j < p.k) wt[j * 256 + tid] = r[j];
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MT_MAXB; ++j) if (j == b) s = sb[j];
    for (int k = 0; k < p.k; ++k) acc[p.tap[b * p.k + k] * 256 + tid] += s * wt[tid * p.k + k];   // own column only
  }
  if (tid < cnt)
    for (int t = 0; t < p.n_taps; ++t) w_eff[(long long)t * p.pairs + i0 + tid] = acc[t * 256 + tid];
}

// sw: the (softmaxed) branch weights the forward pass used.  db_eff (optional): gradient of b_eff; block 0 then also
// writes d bias_b = sw[b] * db_eff and adds <db_eff, bias_b> to its partial of d sw[b].
__global__ __launch_bounds__(256) void merge_taps_bwd_kernel(MergeK p, const float* sw, const float* dw_eff, float* dsw_part,
                                                             const float* db_eff) {
  __shared__ float red[MT_MAXB][256 / 64];
  __shared__ float wt[256 * MT_MAXK], gt[256 * MT_MAXK];
  const int tid = threadIdx.x;
  const long long i0 = (long long)blockIdx.x * 256;
  const int cnt = (int)(p.pairs - i0 < 256 ? p.pairs - i0 : 256);
  const bool ok_i = tid < cnt;
  for (int b = 0; b < p.nb; ++b) {
    const float* wb = p.w[b] + i0 * p.k;
    float* dwb = p.dw[b] + i0 * p.k;
    float r[MT_MAXK], g[MT_MAXK];
#pragma unroll
    for (int j = 0; j < MT_MAXK; ++j) {
      const int e = j * 256 + tid;
      const bool ok = j < p.k && e < cnt * p.k;
      r[j] = wb[ok ? e : 0];
      r[j] = ok ? r[j] : 0.f;
      const bool okg = j < p.k && ok_i;                                   // g[j]: tap of (b, k = j), this pair
      g[j] = dw_eff[okg ? (long long)p.tap[b * p.k + j] * p.pairs + i0 + tid : 0];
      g[j] = okg ? g[j] : 0.f;
    }
    __syncthreads();                       // the previous branch's reads of wt / gt are done
#pragma unroll
    for (int j = 0; j < MT_MAXK; ++j)
      if (j < p.k) wt[j * 256 + tid] = r[j];
    __syncthreads();
    const float s = sw[b];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < MT_MAXK; ++j)
      if (j < p.k) {
        dot += g[j] * wt[tid * p.k + j];
        gt[tid * p.k + j] = s * g[j];        // d W_b[pair][k], laid out as the branch weight is
      }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MT_MAXK; ++j) {
      const int e = j * 256 + tid;
      if (j < p.k && e < cnt * p.k) dwb[e] = gt[e];
    }
    if (db_eff && blockIdx.x == 0)
      for (int c = tid; c < p.co; c += 256) {
        const float g = db_eff[c];
        p.dbias[b][c] = s * g;
        dot += g * p.bias[b][c];
      }
    dot = wave_sum(dot);
    if ((tid & 63) == 0) red[b][tid >> 6] = dot;
  }
  __syncthreads();
  if (tid < p.nb)
    dsw_part[(long long)blockIdx.x * p.nb + tid] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
}

// d sw[b] = sum of the per-block partials (fp64), then through the branch softmax: d raw = s o (d sw - <s, d sw>)
__global__ __launch_bounds__(256) void merge_taps_finish_kernel(const float* dsw_part, int parts, int nb, const float* sw,
                                                                float* d_out, int softmax) {
  __shared__ double red[MT_MAXB][256];
  const int tid = threadIdx.x;
  for (int b = 0; b < nb; ++b) {
    double a = 0.0;
    for (int r = tid; r < parts; r += 256) a += dsw_part[(long long)r * nb + b];
    red[b][tid] = a;
  }
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o)
      for (int b = 0; b < nb; ++b) red[b][tid] += red[b][tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    double dot = 0.0;
    for (int b = 0; b < nb; ++b) dot += (double)sw[b] * red[b][0];
    for (int b = 0; b < nb; ++b) d_out[b] = (float)(softmax ? (double)sw[b] * (red[b][0] - dot) : red[b][0]);
  }
}

int merge_setup(MergeK& p, const float* const* w_host, float* const* dw_host, int nb, int k, int co, int ci,
                const int32_t* tap_host, int n_taps) {
  UNET_REQUIRE(nb > 0 && nb <= MT_MAXB && k > 0 && k <= MT_MAXK && co > 0 && ci > 0, "merge_taps: bad sizes nb=%d k=%d", nb, k);
  UNET_REQUIRE(n_taps > 0 && n_taps <= UNET_MAX_TAPS && tap_host && w_host, "merge_taps: bad tap table");
  p.nb = nb; p.k = k; p.n_taps = n_taps; p.co = co; p.softmax = 0; p.pairs = (long long)co * ci;
  for (int b = 0; b < MT_MAXB; ++b) {
    p.w[b] = b < nb ? w_host[b] : nullptr; p.dw[b] = (dw_host && b < nb) ? dw_host[b] : nullptr;
    p.bias[b] = nullptr; p.dbias[b] = nullptr;
  }
  for (int j = 0; j < MT_MAXB * MT_MAXK; ++j) p.tap[j] = 0;
  for (int j = 0; j < nb * k; ++j) {
    UNET_REQUIRE(tap_host[j] >= 0 && tap_host[j] < n_taps, "merge_taps: tap index %d out of range", tap_host[j]);
    p.tap[j] = tap_host[j];
  }
  for (int b = 0; b < nb; ++b) UNET_REQUIRE(p.w[b] != nullptr, "merge_taps: null branch weight");
  UNET_REQUIRE((p.pairs + 255) / 256 < (1ll << 31), "merge_taps: too many weights");
  return 0;
}
}  // namespace

extern "C" int32_t unet_merge_taps_parts(int32_t co, int32_t ci) { return (int32_t)(((long long)co * ci + 255) / 256); }

extern "C" int unet_merge_branches_fwd(const float* const* w_branch_host, const float* const* bias_branch_host,
                                       const float* sw, int32_t softmax, int32_t n_branch, int32_t k, int32_t co, int32_t ci,
                                       const int32_t* tap_index_host, int32_t n_taps, float* w_eff, float* b_eff,
                                       float* sw_out, void* stream) {
  MergeK p;
  if (merge_setup(p, w_branch_host, nullptr, n_branch, k, co, ci, tap_index_host, n_taps)) return -1;
  UNET_REQUIRE(sw && w_eff, "merge_taps_fwd: null pointer");
  UNET_REQUIRE((bias_branch_host == nullptr) == (b_eff == nullptr), "merge_taps_fwd: biases and b_eff come together");
  p.softmax = softmax ? 1 : 0;
  if (bias_branch_host)
    for (int b = 0; b < n_branch; ++b) {
      UNET_REQUIRE(bias_branch_host[b] != nullptr, "merge_taps_fwd: null branch bias");
      p.bias[b] = bias_branch_host[b];
    }
  hipLaunchKernelGGL(merge_taps_fwd_kernel, dim3((unsigned)((p.pairs + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p, sw, w_eff, b_eff, sw_out);
  UNET_LAUNCH_CHECK("merge_taps_fwd");
  return 0;
}

extern "C" int unet_merge_taps_fwd(const float* const* w_branch_host, const float* sw, int32_t n_branch, int32_t k,
                                   int32_t co, int32_t ci, const int32_t* tap_index_host, int32_t n_taps, float* w_eff,
                                   void* stream) {
  return unet_merge_branches_fwd(w_branch_host, nullptr, sw, 0, n_branch, k, co, ci, tap_index_host, n_taps, w_eff, nullptr,
                                 nullptr, stream);
}

extern "C" int unet_merge_branches_bwd(const float* const* w_branch_host, const float* const* bias_branch_host,
                                       const float* sw, int32_t softmax, int32_t n_branch, int32_t k, int32_t co, int32_t ci,
                                       const int32_t* tap_index_host, int32_t n_taps, const float* dw_eff,
                                       const float* db_eff, float* const* dw_branch_host, float* const* dbias_branch_host,
                                       float* dsw_part, float* dsw, void* stream) {
  MergeK p;
  if (merge_setup(p, w_branch_host, dw_branch_host, n_branch, k, co, ci, tap_index_host, n_taps)) return -1;
  UNET_REQUIRE(sw && dw_eff && dw_branch_host && dsw_part, "merge_taps_bwd: null pointer");
  for (int b = 0; b < n_branch; ++b) UNET_REQUIRE(p.dw[b] != nullptr, "merge_taps_bwd: null branch gradient");
  const bool with_bias = db_eff != nullptr;
  UNET_REQUIRE(!with_bias || (bias_branch_host && dbias_branch_host), "merge_taps_bwd: db_eff without the branch biases");
  if (with_bias)
    for (int b = 0; b < n_branch; ++b) {
      UNET_REQUIRE(bias_branch_host[b] && dbias_branch_host[b], "merge_taps_bwd: null branch bias");
      p.bias[b] = bias_branch_host[b];
      p.dbias[b] = dbias_branch_host[b];
    }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int parts = (int)((p.pairs + 255) / 256);
  hipLaunchKernelGGL(merge_taps_bwd_kernel, dim3((unsigned)parts), dim3(256), 0, st, p, sw, dw_eff, dsw_part,
                     with_bias ? db_eff : nullptr);
  UNET_LAUNCH_CHECK("merge_taps_bwd");
  if (dsw) {
    hipLaunchKernelGGL(merge_taps_finish_kernel, dim3(1), dim3(256), 0, st, dsw_part, parts, n_branch, sw, dsw, softmax ? 1 : 0);
    UNET_LAUNCH_CHECK("merge_taps_finish");
  }
  return 0;
}

extern "C" int unet_merge_taps_bwd(const float* const* w_branch_host, const float* sw, int32_t n_branch, int32_t k,
                                   int32_t co, int32_t ci, const int32_t* tap_index_host, int32_t n_taps,
                                   const float* dw_eff, float* const* dw_branch_host, float* dsw_part, void* stream) {
  return unet_merge_branches_bwd(w_branch_host, nullptr, sw, 0, n_branch, k, co, ci, tap_index_host, n_taps, dw_eff, nullptr,
                                 dw_branch_host, nullptr, dsw_part, nullptr, stream);
}
