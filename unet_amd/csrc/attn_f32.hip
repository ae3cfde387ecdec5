// fp32 band-pair attention: the same operator as csrc/attn.hip (CrossFrequencyCouplingAttentionV2, models.py:395-462;
// SelfAttention1D, models.py:75-97 as one band of heads that share q and k), for the fp32 parity runs:
//   out[n, l, i*C + head*hd + d] = sum_j coupling[i][j] * softmax_keys(q_i k_j^T * scale) v_j
// Exact fp32 products on v_mfma_f32_16x16x4_f32 (an fmaf chain per output, like the fp32 conv kernels), flash style:
// nothing of size L x L is stored.  These kernels exist so that no dtype of the hot path falls back to a library;
// they are laid out for clarity, not tuned (parity runs are small): single-buffered K/V tiles of 32 keys, one
// barrier pair per tile.
//
// Orientation, as in the bf16 kernels: S^T = K Q^T puts a query row on lane (l & 15) and four keys in the
// accumulator registers of lane group g = l >> 4 (keys 4g .. 4g+3 of a 16-key sub-tile).  Register t of that
// accumulator is therefore, as it stands, the B operand (k index g <-> key 4g + t) of the k = 4 step
//   O^T[d][q] += V[key 4g+t][d] * P^T[key 4g+t][q]
// whose A operand is one scalar LDS read per lane.  The 16 x hd operand rows of Q (and of K / V / dO in the
// backward kernels) sit in registers as float4 chunks [16u + 4g, 16u + 4g + 4): chunk element e of step (u, e) pairs
// k index g with column 16u + 4g + e on BOTH operands, so the dot product over hd is complete and every operand
// load is 16 bytes.
#include "common.h"

namespace {

constexpr int A32_THREADS = 256;
constexpr int A32_ROWS = 64;     // rows owned by a workgroup (16 per wave)
constexpr int A32_BK = 32;       // rows of a streamed tile

struct Attn32K {
  const float *q, *k, *v;    // [nb][N][L][C]
  const float* coupling;     // [nb*nb] (already softmaxed)
  float* out;                // [N][L][nb*C]
  float* o_pairs;            // [nb][nb][N][L][C] or null
  float* lse;                // [nb][nb][N][nh][L] or null; log2 domain: p = exp2(s*scale_log2 - lse)
  const float* dout;         // [N][L][nb*C]
  float* delta;              // [nb][nb][N][nh][L]
  float *dq, *dk, *dv;       // [nb][N][L][C]
  int n, l, c, nh, nb, hd;
  float scale, scale_log2;
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// [rows][HDP] tile of a [L][C] slab (head columns hoff .. hoff+hd) -> LDS with row pitch HDP+4; rows past the end and
// columns past hd arrive as zeros
template <int HDP, int ROWS>
__device__ __forceinline__ void load_tile(float* tile, const float* slab, int row0, int l, int ld, int hd, int tid) {
  constexpr int LD = HDP + 4, CH = HDP / 4;
  for (int i = tid; i < ROWS * CH; i += A32_THREADS) {
    const int r = i / CH, ch = i - r * CH;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row0 + r < l && ch * 4 < hd) v = *reinterpret_cast<const f32x4*>(slab + (long long)(row0 + r) * ld + ch * 4);
    *reinterpret_cast<f32x4*>(tile + r * LD + ch * 4) = v;
  }
}

// the wave's 16 operand rows as float4 chunks (see the header): reg[u] = row[16u + 4g .. +3]
template <int HDP>
__device__ __forceinline__ void load_rows(f32x4 (&reg)[HDP / 16], const float* rowp, bool ok, int g, int hd) {
#pragma unroll
  for (int u = 0; u < HDP / 16; ++u) {
    const int col = 16 * u + 4 * g;
    reg[u] = (ok && col < hd) ? *reinterpret_cast<const f32x4*>(rowp + col) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// C[row 4g+i][col l15] (+)= sum_k A[row][k] B[col][k]: A rows from an LDS tile (16 rows starting at `arow`), B rows from
// the register chunks
template <int HDP>
__device__ __forceinline__ f32x4 rows_dot(const float* tile, int arow, const f32x4 (&breg)[HDP / 16], int l15, int g) {
  constexpr int LD = HDP + 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < HDP / 16; ++u) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(tile + (arow + l15) * LD + 16 * u + 4 * g);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = mfma4(a[e], breg[u][e], acc);
  }
  return acc;
}

// acc^T[d = 16dt + 4g + i][col l15] += sum over the 16 rows (arow ..) of  tile[row][d] * b[row][col], where b is an
// accumulator in the C layout (rows 4g+i): register t <-> k index g of step t
template <int HDP>
__device__ __forceinline__ void rows_acc(f32x4 (&acc)[HDP / 16], const float* tile, int arow, const f32x4& b, int l15, int g) {
  constexpr int LD = HDP + 4;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float* rp = tile + (arow + 4 * g + t) * LD + l15;
#pragma unroll
    for (int dt = 0; dt < HDP / 16; ++dt) acc[dt] = mfma4(rp[16 * dt], b[t], acc[dt]);
  }
}

__device__ __forceinline__ float group_max(float v) {      // over the 4 lane groups that share (l & 15)
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// ---------------------------------------------------------------------------------------------
// forward: a workgroup owns 64 query rows of one (window, head, query band) and walks the key/value bands
// ---------------------------------------------------------------------------------------------
template <int HDP>
__global__ __launch_bounds__(A32_THREADS, 1) void attn32_fwd_kernel(const Attn32K p) {
  constexpr int LD = HDP + 4, DT = HDP / 16;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* kt = reinterpret_cast<float*>(smem_raw);
  float* vt = kt + A32_BK * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  int wg = blockIdx.x;
  const int nqt = (p.l + A32_ROWS - 1) / A32_ROWS;
  const int qt = wg % nqt; wg /= nqt;
  const int bi = wg % p.nb; wg /= p.nb;
  const int head = wg % p.nh;
  const int n = wg / p.nh;
  const long long slab = (long long)p.l * p.c;
  const int hoff = head * p.hd;
  const int qrow = qt * A32_ROWS + wave * 16 + l15;
  const bool row_ok = qrow < p.l;

  f32x4 qreg[DT], mix[DT];
  load_rows<HDP>(qreg, p.q + ((long long)bi * p.n + n) * slab + (long long)qrow * p.c + hoff, row_ok, g, p.hd);
#pragma unroll
  for (int d = 0; d < DT; ++d) mix[d] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int j = 0; j < p.nb; ++j) {
    const float* kslab = p.k + ((long long)j * p.n + n) * slab + hoff;
    const float* vslab = p.v + ((long long)j * p.n + n) * slab + hoff;
    f32x4 oacc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) oacc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, lsum = 0.f;
    for (int key0 = 0; key0 < p.l; key0 += A32_BK) {
      __syncthreads();                                  // every wave is done with the previous tile
      load_tile<HDP, A32_BK>(kt, kslab, key0, p.l, p.c, p.hd, tid);
      load_tile<HDP, A32_BK>(vt, vslab, key0, p.l, p.c, p.hd, tid);
      __syncthreads();
      f32x4 st[2];
      float mx = -INFINITY;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        st[s] = rows_dot<HDP>(kt, 16 * s, qreg, l15, g);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          st[s][i] = key0 + 16 * s + 4 * g + i < p.l ? st[s][i] * p.scale_log2 : -INFINITY;
          mx = fmaxf(mx, st[s][i]);
        }
      }
      mx = group_max(mx);                               // finite: key0 < l, so the tile holds at least one key
      const float m_new = fmaxf(m, mx);
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);     // 0 on the first tile (m = -inf)
      float ps = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) { st[s][i] = __builtin_amdgcn_exp2f(st[s][i] - m_new); ps += st[s][i]; }
      lsum = lsum * alpha + group_sum(ps);
      m = m_new;
#pragma unroll
      for (int d = 0; d < DT; ++d) oacc[d] *= alpha;
      rows_acc<HDP>(oacc, vt, 0, st[0], l15, g);
      rows_acc<HDP>(oacc, vt, 16, st[1], l15, g);
    }
    const float inv = 1.f / lsum;
    const float cw = p.coupling[bi * p.nb + j];
    const long long pair = ((long long)bi * p.nb + j) * p.n + n;
    if (p.lse && g == 0 && row_ok) p.lse[(pair * p.nh + head) * p.l + qrow] = m + __builtin_amdgcn_logf(lsum);
    float* op = p.o_pairs ? p.o_pairs + pair * slab + (long long)qrow * p.c + hoff : nullptr;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      const f32x4 o = oacc[d] * inv;
      if (op && row_ok && 16 * d + 4 * g < p.hd) *reinterpret_cast<f32x4*>(op + 16 * d + 4 * g) = o;
      mix[d] += o * cw;
    }
  }
  if (row_ok) {
    float* yp = p.out + ((long long)n * p.l + qrow) * ((long long)p.nb * p.c) + (long long)bi * p.c + hoff;
#pragma unroll
    for (int d = 0; d < DT; ++d)
      if (16 * d + 4 * g < p.hd) *reinterpret_cast<f32x4*>(yp + 16 * d + 4 * g) = mix[d];
  }
}

// ---------------------------------------------------------------------------------------------
// backward 1/3: r = <dO_i[row, head], O_ij[row, head]>;  delta_ij = c_ij * r;  d coupling[i][j] = sum r (per-workgroup
// partials in the layout of unet_cfc_attn_bwd_partials: [pair][n][row block of 64])
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(A32_THREADS) void attn32_rowdot_kernel(const Attn32K p, float* dc_part) {
  __shared__ float wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pair = blockIdx.z, n = blockIdx.y;
  const int bi = pair / p.nb;
  const float cw = p.coupling[pair];
  const long long slab = (long long)p.l * p.c;
  float tot = 0.f;
  for (int rr = 0; rr < 16; ++rr) {
    const int row = blockIdx.x * 64 + wave * 16 + rr;
    if (row >= p.l) break;                         // wave-uniform
    const float* dop = p.dout + ((long long)n * p.l + row) * ((long long)p.nb * p.c) + (long long)bi * p.c;
    const float* op = p.o_pairs + ((long long)pair * p.n + n) * slab + (long long)row * p.c;
    for (int head = 0; head < p.nh; ++head) {
      float acc = 0.f;
      for (int d = lane; d < p.hd; d += 64) acc = fmaf(dop[head * p.hd + d], op[head * p.hd + d], acc);
      acc = wave_sum(acc);
      if (lane == 0) p.delta[(((long long)pair * p.n + n) * p.nh + head) * p.l + row] = cw * acc;
      tot += acc;
    }
  }
  if (lane == 0) wsum[wave] = tot;
  __syncthreads();
  if (threadIdx.x == 0)
    dc_part[((long long)pair * gridDim.y + n) * gridDim.x + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// ---------------------------------------------------------------------------------------------
// backward 2/3: dQ_i = scale * sum_j sum_keys dS_ij K_j,  dS = P o (c_ij dO_i V_j^T - delta_ij)
// ---------------------------------------------------------------------------------------------
template <int HDP>
__global__ __launch_bounds__(A32_THREADS, 1) void attn32_dq_kernel(const Attn32K p) {
  constexpr int LD = HDP + 4, DT = HDP / 16;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* kt = reinterpret_cast<float*>(smem_raw);
  float* vt = kt + A32_BK * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  int wg = blockIdx.x;
  const int nqt = (p.l + A32_ROWS - 1) / A32_ROWS;
  const int qt = wg % nqt; wg /= nqt;
  const int bi = wg % p.nb; wg /= p.nb;
  const int head = wg % p.nh;
  const int n = wg / p.nh;
  const long long slab = (long long)p.l * p.c;
  const int hoff = head * p.hd;
  const int qrow = qt * A32_ROWS + wave * 16 + l15;
  const bool row_ok = qrow < p.l;

  f32x4 qreg[DT], doreg[DT], acc[DT];
  load_rows<HDP>(qreg, p.q + ((long long)bi * p.n + n) * slab + (long long)qrow * p.c + hoff, row_ok, g, p.hd);
  load_rows<HDP>(doreg, p.dout + ((long long)n * p.l + qrow) * ((long long)p.nb * p.c) + (long long)bi * p.c + hoff, row_ok, g,
                 p.hd);
#pragma unroll
  for (int d = 0; d < DT; ++d) acc[d] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int j = 0; j < p.nb; ++j) {
    const float* kslab = p.k + ((long long)j * p.n + n) * slab + hoff;
    const float* vslab = p.v + ((long long)j * p.n + n) * slab + hoff;
    const long long si = ((((long long)bi * p.nb + j) * p.n + n) * p.nh + head) * p.l + qrow;
    const float lse = row_ok ? p.lse[si] : INFINITY;          // rows past the end: p = 0
    const float delta = row_ok ? p.delta[si] : 0.f;
    const float cw = p.coupling[bi * p.nb + j];
    for (int key0 = 0; key0 < p.l; key0 += A32_BK) {
      __syncthreads();
      load_tile<HDP, A32_BK>(kt, kslab, key0, p.l, p.c, p.hd, tid);
      load_tile<HDP, A32_BK>(vt, vslab, key0, p.l, p.c, p.hd, tid);
      __syncthreads();
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        f32x4 st = rows_dot<HDP>(kt, 16 * s, qreg, l15, g);
        const f32x4 dp = rows_dot<HDP>(vt, 16 * s, doreg, l15, g);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float pe = key0 + 16 * s + 4 * g + i < p.l ? __builtin_amdgcn_exp2f(fmaf(st[i], p.scale_log2, -lse)) : 0.f;
          st[i] = pe * fmaf(dp[i], cw, -delta);
        }
        rows_acc<HDP>(acc, kt, 16 * s, st, l15, g);
      }
    }
  }
  if (row_ok) {
    float* yp = p.dq + ((long long)bi * p.n + n) * slab + (long long)qrow * p.c + hoff;
#pragma unroll
    for (int d = 0; d < DT; ++d)
      if (16 * d + 4 * g < p.hd) *reinterpret_cast<f32x4*>(yp + 16 * d + 4 * g) = acc[d] * p.scale;
  }
}

// ---------------------------------------------------------------------------------------------
// backward 3/3: dK_j, dV_j.  A workgroup owns 64 keys of one (window, head, key band); each wave keeps its 16 K and V
// rows in registers and walks the query rows of all query bands in tiles of 32 (Q and dO tiles + their lse / delta)
// ---------------------------------------------------------------------------------------------
template <int HDP>
__global__ __launch_bounds__(A32_THREADS, 1) void attn32_dkv_kernel(const Attn32K p) {
  constexpr int LD = HDP + 4, DT = HDP / 16;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* qtile = reinterpret_cast<float*>(smem_raw);
  float* dtile = qtile + A32_BK * LD;
  float* stats = dtile + A32_BK * LD;                   // [0..31] lse, [32..63] delta of the tile's rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  int wg = blockIdx.x;
  const int nkt = (p.l + A32_ROWS - 1) / A32_ROWS;
  const int ktl = wg % nkt; wg /= nkt;
  const int bj = wg % p.nb; wg /= p.nb;
  const int head = wg % p.nh;
  const int n = wg / p.nh;
  const long long slab = (long long)p.l * p.c;
  const int hoff = head * p.hd;
  const int ldo = p.nb * p.c;
  const int key = ktl * A32_ROWS + wave * 16 + l15;
  const bool key_ok = key < p.l;
  const long long kvo = ((long long)bj * p.n + n) * slab + (long long)key * p.c + hoff;

  f32x4 kreg[DT], vreg[DT], dka[DT], dva[DT];
  load_rows<HDP>(kreg, p.k + kvo, key_ok, g, p.hd);
  load_rows<HDP>(vreg, p.v + kvo, key_ok, g, p.hd);
#pragma unroll
  for (int d = 0; d < DT; ++d) { dka[d] = f32x4{0.f, 0.f, 0.f, 0.f}; dva[d] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  for (int i = 0; i < p.nb; ++i) {
    const float* qslab = p.q + ((long long)i * p.n + n) * slab + hoff;
    const float* dslab = p.dout + (long long)n * p.l * ldo + (long long)i * p.c + hoff;
    const long long sbase = ((((long long)i * p.nb + bj) * p.n + n) * p.nh + head) * p.l;
    const float cw = p.coupling[i * p.nb + bj];
    for (int row0 = 0; row0 < p.l; row0 += A32_BK) {
      __syncthreads();
      load_tile<HDP, A32_BK>(qtile, qslab, row0, p.l, p.c, p.hd, tid);
      load_tile<HDP, A32_BK>(dtile, dslab, row0, p.l, ldo, p.hd, tid);
      if (tid < 2 * A32_BK) {
        const int r = row0 + (tid & (A32_BK - 1));
        stats[tid] = r < p.l ? (tid < A32_BK ? p.lse[sbase + r] : p.delta[sbase + r]) : (tid < A32_BK ? INFINITY : 0.f);
      }
      __syncthreads();
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        // X[q row 4g+i][key l15] = Q K^T ;  dP = dO V^T
        f32x4 st = rows_dot<HDP>(qtile, 16 * s, kreg, l15, g);
        f32x4 dp = rows_dot<HDP>(dtile, 16 * s, vreg, l15, g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rl = 16 * s + 4 * g + r;
          const float pe = __builtin_amdgcn_exp2f(fmaf(st[r], p.scale_log2, -stats[rl]));   // rows past the end: lse = inf
          st[r] = pe * cw;                                        // c_ij P   -> dV
          dp[r] = pe * fmaf(dp[r], cw, -stats[A32_BK + rl]);      // dS       -> dK
        }
        rows_acc<HDP>(dva, dtile, 16 * s, st, l15, g);
        rows_acc<HDP>(dka, qtile, 16 * s, dp, l15, g);
      }
    }
  }
  if (key_ok) {
#pragma unroll
    for (int d = 0; d < DT; ++d)
      if (16 * d + 4 * g < p.hd) {
        *reinterpret_cast<f32x4*>(p.dk + kvo + 16 * d + 4 * g) = dka[d] * p.scale;
        *reinterpret_cast<f32x4*>(p.dv + kvo + 16 * d + 4 * g) = dva[d];
      }
  }
}

template <int HDP> size_t lds_fwd() { return (size_t)2 * A32_BK * (HDP + 4) * sizeof(float); }
template <int HDP> size_t lds_kv() { return lds_fwd<HDP>() + 2 * A32_BK * sizeof(float); }

template <typename K> int set_lds(K kernel, size_t bytes, const char* what) {
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    UNET_REQUIRE(e == hipSuccess, "%s: hipFuncSetAttribute failed: %s", what, hipGetErrorString(e));
  }
  return 0;
}

template <int HDP> int launch_fwd32(const Attn32K& p, hipStream_t st) {
  if (set_lds(&attn32_fwd_kernel<HDP>, lds_fwd<HDP>(), "cfc_attn_fwd (fp32)")) return -1;
  const long long blocks = (long long)p.n * p.nh * p.nb * cdiv(p.l, A32_ROWS);
  hipLaunchKernelGGL(attn32_fwd_kernel<HDP>, dim3((unsigned)blocks), dim3(A32_THREADS), lds_fwd<HDP>(), st, p);
  UNET_LAUNCH_CHECK("cfc_attn_fwd (fp32)");
  return 0;
}

template <int HDP> int launch_bwd32(const Attn32K& p, hipStream_t st) {
  if (set_lds(&attn32_dq_kernel<HDP>, lds_fwd<HDP>(), "cfc_attn_bwd (fp32 dq)")) return -1;
  if (set_lds(&attn32_dkv_kernel<HDP>, lds_kv<HDP>(), "cfc_attn_bwd (fp32 dk, dv)")) return -1;
  const long long blocks = (long long)p.n * p.nh * p.nb * cdiv(p.l, A32_ROWS);
  hipLaunchKernelGGL(attn32_dq_kernel<HDP>, dim3((unsigned)blocks), dim3(A32_THREADS), lds_fwd<HDP>(), st, p);
  UNET_LAUNCH_CHECK("cfc_attn_bwd (fp32 dq)");
  hipLaunchKernelGGL(attn32_dkv_kernel<HDP>, dim3((unsigned)blocks), dim3(A32_THREADS), lds_kv<HDP>(), st, p);
  UNET_LAUNCH_CHECK("cfc_attn_bwd (fp32 dk, dv)");
  return 0;
}

int check_geometry32(const char* what, const Attn32K& p) {
  UNET_REQUIRE(p.n > 0 && p.l > 0 && p.c > 0 && p.nh > 0 && p.nb > 0 && p.nb <= 8, "%s: bad shape n=%d l=%d c=%d heads=%d bands=%d",
               what, p.n, p.l, p.c, p.nh, p.nb);
  UNET_REQUIRE(p.c % p.nh == 0 && p.hd % 4 == 0 && p.hd <= 256,
               "%s: fp32 head_dim %d must be a multiple of 4 and at most 256 (c=%d heads=%d)", what, p.hd, p.c, p.nh);
  UNET_REQUIRE((long long)p.n * p.nh * p.nb * cdiv(p.l, A32_ROWS) < (1ll << 31) && p.n <= 65535 && p.nb * p.nb <= 65535,
               "%s: grid too large", what);
  return 0;
}

#define A32_DISPATCH(fn, p, st)               \
  do {                                        \
    if ((p).hd <= 16) return fn<16>(p, st);   \
    if ((p).hd <= 32) return fn<32>(p, st);   \
    if ((p).hd <= 64) return fn<64>(p, st);   \
    if ((p).hd <= 128) return fn<128>(p, st); \
    return fn<256>(p, st);                    \
  } while (0)

}  // namespace

// called by the extern "C" entry points of attn.hip for dtype == UNET_F32 (same argument meaning)
int unet_cfc_attn_fwd_f32(const void* q, const void* k, const void* v, const float* coupling, int n, int l, int c, int n_heads,
                          int n_bands, float scale, void* out, void* o_pairs, float* lse, hipStream_t st) {
  UNET_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(out) && aligned16(o_pairs),
               "cfc_attn_fwd: pointers must be 16-byte aligned");
  Attn32K p = {};
  p.q = (const float*)q; p.k = (const float*)k; p.v = (const float*)v; p.coupling = coupling;
  p.out = (float*)out; p.o_pairs = (float*)o_pairs; p.lse = lse;
  p.n = n; p.l = l; p.c = c; p.nh = n_heads; p.nb = n_bands; p.hd = n_heads > 0 ? c / n_heads : 0;
  p.scale = scale; p.scale_log2 = scale * 1.4426950408889634f;
  if (check_geometry32("cfc_attn_fwd", p)) return -1;
  A32_DISPATCH(launch_fwd32, p, st);
}

int unet_cfc_attn_bwd_f32(const void* q, const void* k, const void* v, const float* coupling, const void* o_pairs,
                          const float* lse, const void* dout, int n, int l, int c, int n_heads, int n_bands, float scale,
                          void* dq, void* dk, void* dv, float* dcoupling_part, float* delta, hipStream_t st) {
  UNET_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(o_pairs) && aligned16(dout) && aligned16(dq) &&
                   aligned16(dk) && aligned16(dv),
               "cfc_attn_bwd: pointers must be 16-byte aligned");
  Attn32K p = {};
  p.q = (const float*)q; p.k = (const float*)k; p.v = (const float*)v; p.coupling = coupling;
  p.o_pairs = (float*)const_cast<void*>(o_pairs); p.lse = const_cast<float*>(lse);
  p.dout = (const float*)dout; p.delta = delta;
  p.dq = (float*)dq; p.dk = (float*)dk; p.dv = (float*)dv;
  p.n = n; p.l = l; p.c = c; p.nh = n_heads; p.nb = n_bands; p.hd = n_heads > 0 ? c / n_heads : 0;
  p.scale = scale; p.scale_log2 = scale * 1.4426950408889634f;
  if (check_geometry32("cfc_attn_bwd", p)) return -1;
  hipLaunchKernelGGL(attn32_rowdot_kernel, dim3(cdiv(l, 64), n, n_bands * n_bands), dim3(A32_THREADS), 0, st, p, dcoupling_part);
  UNET_LAUNCH_CHECK("cfc_attn_bwd (fp32 rowdot)");
  A32_DISPATCH(launch_bwd32, p, st);
}
