// Band-pair attention of CrossFrequencyCouplingAttentionV2 (models.py:395-462), bf16 MFMA:
//   out[n, l, i*C + head*hd + d] = sum_j coupling[i][j] * softmax_keys(q_i k_j^T * scale) v_j
// The reference issues 25 scaled_dot_product_attention calls per layer (every query band i against every key /
// value band j), scales each result by the softmaxed coupling weight, sums in fp32 and concatenates the bands.
// Here ONE launch does all of it: a workgroup owns 128 query rows of one (window, head, query band), walks the
// five key/value bands with an online softmax each, and mixes the five results in registers; the output lands
// directly in the [N, L, nb*C] layout out_proj consumes.  Nothing of size L x L ever exists.
//
// Orientation (32x32x16 MFMA): S^T = K Q^T puts a query row on the lane and the keys in the registers, so the
// row max / row sum are in-lane reductions plus one exchange between the lane halves, the probabilities are the
// B operand of O^T = V^T P^T straight from the accumulator registers (no LDS round trip), and the per-row
// rescale of O^T is a per-lane scalar.  V^T comes from the row-major V tile by ds_read_b64_tr_b16.
// K / V tiles of 64 keys are staged by buffer_load ... lds into xor-swizzled images (conflict-free for both the
// 16-byte row reads and the transposed reads), double-buffered: one barrier per 64 keys.
//
// Backward = three kernels on the same tiles: rowdot (delta = c_ij <dO_i, O_ij>, d coupling), dQ (workgroup owns
// query rows, walks keys) and dK/dV (workgroup owns keys, walks query rows; X = Q K^T keeps the key on the
// lane so that dV^T = dO^T P and dK^T = Q^T dS again take the accumulator as the B operand).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr int AT_THREADS = 256;
constexpr int AT_BQ = 128;   // rows owned by a workgroup (32 per wave)
constexpr int AT_BK = 64;    // rows of a streamed tile
constexpr unsigned AT_OOB = 0x80000000u;

struct AttnK {
  const __bf16 *q, *k, *v;   // [nb][N][L][C]
  const float* coupling;     // [nb*nb] (already softmaxed)
  __bf16* out;               // [N][L][nb*C]
  __bf16* o_pairs;           // [nb][nb][N][L][C] or null
  float* lse;                // [nb][nb][N][nh][L] or null; log2 domain: p = exp2(s*scale_log2 - lse)
  // backward only
  const __bf16* dout;        // [N][L][nb*C]
  float* delta;              // [nb][nb][N][nh][L]: written by rowdot, read by dQ and dK/dV
  __bf16 *dq, *dk, *dv;      // [nb][N][L][C]
  int n, l, c, nh, nb, hd;
  float scale, scale_log2;
};

template <int HDP> struct Cfg {
  static constexpr int ROWB = HDP * 2;                 // bytes per tile row
  static constexpr int CPR = HDP / 8;                  // 16-byte chunks per row
  static constexpr int XM = CPR >= 16 ? 15 : CPR - 1;  // swizzle mask
  static constexpr int KS = HDP / 16;                  // k-steps over the head dim
  static constexpr int DT = HDP / 32;                  // 32-wide tiles over the head dim
  static constexpr int TILE = AT_BK * ROWB;            // bytes of a [64][HDP] tile
  static constexpr int PPW = TILE / 1024 / 4;          // 1 KiB pieces per wave
  static constexpr int SB = KS < 4 ? KS : 4;           // k-steps per batch of row-read fragments
};

// chunk swizzle of the LDS images: rows that differ in their low two bits land in different 64-byte groups
// (transposed reads), 16 rows of a ds_read_b128 lane group land on 16 different chunks
__device__ __forceinline__ int xv(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ bf16x8 tr_pair(const char* lo, const char* hi) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lo);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)hi);
  const s16x8 t = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, t);
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& x, int s) {   // registers 8s..8s+7 -> bf16 fragment of k-step s
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (__bf16)x[8 * s + j];
  return o;
}

// ---- streamed [64][HDP] tile: global -> LDS by buffer_load ... lds ---------------------------
template <int HDP> struct TileLoader {
  unsigned voff[Cfg<HDP>::PPW];
  __device__ __forceinline__ void setup(int wave, int lane, int c, int hd) {
#pragma unroll
    for (int k = 0; k < Cfg<HDP>::PPW; ++k) {
      const int lin = (wave * Cfg<HDP>::PPW + k) * 1024 + lane * 16;
      const int row = lin / Cfg<HDP>::ROWB;
      const int phys = (lin % Cfg<HDP>::ROWB) >> 4;
      const int ch = phys ^ (xv(row) & Cfg<HDP>::XM);
      voff[k] = ch * 8 < hd ? (unsigned)((row * c + ch * 8) * 2) : AT_OOB;
    }
  }
  // rows row0 .. row0+63 of the slab behind rsrc (rows past its end arrive as zeros)
  __device__ __forceinline__ void issue(const __bf16* base, long long elem_off, int l, int c, int hd, char* tile,
                                        unsigned row0_bytes, int wave) const {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(base + elem_off), 0,
                                                        (unsigned)((((long long)l - 1) * c + hd) * 2), 0x00020000);
#pragma unroll
    for (int k = 0; k < Cfg<HDP>::PPW; ++k)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsrc, (__attribute__((address_space(3))) void*)(tile + (wave * Cfg<HDP>::PPW + k) * 1024), 16,
          voff[k] + row0_bytes, 0, 0, 0);
  }
};

// per-lane LDS byte offsets of the two kinds of reads (see the header comment)
template <int HDP> __device__ __forceinline__ int row_read_base(int lane) {   // + (s << 5) by xor, + 32*t*ROWB
  const int r = lane & 31, h = lane >> 5;
  return r * Cfg<HDP>::ROWB + ((h ^ (xv(r) & Cfg<HDP>::XM)) << 4);
}
template <int HDP> __device__ __forceinline__ int tr_read_base(int lane, int u) {   // ^ (dt << 6), + (32t + 16s)*ROWB
  const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
  const int row = 8 * u + 4 * (g >> 1) + qq;
  const int c_lo = 2 * (g & 1) + (pp >> 1);
  return row * Cfg<HDP>::ROWB + ((c_lo ^ (xv(row) & Cfg<HDP>::XM)) << 4) + 8 * (pp & 1);
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
// MIX: the coupling-weighted sum over the key/value bands is kept in registers and written to `out` (inference:
// nothing else is stored).  !MIX (training: every O_ij is stored for backward anyway): no second accumulator, which
// leaves the registers the compiler needs to keep LDS reads ahead of the MFMAs; cfc_attn_mix_kernel sums the pairs.
template <int HDP, bool MIX>
__global__ __launch_bounds__(AT_THREADS, 1) void cfc_attn_fwd_kernel(const AttnK p) {
  typedef Cfg<HDP> G;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][K tile | V tile]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int nqt = (p.l + AT_BQ - 1) / AT_BQ;
  const int qt = wg % nqt; wg /= nqt;
  const int bi = wg % p.nb; wg /= p.nb;
  const int head = wg % p.nh;
  const int n = wg / p.nh;
  const long long slab = (long long)p.l * p.c;
  const int hoff = head * p.hd;
  const int qrow = qt * AT_BQ + wave * 32 + r;
  const bool row_ok = qrow < p.l;

  bf16x8 qf[G::KS];
  {
    const __bf16* qp = p.q + ((long long)bi * p.n + n) * slab + (long long)qrow * p.c + hoff;
#pragma unroll
    for (int s = 0; s < G::KS; ++s) {
      const int ch = 2 * s + h;
      bf16x8 z;
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.f;
      qf[s] = (row_ok && ch * 8 < p.hd) ? *reinterpret_cast<const bf16x8*>(qp + ch * 8) : z;
    }
  }

  TileLoader<HDP> ld;
  ld.setup(wave, lane, p.c, p.hd);
  const int a0 = row_read_base<HDP>(lane);
  const int vb0 = tr_read_base<HDP>(lane, 0), vb1 = tr_read_base<HDP>(lane, 1);

  f32x16 acc[G::DT], mix[MIX ? G::DT : 1];
#pragma unroll
  for (int d = 0; d < G::DT; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      acc[d][e] = 0.f;
      if (MIX) mix[d][e] = 0.f;
    }
  // Softmax reference point m of a (row, band pair): the row maximum of the pair's FIRST tile, then fixed.  Later
  // scores may exceed it by up to 2^32 (p and the sums stay far inside fp32; O, l and lse = m + log2 l are all
  // relative to the same m), so the accumulators -- which live in AGPRs, out of the VALU's reach -- are never
  // rescaled inside the loop.  A row that does exceed the margin flags the pair, and the whole workgroup walks it
  // a second time with m = the true row maxima found in the first walk (never seen outside the tests).
  float m = -INFINITY, lsum = 0.f, mtrue = -INFINITY;
  int bad = 0;

  const int nkb = (p.l + AT_BK - 1) / AT_BK;
  const unsigned tile_step = (unsigned)AT_BK * p.c * 2;

  auto issue = [&](int j, int kb, int buf) {
    const long long off = ((long long)j * p.n + n) * slab + hoff;
    char* base = smem + buf * 2 * G::TILE;
    ld.issue(p.k, off, p.l, p.c, p.hd, base, (unsigned)kb * tile_step, wave);
    ld.issue(p.v, off, p.l, p.c, p.hd, base + G::TILE, (unsigned)kb * tile_step, wave);
  };

  issue(0, 0, 0);
  int j = 0, buf = 0;
  while (j < p.nb) {
   // the walk over one band pair: nothing but MFMAs touches the accumulators in here
   for (int kb = 0; kb < nkb; ++kb) {
    __syncthreads();   // the tile in `buf` has landed (vmcnt(0) + barrier); every wave is done with the other buffer
    {
      int kb1 = kb + 1, j1 = j;
      if (kb1 == nkb) { kb1 = 0; ++j1; }
      if (j1 < p.nb) issue(j1, kb1, buf ^ 1);
    }
    const char* kt = smem + buf * 2 * G::TILE;
    const char* vt = kt + G::TILE;
    buf ^= 1;

    // ---- S^T = K Q^T : [64 keys][32 query rows] as two 32x32 tiles --------------------------
    f32x16 st[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) st[t][e] = 0.f;
    // fragments are read in batches of SB k-steps, one batch ahead of the MFMAs that consume them
    // (the MIX variant has no registers to spare: one k-step per batch, V^T fragments read as they are used)
    constexpr int SBF = MIX ? 1 : G::SB;
    bf16x8 fk[2][SBF][2];
    auto load_k = [&](int b, int slot) {
#pragma unroll
      for (int s4 = 0; s4 < SBF; ++s4)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          fk[slot][s4][t] = *reinterpret_cast<const bf16x8*>(kt + ((a0 ^ ((b * SBF + s4) << 5)) + t * 32 * G::ROWB));
    };
    load_k(0, 0);
#pragma unroll
    for (int b = 0; b < G::KS / SBF; ++b) {
      if (b + 1 < G::KS / SBF) load_k(b + 1, (b + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);   // keep the reads of the next batch above this batch's MFMAs
#pragma unroll
      for (int s4 = 0; s4 < SBF; ++s4)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk[b & 1][s4][t], qf[b * SBF + s4], st[t], 0, 0, 0);
    }
    // V^T fragments of the first 16 keys: in flight during the softmax arithmetic
    bf16x8 fv[MIX ? 1 : 2][G::DT];
    auto load_v = [&](int g, int slot) {
      const int rowoff = (32 * (g >> 1) + 16 * (g & 1)) * G::ROWB;
#pragma unroll
      for (int d = 0; d < G::DT; ++d)
        fv[slot][d] = tr_pair(vt + ((vb0 ^ (d << 6)) + rowoff), vt + ((vb1 ^ (d << 6)) + rowoff));
    };
    if (!MIX) {
      load_v(0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- online softmax over the keys (registers + the other lane half) ----------------------
    const int key0 = kb * AT_BK;
    if (key0 + AT_BK > p.l) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (key0 + 32 * t + (e & 3) + 8 * (e >> 2) + 4 * h >= p.l) st[t][e] = -INFINITY;
    }
    float mx = st[0][0];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, st[t][e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * p.scale_log2;
    mtrue = fmaxf(mtrue, mx);
    m = m == -INFINITY ? mx : m;
    bad |= mx > m + 32.f;
    // ---- P = exp2(S - m), O^T += V^T P^T.  The exponentials of the next 16 keys are issued between the MFMAs
    // of the current 16 (one wave per SIMD: nothing else would hide them) ----------------------------------------
    auto exp_one = [&](int t, int e) {
      const float pe = __builtin_amdgcn_exp2f(fmaf(st[t][e], p.scale_log2, -m));
      st[t][e] = pe;
      lsum += pe;
    };
#pragma unroll
    for (int e = 0; e < 8; ++e) exp_one(0, e);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int t = g >> 1, s = g & 1, tn = (g + 1) >> 1, sn = (g + 1) & 1;
      const bf16x8 pb = pack8(st[t], s);
      if (MIX) load_v(g, 0);
      else if (g < 3) load_v(g + 1, (g + 1) & 1);
      if (!MIX) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int d = 0; d < G::DT; ++d) {
        acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fv[MIX ? 0 : (g & 1)][d], pb, acc[d], 0, 0, 0);
        if (g < 3) {
#pragma unroll
          for (int x = d * 8 / G::DT; x < (d + 1) * 8 / G::DT; ++x) exp_one(tn, 8 * sn + x);
        }
      }
    }

   }
    // ---- end of a band pair: normalise, keep what backward needs, mix -----------------------------
    if (nkb > 1 && __syncthreads_or(bad)) {
      // second walk over this pair (see above): `buf` now names the buffer that holds the prefetched first tile
      // of the NEXT pair; this pair's first tile replaces it (loads complete in order)
      issue(j, 0, buf);
      m = mtrue;
      lsum = 0.f;
      bad = 0;
#pragma unroll
      for (int d = 0; d < G::DT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[d][e] = 0.f;
    } else {
      const float lt = lsum + __shfl_xor(lsum, 32, 64);
      const float inv = 1.f / lt;
      const float cw = p.coupling[bi * p.nb + j] * inv;
      const long long pair = ((long long)bi * p.nb + j) * p.n + n;
      if (p.lse && h == 0 && row_ok) p.lse[(pair * p.nh + head) * p.l + qrow] = m + __builtin_amdgcn_logf(lt);
      __bf16* op = p.o_pairs ? p.o_pairs + pair * slab + (long long)qrow * p.c + hoff : nullptr;
#pragma unroll
      for (int d = 0; d < G::DT; ++d) {
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          const int dd = 32 * d + 8 * e4 + 4 * h;
          if (op && row_ok && dd < p.hd) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)(acc[d][4 * e4 + e] * inv);
            *reinterpret_cast<bf16x4*>(op + dd) = o;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (MIX) mix[d][4 * e4 + e] = fmaf(acc[d][4 * e4 + e], cw, mix[d][4 * e4 + e]);
            acc[d][4 * e4 + e] = 0.f;
          }
        }
      }
      m = -INFINITY;
      mtrue = -INFINITY;
      lsum = 0.f;
      ++j;
    }
  }

  if (MIX && row_ok) {
    __bf16* yp = p.out + ((long long)n * p.l + qrow) * ((long long)p.nb * p.c) + (long long)bi * p.c + hoff;
#pragma unroll
    for (int d = 0; d < G::DT; ++d)
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const int dd = 32 * d + 8 * e4 + 4 * h;
        if (dd < p.hd) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (__bf16)mix[MIX ? d : 0][4 * e4 + e];
          *reinterpret_cast<bf16x4*>(yp + dd) = o;
        }
      }
  }
}

// out[n][l][i*C + ch] = sum_j coupling[i][j] * O_ij[n][l][ch]   (fp32 sum of the stored bf16 pair outputs, which is
// what the reference adds up: models.py:452-456).  HBM-bound: nb reads + 1 write per element.
__global__ __launch_bounds__(AT_THREADS) void cfc_attn_mix_kernel(const AttnK p) {
  const long long chunks = (long long)p.n * p.l * (p.c >> 3);       // 16-byte chunks of one [N, L, C] slab
  const int bi = blockIdx.y;
  const long long slab_all = (long long)p.n * p.l * p.c;
  float cw[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cw[j] = j < p.nb ? p.coupling[bi * p.nb + j] : 0.f;
  for (long long x = (long long)blockIdx.x * AT_THREADS + threadIdx.x; x < chunks; x += (long long)gridDim.x * AT_THREADS) {
    const long long row = x / (p.c >> 3);
    const int ch = (int)(x - row * (p.c >> 3)) * 8;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int j = 0; j < p.nb; ++j) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(p.o_pairs + ((long long)bi * p.nb + j) * slab_all + row * p.c + ch);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf((float)v[e], cw[j], acc[e]);
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)acc[e];
    *reinterpret_cast<bf16x8*>(p.out + row * ((long long)p.nb * p.c) + (long long)bi * p.c + ch) = o;
  }
}

// ---------------------------------------------------------------------------------------------
// backward 1/3: rowdot.  r = <dO_i[row, head], O_ij[row, head]>;  delta_ij = c_ij * r (the softmax-gradient row
// term of pair (i, j), whose upstream gradient is c_ij dO_i);  d coupling[i][j] = sum r  (per-workgroup partials,
// summed by the caller: no atomics, deterministic).  One wave per row, HBM-bound.
// ---------------------------------------------------------------------------------------------
constexpr int RD_ROWS = 64;   // rows per workgroup (16 per wave)

__global__ __launch_bounds__(AT_THREADS) void cfc_attn_rowdot_kernel(const AttnK p, float* dc_part) {
  __shared__ float part[4][264];    // per wave: one partial per 16-byte chunk (<= 8 heads x 32 chunks + pad)
  __shared__ float wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pair = blockIdx.z, n = blockIdx.y;
  const int bi = pair / p.nb;
  const float cw = p.coupling[pair];
  const int chunks = p.c >> 3, cph = p.hd >> 3;
  const long long slab = (long long)p.l * p.c;
  float tot = 0.f;
  for (int rr = 0; rr < RD_ROWS / 4; ++rr) {
    const int row = blockIdx.x * RD_ROWS + wave * (RD_ROWS / 4) + rr;
    if (row >= p.l) break;                         // wave-uniform
    const __bf16* dop = p.dout + ((long long)n * p.l + row) * ((long long)p.nb * p.c) + (long long)bi * p.c;
    const __bf16* op = p.o_pairs + ((long long)pair * p.n + n) * slab + (long long)row * p.c;
    for (int ch = lane; ch < chunks; ch += 64) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(dop + ch * 8);
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(op + ch * 8);
      float acc = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc = fmaf((float)a[e], (float)b[e], acc);
      part[wave][ch] = acc;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): the wave's own LDS writes are visible to its lanes
    if (lane < p.nh) {
      float r = 0.f;
      for (int k = 0; k < cph; ++k) r += part[wave][lane * cph + k];
      p.delta[(((long long)pair * p.n + n) * p.nh + lane) * p.l + row] = cw * r;
      tot += r;
    }
    __builtin_amdgcn_wave_barrier();
  }
  tot = wave_sum(tot);
  if (lane == 0) wsum[wave] = tot;
  __syncthreads();
  if (threadIdx.x == 0)
    dc_part[((long long)pair * gridDim.y + n) * gridDim.x + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// ---------------------------------------------------------------------------------------------
// backward 2/3: dQ_i = scale * sum_j sum_keys dS_ij K_j,  dS = P o (c_ij dO_i V_j^T - delta_ij).  Same walk as the
// forward kernel; both tiles are read row-wise (S^T = K Q^T, dP^T = V dO^T) and K once more transposed.
// ---------------------------------------------------------------------------------------------
template <int HDP>
__global__ __launch_bounds__(AT_THREADS, 1) void cfc_attn_dq_kernel(const AttnK p) {
  typedef Cfg<HDP> G;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int nqt = (p.l + AT_BQ - 1) / AT_BQ;
  const int qt = wg % nqt; wg /= nqt;
  const int bi = wg % p.nb; wg /= p.nb;
  const int head = wg % p.nh;
  const int n = wg / p.nh;
  const long long slab = (long long)p.l * p.c;
  const int hoff = head * p.hd;
  const int qrow = qt * AT_BQ + wave * 32 + r;
  const bool row_ok = qrow < p.l;

  bf16x8 qf[G::KS], dof[G::KS];
  {
    const __bf16* qp = p.q + ((long long)bi * p.n + n) * slab + (long long)qrow * p.c + hoff;
    const __bf16* dp = p.dout + ((long long)n * p.l + qrow) * ((long long)p.nb * p.c) + (long long)bi * p.c + hoff;
#pragma unroll
    for (int s = 0; s < G::KS; ++s) {
      const int ch = 2 * s + h;
      bf16x8 z;
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.f;
      const bool ok = row_ok && ch * 8 < p.hd;
      qf[s] = ok ? *reinterpret_cast<const bf16x8*>(qp + ch * 8) : z;
      dof[s] = ok ? *reinterpret_cast<const bf16x8*>(dp + ch * 8) : z;
    }
  }

  TileLoader<HDP> ld;
  ld.setup(wave, lane, p.c, p.hd);
  const int a0 = row_read_base<HDP>(lane);
  const int vb0 = tr_read_base<HDP>(lane, 0), vb1 = tr_read_base<HDP>(lane, 1);

  f32x16 acc[G::DT];
#pragma unroll
  for (int d = 0; d < G::DT; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[d][e] = 0.f;

  const int nkb = (p.l + AT_BK - 1) / AT_BK;
  const int total = p.nb * nkb;
  const unsigned tile_step = (unsigned)AT_BK * p.c * 2;
  auto issue = [&](int j, int kb, int buf) {
    const long long off = ((long long)j * p.n + n) * slab + hoff;
    char* base = smem + buf * 2 * G::TILE;
    ld.issue(p.k, off, p.l, p.c, p.hd, base, (unsigned)kb * tile_step, wave);
    ld.issue(p.v, off, p.l, p.c, p.hd, base + G::TILE, (unsigned)kb * tile_step, wave);
  };

  issue(0, 0, 0);
  int j = 0, kb = 0;
  float lse = 0.f, delta = 0.f, cw = 0.f;
  for (int it = 0; it < total; ++it) {
    if (kb == 0) {
      const long long pair = ((long long)bi * p.nb + j) * p.n + n;
      const long long si = (pair * p.nh + head) * p.l + qrow;
      lse = row_ok ? p.lse[si] : INFINITY;        // rows past the end: p = 0
      delta = row_ok ? p.delta[si] : 0.f;
      cw = p.coupling[bi * p.nb + j];
    }
    __syncthreads();
    {
      int kb1 = kb + 1, j1 = j;
      if (kb1 == nkb) { kb1 = 0; ++j1; }
      if (it + 1 < total) issue(j1, kb1, (it + 1) & 1);
    }
    const char* kt = smem + (it & 1) * 2 * G::TILE;
    const char* vt = kt + G::TILE;

    f32x16 st[2], dp[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) { st[t][e] = 0.f; dp[t][e] = 0.f; }
    constexpr int SB2 = G::SB < 2 ? G::SB : 2;     // k-steps per batch (K and V fragments: 8 per batch)
    bf16x8 fk[2][SB2][2], fw[2][SB2][2];
    auto load_kv = [&](int b, int slot) {
#pragma unroll
      for (int s2 = 0; s2 < SB2; ++s2)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int o = (a0 ^ ((b * SB2 + s2) << 5)) + t * 32 * G::ROWB;
          fk[slot][s2][t] = *reinterpret_cast<const bf16x8*>(kt + o);
          fw[slot][s2][t] = *reinterpret_cast<const bf16x8*>(vt + o);
        }
    };
    load_kv(0, 0);
#pragma unroll
    for (int b = 0; b < G::KS / SB2; ++b) {
      if (b + 1 < G::KS / SB2) load_kv(b + 1, (b + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s2 = 0; s2 < SB2; ++s2)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk[b & 1][s2][t], qf[b * SB2 + s2], st[t], 0, 0, 0);
          dp[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[b & 1][s2][t], dof[b * SB2 + s2], dp[t], 0, 0, 0);
        }
    }
    bf16x8 ft[2][G::DT];                           // K^T fragments, one group of 16 keys ahead
    auto load_t = [&](int g, int slot) {
      const int rowoff = (32 * (g >> 1) + 16 * (g & 1)) * G::ROWB;
#pragma unroll
      for (int d = 0; d < G::DT; ++d)
        ft[slot][d] = tr_pair(kt + ((vb0 ^ (d << 6)) + rowoff), kt + ((vb1 ^ (d << 6)) + rowoff));
    };
    load_t(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    const int key0 = kb * AT_BK;
    const bool ragged = key0 + AT_BK > p.l;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float pe = __builtin_amdgcn_exp2f(fmaf(st[t][e], p.scale_log2, -lse));
        if (ragged && key0 + 32 * t + (e & 3) + 8 * (e >> 2) + 4 * h >= p.l) pe = 0.f;
        st[t][e] = pe * fmaf(dp[t][e], cw, -delta);
      }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const bf16x8 db = pack8(st[g >> 1], g & 1);
      if (g < 3) load_t(g + 1, (g + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int d = 0; d < G::DT; ++d) acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ft[g & 1][d], db, acc[d], 0, 0, 0);
    }
    if (++kb == nkb) { kb = 0; ++j; }
  }

  if (row_ok) {
    __bf16* yp = p.dq + ((long long)bi * p.n + n) * slab + (long long)qrow * p.c + hoff;
#pragma unroll
    for (int d = 0; d < G::DT; ++d)
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const int dd = 32 * d + 8 * e4 + 4 * h;
        if (dd < p.hd) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (__bf16)(acc[d][4 * e4 + e] * p.scale);
          *reinterpret_cast<bf16x4*>(yp + dd) = o;
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------
// backward 3/3: dK_j, dV_j.  A workgroup owns 64 keys of one (window, head, key band) and walks the query rows
// of all query bands, 64 at a time (Q and dO tiles, double-buffered).  Waves: 2 key groups of 32 x 2 halves of
// the streamed rows.  The wave's own K and V rows stay in registers as B operands: X = Q K^T has the key on the
// lane and the query rows in the registers, so P and dS are again B operands as they come out of the
// accumulators: dV^T += dO^T (c_ij P), dK^T += Q^T dS (transposed reads of the dO / Q tiles).
// The two row halves are summed through LDS at the end.
// ---------------------------------------------------------------------------------------------
template <int HDP>
__global__ __launch_bounds__(AT_THREADS, 1) void cfc_attn_dkv_kernel(const AttnK p) {
  typedef Cfg<HDP> G;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][Q tile | dO tile], then [2][128] row stats
  float* const stats = reinterpret_cast<float*>(smem + 4 * G::TILE);   // per buffer: [0..63] lse, [64..127] delta
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave & 1, par = wave >> 1;
  const int r = lane & 31, h = lane >> 5;

  int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int nkt = (p.l + AT_BK - 1) / AT_BK;
  const int ktl = wg % nkt; wg /= nkt;
  const int bj = wg % p.nb; wg /= p.nb;
  const int head = wg % p.nh;
  const int n = wg / p.nh;
  const long long slab = (long long)p.l * p.c;
  const int hoff = head * p.hd;
  const int ldo = p.nb * p.c;                          // row pitch of dout
  const int key = ktl * AT_BK + kg * 32 + r;
  const long long kvo = ((long long)bj * p.n + n) * slab + (long long)key * p.c + hoff;

  bf16x8 kf[G::KS], vf[G::KS];
#pragma unroll
  for (int s = 0; s < G::KS; ++s) {
    const int ch = 2 * s + h;
    bf16x8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.f;
    const bool ok = key < p.l && ch * 8 < p.hd;
    kf[s] = ok ? *reinterpret_cast<const bf16x8*>(p.k + kvo + ch * 8) : z;
    vf[s] = ok ? *reinterpret_cast<const bf16x8*>(p.v + kvo + ch * 8) : z;
  }

  TileLoader<HDP> ld, ldd;
  ld.setup(wave, lane, p.c, p.hd);
  ldd.setup(wave, lane, ldo, p.hd);
  const int a0 = row_read_base<HDP>(lane) + par * 32 * G::ROWB;
  const int vb0 = tr_read_base<HDP>(lane, 0) + par * 32 * G::ROWB, vb1 = tr_read_base<HDP>(lane, 1) + par * 32 * G::ROWB;

  f32x16 dka[G::DT], dva[G::DT];
#pragma unroll
  for (int d = 0; d < G::DT; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) { dka[d][e] = 0.f; dva[d][e] = 0.f; }

  const int nqb = (p.l + AT_BK - 1) / AT_BK;
  const int total = p.nb * nqb;

  auto issue = [&](int i, int qb, int buf) {
    const unsigned row0 = (unsigned)qb * AT_BK;
    char* base = smem + buf * 2 * G::TILE;
    ld.issue(p.q, ((long long)i * p.n + n) * slab + hoff, p.l, p.c, p.hd, base, row0 * p.c * 2, wave);
    ldd.issue(p.dout, (long long)n * p.l * ldo + (long long)i * p.c + hoff, p.l, ldo, p.hd, base + G::TILE, row0 * ldo * 2,
              wave);
  };
  auto load_stat = [&](int i, int qb) {            // threads 0..63: lse, 64..127: delta of the tile's rows
    float v = tid < 64 ? INFINITY : 0.f;           // rows past the end: p = 0, delta = 0
    const int row = qb * AT_BK + (tid & 63);
    if (tid < 128 && row < p.l) {
      const long long si = ((((long long)i * p.nb + bj) * p.n + n) * p.nh + head) * p.l + row;
      v = tid < 64 ? p.lse[si] : p.delta[si];
    }
    return v;
  };

  issue(0, 0, 0);
  if (tid < 128) stats[tid] = load_stat(0, 0);
  int i = 0, qb = 0;
  for (int it = 0; it < total; ++it) {
    const float cw = p.coupling[i * p.nb + bj];
    int qb1 = qb + 1, i1 = i;
    if (qb1 == nqb) { qb1 = 0; ++i1; }
    __syncthreads();   // tile `it` and its stats are in LDS; every wave is done with the other buffer
    float stat_next = 0.f;
    if (it + 1 < total) {
      issue(i1, qb1, (it + 1) & 1);
      stat_next = load_stat(i1, qb1);
    }
    const char* qtile = smem + (it & 1) * 2 * G::TILE;
    const char* dtile = qtile + G::TILE;
    const float* st_in = stats + (it & 1) * 128;

    // ---- X = Q K^T, dP = dO V^T : [32 query rows][32 keys] ----------------------------------------
    f32x16 st, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) { st[e] = 0.f; dp[e] = 0.f; }
    constexpr int SBK = G::SB < 2 ? G::SB : 2;       // (256 accumulator + 128 operand registers are taken)
    bf16x8 fq[2][SBK], fd[2][SBK];                   // batches of SBK k-steps, one batch ahead of the MFMAs
    auto load_qd = [&](int b, int slot) {
#pragma unroll
      for (int s4 = 0; s4 < SBK; ++s4) {
        const int o = a0 ^ ((b * SBK + s4) << 5);
        fq[slot][s4] = *reinterpret_cast<const bf16x8*>(qtile + o);
        fd[slot][s4] = *reinterpret_cast<const bf16x8*>(dtile + o);
      }
    };
    load_qd(0, 0);
#pragma unroll
    for (int b = 0; b < G::KS / SBK; ++b) {
      if (b + 1 < G::KS / SBK) load_qd(b + 1, (b + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s4 = 0; s4 < SBK; ++s4) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[b & 1][s4], kf[b * SBK + s4], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fd[b & 1][s4], vf[b * SBK + s4], dp, 0, 0, 0);
      }
    }
    // transposed dO / Q fragments in batches of TB head-dim tiles, one batch ahead
    constexpr int TB = 1, NTB = 2 * (G::DT / TB);     // batches over (k-step s, d)
    bf16x8 tq[2][TB], td[2][TB];
    auto load_tt = [&](int b, int slot) {
      const int s = b / (G::DT / TB), d0 = (b % (G::DT / TB)) * TB;
#pragma unroll
      for (int x = 0; x < TB; ++x) {
        const int o0 = (vb0 ^ ((d0 + x) << 6)) + 16 * s * G::ROWB, o1 = (vb1 ^ ((d0 + x) << 6)) + 16 * s * G::ROWB;
        td[slot][x] = tr_pair(dtile + o0, dtile + o1);
        tq[slot][x] = tr_pair(qtile + o0, qtile + o1);
      }
    };
    load_tt(0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
      const int rl = 32 * par + 8 * e4 + 4 * h;           // local rows rl .. rl+3 live in registers 4*e4 .. 4*e4+3
      const f32x4 ls = *reinterpret_cast<const f32x4*>(st_in + rl);
      const f32x4 de = *reinterpret_cast<const f32x4*>(st_in + 64 + rl);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pe = __builtin_amdgcn_exp2f(fmaf(st[4 * e4 + e], p.scale_log2, -ls[e]));
        st[4 * e4 + e] = pe * cw;                                   // c_ij P      -> dV
        dp[4 * e4 + e] = pe * fmaf(dp[4 * e4 + e], cw, -de[e]);     // dS          -> dK
      }
    }
#pragma unroll
    for (int b = 0; b < NTB; ++b) {
      const int s = b / (G::DT / TB), d0 = (b % (G::DT / TB)) * TB;
      const bf16x8 pb = pack8(st, s), db = pack8(dp, s);
      if (b + 1 < NTB) load_tt(b + 1, (b + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int x = 0; x < TB; ++x) {
        dva[d0 + x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(td[b & 1][x], pb, dva[d0 + x], 0, 0, 0);
        dka[d0 + x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq[b & 1][x], db, dka[d0 + x], 0, 0, 0);
      }
    }
    if (it + 1 < total && tid < 128) stats[((it + 1) & 1) * 128 + tid] = stat_next;
    qb = qb1;
    i = i1;
  }

  // ---- sum the two row halves through LDS (both buffers are free now), store ------------------------
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);         // [kg][dk|dv][DT*16][64 lanes]
  if (par == 1) {
#pragma unroll
    for (int d = 0; d < G::DT; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        red[((kg * 2 + 0) * G::DT * 16 + d * 16 + e) * 64 + lane] = dka[d][e];
        red[((kg * 2 + 1) * G::DT * 16 + d * 16 + e) * 64 + lane] = dva[d][e];
      }
  }
  __syncthreads();
  if (par == 0 && key < p.l) {
#pragma unroll
    for (int d = 0; d < G::DT; ++d)
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const int dd = 32 * d + 8 * e4 + 4 * h;
        if (dd < p.hd) {
          bf16x4 ok, ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int x = d * 16 + 4 * e4 + e;
            ok[e] = (__bf16)((dka[d][4 * e4 + e] + red[((kg * 2 + 0) * G::DT * 16 + x) * 64 + lane]) * p.scale);
            ov[e] = (__bf16)(dva[d][4 * e4 + e] + red[((kg * 2 + 1) * G::DT * 16 + x) * 64 + lane]);
          }
          *reinterpret_cast<bf16x4*>(p.dk + kvo + dd) = ok;
          *reinterpret_cast<bf16x4*>(p.dv + kvo + dd) = ov;
        }
      }
  }
}

#define CFC_INSTANTIATE(HDP)                                             \
  template __global__ void cfc_attn_fwd_kernel<HDP, true>(const AttnK);  \
  template __global__ void cfc_attn_fwd_kernel<HDP, false>(const AttnK); \
  template __global__ void cfc_attn_dq_kernel<HDP>(const AttnK);        \
  template __global__ void cfc_attn_dkv_kernel<HDP>(const AttnK);
CFC_INSTANTIATE(32)
CFC_INSTANTIATE(64)
CFC_INSTANTIATE(128)
CFC_INSTANTIATE(256)
#undef CFC_INSTANTIATE

template <int HDP> int launch_fwd(const AttnK& p, hipStream_t st) {
  const size_t lds = 4 * (size_t)Cfg<HDP>::TILE;
  const bool mix = p.o_pairs == nullptr;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(mix ? reinterpret_cast<const void*>(&cfc_attn_fwd_kernel<HDP, true>)
                                           : reinterpret_cast<const void*>(&cfc_attn_fwd_kernel<HDP, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    UNET_REQUIRE(e == hipSuccess, "cfc_attn_fwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
  }
  const long long blocks = (long long)p.n * p.nh * p.nb * cdiv(p.l, AT_BQ);
  if (mix) {
    hipLaunchKernelGGL((cfc_attn_fwd_kernel<HDP, true>), dim3((unsigned)blocks), dim3(AT_THREADS), lds, st, p);
  } else {
    hipLaunchKernelGGL((cfc_attn_fwd_kernel<HDP, false>), dim3((unsigned)blocks), dim3(AT_THREADS), lds, st, p);
    const long long chunks = (long long)p.n * p.l * (p.c >> 3);
    const unsigned gx = (unsigned)((chunks + AT_THREADS - 1) / AT_THREADS < 4096 ? (chunks + AT_THREADS - 1) / AT_THREADS : 4096);
    hipLaunchKernelGGL(cfc_attn_mix_kernel, dim3(gx, p.nb), dim3(AT_THREADS), 0, st, p);
  }
  UNET_LAUNCH_CHECK("cfc_attn_fwd");
  return 0;
}

template <int HDP> int launch_bwd(const AttnK& p, hipStream_t st) {
  const size_t lds_q = 4 * (size_t)Cfg<HDP>::TILE, lds_kv = 4 * (size_t)Cfg<HDP>::TILE + 1024;
  if (lds_kv > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cfc_attn_dq_kernel<HDP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cfc_attn_dkv_kernel<HDP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);
    UNET_REQUIRE(e == hipSuccess, "cfc_attn_bwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
  }
  const long long per = (long long)p.n * p.nh * p.nb;
  hipLaunchKernelGGL(cfc_attn_dq_kernel<HDP>, dim3((unsigned)(per * cdiv(p.l, AT_BQ))), dim3(AT_THREADS), lds_q, st, p);
  UNET_LAUNCH_CHECK("cfc_attn_bwd (dq)");
  hipLaunchKernelGGL(cfc_attn_dkv_kernel<HDP>, dim3((unsigned)(per * cdiv(p.l, AT_BK))), dim3(AT_THREADS), lds_kv, st, p);
  UNET_LAUNCH_CHECK("cfc_attn_bwd (dk, dv)");
  return 0;
}

int check_geometry(const char* what, int n, int l, int c, int nh, int nb) {
  UNET_REQUIRE(n > 0 && l > 0 && c > 0 && nh > 0 && nb > 0 && nb <= 8, "%s: bad shape n=%d l=%d c=%d heads=%d bands=%d",
               what, n, l, c, nh, nb);
  UNET_REQUIRE(c % nh == 0 && (c / nh) % 8 == 0 && c / nh <= 256,
               "%s: head_dim %d must be a multiple of 8 and at most 256 (c=%d heads=%d)", what, c / nh, c, nh);
  UNET_REQUIRE((long long)l * c * 2 < (1ll << 31), "%s: one [L, C] slab of %lld bytes exceeds a buffer descriptor", what,
               (long long)l * c * 2);
  UNET_REQUIRE((long long)n * nh * nb * cdiv(l, AT_BK) < (1ll << 31), "%s: grid too large", what);
  return 0;
}

}  // namespace

// no nulls except o_pairs / lse
extern "C" int unet_cfc_attn_fwd(int32_t dtype, const void* q, const void* k, const void* v, const float* coupling,
                                 int32_t n, int32_t l, int32_t c, int32_t n_heads, int32_t n_bands, float scale, void* out,
                                 void* o_pairs, float* lse, void* stream) {
  UNET_REQUIRE(dtype == UNET_BF16 || dtype == UNET_F32, "cfc_attn_fwd: bad dtype %d", dtype);
  UNET_REQUIRE(q && k && v && coupling && out, "cfc_attn_fwd: null pointer");
  UNET_REQUIRE(o_pairs || !lse, "cfc_attn_fwd: lse without o_pairs");
  if (dtype == UNET_F32)      // exact-fp32 kernels (csrc/attn_f32.hip)
    return unet_cfc_attn_fwd_f32(q, k, v, coupling, n, l, c, n_heads, n_bands, scale, out, o_pairs, lse,
                                 static_cast<hipStream_t>(stream));
  UNET_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(out) && aligned16(o_pairs),
               "cfc_attn_fwd: pointers must be 16-byte aligned");
  if (check_geometry("cfc_attn_fwd", n, l, c, n_heads, n_bands)) return -1;
  AttnK p = {};
  p.q = (const __bf16*)q; p.k = (const __bf16*)k; p.v = (const __bf16*)v; p.coupling = coupling;
  p.out = (__bf16*)out; p.o_pairs = (__bf16*)o_pairs; p.lse = lse;
  p.n = n; p.l = l; p.c = c; p.nh = n_heads; p.nb = n_bands; p.hd = c / n_heads;
  p.scale = scale; p.scale_log2 = scale * 1.4426950408889634f;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (p.hd <= 32) return launch_fwd<32>(p, st);
  if (p.hd <= 64) return launch_fwd<64>(p, st);
  if (p.hd <= 128) return launch_fwd<128>(p, st);
  return launch_fwd<256>(p, st);
}

extern "C" size_t unet_cfc_attn_bwd_partials(int32_t n, int32_t l, int32_t n_bands) {
  return (size_t)n_bands * n_bands * n * cdiv(l, RD_ROWS);
}

// dcoupling_part: unet_cfc_attn_bwd_partials(n, l, n_bands) floats = [n_bands^2][n * row blocks]; the caller sums
// the last axis.  delta: scratch of lse's shape.
extern "C" int unet_cfc_attn_bwd(int32_t dtype, const void* q, const void* k, const void* v, const float* coupling,
                                 const void* o_pairs, const float* lse, const void* dout, int32_t n, int32_t l, int32_t c,
                                 int32_t n_heads, int32_t n_bands, float scale, void* dq, void* dk, void* dv,
                                 float* dcoupling_part, float* delta, void* stream) {
  UNET_REQUIRE(dtype == UNET_BF16 || dtype == UNET_F32, "cfc_attn_bwd: bad dtype %d", dtype);
  UNET_REQUIRE(q && k && v && coupling && o_pairs && lse && dout && dq && dk && dv && dcoupling_part && delta,
               "cfc_attn_bwd: null pointer");
  if (dtype == UNET_F32)
    return unet_cfc_attn_bwd_f32(q, k, v, coupling, o_pairs, lse, dout, n, l, c, n_heads, n_bands, scale, dq, dk, dv,
                                 dcoupling_part, delta, static_cast<hipStream_t>(stream));
  UNET_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(o_pairs) && aligned16(dout) && aligned16(dq) &&
                   aligned16(dk) && aligned16(dv),
               "cfc_attn_bwd: pointers must be 16-byte aligned");
  if (check_geometry("cfc_attn_bwd", n, l, c, n_heads, n_bands)) return -1;
  UNET_REQUIRE((long long)l * n_bands * c * 2 < (1ll << 31), "cfc_attn_bwd: one [L, bands*C] slab exceeds a buffer descriptor");
  UNET_REQUIRE(n <= 65535 && n_heads <= 64 && c / 8 <= 256, "cfc_attn_bwd: n=%d heads=%d c=%d out of range", n, n_heads, c);
  AttnK p = {};
  p.q = (const __bf16*)q; p.k = (const __bf16*)k; p.v = (const __bf16*)v; p.coupling = coupling;
  p.o_pairs = (__bf16*)const_cast<void*>(o_pairs); p.lse = const_cast<float*>(lse);
  p.dout = (const __bf16*)dout; p.delta = delta;
  p.dq = (__bf16*)dq; p.dk = (__bf16*)dk; p.dv = (__bf16*)dv;
  p.n = n; p.l = l; p.c = c; p.nh = n_heads; p.nb = n_bands; p.hd = c / n_heads;
  p.scale = scale; p.scale_log2 = scale * 1.4426950408889634f;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(cfc_attn_rowdot_kernel, dim3(cdiv(l, RD_ROWS), n, n_bands * n_bands), dim3(AT_THREADS), 0, st, p,
                     dcoupling_part);
  UNET_LAUNCH_CHECK("cfc_attn_bwd (rowdot)");
  if (p.hd <= 32) return launch_bwd<32>(p, st);
  if (p.hd <= 64) return launch_bwd<64>(p, st);
  if (p.hd <= 128) return launch_bwd<128>(p, st);
  return launch_bwd<256>(p, st);
}
