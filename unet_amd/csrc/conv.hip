// Implicit-GEMM tap-list convolution on CDNA4 matrix cores (gfx950).
//
// One kernel family covers Conv1d (k=1/3/merged 4-branch dilated k=7, stride 1/2),
// ConvTranspose1d k4 s2 p1 (two output phases) and all of their data gradients; a second
// family computes weight gradients.  See include/unet_hip.h for the math and the reference
// call sites.
//
// Layout decisions (MI355X-first, not a translation of the reference's NCL tensors):
//   * activations are channels-last [n][row][c]: the GEMM reduction dim (channels) is the
//     contiguous one, so an MFMA operand fragment (8 consecutive k) is ONE 16-byte LDS read
//     at any tap shift -- dilation/stride only move the LDS row index.
//   * per K-chunk the input halo tile (all rows any tap of the phase touches) is staged once
//     in LDS and reused by every tap (23 taps for the merged dilated conv).
//   * weights are pre-packed (unet_conv1d_pack) tile-major in the exact swizzled LDS image, so
//     a 16 KB tile is one contiguous global range streamed with global_load_lds (LDS-DMA,
//     no VGPR round trip), double-buffered against the MFMAs.
//   * LDS rows are 128 B; 16-B chunk index is XORed with (row>>1)&7 so that the 16 rows a
//     ds_read_b128 lane group touches land on 16 distinct 16-B slots of the 256-B bank row.
#include "common.h"
#include <stdlib.h>
#include <mutex>

namespace {

constexpr int BM = 128;            // output channels per block
constexpr int WTILE_BYTES = 16384; // 128 rows x 128 B
constexpr int THREADS = 256;

struct ConvK {
  const char* x1; const char* x2; const char* wp; const float* bias; const char* res; char* y;
  long long x1_ns, x2_ns, y_ns, res_ns;  // elements
  int x1_ld, x2_ld, y_ld, res_ld;
  int c1, c2, kc1, kctot;
  int l_in, l_out, c_out, n;
  int stride, up, act;
  int mtiles;
  int group;             // conv_fwd2 block order: `group` position tiles x all channel slabs are neighbours (0: slab-major)
  int ptiles[2];
  int pbeg[3];
  int mcount[2];
  // LDS image of the input halo tile, per output phase: up to two row segments (input-row parity
  // classes when stride == 2, so that every tap reads LDS rows with unit stride).
  // segment c holds input rows  stride*(m0 + seg_q0[ph][c] + i) + c,  i in [0, seg_rows[ph][c])
  int seg_rows[2][2];
  int seg_q0[2][2];
  int x1_fast, x2_fast;  // 16-byte aligned rows and channel count a multiple of the chunk: LDS-DMA staging
  int x1_vec, x2_vec, y_vec, res_vec;
  int xdb;               // input tile double-buffered
  int xbytes;            // bytes of one input tile buffer (multiple of 1 KiB)
  int tap_row[UNET_MAX_TAPS];  // LDS row that local output position 0 reads for this tap
  float* s1; float* s2;  // conv_fwd2 only, optional: per-(window, position tile, channel) sum / sum of squares of y
  const float* gate; char* y2;  // conv_stream gated epilogue only: y2 = res + gate[n][co] * y (y itself optional)
};

__device__ const u32x4 g_zero_chunk = {0u, 0u, 0u, 0u};  // source of padding rows for LDS-DMA staging

template <bool F32> struct Mma;
template <> struct Mma<false> {
  typedef bf16x8 AB;
  __device__ static __forceinline__ AB ld(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }
  __device__ static __forceinline__ f32x4 mma(AB a, AB b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<true> {
  typedef float AB;
  __device__ static __forceinline__ AB ld(const char* p) { return *reinterpret_cast<const float*>(p); }
  __device__ static __forceinline__ f32x4 mma(AB a, AB b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// ---------------------------------------------------------------------------------------------
// forward / data-grad kernel
// ---------------------------------------------------------------------------------------------
template <bool F32, int BN>
__global__ __launch_bounds__(THREADS, 2) void conv_fwd_kernel(const ConvK p) {
  typedef typename Elem<F32>::T T;
  constexpr int EPC = Elem<F32>::EPC;
  constexpr int KC = EPC * 8;          // K elements per 128-byte LDS row
  constexpr int NI = BN / 32;          // 16-wide position fragments per wave
  constexpr int KSTEPS = F32 ? 8 : 2;  // MFMA k-steps per 128-byte row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wbuf = smem;
  char* xbuf = smem + 2 * WTILE_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lq = lane >> 4;

  // ---- which tile -------------------------------------------------------------------------
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int ppw = p.ptiles[0] + p.ptiles[1];
  const int per_mt = p.n * ppw;
  const int mt = wg / per_mt;
  int rem = wg - mt * per_mt;
  const int n = rem / ppw;
  int pt = rem - n * ppw;
  int phase = 0;
  if (pt >= p.ptiles[0]) { phase = 1; pt -= p.ptiles[0]; }
  const int m0 = pt * BN;
  const int tb = p.pbeg[phase];
  const int nt = p.pbeg[phase + 1] - tb;
  const int rows0 = p.seg_rows[phase][0], rows1 = p.seg_rows[phase][1];
  const int R = rows0 + rows1;
  const int g0a = p.stride * (m0 + p.seg_q0[phase][0]);       // input row of LDS row 0 (segment 0)
  const int g0b = p.stride * (m0 + p.seg_q0[phase][1]) + 1;   // input row of LDS row rows0 (segment 1)

  f32x4 acc[4][NI];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- staging ------------------------------------------------------------------------------
  auto issue_w = [&](int step, int buf) {
    const int kc = step / nt, j = step - kc * nt;
    const char* tile = p.wp + ((size_t)((tb + j) * p.mtiles + mt) * p.kctot + kc) * WTILE_BYTES;
    char* dst = wbuf + buf * WTILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = wave * 4 + i;  // 1 KiB per wave-instruction
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(tile + piece * 1024 + lane * 16),
          (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
    }
  };

  // input halo tile of K-chunk kc -> LDS buffer `buf`.  Fast path: LDS-DMA, 8 rows (1 KiB) per
  // wave-instruction; the chunk swizzle is applied on the per-lane SOURCE address and padding /
  // out-of-range chunks read a 16-byte zero constant.  Slow path (odd channel counts, unaligned
  // views): through registers.
  auto issue_x = [&](int kc, int buf) {
    const char* src; int ld, csrc, cbase, fast, vec; long long ns;
    if (kc < p.kc1) { src = p.x1; ld = p.x1_ld; csrc = p.c1; cbase = kc * KC; ns = p.x1_ns; fast = p.x1_fast; vec = p.x1_vec; }
    else { src = p.x2; ld = p.x2_ld; csrc = p.c2; cbase = (kc - p.kc1) * KC; ns = p.x2_ns; fast = p.x2_fast; vec = p.x2_vec; }
    const T* base = reinterpret_cast<const T*>(src) + (long long)n * ns;
    char* dst = xbuf + buf * p.xbytes;
    if (fast) {
      const int pieces = (R + 7) >> 3;
      const int rl = lane >> 3, cp = lane & 7;
      for (int pc = wave; pc < pieces; pc += THREADS / 64) {
        const int i = pc * 8 + rl;
        const int c = swz(i, cp);
        const int g = i < rows0 ? g0a + p.stride * i : g0b + p.stride * (i - rows0);
        const int e0 = cbase + c * EPC;
        const bool ok = i < R && g >= 0 && g < p.l_in && e0 < csrc;
        const char* sp = ok ? reinterpret_cast<const char*>(base + (long long)g * ld + e0)
                            : reinterpret_cast<const char*>(&g_zero_chunk);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp,
                                         (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, 0, 0);
      }
    } else {
      for (int idx = tid; idx < R * 8; idx += THREADS) {
        const int i = idx >> 3, c = idx & 7;
        const int g = i < rows0 ? g0a + p.stride * i : g0b + p.stride * (i - rows0);
        const int e0 = cbase + c * EPC;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (g >= 0 && g < p.l_in && e0 < csrc) {
          const T* ptr = base + (long long)g * ld + e0;
          if (vec && e0 + EPC <= csrc) {
            v = *reinterpret_cast<const u32x4*>(ptr);
          } else {
            union { u32x4 u; T e[EPC]; } tmp;
            tmp.u = u32x4{0u, 0u, 0u, 0u};
            for (int k = 0; k < EPC; ++k)
              if (e0 + k < csrc) tmp.e[k] = ptr[k];
            v = tmp.u;
          }
        }
        *reinterpret_cast<u32x4*>(dst + i * 128 + (swz(i, c) << 4)) = v;
      }
    }
  };

  // per-lane A-fragment byte offset inside a W tile (tap independent)
  int a_off[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) a_off[mi] = (wm * 64 + mi * 16 + l15) * 128;
  const int a_sw = (l15 >> 1) & 7;  // (row>>1)&7 for rows = 16k + l15

  auto compute = [&](int j, const char* wb, const char* xb) {
    const int r0 = p.tap_row[tb + j] + wn * (BN / 2) + l15;
    int b_off[NI], b_sw[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int row = r0 + ni * 16;
      b_off[ni] = row * 128;
      b_sw[ni] = (row >> 1) & 7;
    }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int chunk = F32 ? ks : ks * 4 + lq;
      const int sub = F32 ? lq * 4 : 0;
      typename Mma<F32>::AB a[4], b[NI];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) a[mi] = Mma<F32>::ld(wb + a_off[mi] + ((chunk ^ a_sw) << 4) + sub);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = Mma<F32>::ld(xb + b_off[ni] + ((chunk ^ b_sw[ni]) << 4) + sub);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = Mma<F32>::mma(a[mi], b[ni], acc[mi][ni]);
    }
  };

  // ---- main loop: steps = K-chunks x taps.  Weight tiles are always prefetched one step ahead by
  // LDS-DMA; the input tile of the next K-chunk is prefetched too when two of them fit (small tap
  // spans: 1x1, k=3, transposed conv), otherwise it is restaged between chunks (23-tap convs, where
  // one tile feeds 23 steps and the second resident block covers the bubble). ---------------------
  const int steps = p.kctot * nt;
  if (steps > 0) {
    issue_w(0, 0);
    issue_x(0, 0);
    int kc = 0, j = 0;
    for (int s = 0; s < steps; ++s) {
      __syncthreads();  // everything issued so far has landed (vmcnt(0)) and the previous step's reads are done
      const bool more = s + 1 < steps;
      const bool newchunk = more && (j + 1 == nt);
      if (more) issue_w(s + 1, (s + 1) & 1);
      if (newchunk && p.xdb) issue_x(kc + 1, (kc + 1) & 1);
      compute(j, wbuf + (s & 1) * WTILE_BYTES, xbuf + (p.xdb ? (kc & 1) * p.xbytes : 0));
      if (newchunk && !p.xdb) {
        __syncthreads();  // every wave is done with the input tile of this K-chunk
        issue_x(kc + 1, 0);
      }
      if (++j == nt) { j = 0; ++kc; }
    }
  }

  // ---- epilogue: bias, residual, activation, store (4 consecutive channels per lane) --------
  T* ybase = reinterpret_cast<T*>(p.y) + (long long)n * p.y_ns;
  const T* rbase = p.res ? reinterpret_cast<const T*>(p.res) + (long long)n * p.res_ns : nullptr;
  const int mcount = p.mcount[phase];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int m = m0 + wn * (BN / 2) + ni * 16 + l15;
    if (m >= mcount) continue;
    const long long o = (long long)p.up * m + phase;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int co = mt * BM + wm * 64 + mi * 16 + lq * 4;
      if (co >= p.c_out) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[mi][ni][r];
      const bool full = co + 4 <= p.c_out;
      if (p.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (co + r < p.c_out) v[r] += p.bias[co + r];
      }
      if (rbase) {
        const T* rp = rbase + o * p.res_ld + co;
        if (full && p.res_vec) {
          float rv[4];
          VecIO<T, 4>::load(rp, rv);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += rv[r];
        } else {
          for (int r = 0; r < 4; ++r)
            if (co + r < p.c_out) v[r] += to_f32(rp[r]);
        }
      }
      if (p.act != UNET_ACT_NONE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = act_fwd(p.act, v[r]);
      }
      T* yp = ybase + o * p.y_ld + co;
      if (full && p.y_vec) {
        VecIO<T, 4>::store(yp, v);
      } else {
        for (int r = 0; r < 4; ++r)
          if (co + r < p.c_out) yp[r] = from_f32<T>(v[r]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// forward / data-grad kernel v2 (bf16, 16-byte aligned channel runs).  Same tiling, LDS images and epilogue as
// conv_fwd_kernel; what changes is the instruction stream around the MFMAs, which in v1 carried ~2 VALU and
// ~2 SALU instructions per MFMA (rocprofv3: SQ_INSTS_VALU / SQ_INSTS_VALU_MFMA_MOPS) and was issue-bound:
//   * both operands are staged with buffer_load ... lds: the per-lane part of every address is computed once
//     per block (rows of the halo tile never change), rows outside the window fail the descriptor's range
//     check and arrive as zeros, the weight tile needs no per-lane address at all (instruction offsets);
//   * LDS fragment addresses: A per block, B once per tap (row swizzle of the tap's first row), all
//     fragment / k-step / buffer offsets are instruction immediates (step loop unrolled by two);
//   * step bookkeeping is incremental scalar arithmetic (no integer divisions in the loop).
// ---------------------------------------------------------------------------------------------
constexpr int F2_MAXP = 12;   // x-tile pieces (8 rows of 128 B) per wave: halo tiles up to 384 rows

// HALF: every K-chunk of the conv holds at most 32 channels (the first conv's Cin = 32): the second k-step of each
// chunk would multiply zeros and is compiled out (as a run-time flag it cost the other convs 0.3-1.5 %)
template <int BN, bool HALF = false>
__device__ __forceinline__ void conv_fwd2_body(const ConvK& p) {
  typedef __bf16 T;
  constexpr int NI = BN / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int XB = 2 * WTILE_BYTES;   // byte offset of the input tile buffers

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lq = lane >> 4;

  // ---- which tile -------------------------------------------------------------------------
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int ppw = p.ptiles[0] + p.ptiles[1];
  const int per_mt = p.n * ppw;
  // Block order.  Slab-major (all tiles of channel slab 0, then slab 1, ...) hands the slabs of one input tile to
  // different XCDs, and every XCD's L2 fetches that tile for itself.  With four or more slabs the order is grouped:
  // `group` position tiles x all slabs are neighbours, i.e. resident on one XCD together, so the input tile is
  // fetched into that L2 once (the blocks of a slab still stream the same weight tiles in lock-step).  Fabric
  // traffic of the 128-wide kernel: 1068 -> 825 MB per launch (profiles/r02_hbm_traffic.json); time unchanged.
  int mt, rem;
  if (p.group) {
    const int gp = p.group, full = per_mt / gp, lim = full * gp * p.mtiles;
    if (wg < lim) {
      const int grp = wg / (gp * p.mtiles), r2 = wg - grp * gp * p.mtiles;
      mt = r2 / gp;
      rem = grp * gp + (r2 - mt * gp);
    } else {
      const int tail = per_mt - full * gp, r2 = wg - lim;
      mt = r2 / tail;
      rem = full * gp + (r2 - mt * tail);
    }
  } else {
    mt = wg / per_mt;
    rem = wg - mt * per_mt;
  }
  const int n = rem / ppw;
  int pt = rem - n * ppw;
  int phase = 0;
  if (pt >= p.ptiles[0]) { phase = 1; pt -= p.ptiles[0]; }
  const int m0 = pt * BN;
  const int tb = p.pbeg[phase];
  const int nt = p.pbeg[phase + 1] - tb;
  const int rows0 = p.seg_rows[phase][0], rows1 = p.seg_rows[phase][1];
  const int R = rows0 + rows1;
  const int g0a = p.stride * (m0 + p.seg_q0[phase][0]);
  const int g0b = p.stride * (m0 + p.seg_q0[phase][1]) + 1;

  f32x4 acc[4][NI];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- staging ------------------------------------------------------------------------------
  constexpr unsigned OOB = 0x80000000u;
  const auto wsrd = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 0x7fffffff, 0x00020000);
  const unsigned w_voff = (unsigned)(wave * 4096 + lane * 16);
  const unsigned w_jstride = (unsigned)p.mtiles * p.kctot * WTILE_BYTES;
  const unsigned w_base = ((unsigned)tb * p.mtiles + mt) * p.kctot * WTILE_BYTES;   // tile (tap tb, mt, kc 0)

  auto issue_w = [&](unsigned tile_off, int buf) {   // 16 KiB tile: 4 wave-instructions of 1 KiB per wave
    auto dst = (__attribute__((address_space(3))) void*)(smem + buf * WTILE_BYTES + wave * 4096);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, dst, 16, w_voff, tile_off, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, dst, 16, w_voff, tile_off, 1024, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, dst, 16, w_voff, tile_off, 2048, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, dst, 16, w_voff, tile_off, 3072, 0);
  };

  // input tile: piece pc (8 rows) of this wave's k-th slot is pc = wave + 4k.  x_voff[k] = byte offset of the
  // lane's 16-byte chunk inside the window for K-chunk 0 of the current source (rows fixed per block).
  unsigned x_voff[F2_MAXP];
  const int rl = lane >> 3, cp = lane & 7;
  const int pieces = (R + 7) >> 3;
  auto setup_x = [&](int ld, int chunks_valid) {   // chunks_valid: 16-byte chunks of this K-chunk that exist (1..8)
#pragma unroll
    for (int k = 0; k < F2_MAXP; ++k) {
      const int i = (wave + 4 * k) * 8 + rl;
      const int c = swz(i, cp);
      const int g = i < rows0 ? g0a + p.stride * i : g0b + p.stride * (i - rows0);
      const bool ok = i < R && g >= 0 && g < p.l_in && c < chunks_valid;
      x_voff[k] = ok ? (unsigned)((g * ld + c * 8) * 2) : OOB;
    }
  };
  auto x_srd = [&](const char* base, long long ns, int ld, int c) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (long long)n * ns * 2), 0,
                                             (unsigned)(((p.l_in - 1) * ld + c) * 2), 0x00020000);
  };
  const auto srd1 = x_srd(p.x1, p.x1_ns, p.x1_ld, p.c1);
  const auto srd2 = p.c2 > 0 ? x_srd(p.x2, p.x2_ns, p.x2_ld, p.c2) : srd1;
  const int tail1 = (p.c1 & 63) ? ((p.c1 & 63) >> 3) : 8, tail2 = (p.c2 & 63) ? ((p.c2 & 63) >> 3) : 8;

  auto issue_x = [&](int kc, int buf) {
    // (re)derive the per-lane offsets when the source or the chunk validity changes: at most 4 times per block
    const bool src2 = kc >= p.kc1;
    const int kcl = src2 ? kc - p.kc1 : kc;
    const int last = src2 ? (p.kctot - p.kc1 - 1) : (p.kc1 - 1);
    const int tail = src2 ? tail2 : tail1;
    if (kcl == 0 || (kcl == last && tail != 8)) setup_x(src2 ? p.x2_ld : p.x1_ld, kcl == last ? tail : 8);
    const unsigned soff = (unsigned)(kcl * 128);
    char* dst = smem + XB + buf * p.xbytes;
#pragma unroll
    for (int k = 0; k < F2_MAXP; ++k) {
      const int pc = wave + 4 * k;
      if (pc < pieces) {
        if (src2) __builtin_amdgcn_raw_ptr_buffer_load_lds(srd2, (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, x_voff[k], soff, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(srd1, (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, x_voff[k], soff, 0, 0);
      }
    }
  };

  // ---- LDS fragment addresses ---------------------------------------------------------------
  int a_addr[2][4];   // [k-step][mi]; hops between the two weight buffers (xor 16 KiB) once per step
  {
    const int a_sw = (l15 >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) a_addr[ks][mi] = (wm * 64 + mi * 16 + l15) * 128 + (((ks * 4 + lq) ^ a_sw) << 4);
  }
  const int b_row_lane = wn * (BN / 2) + l15;

  auto compute = [&](int tap, int xb_off) {
    const int row0 = p.tap_row[tap] + b_row_lane;          // fragment ni reads row0 + 16*ni: same swizzle for all ni
    const int b_sw = (row0 >> 1) & 7;
    const int b0 = xb_off + row0 * 128 + ((lq ^ b_sw) << 4);
    const int b1 = b0 ^ 64;                                  // k-step 1: chunk index bit 2 flips
    bf16x8 a[4], b[NI];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) a[mi] = *reinterpret_cast<const bf16x8*>(smem + a_addr[0][mi]);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const bf16x8*>(smem + b0 + ni * 2048);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    if constexpr (!HALF) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) a[mi] = *reinterpret_cast<const bf16x8*>(smem + a_addr[1][mi]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const bf16x8*>(smem + b1 + ni * 2048);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) a_addr[ks][mi] ^= WTILE_BYTES;
  };

  // ---- main loop: K-chunks (outer) x taps (inner).  The weight tile of the next step is requested right after
  // the barrier that opens a step; the input tile of the next K-chunk is requested at the first step of a chunk
  // when two tiles fit (xdb), otherwise it is restaged between chunks.  All chunk-level work (source switch,
  // per-lane offsets) stays out of the inner tap loop. ---------------------------------------------------------
  if (p.kctot > 0 && nt > 0) {
    issue_w(w_base, 0);
    issue_x(0, 0);
    int wbuf = 0;
    for (int kc = 0; kc < p.kctot; ++kc) {
      const bool last_chunk = kc + 1 == p.kctot;
      const int xoff = XB + (p.xdb ? (kc & 1) * p.xbytes : 0);
      unsigned w_cur = w_base + (unsigned)kc * WTILE_BYTES;
      // first tap of the chunk
      __syncthreads();   // everything issued so far has landed (vmcnt(0)); previous step's reads are done
      if (nt > 1) { w_cur += w_jstride; issue_w(w_cur, wbuf ^ 1); }
      else if (!last_chunk) issue_w(w_base + (unsigned)(kc + 1) * WTILE_BYTES, wbuf ^ 1);
      if (p.xdb && !last_chunk) issue_x(kc + 1, (kc + 1) & 1);
      compute(tb, xoff);
      wbuf ^= 1;
      // remaining taps
      for (int j = 1; j < nt; ++j) {
        __syncthreads();
        if (j + 1 < nt) { w_cur += w_jstride; issue_w(w_cur, wbuf ^ 1); }
        else if (!last_chunk) issue_w(w_base + (unsigned)(kc + 1) * WTILE_BYTES, wbuf ^ 1);
        compute(tb + j, xoff);
        wbuf ^= 1;
      }
      if (!p.xdb && !last_chunk) {
        __syncthreads();  // every wave is done with the input tile of this K-chunk
        issue_x(kc + 1, 0);
      }
    }
  }

  // ---- epilogue: bias, residual, activation, store (4 consecutive channels per lane) --------
  T* ybase = reinterpret_cast<T*>(p.y) + (long long)n * p.y_ns;
  const T* rbase = p.res ? reinterpret_cast<const T*>(p.res) + (long long)n * p.res_ns : nullptr;
  const int mcount = p.mcount[phase];
  const bool stats = p.s1 != nullptr;
  float st1[4][4], st2[4][4];          // per-channel sum / sum of squares of the values this lane stores
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) { st1[mi][r] = 0.f; st2[mi][r] = 0.f; }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int m = m0 + wn * (BN / 2) + ni * 16 + l15;
    if (m >= mcount) continue;
    const long long o = (long long)p.up * m + phase;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int co = mt * BM + wm * 64 + mi * 16 + lq * 4;
      if (co >= p.c_out) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[mi][ni][r];
      const bool full = co + 4 <= p.c_out;
      if (p.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (co + r < p.c_out) v[r] += p.bias[co + r];
      }
      if (rbase) {
        const T* rp = rbase + o * p.res_ld + co;
        if (full && p.res_vec) {
          float rv[4];
          VecIO<T, 4>::load(rp, rv);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += rv[r];
        } else {
          for (int r = 0; r < 4; ++r)
            if (co + r < p.c_out) v[r] += to_f32(rp[r]);
        }
      }
      if (p.act != UNET_ACT_NONE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = act_fwd(p.act, v[r]);
      }
      if (stats) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float q = to_f32(from_f32<T>(v[r]));      // the value as stored: what a pass over y would read
          st1[mi][r] += q;
          st2[mi][r] = fmaf(q, q, st2[mi][r]);
        }
      }
      T* yp = ybase + o * p.y_ld + co;
      if (full && p.y_vec) {
        VecIO<T, 4>::store(yp, v);
      } else {
        for (int r = 0; r < 4; ++r)
          if (co + r < p.c_out) yp[r] = from_f32<T>(v[r]);
      }
    }
  }
  // ---- optional: the tile's per-channel sum and sum of squares (BatchNorm statistics, models.py:801-817), so that
  // no separate pass has to read y back: 16 positions across the lanes of a row, the two position halves of the
  // tile across waves through LDS, one fp32 partial per (window, position tile, channel) ----------------------
  if (stats) {       // block-uniform
    __syncthreads();   // every wave is past its last fragment read: the LDS is free
    float* sred = reinterpret_cast<float*>(smem);          // [2 position halves][128 channels][2]
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = row16_sum(st1[mi][r]), b = row16_sum(st2[mi][r]);
        if (l15 == 0) {
          const int ch = wm * 64 + mi * 16 + lq * 4 + r;
          sred[(wn * BM + ch) * 2] = a;
          sred[(wn * BM + ch) * 2 + 1] = b;
        }
      }
    __syncthreads();
    const int ch = tid >> 1, which = tid & 1, co = mt * BM + ch;
    if (co < p.c_out) {
      float* dst = which ? p.s2 : p.s1;
      dst[((long long)n * ppw + pt) * p.c_out + co] = sred[ch * 2 + which] + sred[(BM + ch) * 2 + which];
    }
  }
}

__global__ __launch_bounds__(THREADS, 2) void conv_fwd2_kernel_128(const ConvK p) { conv_fwd2_body<128>(p); }
__global__ __launch_bounds__(THREADS, 2) void conv_fwd2_kernel_64(const ConvK p) { conv_fwd2_body<64>(p); }
__global__ __launch_bounds__(THREADS, 2) void conv_fwd2_kernel_192(const ConvK p) { conv_fwd2_body<192>(p); }
__global__ __launch_bounds__(THREADS, 2) void conv_fwd2_kernel_256(const ConvK p) { conv_fwd2_body<256>(p); }
__global__ __launch_bounds__(THREADS, 2) void conv_fwd2_kernel_128h(const ConvK p) { conv_fwd2_body<128, true>(p); }
__global__ __launch_bounds__(THREADS, 2) void conv_fwd2_kernel_192h(const ConvK p) { conv_fwd2_body<192, true>(p); }

// ---------------------------------------------------------------------------------------------
// Streaming kernel for single-tap (1x1) convolutions: FiLM q_proj / out_proj, skip_proj, residual_conv and
// their data gradients (models.py:492,495,605,910,3049).  These have K = Cin only (4..16 K-chunks per output
// tile), move 2 bytes per ~C FLOPs and are HBM-latency-bound in the tap-loop kernels, whose blocks live for a
// handful of steps and prefetch one step ahead.  Here a workgroup is PERSISTENT (one per CU, 8 waves): it walks
// its list of 256(co) x 128(pos) output tiles as one continuous stream of (two weight tiles, input chunk) stages
// through a 3-deep LDS ring filled by buffer_load ... lds, with counted s_waitcnt vmcnt(N) and raw s_barrier so
// that two stages (96 KiB per CU) stay in flight across barriers, tile boundaries and epilogues.  Every wave
// issues exactly 6 loads per stage and 8 stores per tile, which is what makes the counts exact (CDNA4 counts
// loads and stores on one counter).  The 256-channel tile halves the input bytes streamed per FLOP, which is
// what bounds these convs (L2 -> LDS bandwidth), relative to the 128 x 128 tile of the tap-loop kernel.
// ---------------------------------------------------------------------------------------------
// Lane (l15, lq) of an accumulator tile pair holds channels 16*t + 4*lq + r (t = 0, 1: registers a, b; r packed as two
// bf16 pairs).  Returns the 8 consecutive channels 8*lq .. 8*lq + 7 of the same position instead, by exchanging 16-lane
// rows between the two registers (v_permlane32_swap, then v_permlane16_swap; rows are the lq groups).
__device__ __forceinline__ u32x4 quad_regroup(const u32x2 a, const u32x2 b) {
  u32x4 w;
#pragma unroll
  for (int pp = 0; pp < 2; ++pp) {
    const u32x2 t = __builtin_amdgcn_permlane32_swap(a[pp], b[pp], false, false);   // [a0 a1 b0 b1], [a2 a3 b2 b3]
    const u32x2 u = __builtin_amdgcn_permlane16_swap(t[0], t[1], false, false);     // [a0 a2 b0 b2], [a1 a3 b1 b3]
    w[pp] = u[0];
    w[2 + pp] = u[1];
  }
  return w;
}

// the inverse: 8 consecutive channels per lane -> the accumulator layout of the two tiles (both swaps are involutions)
__device__ __forceinline__ void quad_ungroup(const u32x4 w, u32x2& a, u32x2& b) {
#pragma unroll
  for (int pp = 0; pp < 2; ++pp) {
    const u32x2 t = __builtin_amdgcn_permlane16_swap(w[pp], w[2 + pp], false, false);
    const u32x2 u = __builtin_amdgcn_permlane32_swap(t[0], t[1], false, false);
    a[pp] = u[0];
    b[pp] = u[1];
  }
}

constexpr int ST_D = 3;                       // ring depth
constexpr int ST_THREADS = 512;               // 8 waves: 4 (output-channel groups of 64) x 2 (position halves)
constexpr int ST_BM = 256;                    // output channels per tile = two packed 128-row weight tiles
constexpr int ST_SLOT = 3 * WTILE_BYTES;      // 32 KiB weights + 16 KiB input chunk (128 positions x 64 channels)
constexpr int ST_BIAS = ST_D * ST_SLOT;       // fp32 bias of every output channel (<= 8 KiB) behind the ring
constexpr int ST_MAXC = 2048;

// Epilogue operands that arrive from memory (the residual tile: 8 x 16 B per lane; FiLM's gate: 16 floats per lane)
// are requested two steps before the epilogue that consumes them, so that their latency hides behind the stream like
// a stage's does.  They cannot live in compiler-managed registers: loads the compiler knows about get s_waitcnt of its
// own placing (it cannot see the counted waits of the pipeline, so it drains the queue: vmcnt(0) inside the loop),
// and loads hidden in inline asm with compiler-allocated outputs are copied and spilled across the loop like any other
// value, i.e. read while still in flight.  So the kernel OWNS v208..v255: amdgpu_num_vgpr caps the compiler's
// allocation at 208 registers (it uses ~150), the loads name the owned registers in their asm text, and the epilogue
// copies them out with v_mov after the pipeline's own wait has covered them.
#define ST_OWNED_LOAD(I, A, B, C, D)                                                                                  \
  case I:                                                                                                              \
    asm volatile("buffer_load_dwordx4 v[" #A ":" #D "], %0, %1, 0 offen" ::"v"(voff), "s"(rs)                        \
                 : "memory", "v" #A, "v" #B, "v" #C, "v" #D);                                                        \
    break;
#define ST_OWNED_GET(I, A, B, C, D)                                                                                   \
  case I:                                                                                                              \
    asm volatile("v_mov_b32 %0, v" #A "\n\tv_mov_b32 %1, v" #B "\n\tv_mov_b32 %2, v" #C "\n\tv_mov_b32 %3, v" #D       \
                 : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]));                                                  \
    break;
#define ST_OWNED_ALL(X)                                                                                               \
  X(0, 208, 209, 210, 211) X(1, 212, 213, 214, 215) X(2, 216, 217, 218, 219) X(3, 220, 221, 222, 223)                 \
  X(4, 224, 225, 226, 227) X(5, 228, 229, 230, 231) X(6, 232, 233, 234, 235) X(7, 236, 237, 238, 239)                 \
  X(8, 240, 241, 242, 243) X(9, 244, 245, 246, 247) X(10, 248, 249, 250, 251) X(11, 252, 253, 254, 255)
constexpr int ST_OWNED_FIRST = 208;
// slot i (0..7: residual rows, 8..11: gate) <- 16 bytes at voff of the buffer rs; i is a compile-time constant after unrolling
template <class RS>
__device__ __forceinline__ void st_owned_load(const int i, const unsigned voff, const RS rs) {
  switch (i) { ST_OWNED_ALL(ST_OWNED_LOAD) default: break; }
}
__device__ __forceinline__ u32x4 st_owned_get(const int i) {
  unsigned r[4] = {0u, 0u, 0u, 0u};
  switch (i) { ST_OWNED_ALL(ST_OWNED_GET) default: break; }
  return u32x4{r[0], r[1], r[2], r[3]};
}

// MODE 0: y = conv + bias.
// MODE 1 / 2: FiLM's out_proj with its gate and residual as the epilogue (models.py:515-517):
//   y2 = res + gate[n][co] * bf16(conv + bias); MODE 1 also stores y (training keeps o for the gate's gradient),
//   MODE 2 does not (inference).
// MODE 3: y = conv + bias + res (the data gradient of q_proj, which receives the residual path's gradient).
template <int MODE>
__global__ __launch_bounds__(ST_THREADS, 1) __attribute__((amdgpu_num_vgpr(ST_OWNED_FIRST)))
void conv_stream_kernel_t(const ConvK p, const unsigned w_bytes) {
  typedef __bf16 T;
  constexpr bool RES = MODE != 0, GATED = MODE == 1 || MODE == 2, KEEP_O = MODE != 2;
  constexpr int BN = 128, NI = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lq = lane >> 4;
  constexpr unsigned OOB = 0x80000000u;

  const int G = gridDim.x;
  const int lb = xcd_remap(blockIdx.x, G);
  const int ptiles = p.ptiles[0];
  const int mt2 = (p.mtiles + 1) >> 1;          // 256-channel tiles
  const int tiles = mt2 * p.n * ptiles;
  const int my_tiles = lb < tiles ? (tiles - lb + G - 1) / G : 0;
  const int kct = p.kctot;
  const int total = my_tiles * kct;
  const int q0 = p.seg_q0[0][0];               // input row of output position 0 (the tap offset)

  // bias -> LDS once (read back per tile with ds_read: no vector-memory wait in the pipeline)
  float* sbias = reinterpret_cast<float*>(smem + ST_BIAS);
  for (int c = tid; c < mt2 * ST_BM; c += ST_THREADS) sbias[c] = (p.bias && c < p.c_out) ? p.bias[c] : 0.f;

  // packed weights: records = the image's exact size, so the absent second half of a last, odd tile reads zeros
  const auto wsrd = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, w_bytes, 0x00020000);
  const unsigned w_voff = (unsigned)((wave & 3) * 4096 + lane * 16);
  const int rl = lane >> 3, cp = lane & 7;
  const int tail = (p.c1 & 63) ? ((p.c1 & 63) >> 3) : 8;
  const unsigned x_win = (unsigned)(((p.l_in - 1) * p.x1_ld + p.c1) * 2);
  const unsigned y_win = (unsigned)(((p.l_out - 1) * p.y_ld + p.c_out) * 2);

  // ---- issue cursor: (tile index in my list, K-chunk) of the next stage to request ------------------
  int i_tile = 0, i_kc = 0, i_mt = 0, i_n = 0, i_m0 = 0;
  unsigned x_voff[2], x_voff_tail[2];
  auto decode = [&](int it, int& mt, int& n, int& m0) {
    const int t = lb + it * G;
    mt = t % mt2;
    const int r = t / mt2;
    const int pt = r % ptiles;
    n = r / ptiles;
    m0 = pt * BN;
  };
  auto setup_issue_tile = [&]() {
    decode(i_tile, i_mt, i_n, i_m0);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = (wave + 8 * k) * 8 + rl;
      const int c = swz(i, cp);
      const int g = i_m0 + q0 + i;
      const bool ok = g >= 0 && g < p.l_in;
      x_voff[k] = ok ? (unsigned)((g * p.x1_ld + c * 8) * 2) : OOB;
      x_voff_tail[k] = (ok && c < tail) ? x_voff[k] : OOB;
    }
  };
  auto issue_slot = [&](int slot) {   // exactly 6 loads per wave
    const auto xsrd = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x1 + (long long)i_n * p.x1_ns * 2), 0, x_win, 0x00020000);
    // waves 0-3 fetch the 128-row tile 2*mt, waves 4-7 the tile 2*mt+1 (if it exists; else out of range -> zeros)
    const unsigned w_off = (unsigned)((2 * i_mt + (wave >> 2)) * kct + i_kc) * WTILE_BYTES;
    auto wdst = (__attribute__((address_space(3))) void*)(smem + slot * ST_SLOT + wave * 4096);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, wdst, 16, w_voff + w_off, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, wdst, 16, w_voff + w_off, 0, 1024, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, wdst, 16, w_voff + w_off, 0, 2048, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, wdst, 16, w_voff + w_off, 0, 3072, 0);
    const unsigned soff = (unsigned)(i_kc * 128);
    const bool last = i_kc == kct - 1;
    char* xdst = smem + slot * ST_SLOT + 2 * WTILE_BYTES;
#pragma unroll
    for (int k = 0; k < 2; ++k)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (__attribute__((address_space(3))) void*)(xdst + (wave + 8 * k) * 1024), 16,
                                               last ? x_voff_tail[k] : x_voff[k], soff, 0, 0);
    if (++i_kc == kct) { i_kc = 0; ++i_tile; if (i_tile < my_tiles) setup_issue_tile(); }
  };

  // ---- LDS fragment addresses (slot 0); the tap row is 0 for a single tap ------------------------------
  int a_addr[2][4], b_addr[2];
  {
    const int a_sw = (l15 >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) a_addr[ks][mi] = (wm * 64 + mi * 16 + l15) * 128 + (((ks * 4 + lq) ^ a_sw) << 4);
    const int row0 = wn * (BN / 2) + l15;
    const int b_sw = (row0 >> 1) & 7;
    b_addr[0] = 2 * WTILE_BYTES + row0 * 128 + ((lq ^ b_sw) << 4);
    b_addr[1] = b_addr[0] ^ 64;
  }

  f32x4 acc[4][NI];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();

  // ---- epilogue operands: residual rows and gate values of a tile, into the owned registers -------------------
  const unsigned r_win = RES ? (unsigned)(((p.l_out - 1) * p.res_ld + p.c_out) * 2) : 0u;
  auto issue_res = [&](int tile) {     // exactly 8 (+ 4 gated) loads per wave (out-of-range rows / channels read zeros)
    int mt, n, m0;
    decode(tile, mt, n, m0);
    const auto rsrd = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res + (long long)n * p.res_ns * 2), 0, r_win, 0x00020000);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int m = m0 + wn * (BN / 2) + ni * 16 + l15;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int co = mt * ST_BM + wm * 64 + h * 32 + lq * 8;
        const unsigned voff = (m < p.l_out && co < p.c_out) ? (unsigned)((m * p.res_ld + co) * 2) : OOB;
        st_owned_load(ni * 2 + h, voff, rsrd);
      }
    }
    if constexpr (GATED) {
      const auto gsrd = __builtin_amdgcn_make_buffer_rsrc((void*)(p.gate + (long long)n * p.c_out), 0,
                                                          (unsigned)p.c_out * 4u, 0x00020000);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int co = mt * ST_BM + wm * 64 + h * 32 + lq * 8;
        const unsigned voff = co < p.c_out ? (unsigned)(co * 4) : OOB;
        st_owned_load(8 + 2 * h, voff, gsrd);
        st_owned_load(9 + 2 * h, voff + 16u, gsrd);
      }
    }
  };

  if (total > 0) {
    setup_issue_tile();
    __syncthreads();                      // bias in LDS
#pragma unroll
    for (int s = 0; s < ST_D - 1; ++s)
      if (s < total) issue_slot(s);

    // The K loop is skewed by half a stage: the second 32-channel half of stage g-1 (fragments already in
    // registers) multiplies while the first half of stage g is read from LDS, and vice versa, so no LDS read
    // latency is exposed between barriers.  A tile's epilogue therefore runs one step after its last stage.
    auto epilogue = [&](int tile) {
      // + bias, bf16, 8 buffer stores per wave (rows / channels outside the tensor are dropped by the range
      // check, so the store COUNT is the same for every wave and tile)
      int mt, n, m0;
      decode(tile, mt, n, m0);
      const auto ysrd = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (long long)n * p.y_ns * 2), 0, y_win, 0x00020000);
      const auto y2srd = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y2 + (long long)n * p.y_ns * 2), 0, y_win, 0x00020000);
      u32x4 gq[2][2];
      if constexpr (GATED) {
#pragma unroll
        for (int h = 0; h < 2; ++h) { gq[h][0] = st_owned_get(8 + 2 * h); gq[h][1] = st_owned_get(9 + 2 * h); }
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int m = m0 + wn * (BN / 2) + ni * 16 + l15;
        // MODE 3: conv + bias + res is rounded ONCE, as the tap-loop kernels do, so the residual (loaded as 8
        // consecutive channels per lane) is first taken back to the accumulator layout
        u32x2 ra[4];
        if constexpr (MODE == 3) {
#pragma unroll
          for (int h = 0; h < 2; ++h) quad_ungroup(st_owned_get(ni * 2 + h), ra[2 * h], ra[2 * h + 1]);
        }
        u32x2 o[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const int co = mt * ST_BM + wm * 64 + mi * 16 + lq * 4;
          const f32x4 bv = *reinterpret_cast<const f32x4*>(sbias + co);
          bf16x4 v;
          if constexpr (MODE == 3) {
            const bf16x4 rv = __builtin_bit_cast(bf16x4, ra[mi]);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (__bf16)(acc[mi][ni][r] + bv[r] + (float)rv[r]);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (__bf16)(acc[mi][ni][r] + bv[r]);
          }
          o[mi] = __builtin_bit_cast(u32x2, v);
        }
        // An accumulator lane holds 4 channels of 4 tiles (8-byte stores, 32 contiguous bytes per row and
        // instruction).  Two lane-row swaps per register pair regroup them into 8 consecutive channels per lane:
        // 16-byte stores, 64 contiguous bytes per row and instruction, half as many store instructions.
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const u32x4 w = quad_regroup(o[2 * h], o[2 * h + 1]);
          const int co = mt * ST_BM + wm * 64 + h * 32 + lq * 8;
          const unsigned voff = (m < p.l_out && co < p.c_out) ? (unsigned)((m * p.y_ld + co) * 2) : OOB;
          if constexpr (GATED) {
            if (KEEP_O) __builtin_amdgcn_raw_buffer_store_b128(w, ysrd, voff, 0, 0);
            const bf16x8 ov = __builtin_bit_cast(bf16x8, w), rv = __builtin_bit_cast(bf16x8, st_owned_get(ni * 2 + h));
            const f32x4 g0 = __builtin_bit_cast(f32x4, gq[h][0]), g1 = __builtin_bit_cast(f32x4, gq[h][1]);
            bf16x8 ov2;
#pragma unroll
            for (int i = 0; i < 8; ++i) ov2[i] = from_f32<T>((float)rv[i] + (i < 4 ? g0[i & 3] : g1[i & 3]) * (float)ov[i]);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ov2), y2srd, voff, 0, 0);
          } else {
            __builtin_amdgcn_raw_buffer_store_b128(w, ysrd, voff, 0, 0);
          }
        }
      }
      zero_acc();
    };
    auto mma = [&](const bf16x8 (&a)[4], const bf16x8 (&b)[NI]) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    };

    int c_tile = 0, c_kc = 0, slot = 0, after = 0;
    bool r_prev = false;                  // the previous step requested a residual tile (12 loads, before its stage)
    bf16x8 a0[4], b0[NI], a1[4], b1[NI];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a1[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; b1[i] = a1[i]; }
#pragma nounroll                          // also keeps the first step from being peeled off (a second copy of the body)
    for (int g = 0; g < total; ++g) {
      // stage g must have landed; the stage requested after it (6 loads per wave) may stay in flight -- plus, for
      // the two steps after a tile epilogue, that tile's 8 stores, which are younger than the stages that were in
      // flight when they were issued (loads and stores share the counter and retire in order)
      // With epilogue operands: a tile's residual (/ gate) request (8 or 12 loads) goes out at the step of its
      // second-to-last stage, BEFORE that step's stage request, so in the queue it sits between the two stages in
      // flight at that point: the next step may leave it and the stage behind it in flight (6 + 8 or 12); the step
      // after that waits for a stage younger than it (6), and that is the step whose second half runs the epilogue.
      // A gated epilogue that keeps o stores 16 times per wave instead of 8.
      const int rem = total - 1 - g;
      if (rem >= 1) {
        if (RES && r_prev) {
          if (GATED) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        } else if (after > 0) {
          if (MODE == 1) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (after > 0) --after;
      r_prev = false;
      __builtin_amdgcn_s_barrier();       // every wave's part of stage g is in LDS; stage g-1's buffer is free
      if constexpr (RES)
        if (c_kc == kct - 2) { issue_res(c_tile); r_prev = true; }
      if (g + ST_D - 1 < total) issue_slot(slot == 0 ? ST_D - 1 : slot - 1);

      const int so = slot * ST_SLOT;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) a0[mi] = *reinterpret_cast<const bf16x8*>(smem + a_addr[0][mi] + so);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b0[ni] = *reinterpret_cast<const bf16x8*>(smem + b_addr[0] + so + ni * 2048);
      mma(a1, b1);                        // second half of stage g-1 (zeros at g == 0)
      // one fragment read per two MFMAs: the eight waves' reads spread over the block instead of arriving as a burst
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (g > 0 && c_kc == 0) {           // ... which closed a tile
        epilogue(c_tile - 1);
        after = 2;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) a1[mi] = *reinterpret_cast<const bf16x8*>(smem + a_addr[1][mi] + so);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b1[ni] = *reinterpret_cast<const bf16x8*>(smem + b_addr[1] + so + ni * 2048);
      mma(a0, b0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      __builtin_amdgcn_sched_barrier(0);
      slot = slot + 1 == ST_D ? 0 : slot + 1;
      if (++c_kc == kct) { c_kc = 0; ++c_tile; }
    }
    mma(a1, b1);
    epilogue(c_tile - 1);
  }
}

constexpr auto conv_stream_kernel = conv_stream_kernel_t<0>;
constexpr auto conv_stream_gated_kernel = conv_stream_kernel_t<1>;        // stores o and out
constexpr auto conv_stream_gated_out_kernel = conv_stream_kernel_t<2>;    // stores out only
constexpr auto conv_stream_res_kernel = conv_stream_kernel_t<3>;          // + residual

// ---------------------------------------------------------------------------------------------
// weight packing:  fp32 w[tap][row][col] (strided)  ->  [tap][row tile][K chunk][128 rows][8 chunks x 16 B]
// in the swizzled LDS image of conv_fwd_kernel.
// ---------------------------------------------------------------------------------------------
struct PackK {
  const float* w; long long ts, rs, cs;
  int n_taps, row_lo, rows, col_lo, c1, c2, kc1, kctot, mtiles;
  int tap_src[UNET_MAX_TAPS];
  char* out;
};

template <bool F32>
__device__ __forceinline__ void pack_job(const PackK& p) {
  typedef typename Elem<F32>::T T;
  constexpr int EPC = Elem<F32>::EPC;
  constexpr int KC = EPC * 8;
  const long long total = (long long)p.n_taps * p.mtiles * p.kctot * 1024;  // 16-B chunks
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * 256) {
    const int cp = (int)(idx & 7);           // physical chunk in the row
    const int row = (int)((idx >> 3) & 127);
    long long t = idx >> 10;
    const int kc = (int)(t % p.kctot); t /= p.kctot;
    const int mt = (int)(t % p.mtiles);
    const int tap = (int)(t / p.mtiles);
    const int c = swz(row, cp);              // logical chunk (XOR is an involution)
    const int r = mt * 128 + row;
    union { u32x4 u; T e[EPC]; } v;
    v.u = u32x4{0u, 0u, 0u, 0u};
    if (r < p.rows) {
      const float* src = p.w + (long long)p.tap_src[tap] * p.ts + (long long)(p.row_lo + r) * p.rs;
      float f[EPC];
#pragma unroll
      for (int k = 0; k < EPC; ++k) {       // the EPC loads issued together (clamped address, masked afterwards)
        int kk, col = -1;
        if (kc < p.kc1) { kk = kc * KC + c * EPC + k; if (kk < p.c1) col = p.col_lo + kk; }
        else { kk = (kc - p.kc1) * KC + c * EPC + k; if (kk < p.c2) col = p.col_lo + p.c1 + kk; }
        f[k] = src[col >= 0 ? (long long)col * p.cs : 0];
        f[k] = col >= 0 ? f[k] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < EPC; ++k) v.e[k] = from_f32<T>(f[k]);
    }
    *reinterpret_cast<u32x4*>(p.out + idx * 16) = v.u;
  }
}

template <bool F32>
__global__ __launch_bounds__(256) void conv_pack_kernel(const PackK p) { pack_job<F32>(p); }

// up to three images in one launch (a conv's forward image and the transposed images its data gradients will need):
// every block walks the jobs in turn, so nothing in a job's loop depends on which job it is
struct PackM { PackK k[UNET_PACK_MAX_JOBS]; int n; };

template <bool F32>
__global__ __launch_bounds__(256) void conv_pack_multi_kernel(const PackM p) {
  pack_job<F32>(p.k[0]);
#pragma unroll
  for (int i = 1; i < UNET_PACK_MAX_JOBS; ++i)
    if (p.n > i) pack_job<F32>(p.k[i]);
}

// ---------------------------------------------------------------------------------------------
// weight-gradient kernel: dW_j[co][ci] = sum_{n,m} dy[n, up*m+phase_j, co] * x[n, stride*m+off_j, ci]
// The reduction runs over positions, i.e. over LDS ROWS of channels-last tiles: both operands
// are read transposed (ds_read_b64_tr_b16 for bf16; plain b32 column reads for fp32).
// Split-K over (window, position-chunk) items into fp32 slabs; a second kernel adds the slabs
// (deterministic, no float atomics).
// ---------------------------------------------------------------------------------------------
struct WgK {
  const char* x1; const char* x2; const char* dy; float* slab;
  long long x1_ns, x2_ns, dy_ns;
  int x1_ld, x2_ld, dy_ld;
  int c1, c2, kt1, kttot;   // 128-channel input tiles per source
  int l_in, c_out, n;
  int stride, up;
  int mtiles, n_taps, splits;
  int mcount[2];
  int x1_vec, x2_vec, dy_vec;
  int tap_off[UNET_MAX_TAPS];
  int tap_phase[UNET_MAX_TAPS];
};

__device__ __forceinline__ int swz_t(int row, int ch) { return ch ^ (((row & 3) << 2) | ((row >> 2) & 3)); }

template <bool F32>
__global__ __launch_bounds__(THREADS, 2) void conv_wgrad_kernel(const WgK p) {
  typedef typename Elem<F32>::T T;
  constexpr int EPC = Elem<F32>::EPC;
  constexpr int BKP = F32 ? 32 : 64;    // positions per staged tile
  constexpr int ROWB = F32 ? 512 : 256; // bytes per LDS row (128 channels)
  constexpr int CPR = ROWB / 16;        // 16-B chunks per row
  constexpr int NLD = BKP * CPR / THREADS;  // chunks per thread per tile (= 4)
  __shared__ __attribute__((aligned(16))) char lds[2 * BKP * ROWB];
  char* dyt = lds;
  char* xt = lds + BKP * ROWB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lq = lane >> 4;

  int wg = xcd_remap(blockIdx.x, gridDim.x);
  // order: (mt, kt) slowest, then split, then tap fastest: blocks sharing an XCD at one time work on
  // the same dy/x rows for different taps -> the operand rows are served from that XCD's L2.
  const int j = wg % p.n_taps; wg /= p.n_taps;
  const int split = wg % p.splits; wg /= p.splits;
  const int kt = wg % p.kttot;
  const int mt = wg / p.kttot;

  const int phase = p.tap_phase[j], off = p.tap_off[j];
  const int mcount = p.mcount[phase];
  const int mchunks = (mcount + BKP - 1) / BKP;
  const int items = p.n * mchunks;
  const int per = (items + p.splits - 1) / p.splits;
  const int it0 = split * per;
  const int it1 = min(items, it0 + per);

  const char* xsrc; int xld, xc, xc0, xvec; long long xns;
  if (kt < p.kt1) { xsrc = p.x1; xld = p.x1_ld; xc = p.c1; xc0 = kt * 128; xns = p.x1_ns; xvec = p.x1_vec; }
  else { xsrc = p.x2; xld = p.x2_ld; xc = p.c2; xc0 = (kt - p.kt1) * 128; xns = p.x2_ns; xvec = p.x2_vec; }
  const int co0 = mt * 128;

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 rdy[NLD], rx[NLD];

  auto load_chunk = [&](const T* base, long long rowoff, bool rowok, int c0, int cmax, int vec) -> u32x4 {
    u32x4 v = {0u, 0u, 0u, 0u};
    if (rowok && c0 < cmax) {
      const T* ptr = base + rowoff + c0;
      if (vec && c0 + EPC <= cmax) v = *reinterpret_cast<const u32x4*>(ptr);
      else {
        union { u32x4 u; T e[EPC]; } tmp;
        tmp.u = u32x4{0u, 0u, 0u, 0u};
        for (int k = 0; k < EPC; ++k)
          if (c0 + k < cmax) tmp.e[k] = ptr[k];
        v = tmp.u;
      }
    }
    return v;
  };

  auto gload = [&](int item) {
    const int n = item / mchunks, mc = item - n * mchunks;
    const T* dyb = reinterpret_cast<const T*>(p.dy) + (long long)n * p.dy_ns;
    const T* xb = reinterpret_cast<const T*>(xsrc) + (long long)n * xns;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * THREADS;
      const int row = idx / CPR, ch = idx % CPR;
      const int m = mc * BKP + row;
      const bool mok = m < mcount;
      const long long o = (long long)p.up * m + phase;
      rdy[i] = load_chunk(dyb, o * p.dy_ld, mok, co0 + ch * EPC, p.c_out, p.dy_vec);
      const long long g = (long long)p.stride * m + off;
      rx[i] = load_chunk(xb, g * xld, mok && g >= 0 && g < p.l_in, xc0 + ch * EPC, xc, xvec);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * THREADS;
      const int row = idx / CPR, ch = idx % CPR;
      const int chp = F32 ? ch : swz_t(row, ch);
      *reinterpret_cast<u32x4*>(dyt + row * ROWB + chp * 16) = rdy[i];
      *reinterpret_cast<u32x4*>(xt + row * ROWB + chp * 16) = rx[i];
    }
  };

  auto compute = [&]() {
    if constexpr (F32) {
#pragma unroll
      for (int kk = 0; kk < BKP / 4; ++kk) {
        const int row = kk * 4 + lq;
        float a[4], b[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          a[mi] = *reinterpret_cast<const float*>(dyt + row * ROWB + (wm * 64 + mi * 16 + l15) * 4);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          b[ni] = *reinterpret_cast<const float*>(xt + row * ROWB + (wn * 64 + ni * 16 + l15) * 4);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
      }
    } else {
      // lane 4q+pp of each 16-lane group addresses row q, columns 4pp..4pp+3 of a 4x16 block;
      // it receives column (lane&15), rows 0..3  (T10 of the CDNA4 guide)
      const int q = l15 >> 2, pp = l15 & 3;
#pragma unroll
      for (int ks = 0; ks < BKP / 32; ++ks) {
        const int r0 = ks * 32 + lq * 8 + q;
        bf16x8 a[4], b[4];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int row = r0 + half * 4;
          const int sw = ((row & 3) << 2) | ((row >> 2) & 3);
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) {
            const int c0 = (wm * 64 + mi * 16) >> 3;
            s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3)))*)(dyt + row * ROWB + (((c0 + (pp >> 1)) ^ sw) << 4) + 8 * (pp & 1)));
            bf16x4 vb = __builtin_bit_cast(bf16x4, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[mi][half * 4 + e] = vb[e];
          }
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            const int c0 = (wn * 64 + ni * 16) >> 3;
            s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3)))*)(xt + row * ROWB + (((c0 + (pp >> 1)) ^ sw) << 4) + 8 * (pp & 1)));
            bf16x4 vb = __builtin_bit_cast(bf16x4, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) b[ni][half * 4 + e] = vb[e];
          }
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
      }
    }
  };

  if (it0 < it1) {
    gload(it0);
    for (int it = it0; it < it1; ++it) {
      __syncthreads();   // previous compute finished reading LDS
      lstore();
      __syncthreads();
      if (it + 1 < it1) gload(it + 1);  // in flight during the MFMAs
      compute();
    }
  }

  // ---- store the fp32 slab tile -------------------------------------------------------------
  const int ctot = p.c1 + p.c2;
  const int ci_base = (kt < p.kt1 ? 0 : p.c1) + xc0;  // logical input-channel index of tile column 0
  float* slab = p.slab + ((long long)split * p.n_taps + j) * p.c_out * ctot;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int cl = xc0 + wn * 64 + ni * 16 + l15;  // channel within its source
    if (cl >= xc) continue;
    const int ci = ci_base - xc0 + cl;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wm * 64 + mi * 16 + lq * 4 + r;
        if (co < p.c_out) slab[(long long)co * ctot + ci] = acc[mi][ni][r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// weight-gradient kernel v2 (bf16, 16-byte aligned channel runs): LDS-DMA staging of both operands into a
// two-stage ring (the loads of position chunk i+1 land while chunk i is in the MFMAs, one barrier per chunk),
// and up to TWO taps per block that share the output-gradient tile: for taps whose offsets differ by
// stride*delta (delta <= 32) the input rows of the second tap are the first tap's rows shifted by delta, so one
// (64+delta)-row input tile feeds both, the dy fragments are read from LDS once for both taps, and L2->LDS
// traffic per FLOP drops by ~40 %.  Same split-K slab output as v1.
// ---------------------------------------------------------------------------------------------
constexpr int WG2_BKP = 64;                    // positions per stage
constexpr int WG2_XROWS = WG2_BKP + 32;        // input rows per stage (max tap distance 32)
constexpr int WG2_DY_BYTES = WG2_BKP * 256;
constexpr int WG2_STAGE = WG2_DY_BYTES + WG2_XROWS * 256;   // 40 KiB

struct Wg2K {
  const char* x1; const char* x2; const char* dy; float* slab;
  long long x1_ns, x2_ns, dy_ns;
  int x1_ld, x2_ld, dy_ld;
  int c1, c2, kt1, kttot;
  int l_in, c_out, n;
  int stride, up;
  int mtiles, n_taps, n_groups, splits;
  int mcount[2];
  int g_tapa[UNET_MAX_TAPS];   // descriptor index of the group's first tap
  int g_tapb[UNET_MAX_TAPS];   // ... of its second tap, or -1
  int g_off[UNET_MAX_TAPS];    // input offset of the first tap
  int g_delta[UNET_MAX_TAPS];  // (off_b - off_a) / stride
  int g_phase[UNET_MAX_TAPS];
  // small-Cin variant (c1 <= 32): up to FOUR taps per group; taps c, d (or -1) and their row distances from tap a
  int g_tapc[UNET_MAX_TAPS], g_tapd[UNET_MAX_TAPS], g_deltac[UNET_MAX_TAPS], g_deltad[UNET_MAX_TAPS];
};

// SMALL: at most 32 input channels (the first conv, inc).  With the regular mapping only a quarter of the 128-wide
// input-channel tile is real and the waves of the upper half (wn = 1) multiply zeros; here a group holds up to four
// taps, each wave computes two of them (wn = 0: taps a, b; wn = 1: taps c, d) on the 32 real channels (two 16-wide
// fragments), so a block does a third of the MFMAs for twice the taps.
template <bool SMALL>
__device__ __forceinline__ void conv_wgrad2_body(const Wg2K& p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lq = lane >> 4;

  int wg = xcd_remap(blockIdx.x, gridDim.x);
  // order: tap group fastest, then the input-channel tile, then the split, output-channel tile slowest: the blocks
  // resident on an XCD at one time reduce the same positions (one split) of one dy slab against all taps and all
  // input-channel tiles, so the dy rows are fetched into that L2 once instead of once per input-channel tile
  const int grp = wg % p.n_groups; wg /= p.n_groups;
  const int kt = wg % p.kttot; wg /= p.kttot;
  const int split = wg % p.splits;
  const int mt = wg / p.splits;

  const int phase = p.g_phase[grp], off0 = p.g_off[grp];
  constexpr int NI = SMALL ? 2 : 4;
  // the (up to) two taps of this wave: LDS row distance from tap a, descriptor index (-1: none)
  const int t0 = (SMALL && wn) ? p.g_tapc[grp] : p.g_tapa[grp];
  const int t1 = (SMALL && wn) ? p.g_tapd[grp] : p.g_tapb[grp];
  const int d0 = (SMALL && wn) ? p.g_deltac[grp] : 0;
  const int delta = (SMALL && wn) ? p.g_deltad[grp] : p.g_delta[grp];
  const bool one = !SMALL || t0 >= 0;
  const bool two = t1 >= 0;
  int dmax = p.g_tapb[grp] >= 0 ? p.g_delta[grp] : 0;
  if (SMALL) {
    if (p.g_tapc[grp] >= 0 && p.g_deltac[grp] > dmax) dmax = p.g_deltac[grp];
    if (p.g_tapd[grp] >= 0 && p.g_deltad[grp] > dmax) dmax = p.g_deltad[grp];
  }
  const int xrows = WG2_BKP + dmax;
  const int mcount = p.mcount[phase];
  const int mchunks = (mcount + WG2_BKP - 1) / WG2_BKP;
  const int items = p.n * mchunks;
  const int per = (items + p.splits - 1) / p.splits;
  const int it0 = split * per;
  const int it1 = min(items, it0 + per);

  const char* xsrc; int xld, xc, xc0; long long xns;
  if (kt < p.kt1) { xsrc = p.x1; xld = p.x1_ld; xc = p.c1; xc0 = kt * 128; xns = p.x1_ns; }
  else { xsrc = p.x2; xld = p.x2_ld; xc = p.c2; xc0 = (kt - p.kt1) * 128; xns = p.x2_ns; }
  const int co0 = mt * 128;

  f32x4 acc0[4][4], acc1[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) { acc0[a][b] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[a][b] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  // ---- staging: buffer_load ... lds.  One descriptor per window (base = first row of the window, records =
  // the window's bytes), so rows before / after the window and channel runs past the tensor fail the hardware
  // range check and arrive in LDS as zeros; the per-lane part of every offset is loop invariant. -----------
  constexpr unsigned OOB = 0x80000000u;
  const int rl = lane >> 4, pc = lane & 15;   // row within a 4-row piece, physical 16-byte chunk
  unsigned dy_voff[WG2_BKP / 16], x_voff[WG2_XROWS / 16];
#pragma unroll
  for (int i = 0; i < WG2_BKP / 16; ++i) {
    const int row = (wave * (WG2_BKP / 16) + i) * 4 + rl;
    const int e0 = co0 + (swz_t(row, pc) << 3);
    dy_voff[i] = e0 < p.c_out ? (unsigned)((p.up * row * p.dy_ld + e0) * 2) : OOB;
  }
#pragma unroll
  for (int i = 0; i < WG2_XROWS / 16; ++i) {
    const int row = (wave + i * (THREADS / 64)) * 4 + rl;
    const int e0 = xc0 + (swz_t(row, pc) << 3);
    x_voff[i] = (e0 < xc && row < xrows) ? (unsigned)((p.stride * row * xld + e0) * 2) : OOB;
  }
  const unsigned dy_win_bytes = (unsigned)(((p.up * (mcount - 1) + phase) * p.dy_ld + p.c_out) * 2);
  const unsigned x_win_bytes = (unsigned)(((p.l_in - 1) * xld + xc) * 2);
  int cur_n = it0 / mchunks, cur_mc = it0 - cur_n * mchunks;     // (window, chunk) of the NEXT stage to issue

  auto issue = [&](int buf) {
    const auto dsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.dy + ((long long)cur_n * p.dy_ns + (long long)phase * p.dy_ld) * 2), 0, dy_win_bytes, 0x00020000);
    const auto xsr = __builtin_amdgcn_make_buffer_rsrc((void*)(xsrc + (long long)cur_n * xns * 2), 0, x_win_bytes, 0x00020000);
    const int mb = cur_mc * WG2_BKP;
    const unsigned dy_io = (unsigned)(p.up * mb * p.dy_ld * 2);
    const unsigned x_io = (unsigned)((p.stride * mb + off0) * xld * 2);   // may be "negative": wraps past the range
    char* sdy = smem + buf * WG2_STAGE;
    char* sx = sdy + WG2_DY_BYTES;
#pragma unroll
    for (int i = 0; i < WG2_BKP / 16; ++i)
#if defined(WG2_ABLATE) && WG2_ABLATE >= 2      // diagnostic builds only (profiles/r03_kernel_experiments.txt): no dy loads
      if (buf < 0)
#endif
      __builtin_amdgcn_raw_ptr_buffer_load_lds(dsrc, (__attribute__((address_space(3))) void*)(sdy + (wave * (WG2_BKP / 16) + i) * 1024),
                                               16, dy_voff[i] + dy_io, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < WG2_XROWS / 16; ++i)
#if defined(WG2_ABLATE) && WG2_ABLATE >= 1      // ... no x loads
      if (buf < 0)
#endif
      if ((wave + i * (THREADS / 64)) * 4 < xrows)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsr, (__attribute__((address_space(3))) void*)(sx + (wave + i * (THREADS / 64)) * 1024),
                                                 16, x_voff[i] + x_io, 0, 0, 0);
    if (++cur_mc == mchunks) { cur_mc = 0; ++cur_n; }
  };

  // ---- per-lane LDS read addresses (loop invariant; stage and k-step offsets are immediates) ----------------
  const int q = l15 >> 2, pp = l15 & 3;
  int a_addr[2][4], b_addr[2][4], c_addr[2][4];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int row = lq * 8 + half * 4 + q;
    const int sw = ((row & 3) << 2) | ((row >> 2) & 3);
    const int rowb = row + d0;
    const int swb = ((rowb & 3) << 2) | ((rowb >> 2) & 3);
    const int rowd = row + delta;
    const int swd = ((rowd & 3) << 2) | ((rowd >> 2) & 3);
    const int cb = SMALL ? 0 : wn * 64;          // first input channel of this wave's fragments
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a_addr[half][i] = row * 256 + ((((wm * 64 + i * 16) >> 3) + (pp >> 1)) ^ sw) * 16 + 8 * (pp & 1);
      b_addr[half][i] = WG2_DY_BYTES + rowb * 256 + ((((cb + i * 16) >> 3) + (pp >> 1)) ^ swb) * 16 + 8 * (pp & 1);
      c_addr[half][i] = WG2_DY_BYTES + rowd * 256 + ((((cb + i * 16) >> 3) + (pp >> 1)) ^ swd) * 16 + 8 * (pp & 1);
    }
  }
  auto tr_read = [&](int addr) -> bf16x4 {
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(smem + addr));
    return __builtin_bit_cast(bf16x4, v);
  };

  auto compute = [&](int stage_off) {
    if (!one) return;                    // SMALL only: a wave whose half of the group holds no tap
#pragma unroll
    for (int ks = 0; ks < WG2_BKP / 32; ++ks) {
      const int o = stage_off + ks * 32 * 256;
      bf16x8 a[4], b[NI];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          bf16x4 vb = tr_read(a_addr[half][mi] + o);
#pragma unroll
          for (int e = 0; e < 4; ++e) a[mi][half * 4 + e] = vb[e];
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          bf16x4 vb = tr_read(b_addr[half][ni] + o);
#pragma unroll
          for (int e = 0; e < 4; ++e) b[ni][half * 4 + e] = vb[e];
        }
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc0[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi], b[ni], acc0[mi][ni], 0, 0, 0);
      if (two) {
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            bf16x4 vb = tr_read(c_addr[half][ni] + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) b[ni][half * 4 + e] = vb[e];
          }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc1[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi], b[ni], acc1[mi][ni], 0, 0, 0);
      }
    }
  };

  // two-stage ring, one barrier per stage.  The LDS read addresses themselves hop between the stages
  // (+/- WG2_STAGE once per stage) so that the k-step offsets inside compute() stay immediates.
  if (it0 < it1) {
    issue(0);
    int hop = WG2_STAGE, nbuf = 1;
    for (int it = it0; it < it1; ++it) {
      __syncthreads();   // this stage has landed (vmcnt(0) + barrier); every wave is done reading the other one
      if (it + 1 < it1) issue(nbuf);
      compute(0);
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int i = 0; i < 4; ++i) { a_addr[half][i] += hop; b_addr[half][i] += hop; c_addr[half][i] += hop; }
      hop = -hop;
      nbuf ^= 1;
    }
  }

  // ---- store the fp32 slab tiles -------------------------------------------------------------
  const int ctot = p.c1 + p.c2;
  const int ci_off = (kt < p.kt1 ? 0 : p.c1);
  auto store = [&](const f32x4 (&acc)[4][4], int tap) {
    float* slab = p.slab + ((long long)split * p.n_taps + tap) * p.c_out * ctot;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int cl = xc0 + (SMALL ? 0 : wn * 64) + ni * 16 + l15;  // channel within its source
      if (cl >= xc) continue;
      const int ci = ci_off + cl;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + wm * 64 + mi * 16 + lq * 4 + r;
          if (co < p.c_out) slab[(long long)co * ctot + ci] = acc[mi][ni][r];
        }
      }
    }
  };
  if (one) store(acc0, t0);
  if (two) store(acc1, t1);
}

__global__ __launch_bounds__(THREADS, 2) void conv_wgrad2_kernel(const Wg2K p) { conv_wgrad2_body<false>(p); }
__global__ __launch_bounds__(THREADS, 2) void conv_wgrad2s_kernel(const Wg2K p) { conv_wgrad2_body<true>(p); }

// ---------------------------------------------------------------------------------------------
// Weight gradient of SINGLE-TAP convs with wide channel counts (the FiLM q_proj / out_proj at C >= 512,
// models.py:492,495): 256(co) x 256(ci) output tile per workgroup.  With the 128 x 128 tiles of conv_wgrad2 every
// dy slab is staged once per input-channel tile and every x slab once per output-channel tile -- at C = 512 that is
// 1 GiB moved from L2 to LDS for 256 MiB of operands, and the kernel is bound by exactly that fill.  Eight waves
// (4 output-channel groups of 64 x 2 input-channel halves of 128; 128 accumulator registers per lane), one workgroup
// per CU, stages of 64 positions = four 16 KiB sub-tiles [dy co 0-127 | dy co 128-255 | x ci 0-127 | x ci 128-255] in the
// layout of conv_wgrad2 (so the transposed ds_read_b64_tr_b16 fragment reads and their swizzle carry over), two stages.
// (A ring of four 32-position stages with counted vmcnt waits, three stages in flight, measured 15 % SLOWER: twice the
// barriers per MFMA.)  film512 121 -> 109 us, film1024 222 -> 197 us incl. the slab sum (scripts/ab_conv.sh).
// ---------------------------------------------------------------------------------------------
constexpr int WG3_THREADS = 512;
constexpr int WG3_BKP = 64;
constexpr int WG3_SUB = WG3_BKP * 256;          // one sub-tile: 64 rows x 256 B
constexpr int WG3_STAGE = 4 * WG3_SUB;          // 64 KiB

struct Wg3K {
  const char* x; const char* dy; float* slab;
  long long x_ns, dy_ns;
  int x_ld, dy_ld;
  int c_in, c_out, l_in, mcount, n;
  int off0;            // input row of output position 0
  int mt2, kt2, splits;
};

__global__ __launch_bounds__(WG3_THREADS, 1) void conv_wgrad3_kernel(const Wg3K p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lq = lane >> 4;

  int wg = xcd_remap(blockIdx.x, gridDim.x);
  // tiles fastest, split slowest: the (few) tiles of one split are neighbours on one XCD, so that split's dy and x rows
  // are fetched into that L2 once and serve every tile
  const int kt = wg % p.kt2; wg /= p.kt2;
  const int mt = wg % p.mt2;
  const int split = wg / p.mt2;
  const int co0 = mt * 256, ci0 = kt * 256;
  const int mchunks = (p.mcount + WG3_BKP - 1) / WG3_BKP;
  const int items = p.n * mchunks;
  const int per = (items + p.splits - 1) / p.splits;
  const int it0 = split * per;
  const int it1 = min(items, it0 + per);

  f32x4 acc[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- staging: a wave fills half of ONE sub-tile (8 pieces of 4 rows); rows / channels outside the window fail the
  // descriptor's range check and arrive as zeros ----------------------------------------------------------------
  constexpr unsigned OOB = 0x80000000u;
  const int sub = wave >> 1;                         // 0, 1: dy halves; 2, 3: x halves
  const bool is_x = sub >= 2;
  const int ld = is_x ? p.x_ld : p.dy_ld;
  const int cbase = is_x ? ci0 + (sub - 2) * 128 : co0 + sub * 128;
  const int climit = is_x ? p.c_in : p.c_out;
  const int rl = lane >> 4, pc = lane & 15;
  unsigned voff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = ((wave & 1) * 8 + i) * 4 + rl;
    const int e0 = cbase + (swz_t(row, pc) << 3);
    voff[i] = e0 < climit ? (unsigned)((row * ld + e0) * 2) : OOB;
  }
  const unsigned win_bytes = is_x ? (unsigned)(((p.l_in - 1) * p.x_ld + p.c_in) * 2)
                                  : (unsigned)(((p.mcount - 1) * p.dy_ld + p.c_out) * 2);
  const char* src = is_x ? p.x : p.dy;
  const long long ns = is_x ? p.x_ns : p.dy_ns;
  const int roff = is_x ? p.off0 : 0;
  int cur_n = it0 / mchunks, cur_mc = it0 - cur_n * mchunks;

  auto issue = [&](int buf) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (long long)cur_n * ns * 2), 0, win_bytes, 0x00020000);
    const unsigned io = (unsigned)((cur_mc * WG3_BKP + roff) * ld * 2);     // may be "negative": wraps past the range
    char* dst = smem + buf * WG3_STAGE + sub * WG3_SUB + (wave & 1) * 8 * 1024;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16,
                                               voff[i] + io, 0, 0, 0);
    if (++cur_mc == mchunks) { cur_mc = 0; ++cur_n; }
  };

  // ---- per-lane LDS read addresses (stage 0) -------------------------------------------------------------------
  const int q = l15 >> 2, pp = l15 & 3;
  int a_addr[2][4], b_addr[2][8];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int row = lq * 8 + half * 4 + q;
    const int sw = ((row & 3) << 2) | ((row >> 2) & 3);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      a_addr[half][i] = (wm >> 1) * WG3_SUB + row * 256 + (((((wm & 1) * 64 + i * 16) >> 3) + (pp >> 1)) ^ sw) * 16 + 8 * (pp & 1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      b_addr[half][i] = (2 + wn) * WG3_SUB + row * 256 + ((((i * 16) >> 3) + (pp >> 1)) ^ sw) * 16 + 8 * (pp & 1);
  }
  auto tr_read = [&](int addr) -> bf16x4 {
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(smem + addr));
    return __builtin_bit_cast(bf16x4, v);
  };

  auto compute = [&](int stage_off) {
#pragma unroll
    for (int ks = 0; ks < WG3_BKP / 32; ++ks) {
      const int o = stage_off + ks * 32 * 256;
      bf16x8 a[4], b[8];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          bf16x4 vb = tr_read(a_addr[half][mi] + o);
#pragma unroll
          for (int e = 0; e < 4; ++e) a[mi][half * 4 + e] = vb[e];
        }
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
          bf16x4 vb = tr_read(b_addr[half][ni] + o);
#pragma unroll
          for (int e = 0; e < 4; ++e) b[ni][half * 4 + e] = vb[e];
        }
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 8; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
  };

  if (it0 < it1) {
    issue(0);
    int nbuf = 1;
    for (int it = it0; it < it1; ++it) {
      __syncthreads();   // this stage has landed (vmcnt(0) + barrier); every wave is done reading the other one
      if (it + 1 < it1) issue(nbuf);
      compute((nbuf ^ 1) * WG3_STAGE);
      nbuf ^= 1;
    }
  }

  float* slab = p.slab + (long long)split * p.c_out * p.c_in;
#pragma unroll
  for (int ni = 0; ni < 8; ++ni) {
    const int ci = ci0 + wn * 128 + ni * 16 + l15;
    if (ci >= p.c_in) continue;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wm * 64 + mi * 16 + lq * 4 + r;
        if (co < p.c_out) slab[(long long)co * p.c_in + ci] = acc[mi][ni][r];
      }
  }
}

// out = sum of the split-K slabs, in slab order (deterministic).  16 bytes per lane and eight slabs' loads in flight
// per thread: as one scalar load per slab in a run-time loop the sum of a 1x1 conv's 32 slabs was 32 dependent
// memory round trips (16 us for 256 KiB of output).
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* slab, float* out, long long count, int splits) {
  const bool vec = !(count & 3) && !(reinterpret_cast<uintptr_t>(slab) & 15) && !(reinterpret_cast<uintptr_t>(out) & 15);
  const long long c4 = vec ? count >> 2 : 0;               // float4 columns (count is a multiple of 4 in every config)
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < c4; i += (long long)gridDim.x * 256) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 8 <= splits; k += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(slab + (long long)(k + u) * count + 4 * i);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < splits; ++k) s += *reinterpret_cast<const f32x4*>(slab + (long long)k * count + 4 * i);
    *reinterpret_cast<f32x4*>(out + 4 * i) = s;
  }
  for (long long i = 4 * c4 + (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += slab[k * count + i];
    out[i] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct Phases {
  int pbeg[3]; int omin[2]; int omax[2]; int mcount[2];
};

int analyse(const UnetConvDesc* d, Phases* ph) {
  UNET_REQUIRE(d->dtype == UNET_BF16 || d->dtype == UNET_F32, "conv: bad dtype %d", d->dtype);
  UNET_REQUIRE(d->n > 0 && d->l_in > 0 && d->l_out > 0 && d->c1 > 0 && d->c2 >= 0 && d->c_out > 0,
               "conv: non-positive dimension (n=%d l_in=%d l_out=%d c1=%d c2=%d c_out=%d)", d->n, d->l_in,
               d->l_out, d->c1, d->c2, d->c_out);
  UNET_REQUIRE(d->up == 1 || d->up == 2, "conv: up must be 1 or 2 (got %d)", d->up);
  UNET_REQUIRE(d->stride == 1 || d->stride == 2, "conv: stride must be 1 or 2 (got %d)", d->stride);
  UNET_REQUIRE(d->n_taps >= 0 && d->n_taps <= UNET_MAX_TAPS, "conv: n_taps %d out of range", d->n_taps);
  UNET_REQUIRE(d->x1_ld >= d->c1 && (d->c2 == 0 || d->x2_ld >= d->c2) && d->y_ld >= d->c_out,
               "conv: row pitch smaller than channel count");
  int prev = 0;
  ph->pbeg[0] = 0; ph->pbeg[1] = d->n_taps; ph->pbeg[2] = d->n_taps;
  for (int r = 0; r < 2; ++r) { ph->omin[r] = 0; ph->omax[r] = 0; ph->mcount[r] = 0; }
  bool seen1 = false;
  for (int j = 0; j < d->n_taps; ++j) {
    int r = d->tap_phase[j];
    UNET_REQUIRE(r >= 0 && r < d->up, "conv: tap %d has phase %d but up=%d", j, r, d->up);
    UNET_REQUIRE(r >= prev, "conv: taps must be grouped by phase (phase 0 first)");
    if (r == 1 && !seen1) { ph->pbeg[1] = j; seen1 = true; }
    prev = r;
    UNET_REQUIRE(d->tap_off[j] > -4096 && d->tap_off[j] < 4096, "conv: tap offset %d too large", d->tap_off[j]);
  }
  for (int r = 0; r < 2; ++r) {
    int b = ph->pbeg[r], e = ph->pbeg[r + 1];
    if (e > b) {
      int lo = d->tap_off[b], hi = d->tap_off[b];
      for (int j = b; j < e; ++j) { lo = d->tap_off[j] < lo ? d->tap_off[j] : lo; hi = d->tap_off[j] > hi ? d->tap_off[j] : hi; }
      ph->omin[r] = lo; ph->omax[r] = hi;
    }
    // m such that up*m + r < l_out
    ph->mcount[r] = r < d->up ? (d->l_out - r + d->up - 1) / d->up : 0;
    if (ph->mcount[r] < 0) ph->mcount[r] = 0;
  }
  return 0;
}

int kchunks(int dtype, int c) { int kc = dtype == UNET_F32 ? 32 : 64; return (c + kc - 1) / kc; }
int esize(int dtype) { return dtype == UNET_F32 ? 4 : 2; }
bool vec_ok(const void* p, long long ld, long long ns, int es) {
  return aligned16(p) && (ld * es) % 16 == 0 && (ns * es) % 16 == 0;
}
bool vec8_ok(const void* p, long long ld, long long ns, int es) {  // 4-element vectors
  const int b = 4 * es;
  return (reinterpret_cast<uintptr_t>(p) % b) == 0 && (ld * es) % b == 0 && (ns * es) % b == 0;
}

// per-phase LDS image geometry for a tile of `bn` output positions
struct TileGeom { int seg_rows[2][2]; int seg_q0[2][2]; int tap_row[UNET_MAX_TAPS]; int rmax; };

void tile_geom(const UnetConvDesc* d, const Phases& ph, int bn, TileGeom* tg) {
  tg->rmax = 0;
  for (int j = 0; j < UNET_MAX_TAPS; ++j) tg->tap_row[j] = 0;
  for (int r = 0; r < 2; ++r) {
    int qmin[2] = {1 << 30, 1 << 30}, qmax[2] = {-(1 << 30), -(1 << 30)};
    for (int j = ph.pbeg[r]; j < ph.pbeg[r + 1]; ++j) {
      const int off = d->tap_off[j];
      const int c = d->stride == 2 ? ((off % 2) + 2) % 2 : 0;
      const int q = d->stride == 2 ? (off - c) / 2 : off;
      qmin[c] = q < qmin[c] ? q : qmin[c];
      qmax[c] = q > qmax[c] ? q : qmax[c];
    }
    int base = 0;
    for (int c = 0; c < 2; ++c) {
      const bool used = qmax[c] >= qmin[c];
      tg->seg_rows[r][c] = used ? bn + qmax[c] - qmin[c] : 0;
      tg->seg_q0[r][c] = used ? qmin[c] : 0;
    }
    for (int j = ph.pbeg[r]; j < ph.pbeg[r + 1]; ++j) {
      const int off = d->tap_off[j];
      const int c = d->stride == 2 ? ((off % 2) + 2) % 2 : 0;
      const int q = d->stride == 2 ? (off - c) / 2 : off;
      tg->tap_row[j] = (c ? tg->seg_rows[r][0] : 0) + q - qmin[c];
    }
    (void)base;
    const int rows = tg->seg_rows[r][0] + tg->seg_rows[r][1];
    tg->rmax = rows > tg->rmax ? rows : tg->rmax;
  }
}

template <int BN>
int launch_fwd2(ConvK& k, int rmax, hipStream_t st) {
  const int xbytes = ((rmax + 7) / 8) * 1024;
  k.xbytes = xbytes;
  k.xdb = (2 * WTILE_BYTES + 2 * (size_t)xbytes <= 80 * 1024) ? 1 : 0;
  const size_t lds = 2 * WTILE_BYTES + (size_t)xbytes * (k.xdb ? 2 : 1);
  // one source of at most 32 channels: the variant without the second k-step (BN 128 / 192)
  const bool half = BN != 64 && BN != 256 && k.c2 == 0 && k.c1 <= 32;
  void (*kern)(const ConvK) = BN == 256 ? conv_fwd2_kernel_256
                            : BN == 192 ? (half ? conv_fwd2_kernel_192h : conv_fwd2_kernel_192)
                            : BN == 128 ? (half ? conv_fwd2_kernel_128h : conv_fwd2_kernel_128)
                                        : conv_fwd2_kernel_64;
  static bool attr_set[2] = {false, false};
  if (!attr_set[half]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    UNET_REQUIRE(e == hipSuccess, "conv: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    attr_set[half] = true;
  }
  const long long blocks = (long long)k.mtiles * k.n * (k.ptiles[0] + k.ptiles[1]);
  UNET_REQUIRE(blocks > 0 && blocks < (1ll << 31), "conv: grid of %lld blocks", blocks);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(THREADS), lds, st, k);
  UNET_LAUNCH_CHECK("conv_fwd2");
  return 0;
}

int launch_stream(ConvK& k, hipStream_t st) {
  static bool attr_set = false;
  static int n_cu = 0;
  const int mt2 = (k.mtiles + 1) / 2;
  const size_t lds = ST_BIAS + (size_t)mt2 * ST_BM * sizeof(float);
  const unsigned w_bytes = (unsigned)((size_t)k.mtiles * k.kctot * WTILE_BYTES);
  void (*kern)(const ConvK, const unsigned) =
      k.gate ? (k.y ? conv_stream_gated_kernel : conv_stream_gated_out_kernel) : (k.res ? conv_stream_res_kernel : conv_stream_kernel);
  if (!attr_set) {
    hipError_t e = hipSuccess;
    void (*all[4])(const ConvK, const unsigned) = {conv_stream_kernel, conv_stream_gated_kernel, conv_stream_gated_out_kernel,
                                                   conv_stream_res_kernel};
    for (int i = 0; i < 4 && e == hipSuccess; ++i)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(all[i]), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    UNET_REQUIRE(e == hipSuccess, "conv: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    int dev = 0;
    e = hipGetDevice(&dev);
    UNET_REQUIRE(e == hipSuccess, "conv: hipGetDevice failed: %s", hipGetErrorString(e));
    e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    UNET_REQUIRE(e == hipSuccess && n_cu > 0, "conv: cannot read the CU count");
    attr_set = true;
  }
  const long long tiles = (long long)mt2 * k.n * k.ptiles[0];
  UNET_REQUIRE(tiles > 0 && tiles < (1ll << 31), "conv: %lld tiles", tiles);
  const unsigned blocks = (unsigned)(tiles < n_cu ? tiles : n_cu);   // one persistent workgroup per CU
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(ST_THREADS), lds, st, k, w_bytes);
  UNET_LAUNCH_CHECK("conv_stream");
  return 0;
}

template <bool F32, int BN>
int launch_fwd(ConvK& k, int rmax, hipStream_t st) {
  static bool attr_set = false;
  const int xbytes = ((rmax + 7) / 8) * 1024;
  k.xbytes = xbytes;
  k.xdb = (2 * WTILE_BYTES + 2 * (size_t)xbytes <= 80 * 1024) ? 1 : 0;
  const size_t lds = 2 * WTILE_BYTES + (size_t)xbytes * (k.xdb ? 2 : 1);
  UNET_REQUIRE(lds <= 160 * 1024, "conv: tap span needs %zu bytes of LDS (max 163840)", lds);
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_fwd_kernel<F32, BN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    UNET_REQUIRE(e == hipSuccess, "conv: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    attr_set = true;
  }
  const long long blocks = (long long)k.mtiles * k.n * (k.ptiles[0] + k.ptiles[1]);
  UNET_REQUIRE(blocks > 0 && blocks < (1ll << 31), "conv: grid of %lld blocks", blocks);
  hipLaunchKernelGGL((conv_fwd_kernel<F32, BN>), dim3((unsigned)blocks), dim3(THREADS), lds, st, k);
  UNET_LAUNCH_CHECK("conv_fwd");
  return 0;
}

}  // namespace

extern "C" size_t unet_conv1d_packed_bytes(int32_t dtype, int32_t n_taps, int32_t rows, int32_t c1, int32_t c2) {
  const int kct = kchunks(dtype, c1) + (c2 > 0 ? kchunks(dtype, c2) : 0);
  return (size_t)n_taps * cdiv(rows, BM) * kct * WTILE_BYTES;
}

namespace {
int fill_pack(PackK* k, int32_t dtype, const float* w, int64_t tap_stride, int64_t row_stride, int64_t col_stride,
              int32_t n_taps, const int32_t* tap_src_host, int32_t row_lo, int32_t rows, int32_t col_lo, int32_t c1,
              int32_t c2, void* packed) {
  UNET_REQUIRE(n_taps > 0 && n_taps <= UNET_MAX_TAPS && rows > 0 && c1 > 0 && c2 >= 0, "pack: bad sizes");
  UNET_REQUIRE(w && packed && aligned16(packed), "pack: null or misaligned pointer");
  k->w = w; k->ts = tap_stride; k->rs = row_stride; k->cs = col_stride;
  k->n_taps = n_taps; k->row_lo = row_lo; k->rows = rows; k->col_lo = col_lo; k->c1 = c1; k->c2 = c2;
  k->kc1 = kchunks(dtype, c1); k->kctot = k->kc1 + (c2 > 0 ? kchunks(dtype, c2) : 0);
  k->mtiles = cdiv(rows, BM);
  for (int j = 0; j < UNET_MAX_TAPS; ++j) k->tap_src[j] = j < n_taps ? (tap_src_host ? tap_src_host[j] : j) : 0;
  k->out = static_cast<char*>(packed);
  return 0;
}
long long pack_chunks(const PackK& k) { return (long long)k.n_taps * k.mtiles * k.kctot * 1024; }
}  // namespace

extern "C" int unet_conv1d_pack(int32_t dtype, const float* w, int64_t tap_stride, int64_t row_stride,
                                int64_t col_stride, int32_t n_taps, const int32_t* tap_src_host,
                                int32_t row_lo, int32_t rows, int32_t col_lo, int32_t c1, int32_t c2,
                                void* packed, void* stream) {
  UNET_REQUIRE(dtype == UNET_BF16 || dtype == UNET_F32, "pack: bad dtype %d", dtype);
  PackK k;
  if (fill_pack(&k, dtype, w, tap_stride, row_stride, col_stride, n_taps, tap_src_host, row_lo, rows, col_lo, c1, c2,
                packed)) return -1;
  const long long total = pack_chunks(k);
  const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == UNET_F32) hipLaunchKernelGGL(conv_pack_kernel<true>, dim3(blocks), dim3(256), 0, st, k);
  else hipLaunchKernelGGL(conv_pack_kernel<false>, dim3(blocks), dim3(256), 0, st, k);
  UNET_LAUNCH_CHECK("conv_pack");
  return 0;
}

extern "C" int unet_conv1d_pack_multi(int32_t dtype, const UnetPackJob* jobs, int32_t n_jobs, void* stream) {
  UNET_REQUIRE(dtype == UNET_BF16 || dtype == UNET_F32, "pack_multi: bad dtype %d", dtype);
  UNET_REQUIRE(jobs && n_jobs >= 1 && n_jobs <= UNET_PACK_MAX_JOBS, "pack_multi: 1..%d jobs per call, got %d",
               UNET_PACK_MAX_JOBS, n_jobs);
  PackM m;
  m.n = n_jobs;
  long long most = 0;
  for (int i = 0; i < UNET_PACK_MAX_JOBS; ++i) {
    const UnetPackJob& j = jobs[i < n_jobs ? i : 0];     // unused slots repeat job 0 (never run: n guards them)
    if (fill_pack(&m.k[i], dtype, j.w, j.tap_stride, j.row_stride, j.col_stride, j.n_taps,
                  j.has_tap_src ? j.tap_src : nullptr, j.row_lo, j.rows, j.col_lo, j.c1, j.c2, j.packed)) return -1;
    if (i < n_jobs && pack_chunks(m.k[i]) > most) most = pack_chunks(m.k[i]);
  }
  const int blocks = (int)((most + 255) / 256 > 8192 ? 8192 : (most + 255) / 256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == UNET_F32) hipLaunchKernelGGL(conv_pack_multi_kernel<true>, dim3(blocks), dim3(256), 0, st, m);
  else hipLaunchKernelGGL(conv_pack_multi_kernel<false>, dim3(blocks), dim3(256), 0, st, m);
  UNET_LAUNCH_CHECK("conv_pack_multi");
  return 0;
}

namespace { thread_local int g_last_variant = 0; }

// which kernel the calling thread's most recent conv launch used (profiling aid; see include/unet_hip.h)
extern "C" int unet_conv1d_last_variant(void) { return g_last_variant; }

namespace {
int conv_fwd_impl(const UnetConvDesc* d, const void* x1, const void* x2, const void* w_packed, const float* bias,
                  const void* res, void* y, float* s1_part, float* s2_part, int32_t* parts_out, void* stream,
                  const float* gate = nullptr, void* y2 = nullptr);
}

extern "C" int unet_conv1d_fwd(const UnetConvDesc* d, const void* x1, const void* x2, const void* w_packed,
                               const float* bias, const void* res, void* y, void* stream) {
  return conv_fwd_impl(d, x1, x2, w_packed, bias, res, y, nullptr, nullptr, nullptr, stream);
}

extern "C" int unet_conv1d_fwd_gated(const UnetConvDesc* d, const void* x1, const void* w_packed, const float* bias,
                                     const void* res, const float* gate, void* y_o, void* y_out, void* stream) {
  UNET_REQUIRE(gate && res && y_out, "conv_fwd_gated: null pointer");
  return conv_fwd_impl(d, x1, nullptr, w_packed, bias, res, y_o, nullptr, nullptr, nullptr, stream, gate, y_out);
}

extern "C" int32_t unet_conv1d_stats_max_parts(const UnetConvDesc* d) {
  UNET_REQUIRE(d, "conv_stats_max_parts: null pointer");
  return d->up == 1 && d->dtype == UNET_BF16 ? cdiv(d->l_out, 64) : 0;
}

extern "C" int unet_conv1d_fwd_stats(const UnetConvDesc* d, const void* x1, const void* x2, const void* w_packed,
                                     const float* bias, const void* res, void* y, float* s1_part, float* s2_part,
                                     int32_t* parts_out, void* stream) {
  UNET_REQUIRE(s1_part && s2_part && parts_out, "conv_fwd_stats: null pointer");
  return conv_fwd_impl(d, x1, x2, w_packed, bias, res, y, s1_part, s2_part, parts_out, stream);
}

namespace {
// gate != nullptr: the gated form (unet_conv1d_fwd_gated); returns 1 without launching when the streaming kernel does
// not take the shape
int conv_fwd_impl(const UnetConvDesc* d, const void* x1, const void* x2, const void* w_packed, const float* bias,
                  const void* res, void* y, float* s1_part, float* s2_part, int32_t* parts_out, void* stream,
                  const float* gate, void* y2) {
  UNET_REQUIRE(d && x1 && w_packed && (gate ? y2 != nullptr : y != nullptr), "conv_fwd: null pointer");
  UNET_REQUIRE(!gate || (res && !x2 && d->c2 == 0), "conv_fwd_gated: needs the residual and a single source");
  if (parts_out) *parts_out = 0;
  Phases ph;
  if (analyse(d, &ph)) return -1;
  UNET_REQUIRE(d->c2 == 0 || x2, "conv_fwd: c2>0 but x2 is null");
  UNET_REQUIRE(aligned16(w_packed), "conv_fwd: packed weights must be 16-byte aligned");
  const int es = esize(d->dtype);
  ConvK k;
  k.x1 = static_cast<const char*>(x1); k.x2 = static_cast<const char*>(x2);
  k.wp = static_cast<const char*>(w_packed); k.bias = bias; k.res = static_cast<const char*>(res);
  k.y = static_cast<char*>(y);
  k.s1 = k.s2 = nullptr;
  k.gate = gate; k.y2 = static_cast<char*>(y2);
  const bool want_stats = s1_part && d->up == 1;       // only the tiled bf16 kernel below produces them
  k.x1_ns = d->x1_ns; k.x2_ns = d->x2_ns; k.y_ns = d->y_ns; k.res_ns = d->res_ns;
  k.x1_ld = d->x1_ld; k.x2_ld = d->x2_ld; k.y_ld = d->y_ld; k.res_ld = d->res_ld;
  k.c1 = d->c1; k.c2 = d->c2;
  k.kc1 = kchunks(d->dtype, d->c1); k.kctot = k.kc1 + (d->c2 > 0 ? kchunks(d->dtype, d->c2) : 0);
  k.l_in = d->l_in; k.l_out = d->l_out; k.c_out = d->c_out; k.n = d->n;
  k.stride = d->stride; k.up = d->up; k.act = d->act;
  k.mtiles = cdiv(d->c_out, BM);
  k.group = k.mtiles >= 4 ? 16 : 0;
  for (int r = 0; r < 3; ++r) k.pbeg[r] = ph.pbeg[r];
  k.x1_vec = vec_ok(x1, d->x1_ld, d->x1_ns, es);
  k.x2_vec = x2 ? vec_ok(x2, d->x2_ld, d->x2_ns, es) : 0;
  const int epc = d->dtype == UNET_F32 ? 4 : 8;
  k.x1_fast = k.x1_vec && d->c1 % epc == 0;
  k.x2_fast = k.x2_vec && d->c2 % epc == 0;
  k.y_vec = vec8_ok(y ? y : y2, d->y_ld, d->y_ns, es);
  k.res_vec = res ? vec8_ok(res, d->res_ld, d->res_ns, es) : 0;
  for (int r = 0; r < 2; ++r) k.mcount[r] = ph.mcount[r];
  // tile width: 128 positions unless the halo tile would not leave room for two blocks per CU
  TileGeom tg;
  int bn = 128;
  tile_geom(d, ph, 128, &tg);
  {
    const char* env = getenv("UNET_CONV_BN");
    if (env && atoi(env) == 64) bn = 64;
    else if (!env && 2 * WTILE_BYTES + (size_t)((tg.rmax + 7) / 8) * 1024 > 80 * 1024) bn = 64;
  }
  if (bn != 128) tile_geom(d, ph, bn, &tg);
  for (int r = 0; r < 2; ++r) {
    k.ptiles[r] = (r < d->up) ? cdiv(ph.mcount[r], bn) : 0;
    for (int c = 0; c < 2; ++c) { k.seg_rows[r][c] = tg.seg_rows[r][c]; k.seg_q0[r][c] = tg.seg_q0[r][c]; }
  }
  for (int j = 0; j < UNET_MAX_TAPS; ++j) k.tap_row[j] = tg.tap_row[j];
  hipStream_t st = static_cast<hipStream_t>(stream);
  {
    // v2 (lean instruction stream, buffer LDS-DMA): bf16, aligned channel runs, halo tile within 12 pieces per wave,
    // window and packed-weight byte offsets that fit the 32-bit buffer offsets
    const char* env = getenv("UNET_CONV_V1");
    const long long lim = 1ll << 30;
    const bool v2 = !(env && atoi(env) == 1) && d->dtype == UNET_BF16 && k.x1_fast && (d->c2 == 0 || k.x2_fast) &&
                    (tg.rmax + 7) / 8 <= 4 * F2_MAXP && (long long)d->l_in * d->x1_ld * 2 < lim &&
                    (d->c2 == 0 || (long long)d->l_in * d->x2_ld * 2 < lim) &&
                    (long long)d->n_taps * k.mtiles * k.kctot * WTILE_BYTES < lim &&
                    2 * WTILE_BYTES + (size_t)((tg.rmax + 7) / 8) * 1024 <= 160 * 1024;
    // single-tap convs with enough K-chunks: the persistent streaming kernel (deep LDS ring)
    const char* envs = getenv("UNET_CONV_NOSTREAM");
    const char* envr = getenv("UNET_STREAM_NORES");     // diagnostic: convs with a residual keep to the tap-loop kernel
    const bool res_ok = res && vec_ok(res, d->res_ld, d->res_ns, 2) && (long long)d->l_out * d->res_ld * 2 < lim;
    const bool stream = v2 && !(envs && atoi(envs) == 1) && d->n_taps == 1 && d->up == 1 && d->stride == 1 && d->c2 == 0 &&
                        k.kctot >= 3 && d->c_out <= ST_MAXC && d->c_out % 8 == 0 && d->act == UNET_ACT_NONE &&
                        (!y || vec_ok(y, d->y_ld, d->y_ns, 2)) && bn == 128 && (long long)d->l_out * d->y_ld * 2 < lim &&
                        (gate ? res_ok && vec_ok(y2, d->y_ld, d->y_ns, 2) && aligned16(gate)
                              : (!res || (res_ok && !(envr && atoi(envr) == 1))));
    if (stream) { g_last_variant = UNET_VARIANT_FWD_STREAM; return launch_stream(k, st); }
    if (gate) return 1;
    if (v2 && bn == 128) {
      // 192 positions per tile (2/3 of the weight bytes and LDS-DMA instructions per FLOP) for many-tap convs when
      // the halo tile still leaves two workgroups per CU and the last, partial tile wastes under 4 % of the work
      const char* envb = getenv("UNET_CONV_BN");
      // 256 positions per tile for convs with few taps per phase (k = 3, the transposed conv's two taps per phase):
      // their blocks live for a handful of (K-chunk, tap) steps, the weight tiles are most of what a step stages and
      // every step pays a barrier; the wide tile halves the weight bytes and the barriers per FLOP (128 accumulator
      // registers per lane, the halo tile single-buffered so that two workgroups still fit a CU)
      // (unit stride only: the stride-2 data gradient of the transposed conv measured 3 % slower, scripts/bench_conv.py)
      bool want256 = d->n_taps <= 4 * d->up && d->stride == 1 && d->c_out >= 128 && !(envb && atoi(envb) != 256) &&
                     !(k.c2 == 0 && k.c1 <= 32) &&
                     !(d->n_taps == 1 && k.kctot <= 4);   // single tap over <= 256 channels (FiLM's q_proj data gradient
                                                          // with its residual at C = 256): 110 us against 100 on 128
      for (int r = 0; r < d->up && want256; ++r)
        want256 = ph.mcount[r] > 0 && (long long)cdiv(ph.mcount[r], 256) * 256 * 100 <= (long long)ph.mcount[r] * 104;
      if (want256) {
        TileGeom tg2;
        tile_geom(d, ph, 256, &tg2);
        const int pieces = (tg2.rmax + 7) / 8;
        if (pieces <= 4 * F2_MAXP && 2 * WTILE_BYTES + (size_t)pieces * 1024 <= 80 * 1024) {
          for (int r = 0; r < 2; ++r) {
            k.ptiles[r] = (r < d->up) ? cdiv(ph.mcount[r], 256) : 0;
            for (int c = 0; c < 2; ++c) { k.seg_rows[r][c] = tg2.seg_rows[r][c]; k.seg_q0[r][c] = tg2.seg_q0[r][c]; }
          }
          for (int j = 0; j < UNET_MAX_TAPS; ++j) k.tap_row[j] = tg2.tap_row[j];
          g_last_variant = UNET_VARIANT_FWD_TILED;
          if (want_stats) { k.s1 = s1_part; k.s2 = s2_part; *parts_out = k.ptiles[0]; }
          return launch_fwd2<256>(k, tg2.rmax, st);
        }
      }
      bool want = d->n_taps >= 8 && !(envb && atoi(envb) == 128);
      for (int r = 0; r < d->up && want; ++r)
        want = ph.mcount[r] > 0 && (long long)cdiv(ph.mcount[r], 192) * 192 * 100 <= (long long)ph.mcount[r] * 104;
      if (want) {
        TileGeom tg2;
        tile_geom(d, ph, 192, &tg2);
        const int pieces = (tg2.rmax + 7) / 8;
        if (pieces <= 4 * F2_MAXP && 2 * WTILE_BYTES + (size_t)pieces * 1024 <= 80 * 1024) {
          for (int r = 0; r < 2; ++r) {
            k.ptiles[r] = (r < d->up) ? cdiv(ph.mcount[r], 192) : 0;
            for (int c = 0; c < 2; ++c) { k.seg_rows[r][c] = tg2.seg_rows[r][c]; k.seg_q0[r][c] = tg2.seg_q0[r][c]; }
          }
          for (int j = 0; j < UNET_MAX_TAPS; ++j) k.tap_row[j] = tg2.tap_row[j];
          g_last_variant = UNET_VARIANT_FWD_TILED;
          if (want_stats) { k.s1 = s1_part; k.s2 = s2_part; *parts_out = k.ptiles[0]; }
          return launch_fwd2<192>(k, tg2.rmax, st);
        }
      }
    }
    if (v2) {
      g_last_variant = UNET_VARIANT_FWD_TILED;
      if (want_stats) { k.s1 = s1_part; k.s2 = s2_part; *parts_out = k.ptiles[0]; }
      return bn == 128 ? launch_fwd2<128>(k, tg.rmax, st) : launch_fwd2<64>(k, tg.rmax, st);
    }
  }
  g_last_variant = UNET_VARIANT_FWD_GENERIC;
  if (d->dtype == UNET_F32) return bn == 128 ? launch_fwd<true, 128>(k, tg.rmax, st) : launch_fwd<true, 64>(k, tg.rmax, st);
  return bn == 128 ? launch_fwd<false, 128>(k, tg.rmax, st) : launch_fwd<false, 64>(k, tg.rmax, st);
}
}  // namespace

namespace {
int wgrad_splits(const UnetConvDesc* d, const Phases& ph) {
  const int bkp = d->dtype == UNET_F32 ? 32 : 64;
  const int kt = cdiv(d->c1, 128) + (d->c2 > 0 ? cdiv(d->c2, 128) : 0);
  const long long tiles = (long long)d->n_taps * cdiv(d->c_out, BM) * kt;
  int mc = ph.mcount[0] > ph.mcount[1] ? ph.mcount[0] : ph.mcount[1];
  long long items = (long long)d->n * cdiv(mc, bkp);
  long long s = (2048 + tiles - 1) / tiles;
  if (s > 64) s = 64;
  if (s > items) s = items;
  if (s < 1) s = 1;
  return (int)s;
}
}  // namespace

namespace {
// groups of taps that can share one input tile: same phase, same (offset mod stride), at most 32 rows from the
// group's first tap.  Two taps per group; four for the small-Cin variant (conv_wgrad2s_kernel).
struct WgGroups {
  int n;
  int tapa[UNET_MAX_TAPS], tapb[UNET_MAX_TAPS], tapc[UNET_MAX_TAPS], tapd[UNET_MAX_TAPS];
  int off[UNET_MAX_TAPS], delta[UNET_MAX_TAPS], deltac[UNET_MAX_TAPS], deltad[UNET_MAX_TAPS], phase[UNET_MAX_TAPS];
};

bool wgrad2_small(const UnetConvDesc* d) {
  const char* env = getenv("UNET_WGRAD_NOSMALL");
  return d->c2 == 0 && d->c1 <= 32 && !(env && atoi(env) == 1);
}

void wgrad_groups(const UnetConvDesc* d, WgGroups* g) {
  const int per = wgrad2_small(d) ? 4 : 2;
  int order[UNET_MAX_TAPS];
  for (int j = 0; j < d->n_taps; ++j) order[j] = j;
  auto cls = [&](int j) { return ((d->tap_off[j] % d->stride) + d->stride) % d->stride; };
  auto less = [&](int a, int b) {
    if (d->tap_phase[a] != d->tap_phase[b]) return d->tap_phase[a] < d->tap_phase[b];
    if (cls(a) != cls(b)) return cls(a) < cls(b);
    return d->tap_off[a] < d->tap_off[b];
  };
  for (int i = 1; i < d->n_taps; ++i)
    for (int k = i; k > 0 && less(order[k], order[k - 1]); --k) { int t = order[k]; order[k] = order[k - 1]; order[k - 1] = t; }
  g->n = 0;
  for (int i = 0; i < d->n_taps;) {
    const int a = order[i];
    int member[4] = {a, -1, -1, -1}, dl[4] = {0, 0, 0, 0}, cnt = 1;
    while (cnt < per && i + cnt < d->n_taps) {
      const int c = order[i + cnt];
      if (d->tap_phase[c] != d->tap_phase[a] || cls(c) != cls(a) || (d->tap_off[c] - d->tap_off[a]) / d->stride > 32) break;
      member[cnt] = c;
      dl[cnt] = (d->tap_off[c] - d->tap_off[a]) / d->stride;
      ++cnt;
    }
    const int k = g->n++;
    g->tapa[k] = member[0]; g->tapb[k] = member[1]; g->tapc[k] = member[2]; g->tapd[k] = member[3];
    g->off[k] = d->tap_off[a]; g->phase[k] = d->tap_phase[a];
    g->delta[k] = dl[1]; g->deltac[k] = dl[2]; g->deltad[k] = dl[3];
    i += cnt;
  }
}

bool wgrad2_ok(const UnetConvDesc* d, const void* x1, const void* x2, const void* dy) {
  if (d->dtype != UNET_BF16) return false;
  const char* env = getenv("UNET_WGRAD_V1");
  if (env && atoi(env) == 1) return false;
  if (d->c1 % 8 || d->c2 % 8 || d->c_out % 8) return false;
  if (!vec_ok(x1, d->x1_ld, d->x1_ns, 2) || !vec_ok(dy, d->y_ld, d->y_ns, 2)) return false;
  if (d->c2 > 0 && !vec_ok(x2, d->x2_ld, d->x2_ns, 2)) return false;
  // buffer descriptors cover one window: offsets must stay below 2^31 so that "negative" rows wrap out of range
  const long long lim = 1ll << 30;
  if ((long long)d->l_in * d->x1_ld * 2 >= lim || (long long)d->l_out * d->y_ld * 2 >= lim) return false;
  if (d->c2 > 0 && (long long)d->l_in * d->x2_ld * 2 >= lim) return false;
  return true;
}

// the 256 x 256-tile kernel: one tap at unit stride, one source, whole 256-channel tiles
bool wgrad3_ok(const UnetConvDesc* d) {
  const char* env = getenv("UNET_WGRAD3");
  if (env && atoi(env) == 0) return false;
  return d->dtype == UNET_BF16 && d->n_taps == 1 && d->stride == 1 && d->up == 1 && d->c2 == 0 && d->c1 % 256 == 0 &&
         d->c_out % 256 == 0;
}
int wgrad3_splits(const UnetConvDesc* d) {
  // one workgroup per CU: enough splits to give every CU one, never more items than there are
  const long long tiles = (long long)(d->c1 / 256) * (d->c_out / 256);
  const long long items = (long long)d->n * cdiv(d->l_out, WG3_BKP);
  long long s = (256 + tiles - 1) / tiles;       // up to 256 splits for the single 256 x 256 tile of a C = 256 conv:
  if (s > 256) s = 256;                          // 101 -> 81 us there (64 splits: 146 us, a quarter of the CUs)
  if (s > items) s = items;
  return (int)(s < 1 ? 1 : s);
}

int wgrad2_splits(const UnetConvDesc* d, const Phases& ph, int groups) {
  // Split-K factor from a small cost model: rounds of 512 resident blocks (2 per CU) x steps per block, plus the
  // HBM round trip of the fp32 slabs (written once, read once by the reducer).  Few, long blocks keep the slab
  // traffic below the operand traffic; the split count is chosen so that the grid fills whole rounds.
  const int kt = cdiv(d->c1, 128) + (d->c2 > 0 ? cdiv(d->c2, 128) : 0);
  const long long tiles = (long long)groups * cdiv(d->c_out, BM) * kt;
  int mc = ph.mcount[0] > ph.mcount[1] ? ph.mcount[0] : ph.mcount[1];
  const long long items = (long long)d->n * cdiv(mc, WG2_BKP);
  const double t_step = 0.45 * (double)d->n_taps / groups;                       // us, ~32 MFMAs per wave and tap
  const double slab_mb = (double)d->n_taps * d->c_out * (d->c1 + d->c2) * 4.0 / 1e6;
  int best = 1;
  double best_cost = 1e30;
  for (int s = 1; s <= 64 && s <= items; ++s) {
    const double rounds = (double)((tiles * s + 511) / 512);
    const double steps = (double)((items + s - 1) / s);
    const double cost = rounds * (steps * t_step + 3.0) + 2.0 * s * slab_mb / 4.0;   // 4 MB/us ~ 4 TB/s
    if (cost < best_cost) { best_cost = cost; best = s; }
  }
  return best;
}
}  // namespace

namespace {
// wide: count the 256 x 256-tile kernel's slabs too (it is skipped at call time when the workspace cannot hold them)
size_t wgrad_ws_bytes(const UnetConvDesc* d, bool wide) {
  Phases ph;
  if (!d || analyse(d, &ph) || d->n_taps <= 0) return 0;
  // the kernel (v1 / v2) is chosen at call time from pointer alignment: size for the larger slab count
  int s = wgrad_splits(d, ph);
  if (d->dtype == UNET_BF16) {
    WgGroups g;
    wgrad_groups(d, &g);
    const int s2 = wgrad2_splits(d, ph, g.n);
    if (s2 > s) s = s2;
    if (wide && wgrad3_ok(d) && wgrad3_splits(d) > s) s = wgrad3_splits(d);
  }
  return (size_t)s * d->n_taps * d->c_out * (d->c1 + d->c2) * sizeof(float);
}
}  // namespace

extern "C" size_t unet_conv1d_wgrad_workspace_bytes(const UnetConvDesc* d) { return wgrad_ws_bytes(d, true); }

namespace {
int wgrad_impl(const UnetConvDesc* d, const void* x1, const void* x2, const void* dy, float* dw, void* workspace,
               size_t ws_bytes, void* stream);
}

extern "C" int unet_conv1d_wgrad(const UnetConvDesc* d, const void* x1, const void* x2, const void* dy,
                                 float* dw, void* workspace, size_t ws_bytes, void* stream) {
  return wgrad_impl(d, x1, x2, dy, dw, workspace, ws_bytes, stream);
}

// ---------------------------------------------------------------------------------------------
// Allocation-free variant of unet_conv1d_wgrad (see include/unet_hip.h): the split-K slabs live in a
// file-scope device array instead of a caller-provided workspace.  Every GPU the library runs on holds its own
// copy of the array at its own address, resolved once per device with hipGetSymbolAddress; host code never
// passes the array's name to a kernel, only that resolved pointer.
// ---------------------------------------------------------------------------------------------
static __device__ __align__(256) unsigned char scratch[UNET_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int MAX_DEVICES = 64;
void* g_scratch_ptr[MAX_DEVICES];          // per device: address of that device's copy of `scratch`
std::mutex g_scratch_mutex;

int scratch_for_current_device(void** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  UNET_REQUIRE(e == hipSuccess, "conv_wgrad_noalloc: hipGetDevice failed: %s", hipGetErrorString(e));
  UNET_REQUIRE(dev >= 0 && dev < MAX_DEVICES, "conv_wgrad_noalloc: device index %d out of range", dev);
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    void* ptr = nullptr;
    e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));   // the current device's copy
    UNET_REQUIRE(e == hipSuccess && ptr, "conv_wgrad_noalloc: hipGetSymbolAddress failed: %s", hipGetErrorString(e));
    g_scratch_ptr[dev] = ptr;
  }
  *out = g_scratch_ptr[dev];
  return 0;
}
}  // namespace

extern "C" size_t unet_conv1d_wgrad_noalloc_max_bytes(void) { return (size_t)UNET_NOALLOC_SCRATCH_BYTES; }

extern "C" int unet_conv1d_wgrad_noalloc(const UnetConvDesc* d, const void* x1, const void* x2, const void* dy,
                                         float* dw, void* stream) {
  UNET_REQUIRE(d && x1 && dy && dw, "conv_wgrad_noalloc: null pointer");
  size_t need = wgrad_ws_bytes(d, true);
  if (need > (size_t)UNET_NOALLOC_SCRATCH_BYTES) need = wgrad_ws_bytes(d, false);   // without the wide kernel's slabs
  UNET_REQUIRE(need > 0, "conv_wgrad_noalloc: bad descriptor");
  UNET_REQUIRE(need <= (size_t)UNET_NOALLOC_SCRATCH_BYTES,
               "conv_wgrad_noalloc: the split-K slabs of this conv need %zu bytes, the built-in scratch holds %zu; "
               "use unet_conv1d_wgrad with a caller-provided workspace", need, (size_t)UNET_NOALLOC_SCRATCH_BYTES);
  void* ptr = nullptr;
  if (scratch_for_current_device(&ptr)) return -1;
  // every slab element that slab_reduce_kernel reads is written by the weight-gradient kernel of the same call
  // (blocks with an empty split store zeros), so nothing in the scratch has to be cleared between calls
  return wgrad_impl(d, x1, x2, dy, dw, ptr, (size_t)UNET_NOALLOC_SCRATCH_BYTES, stream);
}

namespace {
int wgrad_impl(const UnetConvDesc* d, const void* x1, const void* x2, const void* dy, float* dw, void* workspace,
               size_t ws_bytes, void* stream) {
  UNET_REQUIRE(d && x1 && dy && dw, "conv_wgrad: null pointer");
  Phases ph;
  if (analyse(d, &ph)) return -1;
  UNET_REQUIRE(d->n_taps > 0, "conv_wgrad: no taps");
  UNET_REQUIRE(d->c2 == 0 || x2, "conv_wgrad: c2>0 but x2 is null");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long count = (long long)d->n_taps * d->c_out * (d->c1 + d->c2);
  g_last_variant = UNET_VARIANT_WGRAD_GENERIC;
  // (a workspace too small for the wide kernel's slabs -- the allocation-free entry point's fixed scratch -- takes the
  // paired kernel below, which needs fewer)
  if (wgrad2_ok(d, x1, x2, dy) && wgrad3_ok(d) && workspace &&
      ws_bytes >= (size_t)wgrad3_splits(d) * count * sizeof(float)) {
    g_last_variant = UNET_VARIANT_WGRAD_WIDE;
    const int splits = wgrad3_splits(d);
    Wg3K k;
    k.x = static_cast<const char*>(x1); k.dy = static_cast<const char*>(dy); k.slab = static_cast<float*>(workspace);
    k.x_ns = d->x1_ns; k.dy_ns = d->y_ns; k.x_ld = d->x1_ld; k.dy_ld = d->y_ld;
    k.c_in = d->c1; k.c_out = d->c_out; k.l_in = d->l_in; k.mcount = ph.mcount[0]; k.n = d->n;
    k.off0 = d->tap_off[0];
    k.mt2 = d->c_out / 256; k.kt2 = d->c1 / 256; k.splits = splits;
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad3_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WG3_STAGE);
      UNET_REQUIRE(e == hipSuccess, "conv_wgrad: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      attr_set = true;
    }
    const long long blocks = (long long)k.mt2 * k.kt2 * splits;
    hipLaunchKernelGGL(conv_wgrad3_kernel, dim3((unsigned)blocks), dim3(WG3_THREADS), 2 * WG3_STAGE, st, k);
    UNET_LAUNCH_CHECK("conv_wgrad3");
    const long long cols = (count & 3) ? count : count >> 2;
    int rb = (int)((cols + 255) / 256 > 4096 ? 4096 : (cols + 255) / 256);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(rb), dim3(256), 0, st, k.slab, dw, count, splits);
    UNET_LAUNCH_CHECK("slab_reduce");
    return 0;
  }
  if (wgrad2_ok(d, x1, x2, dy)) {
    g_last_variant = UNET_VARIANT_WGRAD_PAIRED;
    WgGroups g;
    wgrad_groups(d, &g);
    const int splits = wgrad2_splits(d, ph, g.n);
    const size_t need = (size_t)splits * count * sizeof(float);
    UNET_REQUIRE(workspace && ws_bytes >= need, "conv_wgrad: workspace %zu bytes < %zu needed", ws_bytes, need);
    Wg2K k;
    k.x1 = static_cast<const char*>(x1); k.x2 = static_cast<const char*>(x2); k.dy = static_cast<const char*>(dy);
    k.slab = static_cast<float*>(workspace);
    k.x1_ns = d->x1_ns; k.x2_ns = d->x2_ns; k.dy_ns = d->y_ns;
    k.x1_ld = d->x1_ld; k.x2_ld = d->x2_ld; k.dy_ld = d->y_ld;
    k.c1 = d->c1; k.c2 = d->c2; k.kt1 = cdiv(d->c1, 128); k.kttot = k.kt1 + (d->c2 > 0 ? cdiv(d->c2, 128) : 0);
    k.l_in = d->l_in; k.c_out = d->c_out; k.n = d->n; k.stride = d->stride; k.up = d->up;
    k.mtiles = cdiv(d->c_out, BM); k.n_taps = d->n_taps; k.n_groups = g.n; k.splits = splits;
    k.mcount[0] = ph.mcount[0]; k.mcount[1] = ph.mcount[1];
    for (int j = 0; j < UNET_MAX_TAPS; ++j) {
      const bool v = j < g.n;
      k.g_tapa[j] = v ? g.tapa[j] : 0; k.g_tapb[j] = v ? g.tapb[j] : -1; k.g_off[j] = v ? g.off[j] : 0;
      k.g_delta[j] = v ? g.delta[j] : 0; k.g_phase[j] = v ? g.phase[j] : 0;
      k.g_tapc[j] = v ? g.tapc[j] : -1; k.g_tapd[j] = v ? g.tapd[j] : -1;
      k.g_deltac[j] = v ? g.deltac[j] : 0; k.g_deltad[j] = v ? g.deltad[j] : 0;
    }
    const bool small = wgrad2_small(d);
    void (*kern)(const Wg2K) = small ? conv_wgrad2s_kernel : conv_wgrad2_kernel;
    static bool attr_set[2] = {false, false};
    if (!attr_set[small]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WG2_STAGE);
      UNET_REQUIRE(e == hipSuccess, "conv_wgrad: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      attr_set[small] = true;
    }
    const long long blocks = (long long)k.mtiles * k.kttot * splits * g.n;
    UNET_REQUIRE(blocks < (1ll << 31), "conv_wgrad: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(THREADS), 2 * WG2_STAGE, st, k);
    UNET_LAUNCH_CHECK("conv_wgrad2");
    const long long cols = (count & 3) ? count : count >> 2;
    int rb = (int)((cols + 255) / 256 > 4096 ? 4096 : (cols + 255) / 256);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(rb), dim3(256), 0, st, k.slab, dw, count, splits);
    UNET_LAUNCH_CHECK("slab_reduce");
    return 0;
  }
  const int splits = wgrad_splits(d, ph);
  const size_t need = (size_t)splits * d->n_taps * d->c_out * (d->c1 + d->c2) * sizeof(float);
  UNET_REQUIRE(workspace && ws_bytes >= need, "conv_wgrad: workspace %zu bytes < %zu needed", ws_bytes, need);
  const int es = esize(d->dtype);
  WgK k;
  k.x1 = static_cast<const char*>(x1); k.x2 = static_cast<const char*>(x2); k.dy = static_cast<const char*>(dy);
  k.slab = static_cast<float*>(workspace);
  k.x1_ns = d->x1_ns; k.x2_ns = d->x2_ns; k.dy_ns = d->y_ns;
  k.x1_ld = d->x1_ld; k.x2_ld = d->x2_ld; k.dy_ld = d->y_ld;
  k.c1 = d->c1; k.c2 = d->c2; k.kt1 = cdiv(d->c1, 128); k.kttot = k.kt1 + (d->c2 > 0 ? cdiv(d->c2, 128) : 0);
  k.l_in = d->l_in; k.c_out = d->c_out; k.n = d->n; k.stride = d->stride; k.up = d->up;
  k.mtiles = cdiv(d->c_out, BM); k.n_taps = d->n_taps; k.splits = splits;
  k.mcount[0] = ph.mcount[0]; k.mcount[1] = ph.mcount[1];
  k.x1_vec = vec_ok(x1, d->x1_ld, d->x1_ns, es);
  k.x2_vec = x2 ? vec_ok(x2, d->x2_ld, d->x2_ns, es) : 0;
  k.dy_vec = vec_ok(dy, d->y_ld, d->y_ns, es);
  for (int j = 0; j < UNET_MAX_TAPS; ++j) {
    k.tap_off[j] = j < d->n_taps ? d->tap_off[j] : 0;
    k.tap_phase[j] = j < d->n_taps ? d->tap_phase[j] : 0;
  }
  const long long blocks = (long long)k.mtiles * k.kttot * splits * d->n_taps;
  UNET_REQUIRE(blocks < (1ll << 31), "conv_wgrad: grid too large");
  if (d->dtype == UNET_F32) hipLaunchKernelGGL(conv_wgrad_kernel<true>, dim3((unsigned)blocks), dim3(THREADS), 0, st, k);
  else hipLaunchKernelGGL(conv_wgrad_kernel<false>, dim3((unsigned)blocks), dim3(THREADS), 0, st, k);
  UNET_LAUNCH_CHECK("conv_wgrad");
  int rb = (int)((count + 255) / 256 > 4096 ? 4096 : (count + 255) / 256);
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(rb), dim3(256), 0, st, k.slab, dw, count, splits);
  UNET_LAUNCH_CHECK("slab_reduce");
  return 0;
}
}  // namespace
