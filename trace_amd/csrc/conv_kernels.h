// Implicit-GEMM Conv1d kernels on fp32 MFMA (v_mfma_f32_16x16x4_f32) for the TRACE temporal
// UNet, with the GroupNorm + Mish + (time/cond bias | residual) epilogue fused.
//
// GEMM view of one layer:   out[(traj,l), co] = bias[co] + sum_{tap,ci} in[traj, ci, l*IS + tap] * W[tap,ci,co]
//   rows   = (trajectory, output position) flattened; a workgroup owns TB whole trajectories
//            (TB*L_OUT = 208 rows = 13 MFMA row tiles) so GroupNorm statistics never leave it
//   cols   = 64 output channels per workgroup (4 waves x one 16-wide MFMA column tile)
//   K      = taps x input channels, streamed in chunks of CH (<=64) channels through LDS
//
// A operand: activations staged once per chunk into LDS as [position][channel] (zeroed halo /
//            zero-stuffing, position stride CH+4 floats), read with one conflict-free ds_read_b128
//            per (row tile, k-block) that feeds 4 MFMA passes; the reads of the next k-block are
//            issued during the last pass of the current one
// B operand: weights pre-packed on the host in MFMA fragment order, streamed L2 -> VGPR with one
//            coalesced dwordx4 per 16 k (no LDS, no barrier in the K loop), prefetched two
//            k-blocks ahead so that its wait never covers the (older) staging loads
// staging  : the next chunk's activations are fetched global -> VGPR (buffer_load b128) before
//            the K loop of the current chunk and written to LDS after it (issue early / write
//            late), so HBM / L2 latency hides under the MFMAs
// epilogue : accumulators -> LDS tile laid out like the global tensor; GroupNorm statistics from
//            registers (one (trajectory, channel) row per thread, butterfly over the group);
//            normalise + Mish + cond/time bias + residual(s) and 128-bit stores in one flat pass
#pragma once
#include "common.h"

namespace trace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
  const float* in;     // [rows][in_ctot][L_IN]
  int in_tstride;      // floats per trajectory (multiple of 4)
  int in_coff;         // first input channel (multiple of 16)
  int cin;             // input channels (multiple of chunk)
  int chunk;           // channels per LDS stage (selects the template instance)
  const f32x4* wp;     // packed weights [cout/16][nkb][64]
  int nkb;             // k-blocks (16 k each) = taps * cin / 16
  const float* bias;   // [cout] or null
  float* out;          // [rows][out_ctot][L_OUT]
  int out_tstride;
  int out_coff;
  int cout;
  int rows;
  // epilogue
  int gn;              // GroupNorm + Mish
  int cg;              // channels per group (power of two)
  const float* gamma;
  const float* beta;
  float* ysave;        // pre-norm conv output [rows][cout][L_OUT] (input-VJP) or null
  float* stats;        // [rows][cout/cg][2] (mean, rstd) or null
  const float* pc;     // per-cond-row additive table or null
  int pc_stride;
  int pc_off;
  int rows_per_cond;
  const float* pt;     // per-step additive vector [cout] (already offset) or null
  const float* res;    // residual [rows][res_ctot][L_OUT] or null
  int res_tstride;
  int res_coff;
  const float* res2;
  int res2_tstride;
  int res2_coff;
  double alg_macs_per_pos;   // algorithmic MACs per (row, output position): cout * cin_real * taps_alg
  // second K source: cin2 more input channels taken from in2 at the centre tap only (the 1x1
  // residual conv of a block folded into the k5 dgrad of the same block); weights appended
  const float* in2;
  int in2_tstride;
  int in2_coff;
  int cin2;            // multiple of chunk, 0 = none
  // GroupNorm + Mish BACKWARD epilogue: v = acc + res + res2 is the gradient w.r.t. the Mish output
  // of the layer that produced this conv's (forward) input; out <- d/dy of that layer's
  // pre-norm output y, out_raw <- v (needed again by the residual / skip paths)
  int gnb;
  int gb_cmax;         // only output channels < gb_cmax go through the norm (rest: out_raw only)
  const float* gb_y;   // saved pre-norm output [rows][gb_c][L_OUT]
  const float* gb_stats;
  const float* gb_gamma;
  const float* gb_beta;
  int gb_cg;
  int gb_c;
  float* out_raw;      // or null
  int raw_tstride;
  int raw_coff;
};

template <int L_OUT, int IN_STRIDE, int UP, int TAPS, int PADL, int TB, int CH>
struct ConvGeom {
  static constexpr int R = TB * L_OUT;
  static constexpr int NRT = R / 16;
  static constexpr int LP = (L_OUT - 1) * IN_STRIDE + TAPS;
  static constexpr int S = TB * LP;                        // staged positions (with halo)
  static constexpr int CHS = CH + 4;                       // LDS floats per position: CHS/4 odd
  static constexpr int L_IN = (L_OUT * IN_STRIDE) / UP;
  static constexpr int RS = R + 4;
  static constexpr int F4_PER_TRAJ = CH * L_IN / 4;        // float4 per trajectory per chunk
  static constexpr int F4_TOTAL = TB * F4_PER_TRAJ;
  static constexpr int NF4 = (F4_TOTAL + 255) / 256;       // staging registers (float4) per thread
  static constexpr int KBPC = CH / 16;
  static constexpr int NKK = TAPS * KBPC;
  static_assert(R % 16 == 0, "row block must be a whole number of MFMA tiles");
  static_assert((CH * L_IN) % 4 == 0 && CH % 16 == 0, "chunk must be float4 / k-block aligned");
  static_assert((CHS / 4) % 2 == 1, "position stride must be an odd number of 16-byte slots");
  static constexpr int NQ = R * 64 / 4 / 256;              // output float4 per thread
  static_assert(R * 64 % 1024 == 0 && TB % 4 == 0, "output tile must split evenly over 256 threads");
  static constexpr size_t EPI_FLOATS = (size_t)R * 64 + (size_t)TB * 64 * 4;   // tile + tables
  static constexpr size_t LDS_FLOATS = ((size_t)CHS * S > EPI_FLOATS) ? (size_t)CHS * S : EPI_FLOATS;
};

template <int L_OUT, int IN_STRIDE, int UP, int TAPS, int PADL, int TB, int CH>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(ConvArgs a) {
  using G = ConvGeom<L_OUT, IN_STRIDE, UP, TAPS, PADL, TB, CH>;
  constexpr int NRT = G::NRT, LP = G::LP, S = G::S, L_IN = G::L_IN, R = G::R, CHS = G::CHS;
  constexpr int NF4 = G::NF4, KBPC = G::KBPC, NKK = G::NKK;
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int traj0 = blockIdx.x * TB;
  const int co0 = blockIdx.y * 64;

  // ---- staging: thread handles float4 #(tid + 256 j) of the chunk -------------------------
  // Buffer loads with a 32-bit byte offset: the descriptor's range check returns 0 for
  // trajectories past `rows` (tail workgroup), so there is no per-load predicate.
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in), 0, (unsigned)((size_t)a.rows * a.in_tstride * sizeof(float)),
      0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in2 ? a.in2 : a.in), 0,
      a.in2 ? (unsigned)((size_t)a.rows * a.in2_tstride * sizeof(float)) : 0u, 0x00020000);
  const int nch1 = a.cin / CH;
  const int nch = nch1 + a.cin2 / CH;
  unsigned spk[NF4];            // (trajectory in tile) << 20 | float offset inside the chunk
#pragma unroll
  for (int j = 0; j < NF4; ++j) {
    const int q = tid + 256 * j;
    const int tr = q / G::F4_PER_TRAJ;
    const int rem = q - tr * G::F4_PER_TRAJ;
    spk[j] = q < G::F4_TOTAL ? ((unsigned)tr << 20) | (unsigned)(rem * 4) : 0xffffffffu;
  }
  f32x4 stg[NF4];
  auto stage_load = [&](int ch) {
    const bool second = ch >= nch1;
    const int tstride = second ? a.in2_tstride : a.in_tstride;
    const int cbase = ((second ? a.in2_coff + (ch - nch1) * CH : a.in_coff + ch * CH) * L_IN) +
                      traj0 * tstride;
#pragma unroll
    for (int j = 0; j < NF4; ++j) {
      const unsigned off = spk[j] == 0xffffffffu
                               ? 0xffffffffu     // out of range -> zeros
                               : (unsigned)(((int)(spk[j] >> 20) * tstride + (int)(spk[j] & 0xfffffu) + cbase) * 4);
      auto v = second ? __builtin_amdgcn_raw_buffer_load_b128(rsrc2, off, 0, 0)
                      : __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
      stg[j] = __builtin_bit_cast(f32x4, v);
    }
  };
  auto stage_write = [&]() {
#pragma unroll
    for (int j = 0; j < NF4; ++j) {
      const int q = tid + 256 * j;
      if (q < G::F4_TOTAL) {
        const int tr = q / G::F4_PER_TRAJ;
        const int e0 = (q - tr * G::F4_PER_TRAJ) * 4;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = e0 + u;
          const int ci = e / L_IN;
          const int i = e - ci * L_IN;
          lds[(tr * LP + i * UP + PADL) * CHS + ci] = stg[j][u];
        }
      }
    }
  };

  stage_load(0);
  // halo / zero-stuffing positions are written once; data positions are the same for every chunk
#pragma unroll 2
  for (int i = tid; i < CHS * S / 4; i += 256)
    reinterpret_cast<f32x4*>(lds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // LDS image is [position][channel]; lane (row = lane&15, kq = lane>>4) reads the 4 channels
  // 16m + 4kq .. +3 of its output row's input position with one ds_read_b128, and feeds element
  // u to MFMA pass u (the host packs the weights in the matching k order)
  int rowoff[NRT];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) {
    int r = rt * 16 + (lane & 15);
    int tr = r / L_OUT;
    int l = r - tr * L_OUT;
    rowoff[rt] = (tr * LP + l * IN_STRIDE) * CHS + (lane >> 4) * 4;
  }
  f32x4 acc[NRT];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const f32x4* wp = a.wp + ((size_t)(blockIdx.y * 4 + wave) * a.nkb) * 64 + lane;
  f32x4 bnext = *wp;
  f32x4 bnext2 = a.nkb > 1 ? wp[64] : bnext;
  int kdone = 0;

  for (int ch = 0; ch < nch; ++ch) {
    __syncthreads();            // previous chunk fully consumed (and zero fill done)
    stage_write();
    if (ch + 1 < nch) stage_load(ch + 1);       // in flight during this chunk's K loop
    __syncthreads();

    // K loop over the staged chunk.  One ds_read_b128 per (row tile, k-block) supplies the A
    // values of 4 MFMA passes; the read for the NEXT k-block is issued right after the last pass
    // that consumes the register, i.e. 13 MFMAs (>400 cycles) ahead of its first use.
    // Chunks of the second source contribute at the centre tap only.
    const bool second = ch >= nch1;
    int kk = second ? PADL * KBPC : 0;
    const int kk_end = second ? (PADL + 1) * KBPC : NKK;
    int tap = second ? PADL : 0, kb = 0;
    f32x4 av[NRT];
    const float* lp = lds + tap * CHS;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) av[rt] = *reinterpret_cast<const f32x4*>(lp + rowoff[rt]);
    for (; kk < kk_end; ++kk) {
      const f32x4 b = bnext;
      bnext = bnext2;
      ++kdone;
      wp += 64;
      if (kdone + 1 < a.nkb) bnext2 = wp[64];
      int kb2 = kb + 1, tap2 = tap;
      if (kb2 == KBPC) { kb2 = 0; ++tap2; }
      const float* lpn = lds + tap2 * CHS + kb2 * 16;
      const bool more = kk + 1 < kk_end;
#pragma unroll
      for (int u = 0; u < 3; ++u) {
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
          acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][u], b[u], acc[rt], 0, 0, 0);
      }
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) {
        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][3], b[3], acc[rt], 0, 0, 0);
        if (more) av[rt] = *reinterpret_cast<const f32x4*>(lpn + rowoff[rt]);
      }
      lp = lpn;
      kb = kb2;
      tap = tap2;
    }
  }

  // ---- epilogue --------------------------------------------------------------------------
  // The 208 x 64 accumulator tile goes through LDS once, laid out exactly like the global
  // tensors ([trajectory][channel][position], TILE floats per trajectory), so that
  //   E1  per-(trajectory, channel) rows are contiguous  -> GroupNorm statistics in registers
  //   E2  the whole tile is one flat stream              -> 128-bit LDS reads and global accesses
  // Residual operands are fetched into registers before anything else: their latency hides
  // under E0/E1 instead of serialising the store loop.
  constexpr int TILE = 64 * L_OUT;
  constexpr int NQ = G::NQ;
  const bool has_tab = a.gn || a.pc;
  auto flat_off = [&](int k, int tstride, int coff, int& rem_out) -> unsigned {
    const int f = (tid + 256 * k) * 4;
    const int tr = f / TILE;
    const int rem = f - tr * TILE;
    rem_out = rem;
    const int tg = traj0 + tr;
    return tg < a.rows ? (unsigned)((tg * tstride + (coff + co0) * L_OUT + rem) * 4) : 0xffffffffu;
  };
  f32x4 rv[NQ];
  if (a.res) {
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.res), 0, (unsigned)((size_t)a.rows * a.res_tstride * sizeof(float)), 0x00020000);
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      int rem;
      rv[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                            rr, flat_off(k, a.res_tstride, a.res_coff, rem), 0, 0));
    }
  } else {
#pragma unroll
    for (int k = 0; k < NQ; ++k) rv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (a.res2) {
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.res2), 0, (unsigned)((size_t)a.rows * a.res2_tstride * sizeof(float)), 0x00020000);
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      int rem;
      f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                              rr, flat_off(k, a.res2_tstride, a.res2_coff, rem), 0, 0));
      rv[k] += t;
    }
  }

  // E0: accumulators (+ conv bias) -> LDS tile
  __syncthreads();
  {
    const int col = lane & 15, q = lane >> 4;
    const int co_l = wave * 16 + col;
    const float bv = a.bias ? a.bias[co0 + co_l] : 0.f;
    float* tcol = lds + co_l * L_OUT;
    int tr = 0, l = q * 4;                      // row r = rt*16 + q*4 + reg = tr*L_OUT + l
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      if constexpr (L_OUT % 4 == 0) {           // a quad of rows never straddles two trajectories
        f32x4 v = acc[rt];
        v[0] += bv; v[1] += bv; v[2] += bv; v[3] += bv;
        *reinterpret_cast<f32x4*>(tcol + tr * TILE + l) = v;
      } else {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          int ll = l + reg, t2 = tr;
          if (ll >= L_OUT) { ll -= L_OUT; ++t2; }
          tcol[t2 * TILE + ll] = acc[rt][reg] + bv;
        }
      }
      l += 16;
      tr += l / L_OUT;
      l %= L_OUT;
    }
  }
  __syncthreads();

  typedef unsigned u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
  if (a.gnb) {
    // ---- GroupNorm + Mish backward epilogue (input-VJP) ---------------------------------
    // G1 (flat): v = tile + residuals; raw copy out; v back into the tile
    const bool normed = co0 < a.gb_cmax;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        a.out_raw ? a.out_raw : a.out, 0,
        a.out_raw ? (unsigned)((size_t)a.rows * a.raw_tstride * sizeof(float)) : 0u, 0x00020000);
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      const int f = (tid + 256 * k) * 4;
      f32x4 y = *reinterpret_cast<const f32x4*>(lds + f);
      y += rv[k];
      if (a.out_raw) {
        int rem;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), rw,
                                               flat_off(k, a.raw_tstride, a.raw_coff, rem), 0, 0);
      }
      *reinterpret_cast<f32x4*>(lds + f) = y;
    }
    if (!normed) return;                       // uniform per workgroup
    __syncthreads();
    // G2 (rows): thread = one (trajectory, channel) row, lanes = channels
    {
      constexpr int NP = TB / 4;
      const int c = lane, co = co0 + c;
      const int cg = a.gb_cg, cgs = __builtin_ctz(cg);
      const float gam = a.gb_gamma[co], bet = a.gb_beta[co];
      const float inv_n = 1.f / (float)(cg * L_OUT);
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int tr = wave + 4 * j;
        const int tg = min(traj0 + tr, a.rows - 1);
        float* row = lds + (tr * 64 + c) * L_OUT;
        const float* yrow = a.gb_y + ((size_t)tg * a.gb_c + co) * L_OUT;
        const float* stp = a.gb_stats + ((size_t)tg * (a.gb_c >> cgs) + (co >> cgs)) * 2;
        const float mean = stp[0], rstd = stp[1];
        float gz[L_OUT], xh[L_OUT];
        if constexpr (L_OUT % 4 == 0) {
#pragma unroll
          for (int i = 0; i < L_OUT / 4; ++i) {
            const f32x4 t = reinterpret_cast<const f32x4*>(row)[i];
            const f32x4 w = reinterpret_cast<const f32x4*>(yrow)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) { gz[4 * i + e] = t[e]; xh[4 * i + e] = w[e]; }
          }
        } else if constexpr (L_OUT % 2 == 0) {
#pragma unroll
          for (int i = 0; i < L_OUT / 2; ++i) {
            const float2 t = reinterpret_cast<const float2*>(row)[i];
            const float2 w = reinterpret_cast<const float2*>(yrow)[i];
            gz[2 * i] = t.x; gz[2 * i + 1] = t.y; xh[2 * i] = w.x; xh[2 * i + 1] = w.y;
          }
        } else {
#pragma unroll
          for (int i = 0; i < L_OUT; ++i) { gz[i] = row[i]; xh[i] = yrow[i]; }
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < L_OUT; ++i) {
          const float x = (xh[i] - mean) * rstd;
          const float z = fmaf(x, gam, bet);
          const float d = gz[i] * mish_grad_fast_f(z) * gam;
          gz[i] = d;
          xh[i] = x;
          s1 += d;
          s2 = fmaf(d, x, s2);
        }
        for (int m = 1; m < cg; m <<= 1) {
          s1 += __shfl_xor(s1, m);
          s2 += __shfl_xor(s2, m);
        }
        s1 *= inv_n;
        s2 *= inv_n;
#pragma unroll
        for (int i = 0; i < L_OUT; ++i) gz[i] = rstd * (gz[i] - s1 - xh[i] * s2);
        if constexpr (L_OUT % 4 == 0) {
#pragma unroll
          for (int i = 0; i < L_OUT / 4; ++i)
            reinterpret_cast<f32x4*>(row)[i] = f32x4{gz[4 * i], gz[4 * i + 1], gz[4 * i + 2], gz[4 * i + 3]};
        } else if constexpr (L_OUT % 2 == 0) {
#pragma unroll
          for (int i = 0; i < L_OUT / 2; ++i)
            reinterpret_cast<float2*>(row)[i] = float2{gz[2 * i], gz[2 * i + 1]};
        } else {
#pragma unroll
          for (int i = 0; i < L_OUT; ++i) row[i] = gz[i];
        }
      }
    }
    __syncthreads();
    // G3 (flat): d/dy out
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        a.out, 0, (unsigned)((size_t)a.rows * a.out_tstride * sizeof(float)), 0x00020000);
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      const int f = (tid + 256 * k) * 4;
      const f32x4 y = *reinterpret_cast<const f32x4*>(lds + f);
      int rem;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), ro,
                                             flat_off(k, a.out_tstride, a.out_coff, rem), 0, 0);
    }
    return;
  }

  // E1: one (trajectory, channel) row per thread and pass; lanes = the 64 channels, so the
  // channels of a group are neighbouring lanes and the group reduction is a butterfly.
  f32x4* tab = reinterpret_cast<f32x4*>(lds + R * 64);      // [TB][64] {scale, shift, add, -}
  if (has_tab) {
    constexpr int NP = TB / 4;
    const int c = lane, co = co0 + c;
    const float gam = a.gn ? a.gamma[co] : 1.f;
    const float bet = a.gn ? a.beta[co] : 0.f;
    const float ptv = a.pc ? a.pt[co] : 0.f;
    const int cg = a.gn ? a.cg : 1;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int tr = wave + 4 * j;
      const int tg = min(traj0 + tr, a.rows - 1);
      float scale = 1.f, shift = 0.f;
      if (a.gn) {
        const float* row = lds + (tr * 64 + c) * L_OUT;
        float v[L_OUT];
        if constexpr (L_OUT % 4 == 0) {
#pragma unroll
          for (int i = 0; i < L_OUT / 4; ++i) {
            f32x4 t = reinterpret_cast<const f32x4*>(row)[i];
            v[4 * i] = t[0]; v[4 * i + 1] = t[1]; v[4 * i + 2] = t[2]; v[4 * i + 3] = t[3];
          }
        } else if constexpr (L_OUT % 2 == 0) {
#pragma unroll
          for (int i = 0; i < L_OUT / 2; ++i) {
            float2 t = reinterpret_cast<const float2*>(row)[i];
            v[2 * i] = t.x; v[2 * i + 1] = t.y;
          }
        } else {
#pragma unroll
          for (int i = 0; i < L_OUT; ++i) v[i] = row[i];
        }
        // exact two-level moments: row mean / centred sum of squares, then Chan's combination
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < L_OUT; ++i) s += v[i];
        const float rmean = s * (1.f / (float)L_OUT);
        float m2 = 0.f;
#pragma unroll
        for (int i = 0; i < L_OUT; ++i) {
          const float d = v[i] - rmean;
          m2 = fmaf(d, d, m2);
        }
        float gs = rmean;
        for (int m = 1; m < cg; m <<= 1) gs += __shfl_xor(gs, m);
        const float mean = gs / (float)cg;
        const float dm = rmean - mean;
        float gv = fmaf(dm * dm, (float)L_OUT, m2);
        for (int m = 1; m < cg; m <<= 1) gv += __shfl_xor(gv, m);
        const float var = gv / (float)(cg * L_OUT);
        const float rstd = 1.f / sqrtf(var + 1e-5f);
        scale = rstd * gam;
        shift = bet - mean * scale;
        if (a.stats && (c & (cg - 1)) == 0 && traj0 + tr < a.rows) {
          const int cgs = __builtin_ctz(cg);
          const size_t si = ((size_t)(traj0 + tr) * (a.cout >> cgs) + (co >> cgs)) * 2;
          a.stats[si] = mean;
          a.stats[si + 1] = rstd;
        }
      }
      const float add =
          a.pc ? a.pc[(size_t)(tg / a.rows_per_cond) * a.pc_stride + a.pc_off + co] + ptv : 0.f;
      tab[tr * 64 + c] = f32x4{scale, shift, add, 0.f};
    }
    __syncthreads();
  }

  // E2: flat pass -- normalise / activate / add / store, 4 consecutive outputs per access
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
      a.out, 0, (unsigned)((size_t)a.rows * a.out_tstride * sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
      a.ysave ? a.ysave : a.out, 0,
      a.ysave ? (unsigned)((size_t)a.rows * a.cout * L_OUT * sizeof(float)) : 0u, 0x00020000);
#pragma unroll
  for (int k = 0; k < NQ; ++k) {
    const int f = (tid + 256 * k) * 4;
    f32x4 y = *reinterpret_cast<const f32x4*>(lds + f);
    int rem;
    const unsigned oo = flat_off(k, a.out_tstride, a.out_coff, rem);
    if (a.ysave) {
      int rem2;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, y),
                                             ry, flat_off(k, a.cout * L_OUT, 0, rem2), 0, 0);
    }
    if (has_tab) {
      const int tr = f / TILE;
      const int c0 = rem / L_OUT;
      const int l0 = rem - c0 * L_OUT;
      const f32x4 t0 = tab[tr * 64 + c0];
      f32x4 t1 = t0;
      if constexpr (L_OUT % 4 != 0) t1 = tab[tr * 64 + min(c0 + 1, 63)];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool hi = (L_OUT % 4 != 0) && (l0 + i >= L_OUT);
        const float sc = hi ? t1[0] : t0[0], sh = hi ? t1[1] : t0[1], ad = hi ? t1[2] : t0[2];
        float z = fmaf(y[i], sc, sh);
        if (a.gn) z = mish_fast_f(z);
        y[i] = z + ad;
      }
    }
    y += rv[k];
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, y), ro, oo, 0, 0);
  }
}

template <int L_OUT, int IN_STRIDE, int UP, int TAPS, int PADL, int TB, int CH>
int launch_conv(const ConvArgs& a_in, hipStream_t st) {
  using G = ConvGeom<L_OUT, IN_STRIDE, UP, TAPS, PADL, TB, CH>;
  ConvArgs a = a_in;
  if (a.rows <= 0) return TRACE_OK;
  TRACE_REQUIRE(a.cout % 64 == 0 && a.chunk == CH && a.cin % CH == 0 && a.cin2 % CH == 0 &&
                    a.nkb == TAPS * a.cin / 16 + a.cin2 / 16,
                "conv: bad channel geometry cin=%d cin2=%d chunk=%d cout=%d nkb=%d", a.cin, a.cin2,
                a.chunk, a.cout, a.nkb);
  TRACE_REQUIRE(a.cin2 == 0 || (a.in2 && IN_STRIDE == 1 && UP == 1 && a.in2_tstride % 4 == 0 &&
                                a.in2_coff % 16 == 0 && (reinterpret_cast<uintptr_t>(a.in2) & 15) == 0 &&
                                (size_t)(a.rows + TB) * a.in2_tstride * sizeof(float) < 0xffffffffull),
                "conv: second K source needs a stride-1 conv and an aligned tensor below 4 GiB");
  TRACE_REQUIRE(!a.gnb || (!a.gn && !a.pc && !a.ysave && a.gb_y && a.gb_stats && a.gb_gamma && a.gb_beta &&
                           a.gb_cg > 0 && (a.gb_cg & (a.gb_cg - 1)) == 0 && a.gb_cg <= 64 &&
                           a.gb_cmax % 64 == 0 && a.gb_cmax <= a.gb_c &&
                           (reinterpret_cast<uintptr_t>(a.gb_y) & 15) == 0),
                "conv: bad GroupNorm-backward epilogue arguments");
  TRACE_REQUIRE(!a.gn || (a.cg > 0 && (a.cg & (a.cg - 1)) == 0 && a.cg <= 64), "conv: group size %d",
                a.cg);
  TRACE_REQUIRE(a.in_tstride % 4 == 0 && a.in_coff % 16 == 0 &&
                    (reinterpret_cast<uintptr_t>(a.in) & 15) == 0,
                "conv: input must be 16-byte aligned per trajectory chunk");
  TRACE_REQUIRE((size_t)(a.rows + TB) * a.in_tstride * sizeof(float) < 0xffffffffull,
                "conv: input of %d rows exceeds the 4 GiB buffer-descriptor range; split the rows",
                a.rows);
  // the epilogue moves 4 consecutive outputs per access through buffer descriptors
  auto ok128 = [&](const float* p, int tstride, int coff) {
    return p == nullptr || ((reinterpret_cast<uintptr_t>(p) & 15) == 0 && tstride % 4 == 0 &&
                            (coff * L_OUT) % 4 == 0 &&
                            (size_t)a.rows * tstride * sizeof(float) < 0xffffffffull);
  };
  TRACE_REQUIRE(ok128(a.out, a.out_tstride, a.out_coff) && ok128(a.res, a.res_tstride, a.res_coff) &&
                    ok128(a.res2, a.res2_tstride, a.res2_coff) && ok128(a.ysave, a.cout * L_OUT, 0) &&
                    ok128(a.out_raw, a.raw_tstride, a.raw_coff),
                "conv: output / residual tensors must be 16-byte aligned per trajectory tile and "
                "below 4 GiB (%d rows)", a.rows);
  TRACE_REQUIRE(!a.pc || a.pt, "conv: cond table without step table");
  constexpr size_t lds_bytes = G::LDS_FLOATS * sizeof(float);
  // two workgroups per CU must fit in the 160 KiB LDS
  static_assert(lds_bytes <= 80 * 1024, "conv: LDS per workgroup too large");
  static bool attr_set = false;
  if (!attr_set) {
    TRACE_CHECK_HIP(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&conv_gemm_kernel<L_OUT, IN_STRIDE, UP, TAPS, PADL, TB, CH>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    attr_set = true;
  }
  dim3 grid((a.rows + TB - 1) / TB, a.cout / 64);
  const bool prof = prof_active();
  if (prof) prof_pre(st);
  hipLaunchKernelGGL((conv_gemm_kernel<L_OUT, IN_STRIDE, UP, TAPS, PADL, TB, CH>), grid, dim3(256),
                     lds_bytes, st, a);
  if (prof) prof_post(st, 2.0 * a.alg_macs_per_pos * (double)a.rows * L_OUT);
  return launch_status("conv_gemm_kernel");
}

}  // namespace trace
