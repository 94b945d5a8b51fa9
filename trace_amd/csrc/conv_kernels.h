// Shared pieces of the implicit-GEMM Conv1d kernels of the TRACE temporal UNet (conv_bf3_kernels.h holds the K loop):
// the launch arguments, the operand splits, and the fused epilogue -- GroupNorm + Mish + (time/cond bias | residual)
// forward, GroupNorm + Mish backward for the input-VJP.
//
// GEMM view of one layer:   out[(traj,l), co] = bias[co] + sum_{tap,ci} in[traj, ci, l*IS + tap] * W[tap,ci,co]
//   rows   = (trajectory, output position) flattened; a workgroup owns TB whole trajectories
//            (TB*L_OUT = 208 rows = 13 MFMA row tiles) so GroupNorm statistics never leave it
//   cols   = 64 output channels per workgroup (4 waves x one 16-wide MFMA column tile)
// epilogue : accumulators -> LDS tile laid out like the global tensor; GroupNorm statistics from
//            registers (one (trajectory, channel) row per thread, butterfly over the group);
//            normalise + Mish + cond/time bias + residual(s) and 128-bit stores in one flat pass
// (Round 1's fp32-MFMA K loop, conv_gemm_kernel, left the tree in round 3: no launch had used it since the stride-2
// convs moved to conv_bf3_kernel; DESIGN.md section 5.1 keeps its measurements.)
#pragma once
#include "common.h"

namespace trace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
  const float* in;     // [rows][in_ctot][L_IN]
  int in_tstride;      // floats per trajectory (multiple of 4)
  int in_coff;         // first input channel (multiple of 16)
  int cin;             // input channels (multiple of 32; channels past the layer's own carry zero weights)
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
  unsigned pc_magic;   // 2^32 / rows_per_cond + 1 (0 when rows_per_cond == 1): cond row by multiply-high
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
  // split-bf16 weight stream (conv_bf3_kernels.h): [cout/16][nsl][3 planes][64 lanes] x 8 bf16, or null
  const void* wb;
  int nsl;             // k-slices of 32 = taps * cin / 32 + cin2 / 32
  unsigned long long* dbg;   // diagnostic builds only (TRACE_CONV_STAMPS): 16 time stamps per workgroup
  // Split-fp16 K loop (conv_bf3_kernels.h, SCH = 1): operands as TWO fp16 terms, three MFMAs per product.  fp16 has
  // 5 exponent bits, so both operands are brought into range by exact powers of two: the weights on the host
  // (wexp, per layer), the activations per group of 16 trajectories from the largest |value| their PRODUCER saw
  // (amax arrays: one uint = fp32 bit pattern per 16 trajectories, atomically max-ed by the epilogues that write
  // the tensor, zeroed once per pass).  A launch runs on the fp16 pipe only when wh and the amax of every input
  // are given; the accumulators are scaled back before the epilogue.
  const void* wh;            // split-fp16 weight stream [cout/16][nsl][2 planes][64 lanes] x 8 fp16 (scaled by 2^wexp), or null
  int wexp;
  const unsigned* in_amax;   // max |in| per 16 trajectories (index row / 16), or null
  const unsigned* in2_amax;
  // Fused head (final block only, cout == 64): pred[row][l][o] = head_b[o] + sum_c head_w[o][c] out[row][c][l], formed
  // from the finished output tile in LDS; `out` may then be null (the activation itself is never read again)
  const float* head_w;       // [2][64] or null
  const float* head_b;       // [2]
  float* head_pred;          // [rows][L_OUT][2]
  unsigned* out_amax;        // where this launch's epilogue records max |out| (null: not recorded)
  unsigned* raw_amax;        // ... and max |out_raw| of the GroupNorm-backward epilogue
};

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// 8 fp32 values -> three bf16 planes (exact three-way split)
__device__ __forceinline__ void split8_bf16(const float (&v)[8], u32x4v& h, u32x4v& m, u32x4v& l) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a0 = v[2 * i], a1 = v[2 * i + 1];
    const unsigned ph = cvt_pk_bf16(a0, a1);
    const float r0 = a0 - __builtin_bit_cast(float, ph << 16);
    const float r1 = a1 - __builtin_bit_cast(float, ph & 0xffff0000u);
    const unsigned pm = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, pm << 16);
    const float s1 = r1 - __builtin_bit_cast(float, pm & 0xffff0000u);
    h[i] = ph;
    m[i] = pm;
    l[i] = cvt_pk_bf16(s0, s1);
  }
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// a - float(low / high half of h): one mixed-precision FMA (exact: h is a rounded to 11 bits)
__device__ __forceinline__ float sub_f16_lo(float a, unsigned h) {
  float r;
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(h));
  return r;
}
__device__ __forceinline__ float sub_f16_hi(float a, unsigned h) {
  float r;
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(h));
  return r;
}
// 8 fp32 values, times the power of two s -> two fp16 planes: v s = h + l up to 2^-24 |v s| (l keeps 11 of the 13
// bits h leaves; below |v s| = 2^-3 the low term is subnormal and the absolute error stays at 2^-25)
__device__ __forceinline__ void split8_f16(const float (&v)[8], const float s, u32x4v& h, u32x4v& l) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // h = fp16(v s) in one rounding (mixed-precision FMA writing one half of the register each), l = v s - h
    // exactly (FMA), then rounded to fp16: five instructions per pair
    unsigned ph;
    float l0, l1;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(ph) : "v"(v[2 * i]), "s"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(ph) : "v"(v[2 * i + 1]), "s"(s));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l0) : "v"(v[2 * i]), "s"(s), "v"(ph));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(l1) : "v"(v[2 * i + 1]), "s"(s), "v"(ph));
    h[i] = ph;
    l[i] = cvt_pk_f16(l0, l1);
  }
}

// max over the wave on DPP modifiers (16-lane rows) + v_readlane across rows; v >= 0
__device__ __forceinline__ float wave_max_nonneg(float v) {
  auto step = [](float x, int tag) {
    const int xi = __builtin_bit_cast(int, x);
    int y;
    switch (tag) {
      case 1: y = __builtin_amdgcn_update_dpp(0, xi, 0xB1, 0xF, 0xF, false); break;
      case 2: y = __builtin_amdgcn_update_dpp(0, xi, 0x4E, 0xF, 0xF, false); break;
      case 4: y = __builtin_amdgcn_update_dpp(0, xi, 0x141, 0xF, 0xF, false); break;
      default: y = __builtin_amdgcn_update_dpp(0, xi, 0x140, 0xF, 0xF, false); break;
    }
    return fmaxf(x, __builtin_bit_cast(float, y));
  };
  v = step(v, 1); v = step(v, 2); v = step(v, 4); v = step(v, 8);
  const int x = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// the workgroup's trajectories (at most 16, aligned) share one amax slot
__device__ __forceinline__ void record_amax(unsigned* amax, const int traj0, float m) {
  m = wave_max_nonneg(m);
  if ((threadIdx.x & 63) == 0) atomicMax(amax + (traj0 >> 4), __builtin_bit_cast(unsigned, m));
}
__device__ __forceinline__ float amax4(float m, const f32x4 y) {
  return fmaxf(fmaxf(m, fmaxf(fabsf(y[0]), fabsf(y[1]))), fmaxf(fabsf(y[2]), fabsf(y[3])));
}

#ifdef TRACE_CONV_STAMPS
#define TRACE_STAMP(i)                                                                \
  do {                                                                                \
    if (a.dbg && threadIdx.x == 0)                                                    \
      a.dbg[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime();            \
  } while (0)
unsigned long long* stamp_buffer_for_next_launch();
#else
#define TRACE_STAMP(i) do { } while (0)
#endif

// Sum over the cg neighbouring lanes of a GroupNorm group (cg a power of two <= 64), for NP
// independent values at once.  Inside a row of 16 lanes the butterfly runs on DPP modifiers (pure
// VALU: quad_perm for xor 1 / 2, row_half_mirror and row_mirror for the 8- and 16-lane steps, which
// act as xor 4 / 8 once the lanes of a quad / half-row agree); rows are combined through
// v_readlane.  A ds_bpermute round trip per step costs ~350 cycles on the loaded LDS (measured
// with in-kernel stamps), ten dependent ones per statistic.
__device__ __forceinline__ float dpp_add(float v, int ctrl_tag) {
  const int x = __builtin_bit_cast(int, v);
  int y;
  switch (ctrl_tag) {
    case 1: y = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false); break;    // quad_perm [1,0,3,2]
    case 2: y = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false); break;    // quad_perm [2,3,0,1]
    case 4: y = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false); break;   // row_half_mirror
    default: y = __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false); break;  // row_mirror
  }
  return v + __builtin_bit_cast(float, y);
}

template <int NP>
__device__ __forceinline__ void group_sum(float (&v)[NP], int cg) {
  if (cg > 1) {
#pragma unroll
    for (int j = 0; j < NP; ++j) v[j] = dpp_add(v[j], 1);
  }
  if (cg > 2) {
#pragma unroll
    for (int j = 0; j < NP; ++j) v[j] = dpp_add(v[j], 2);
  }
  if (cg > 4) {
#pragma unroll
    for (int j = 0; j < NP; ++j) v[j] = dpp_add(v[j], 4);
  }
  if (cg > 8) {
#pragma unroll
    for (int j = 0; j < NP; ++j) v[j] = dpp_add(v[j], 8);
  }
  if (cg > 16) {
    const bool upper = (threadIdx.x & 32) != 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int x = __builtin_bit_cast(int, v[j]);
      const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 0));
      const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 16));
      const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 32));
      const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 48));
      const float lo = r0 + r1, hi = r2 + r3;
      v[j] = cg > 32 ? lo + hi : (upper ? hi : lo);
    }
  }
}

// ---- epilogue --------------------------------------------------------------------------------
// The 208 x 64 accumulator tile of a workgroup (wave w owns output channels 16w .. 16w+15 and all
// NRT row tiles, MFMA C layout: column = lane & 15, rows 4 (lane >> 4) + reg) goes through LDS once,
// laid out exactly like the global tensors ([trajectory][channel][position], TILE floats per
// trajectory), so that
//   E1  per-(trajectory, channel) rows are contiguous  -> GroupNorm statistics in registers
//   E2  the whole tile is one flat stream              -> 128-bit LDS reads and global accesses
// Residual operands are fetched into registers before anything else: their latency hides
// under E0/E1 instead of serialising the store loop.
// ROWMAP says which output row MFMA row i of row tile rt is:
//   0  rows in natural order:  r = 16 rt + i = trajectory * L_OUT + position        (conv_gemm_kernel)
//   1  trajectory-major tiles: tile rt holds positions PPT rt .. PPT rt + PPT - 1 (PPT = L_OUT / 13) of
//      all TB trajectories, i = trajectory * PPT + (position - PPT rt)               (conv_bf3_kernel)
// NW = waves per workgroup.  With NW = 8 (conv_bf3_kernel) wave w owns column tile w & 3 and the row tiles
// rt0 .. rt0 + nrtw - 1 (the 13 row tiles split 7 / 6 between the two waves of a column tile): NACC
// accumulator tiles, of which the first nrtw are live.
template <int L_OUT, int TB, int ROWMAP = 0, int NW = 4, int NACC = TB * L_OUT / 16>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x4 (&acc)[NACC], float* lds,
                                              const int traj0, const int co0, const int rt0 = 0,
                                              const int nrtw = NACC, const float oscale = 1.f) {
  // explicit FMAs only: the epilogue's arithmetic must not depend on which candidates the compiler contracts in
  // this instantiation (common.h, mish_fast_f)
#pragma clang fp contract(off)
  constexpr int R = TB * L_OUT;
  constexpr int TILE = 64 * L_OUT;
  constexpr int NTH = NW * 64;                       // threads of the workgroup
  constexpr int NF4 = R * 64 / 4;                    // 16-byte cells of the output tile
  constexpr int NQ = (NF4 + NTH - 1) / NTH;
  constexpr bool QTAIL = NF4 % NTH != 0;             // the last flat pass is partial
  static_assert(NW == 4 || ROWMAP == 1, "8-wave workgroups use the trajectory-major tiles");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int cwave = wave & 3;                        // column tile of the wave
  const bool has_tab = a.gn || a.pc;
  auto flat_live = [&](int k) -> bool { return !QTAIL || (tid + NTH * k) < NF4; };
  // Flat cell tid + NTH k of the tile = trajectory ftg[k] (global index; -1: past `rows` or past the tile), float
  // offset frem[k] inside that trajectory's 64-channel slab.  One division by a constant per thread, the passes
  // differ by compile-time carries; every flat pass (residuals, saved output, raw gradient, output) shares them.
  constexpr int CPT = TILE / 4;                      // 16-byte cells per trajectory
  int ftg[NQ], frem[NQ];
  {
    const int q0 = tid / CPT, r0 = tid - q0 * CPT;
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      const int A = (NTH * k) / CPT, B = (NTH * k) % CPT;
      int r = r0 + B, t = q0 + A;
      if (r >= CPT) { r -= CPT; ++t; }
      const int tg = traj0 + t;
      ftg[k] = (tg < a.rows && flat_live(k)) ? tg : -1;
      frem[k] = r * 4;
    }
  }
  // unsigned 32-bit byte offset: the launcher guarantees rows * tstride * 4 < 4 GiB (so the 24-bit multiply is
  // exact: rows, tstride < 2^24); invalid cells fall outside every descriptor's range
  auto flat_off = [&](int k, int tstride, int coff) -> unsigned {
    const unsigned o = (__umul24((unsigned)ftg[k], (unsigned)tstride) + (unsigned)((coff + co0) * L_OUT + frem[k])) * 4u;
    return ftg[k] >= 0 ? o : 0xffffffffu;
  };
  auto lds_cell = [&](int k) -> float* { return lds + (flat_live(k) ? (tid + NTH * k) * 4 : 0); };
  f32x4 rv[NQ];
  if (a.res) {
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.res), 0, (unsigned)((size_t)a.rows * a.res_tstride * sizeof(float)), 0x00020000);
#pragma unroll
    for (int k = 0; k < NQ; ++k)
      rv[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                            rr, flat_off(k, a.res_tstride, a.res_coff), 0, 0));
  } else {
#pragma unroll
    for (int k = 0; k < NQ; ++k) rv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (a.res2) {
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.res2), 0, (unsigned)((size_t)a.rows * a.res2_tstride * sizeof(float)), 0x00020000);
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                              rr, flat_off(k, a.res2_tstride, a.res2_coff), 0, 0));
      rv[k] += t;
    }
  }

  // E0: accumulators (+ conv bias) -> LDS tile
  __syncthreads();
  {
    const int col = lane & 15, q = lane >> 4;
    const int co_l = cwave * 16 + col;
    const float bv = a.bias ? a.bias[co0 + co_l] : 0.f;
    float* tcol = lds + co_l * L_OUT;
    int tr = 0, l = q * 4;                      // row r = rt*16 + q*4 + reg = tr*L_OUT + l
#pragma unroll
    for (int ri = 0; ri < NACC; ++ri) {
      if (ri >= nrtw) break;
      const int rt = rt0 + ri;
      if constexpr (ROWMAP == 1) {
        // row tile rt = positions PPT rt .. PPT rt + PPT - 1 of all TB trajectories; a half tile's last row tile
        // reaches past L_OUT (Bf3Geom::LPOS): those phantom positions are never written
        constexpr int PPT = 16 / TB;
        static_assert(PPT * TB == 16 && (PPT == 1 || PPT == 2 || PPT == 4 || PPT == 8),
                      "trajectory-major tiles: TB = 16 / PPT trajectories");
        f32x4 v = acc[ri];                      // oscale: the fp16 scheme's power of two (1 otherwise: exact)
        v[0] = fmaf(v[0], oscale, bv); v[1] = fmaf(v[1], oscale, bv);
        v[2] = fmaf(v[2], oscale, bv); v[3] = fmaf(v[3], oscale, bv);
        if constexpr (PPT == 8) {               // lane groups 2t, 2t + 1 = trajectory t, registers = 4 of its 8 positions
          const int p0 = 8 * rt + 4 * (q & 1);
          if (p0 < L_OUT) *reinterpret_cast<f32x4*>(tcol + (q >> 1) * TILE + p0) = v;
        } else if constexpr (PPT == 4) {        // lane group q = trajectory, registers = 4 positions
          if (4 * rt + 3 < L_OUT) {
            if constexpr (L_OUT % 4 == 0) {
              *reinterpret_cast<f32x4*>(tcol + q * TILE + 4 * rt) = v;
            } else {                            // rows of L_OUT = 26 floats start at 8-byte boundaries only
              *reinterpret_cast<float2*>(tcol + q * TILE + 4 * rt) = float2{v[0], v[1]};
              *reinterpret_cast<float2*>(tcol + q * TILE + 4 * rt + 2) = float2{v[2], v[3]};
            }
          } else {                              // L_OUT = 26 on half tiles: two positions left
            if constexpr (L_OUT % 4 == 2) {
              *reinterpret_cast<float2*>(tcol + q * TILE + 4 * rt) = float2{v[0], v[1]};
            } else {
#pragma unroll
              for (int reg = 0; reg < 4; ++reg)
                if (4 * rt + reg < L_OUT) tcol[q * TILE + 4 * rt + reg] = v[reg];
            }
          }
        } else if constexpr (PPT == 2) {        // registers 0,1 / 2,3 = two positions of trajectories 2q / 2q+1
          if constexpr (L_OUT % 2 == 0) {
            *reinterpret_cast<float2*>(tcol + (2 * q) * TILE + 2 * rt) = float2{v[0], v[1]};
            *reinterpret_cast<float2*>(tcol + (2 * q + 1) * TILE + 2 * rt) = float2{v[2], v[3]};
          } else {                              // L_OUT = 13 on half tiles: rows start at odd offsets, last position single
            tcol[(2 * q) * TILE + 2 * rt] = v[0];
            tcol[(2 * q + 1) * TILE + 2 * rt] = v[2];
            if (2 * rt + 1 < L_OUT) {
              tcol[(2 * q) * TILE + 2 * rt + 1] = v[1];
              tcol[(2 * q + 1) * TILE + 2 * rt + 1] = v[3];
            }
          }
        } else {                                // registers = trajectories 4q .. 4q+3, one position
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) tcol[(4 * q + reg) * TILE + rt] = v[reg];
        }
      } else if constexpr (L_OUT % 4 == 0) {    // a quad of rows never straddles two trajectories
        f32x4 v = acc[ri];
        v[0] += bv; v[1] += bv; v[2] += bv; v[3] += bv;
        *reinterpret_cast<f32x4*>(tcol + tr * TILE + l) = v;
      } else {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          int ll = l + reg, t2 = tr;
          if (ll >= L_OUT) { ll -= L_OUT; ++t2; }
          tcol[t2 * TILE + ll] = acc[ri][reg] + bv;
        }
      }
      l += 16;
      tr += l / L_OUT;
      l %= L_OUT;
    }
  }
  __syncthreads();
  TRACE_STAMP(10);

  typedef unsigned u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
  if (a.gnb) {
    // ---- GroupNorm + Mish backward epilogue (input-VJP) ---------------------------------
    // G1 (flat): v = tile + residuals; raw copy out; v back into the tile
    const bool normed = co0 < a.gb_cmax;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        a.out_raw ? a.out_raw : a.out, 0,
        a.out_raw ? (unsigned)((size_t)a.rows * a.raw_tstride * sizeof(float)) : 0u, 0x00020000);
    if (a.out_raw) {                           // uniform; two loops so that neither carries a branch per cell
      float mraw = 0.f;
#pragma unroll
      for (int k = 0; k < NQ; ++k) {
        float* cell = lds_cell(k);
        f32x4 y = *reinterpret_cast<const f32x4*>(cell);
        y += rv[k];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), rw,
                                               flat_off(k, a.raw_tstride, a.raw_coff), 0, 0);
        mraw = amax4(mraw, y);       // cells past the tile repeat cell 0, rows past `rows` hold the conv of zeros
        if (flat_live(k)) *reinterpret_cast<f32x4*>(cell) = y;
      }
      if (a.raw_amax) record_amax(a.raw_amax, traj0, mraw);
    } else {
#pragma unroll
      for (int k = 0; k < NQ; ++k) {
        float* cell = lds_cell(k);
        f32x4 y = *reinterpret_cast<const f32x4*>(cell);
        y += rv[k];
        if (flat_live(k)) *reinterpret_cast<f32x4*>(cell) = y;
      }
    }
    if (!normed) {                             // uniform per workgroup
      TRACE_STAMP(11);
      return;
    }
    __syncthreads();
    // G2 (rows): thread = one (trajectory, channel) row per pass, lanes = channels; with more waves than
    // trajectories (L = 52 on 8 waves) the upper waves have no row
    if (TB >= NW || wave < TB) {
      constexpr int NP = (TB + NW - 1) / NW;
      const int c = lane, co = co0 + c;
      const int cg = a.gb_cg, cgs = __builtin_ctz(cg);
      const float gam = a.gb_gamma[co], bet = a.gb_beta[co];
      const float inv_n = 1.f / (float)(cg * L_OUT);
      float gz[NP][L_OUT], xh[NP][L_OUT], s1[NP], s2[NP], rs[NP];
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int tr = wave + NW * j;
        const int tg = min(traj0 + tr, a.rows - 1);
        const float* row = lds + (tr * 64 + c) * L_OUT;
        const float* yrow = a.gb_y + ((size_t)tg * a.gb_c + co) * L_OUT;
        const float* stp = a.gb_stats + ((size_t)tg * (a.gb_c >> cgs) + (co >> cgs)) * 2;
        const float mean = stp[0], rstd = stp[1];
        rs[j] = rstd;
        if constexpr (L_OUT % 4 == 0) {
#pragma unroll
          for (int i = 0; i < L_OUT / 4; ++i) {
            const f32x4 t = reinterpret_cast<const f32x4*>(row)[i];
            const f32x4 w = reinterpret_cast<const f32x4*>(yrow)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) { gz[j][4 * i + e] = t[e]; xh[j][4 * i + e] = w[e]; }
          }
        } else if constexpr (L_OUT % 2 == 0) {
#pragma unroll
          for (int i = 0; i < L_OUT / 2; ++i) {
            const float2 t = reinterpret_cast<const float2*>(row)[i];
            const float2 w = reinterpret_cast<const float2*>(yrow)[i];
            gz[j][2 * i] = t.x; gz[j][2 * i + 1] = t.y; xh[j][2 * i] = w.x; xh[j][2 * i + 1] = w.y;
          }
        } else {
#pragma unroll
          for (int i = 0; i < L_OUT; ++i) { gz[j][i] = row[i]; xh[j][i] = yrow[i]; }
        }
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int i = 0; i < L_OUT; ++i) {
          const float x = (xh[j][i] - mean) * rstd;
          const float z = fmaf(x, gam, bet);
          const float d = gz[j][i] * mish_grad_fast_f(z) * gam;
          gz[j][i] = d;
          xh[j][i] = x;
          t1 += d;
          t2 = fmaf(d, x, t2);
        }
#ifndef TRACE_NO_ROWSUM_FENCE
        asm volatile("" : "+v"(t1), "+v"(t2));      // same guard as the forward moments
#endif
        s1[j] = t1;
        s2[j] = t2;
      }
      group_sum<NP>(s1, cg);
      group_sum<NP>(s2, cg);
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        float* row = lds + ((wave + NW * j) * 64 + c) * L_OUT;
        const float m1 = s1[j] * inv_n, m2 = s2[j] * inv_n, rstd = rs[j];
#pragma unroll
        for (int i = 0; i < L_OUT; ++i) gz[j][i] = rstd * (gz[j][i] - m1 - xh[j][i] * m2);
        if constexpr (L_OUT % 4 == 0) {
#pragma unroll
          for (int i = 0; i < L_OUT / 4; ++i)
            reinterpret_cast<f32x4*>(row)[i] =
                f32x4{gz[j][4 * i], gz[j][4 * i + 1], gz[j][4 * i + 2], gz[j][4 * i + 3]};
        } else if constexpr (L_OUT % 2 == 0) {
#pragma unroll
          for (int i = 0; i < L_OUT / 2; ++i)
            reinterpret_cast<float2*>(row)[i] = float2{gz[j][2 * i], gz[j][2 * i + 1]};
        } else {
#pragma unroll
          for (int i = 0; i < L_OUT; ++i) row[i] = gz[j][i];
        }
      }
    }
    __syncthreads();
    // G3 (flat): d/dy out
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        a.out, 0, (unsigned)((size_t)a.rows * a.out_tstride * sizeof(float)), 0x00020000);
    float mout = 0.f;
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      const f32x4 y = *reinterpret_cast<const f32x4*>(lds_cell(k));
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), ro,
                                             flat_off(k, a.out_tstride, a.out_coff), 0, 0);
      mout = amax4(mout, y);
    }
    if (a.out_amax) record_amax(a.out_amax, traj0, mout);
    TRACE_STAMP(11);
    return;
  }

  // Raw conv outputs for the backward (clean guidance only): flat copy of the tile before it is
  // normalised in place.
  if (a.ysave) {
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        a.ysave, 0, (unsigned)((size_t)a.rows * a.cout * L_OUT * sizeof(float)), 0x00020000);
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      const f32x4 y = *reinterpret_cast<const f32x4*>(lds_cell(k));
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), ry,
                                             flat_off(k, a.cout * L_OUT, 0), 0, 0);
    }
    if (has_tab) __syncthreads();          // every flat read done before rows are rewritten
  }

  // E1: one (trajectory, channel) row per thread and pass; lanes = the 64 channels, so the
  // channels of a group are neighbouring lanes and the group reduction is a butterfly.  The thread
  // keeps its rows in registers, normalises / activates / adds the condition there and writes the
  // finished rows back into the tile: no parameter table in LDS, so the tile alone (R x 64 floats =
  // 52 KB) is the epilogue's footprint and three workgroups fit a CU.
  if (has_tab && (TB >= NW || wave < TB)) {
    constexpr int NP = (TB + NW - 1) / NW;
    const int c = lane, co = co0 + c;
    const float gam = a.gn ? a.gamma[co] : 1.f;
    const float bet = a.gn ? a.beta[co] : 0.f;
    const float ptv = a.pc ? a.pt[co] : 0.f;
    const int cg = a.gn ? a.cg : 1;
    float add[NP], rmean[NP], m2[NP], gs[NP];
    float v[NP][L_OUT];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int tg = min(traj0 + wave + NW * j, a.rows - 1);
      // cond row = tg / rows_per_cond by multiply-high (pc_magic = 2^32 / rows_per_cond + 1, exact below 2^24 rows;
      // 0 = one row per trajectory)
      const unsigned cr = a.pc_magic ? __umulhi((unsigned)tg, a.pc_magic) : (unsigned)tg;
      add[j] = a.pc ? a.pc[(size_t)cr * a.pc_stride + a.pc_off + co] + ptv : 0.f;
      rmean[j] = 0.f;
      m2[j] = 0.f;
      const float* row = lds + ((wave + NW * j) * 64 + c) * L_OUT;
      if constexpr (L_OUT % 4 == 0) {
#pragma unroll
        for (int i = 0; i < L_OUT / 4; ++i) {
          f32x4 t = reinterpret_cast<const f32x4*>(row)[i];
          v[j][4 * i] = t[0]; v[j][4 * i + 1] = t[1]; v[j][4 * i + 2] = t[2]; v[j][4 * i + 3] = t[3];
        }
      } else if constexpr (L_OUT % 2 == 0) {
#pragma unroll
        for (int i = 0; i < L_OUT / 2; ++i) {
          float2 t = reinterpret_cast<const float2*>(row)[i];
          v[j][2 * i] = t.x; v[j][2 * i + 1] = t.y;
        }
      } else {
#pragma unroll
        for (int i = 0; i < L_OUT; ++i) v[j][i] = row[i];
      }
    }
    float scale[NP], shift[NP];
    if (a.gn) {
      // exact two-level moments: row mean / centred sum of squares per (trajectory, channel)
      // row, then Chan's combination over the channels of the group
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < L_OUT; ++i) sacc += v[j][i];
        float rm = sacc * (1.f / (float)L_OUT);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < L_OUT; ++i) {
          const float d = v[j][i] - rm;
          q = fmaf(d, d, q);
        }
        // Each row's moments stay in registers of their own: with the two rows of a thread SLP-packed
        // (v_pk_add_f32 chains fed by v_mov pairs of just-returned LDS registers behind partial
        // s_waitcnt lgkmcnt(N)) lanes 48..63 of the sums differed from run to run on gfx950 (DESIGN.md
        // section 5.1, "run-to-run differences"); the opaque use keeps the sums scalar whatever the flags.
#ifndef TRACE_NO_ROWSUM_FENCE      // diagnostic builds only (profiles/r3_groupnorm_isa_*): variant A of DESIGN.md section 5.1
        asm volatile("" : "+v"(rm), "+v"(q));
#endif
        rmean[j] = rm;
        m2[j] = q;
        gs[j] = rm;
      }
      TRACE_STAMP(6);
      group_sum<NP>(gs, cg);
      float gv[NP];
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        gs[j] = gs[j] / (float)cg;                       // group mean
        const float dm = rmean[j] - gs[j];
        gv[j] = fmaf(dm * dm, (float)L_OUT, m2[j]);
      }
      group_sum<NP>(gv, cg);
      TRACE_STAMP(7);
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int tr = wave + NW * j;
        const float mean = gs[j];
        const float var = gv[j] / (float)(cg * L_OUT);
        const float rstd = 1.f / sqrtf(var + 1e-5f);
        scale[j] = rstd * gam;
        shift[j] = bet - mean * scale[j];
        if (a.stats && (c & (cg - 1)) == 0 && traj0 + tr < a.rows) {
          const int cgs = __builtin_ctz(cg);
          const size_t si = ((size_t)(traj0 + tr) * (a.cout >> cgs) + (co >> cgs)) * 2;
          a.stats[si] = mean;
          a.stats[si + 1] = rstd;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NP; ++j) { scale[j] = 1.f; shift[j] = 0.f; }
    }
    TRACE_STAMP(8);
    // (uniform condition hoisted: a branch per element kept the Mish chains of a row from overlapping)
    if (a.gn) {
#pragma unroll
      for (int j = 0; j < NP; ++j)
#pragma unroll
        for (int i = 0; i < L_OUT; ++i) v[j][i] = mish_fast_f(fmaf(v[j][i], scale[j], shift[j])) + add[j];
    } else {
#pragma unroll
      for (int j = 0; j < NP; ++j)
#pragma unroll
        for (int i = 0; i < L_OUT; ++i) v[j][i] = fmaf(v[j][i], scale[j], shift[j]) + add[j];
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      float* row = lds + ((wave + NW * j) * 64 + c) * L_OUT;
      if constexpr (L_OUT % 4 == 0) {
#pragma unroll
        for (int i = 0; i < L_OUT / 4; ++i)
          reinterpret_cast<f32x4*>(row)[i] = f32x4{v[j][4 * i], v[j][4 * i + 1], v[j][4 * i + 2], v[j][4 * i + 3]};
      } else if constexpr (L_OUT % 2 == 0) {
#pragma unroll
        for (int i = 0; i < L_OUT / 2; ++i) reinterpret_cast<float2*>(row)[i] = float2{v[j][2 * i], v[j][2 * i + 1]};
      } else {
#pragma unroll
        for (int i = 0; i < L_OUT; ++i) row[i] = v[j][i];
      }
    }
  }
  if (has_tab) __syncthreads();
  TRACE_STAMP(9);

  // Fused head: the tile holds the finished activation (no residual on this layer); one (trajectory, position) per
  // thread, the 64 channels in the order and with the FMAs of head_fwd_kernel (same bits)
  if (a.head_pred) {
    for (int idx = tid; idx < R; idx += NTH) {
      const int tr = idx / L_OUT, l = idx - tr * L_OUT;
      const float* col = lds + tr * TILE + l;
      float p0 = a.head_b[0], p1 = a.head_b[1];
#pragma unroll 16
      for (int ch = 0; ch < 64; ++ch) {
        const float v = col[ch * L_OUT];
        p0 = fmaf(a.head_w[ch], v, p0);
        p1 = fmaf(a.head_w[64 + ch], v, p1);
      }
      if (traj0 + tr < a.rows)
        *reinterpret_cast<float2*>(a.head_pred + ((size_t)(traj0 + tr) * L_OUT + l) * 2) = float2{p0, p1};
    }
    if (!a.out) { TRACE_STAMP(11); return; }
  }
  // E2: flat pass -- + residual(s), store; 4 consecutive outputs per access
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
      a.out, 0, (unsigned)((size_t)a.rows * a.out_tstride * sizeof(float)), 0x00020000);
  float mout = 0.f;
#pragma unroll
  for (int k = 0; k < NQ; ++k) {
    f32x4 y = *reinterpret_cast<const f32x4*>(lds_cell(k));
    y += rv[k];
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), ro, flat_off(k, a.out_tstride, a.out_coff), 0, 0);
    mout = amax4(mout, y);
  }
  if (a.out_amax) record_amax(a.out_amax, traj0, mout);
  TRACE_STAMP(11);
}

}  // namespace trace
