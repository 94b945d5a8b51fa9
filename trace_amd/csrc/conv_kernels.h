// Implicit-GEMM Conv1d kernels on fp32 MFMA (v_mfma_f32_16x16x4_f32) for the TRACE temporal
// UNet, with the GroupNorm + Mish + (time/cond bias | residual) epilogue fused.
//
// GEMM view of one layer:   out[(traj,l), co] = bias[co] + sum_{tap,ci} in[traj, ci, l*IS + tap] * W[tap,ci,co]
//   rows   = (trajectory, output position) flattened; a workgroup owns TB whole trajectories
//            (TB*L_OUT = 208 rows = 13 MFMA row tiles) so GroupNorm statistics never leave it
//   cols   = 64 output channels per workgroup (4 waves x one 16-wide MFMA column tile)
//   K      = taps x input channels, streamed in chunks of CH (<=64) channels through LDS
//
// A operand: activations staged once per chunk into LDS, channel-major with zeroed halo /
//            zero-stuffing, read with ds_read_b32 (lanes = consecutive positions: conflict-free),
//            software-pipelined one k-step (13 reads) ahead of the MFMAs that consume them
// B operand: weights pre-packed on the host in MFMA fragment order, streamed L2 -> VGPR with one
//            coalesced dwordx4 per 16 k (no LDS, no barrier in the K loop), prefetched one
//            k-block ahead
// staging  : the next chunk's activations are fetched global -> VGPR (dwordx4) before the K loop
//            of the current chunk and written to LDS after it (issue early / write late), so HBM
//            / L2 latency hides under the MFMAs
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
  static constexpr size_t LDS_FLOATS =
      ((size_t)CHS * S > (size_t)64 * RS + 2 * TB * 8 + 3 * TB * 64) ? (size_t)CHS * S
                                                                      : (size_t)64 * RS + 2 * TB * 8 + 3 * TB * 64;
};

template <int L_OUT, int IN_STRIDE, int UP, int TAPS, int PADL, int TB, int CH>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(ConvArgs a) {
  using G = ConvGeom<L_OUT, IN_STRIDE, UP, TAPS, PADL, TB, CH>;
  constexpr int NRT = G::NRT, LP = G::LP, S = G::S, L_IN = G::L_IN, RS = G::RS, CHS = G::CHS;
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
  unsigned soff[NF4];
#pragma unroll
  for (int j = 0; j < NF4; ++j) {
    const int q = tid + 256 * j;
    const int tr = q / G::F4_PER_TRAJ;
    const int rem = q - tr * G::F4_PER_TRAJ;
    soff[j] = q < G::F4_TOTAL
                  ? (unsigned)(((traj0 + tr) * a.in_tstride + a.in_coff * L_IN + rem * 4) * 4)
                  : 0xffffffffu;     // out of range -> zeros
  }
  f32x4 stg[NF4];
  auto stage_load = [&](int c0) {
    const unsigned cb = (unsigned)(c0 * L_IN * 4);
#pragma unroll
    for (int j = 0; j < NF4; ++j) {
      auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, soff[j] == 0xffffffffu ? soff[j] : soff[j] + cb,
                                                     0, 0);
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
  int kdone = 0;

  for (int c0 = 0; c0 < a.cin; c0 += CH) {
    __syncthreads();            // previous chunk fully consumed (and zero fill done)
    stage_write();
    if (c0 + CH < a.cin) stage_load(c0 + CH);   // in flight during this chunk's K loop
    __syncthreads();

    // K loop over the staged chunk.  One ds_read_b128 per (row tile, k-block) supplies the A
    // values of 4 MFMA passes; the read for the NEXT k-block is issued right after the last pass
    // that consumes the register, i.e. 13 MFMAs (>400 cycles) ahead of its first use.
    f32x4 av[NRT];
    const float* lp = lds;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) av[rt] = *reinterpret_cast<const f32x4*>(lp + rowoff[rt]);
    int tap = 0, kb = 0;
    for (int kk = 0; kk < NKK; ++kk) {
      const f32x4 b = bnext;
      ++kdone;
      wp += 64;
      if (kdone < a.nkb) bnext = *wp;
      int kb2 = kb + 1, tap2 = tap;
      if (kb2 == KBPC) { kb2 = 0; ++tap2; }
      const float* lpn = lds + tap2 * CHS + kb2 * 16;
      const bool more = kk + 1 < NKK;
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

  // ---- epilogue: accumulators -> LDS [co][row] -------------------------------------
  __syncthreads();
  {
    const int col = lane & 15, q = lane >> 4;
    const int co_l = wave * 16 + col;
    const float bv = a.bias ? a.bias[co0 + co_l] : 0.f;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      f32x4 v = acc[rt];
      v[0] += bv; v[1] += bv; v[2] += bv; v[3] += bv;
      *reinterpret_cast<f32x4*>(&lds[co_l * RS + rt * 16 + q * 4]) = v;
    }
  }
  __syncthreads();
  float* stat = lds + 64 * RS;                 // [TB*8][2] group statistics
  float* tab_scale = stat + 2 * TB * 8;        // [TB][64] rstd*gamma
  float* tab_shift = tab_scale + TB * 64;      // [TB][64] beta - mean*rstd*gamma
  float* tab_add = tab_shift + TB * 64;        // [TB][64] time/cond bias added after the Mish
  const int cg_shift = a.gn ? __builtin_ctz(a.cg) : 0;
  const int gps = a.gn ? (64 >> cg_shift) : 1;
  if (a.gn) {
    const int nunits = TB * gps;
    const int n = a.cg * L_OUT;
    const float inv_n = 1.f / (float)n;
    for (int u = wave; u < nunits; u += 4) {
      int tr = u / gps;
      int g = u - tr * gps;
      const float* base = lds + (g * a.cg) * RS + tr * L_OUT;
      float s = 0.f;
      for (int e = lane; e < n; e += 64) {
        int c = e / L_OUT;
        int l = e - c * L_OUT;
        s += base[c * RS + l];
      }
      float mean = wave_sum(s) * inv_n;
      float v = 0.f;
      for (int e = lane; e < n; e += 64) {
        int c = e / L_OUT;
        int l = e - c * L_OUT;
        float d = base[c * RS + l] - mean;
        v += d * d;
      }
      float var = wave_sum(v) * inv_n;
      float rstd = 1.f / sqrtf(var + 1e-5f);
      if (lane == 0) {
        stat[2 * u] = mean;
        stat[2 * u + 1] = rstd;
        int tg = traj0 + tr;
        if (a.stats && tg < a.rows) {
          size_t si = ((size_t)tg * (a.cout >> cg_shift) + ((co0 >> cg_shift) + g)) * 2;
          a.stats[si] = mean;
          a.stats[si + 1] = rstd;
        }
      }
    }
    __syncthreads();
  }
  // per-(trajectory, channel) affine of the norm and the additive bias, built once
  if (a.gn || a.pc) {
    for (int idx = tid; idx < TB * 64; idx += 256) {
      const int tr = idx >> 6, c = idx & 63, co = co0 + c;
      const int tg = min(traj0 + tr, a.rows - 1);
      if (a.gn) {
        const int u = tr * gps + (c >> cg_shift);
        const float sc = stat[2 * u + 1] * a.gamma[co];
        tab_scale[idx] = sc;
        tab_shift[idx] = a.beta[co] - stat[2 * u] * sc;
      }
      tab_add[idx] = a.pc ? a.pc[(size_t)(tg / a.rows_per_cond) * a.pc_stride + a.pc_off + co] + a.pt[co]
                          : 0.f;
    }
    __syncthreads();
  }
  // ---- normalise / activate / add / store (contiguous per trajectory) ---------------
  constexpr int STEP_C = 256 / L_OUT, STEP_L = 256 % L_OUT;
  for (int tr = 0; tr < TB; ++tr) {
    const int tg = traj0 + tr;
    if (tg >= a.rows) break;
    const float* resp =
        a.res ? a.res + (size_t)tg * a.res_tstride + (size_t)(a.res_coff + co0) * L_OUT : nullptr;
    const float* res2p =
        a.res2 ? a.res2 + (size_t)tg * a.res2_tstride + (size_t)(a.res2_coff + co0) * L_OUT : nullptr;
    float* outp = a.out + (size_t)tg * a.out_tstride + (size_t)(a.out_coff + co0) * L_OUT;
    float* ysp = a.ysave ? a.ysave + ((size_t)tg * a.cout + co0) * L_OUT : nullptr;
    int c = tid / L_OUT;
    int l = tid - c * L_OUT;
    const float* ytile = lds + tr * L_OUT;
    const int tb = tr * 64;
    for (int e = tid; e < 64 * L_OUT; e += 256) {
      float y = ytile[c * RS + l];
      if (ysp) ysp[e] = y;
      if (a.gn) y = mish_fast_f(fmaf(y, tab_scale[tb + c], tab_shift[tb + c]));
      if (a.pc) y += tab_add[tb + c];
      if (resp) y += resp[e];
      if (res2p) y += res2p[e];
      outp[e] = y;
      c += STEP_C;
      l += STEP_L;
      if (l >= L_OUT) { l -= L_OUT; ++c; }
    }
  }
}

template <int L_OUT, int IN_STRIDE, int UP, int TAPS, int PADL, int TB, int CH>
int launch_conv(const ConvArgs& a, hipStream_t st) {
  using G = ConvGeom<L_OUT, IN_STRIDE, UP, TAPS, PADL, TB, CH>;
  if (a.rows <= 0) return TRACE_OK;
  TRACE_REQUIRE(a.cout % 64 == 0 && a.chunk == CH && a.cin % CH == 0 && a.nkb == TAPS * a.cin / 16,
                "conv: bad channel geometry cin=%d chunk=%d cout=%d nkb=%d", a.cin, a.chunk, a.cout,
                a.nkb);
  TRACE_REQUIRE(!a.gn || (a.cg > 0 && (a.cg & (a.cg - 1)) == 0 && a.cg <= 64), "conv: group size %d",
                a.cg);
  TRACE_REQUIRE(a.in_tstride % 4 == 0 && a.in_coff % 16 == 0 &&
                    (reinterpret_cast<uintptr_t>(a.in) & 15) == 0,
                "conv: input must be 16-byte aligned per trajectory chunk");
  TRACE_REQUIRE((size_t)a.rows * a.in_tstride * sizeof(float) < 0xffffffffull,
                "conv: input of %d rows exceeds the 4 GiB buffer-descriptor range; split the rows",
                a.rows);
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
