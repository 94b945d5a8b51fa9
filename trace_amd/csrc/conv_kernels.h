// Implicit-GEMM Conv1d kernels on fp32 MFMA (v_mfma_f32_16x16x4_f32) for the TRACE temporal
// UNet, with the GroupNorm + Mish + (time/cond bias | residual) epilogue fused.
//
// GEMM view of one layer:   out[(traj,l), co] = bias[co] + sum_{tap,ci} in[traj, ci, l*IS + tap] * W[tap,ci,co]
//   rows   = (trajectory, output position) flattened; a workgroup owns TB whole trajectories
//            (TB*L_OUT = 208 rows = 13 MFMA row tiles) so GroupNorm statistics never leave it
//   cols   = 64 output channels per workgroup (4 waves x one 16-wide MFMA column tile)
//   K      = taps x input channels, streamed in chunks of <=64 channels through LDS
//
// A operand: activations staged once per chunk into LDS, channel-major with zeroed halo /
//            zero-stuffing, read with ds_read_b32 (lanes = consecutive positions: conflict-free)
// B operand: weights pre-packed on the host in MFMA fragment order, streamed L2 -> VGPR with one
//            coalesced dwordx4 per 16 k (no LDS, no barrier in the K loop), prefetched one
//            k-block ahead
#pragma once
#include "common.h"

namespace trace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
  const float* in;     // [rows][in_ctot][L_IN]
  int in_tstride;      // floats per trajectory
  int in_coff;         // first input channel
  int cin;             // input channels (multiple of chunk)
  int chunk;           // channels per LDS stage (multiple of 16)
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
  int cg;              // channels per group
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
};

template <int L_OUT, int IN_STRIDE, int UP, int TAPS, int PADL, int TB>
struct ConvGeom {
  static constexpr int R = TB * L_OUT;
  static constexpr int NRT = R / 16;
  static constexpr int LP = (L_OUT - 1) * IN_STRIDE + TAPS;
  static constexpr int S = TB * LP;
  static constexpr int L_IN = (L_OUT * IN_STRIDE) / UP;
  static constexpr int RS = R + 4;
  static_assert(R % 16 == 0, "row block must be a whole number of MFMA tiles");
  static size_t lds_floats(int chunk) {
    size_t a = (size_t)chunk * S, b = (size_t)64 * RS + 2 * TB * 8;
    return a > b ? a : b;
  }
};

template <int L_OUT, int IN_STRIDE, int UP, int TAPS, int PADL, int TB>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(ConvArgs a) {
  using G = ConvGeom<L_OUT, IN_STRIDE, UP, TAPS, PADL, TB>;
  constexpr int NRT = G::NRT, LP = G::LP, S = G::S, L_IN = G::L_IN, RS = G::RS;
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int traj0 = blockIdx.x * TB;
  const int co0 = blockIdx.y * 64;

  for (int i = tid; i < a.chunk * S; i += 256) lds[i] = 0.f;

  int rowoff[NRT];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) {
    int r = rt * 16 + (lane & 15);
    int tr = r / L_OUT;
    int l = r - tr * L_OUT;
    rowoff[rt] = tr * LP + l * IN_STRIDE + (lane >> 4) * S;
  }
  f32x4 acc[NRT];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const f32x4* wp = a.wp + ((size_t)(blockIdx.y * 4 + wave) * a.nkb) * 64 + lane;
  const int kbpc = a.chunk >> 4;
  const int nkk = TAPS * kbpc;
  f32x4 bnext = *wp;
  int kdone = 0;

  for (int c0 = 0; c0 < a.cin; c0 += a.chunk) {
    __syncthreads();
    {
      const int n_el = a.chunk * L_IN;
      for (int idx = tid; idx < TB * n_el; idx += 256) {
        int tr = idx / n_el;
        int rem = idx - tr * n_el;
        int ci = rem / L_IN;
        int i = rem - ci * L_IN;
        int tg = traj0 + tr;
        if (tg < a.rows)
          lds[ci * S + tr * LP + i * UP + PADL] =
              a.in[(size_t)tg * a.in_tstride + (size_t)(a.in_coff + c0 + ci) * L_IN + i];
      }
    }
    __syncthreads();
    int tap = 0, kb = 0;
    for (int kk = 0; kk < nkk; ++kk) {
      f32x4 b = bnext;
      ++kdone;
      wp += 64;
      if (kdone < a.nkb) bnext = *wp;
      const float* lp = lds + kb * 16 * S + tap;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
          float av = lp[rowoff[rt] + s * 4 * S];
          acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[s], acc[rt], 0, 0, 0);
        }
      }
      if (++kb == kbpc) { kb = 0; ++tap; }
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
  float* stat = lds + 64 * RS;
  const int gps = a.gn ? 64 / a.cg : 1;
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
          size_t si = ((size_t)tg * (a.cout / a.cg) + (co0 / a.cg + g)) * 2;
          a.stats[si] = mean;
          a.stats[si + 1] = rstd;
        }
      }
    }
    __syncthreads();
  }
  // ---- normalise / activate / add / store (contiguous per trajectory) ---------------
  for (int tr = 0; tr < TB; ++tr) {
    const int tg = traj0 + tr;
    if (tg >= a.rows) break;
    const float* pcrow =
        a.pc ? a.pc + (size_t)(tg / a.rows_per_cond) * a.pc_stride + a.pc_off + co0 : nullptr;
    for (int e = tid; e < 64 * L_OUT; e += 256) {
      int c = e / L_OUT;
      int l = e - c * L_OUT;
      int co = co0 + c;
      float y = lds[c * RS + tr * L_OUT + l];
      if (a.ysave) a.ysave[((size_t)tg * a.cout + co) * L_OUT + l] = y;
      if (a.gn) {
        int u = tr * gps + c / a.cg;
        float z = (y - stat[2 * u]) * stat[2 * u + 1] * a.gamma[co] + a.beta[co];
        y = mish_f(z);
      }
      if (pcrow) y += pcrow[c] + a.pt[co];
      if (a.res) y += a.res[(size_t)tg * a.res_tstride + (size_t)(a.res_coff + co) * L_OUT + l];
      if (a.res2) y += a.res2[(size_t)tg * a.res2_tstride + (size_t)(a.res2_coff + co) * L_OUT + l];
      a.out[(size_t)tg * a.out_tstride + (size_t)(a.out_coff + co) * L_OUT + l] = y;
    }
  }
}

template <int L_OUT, int IN_STRIDE, int UP, int TAPS, int PADL, int TB>
int launch_conv(const ConvArgs& a, hipStream_t st) {
  using G = ConvGeom<L_OUT, IN_STRIDE, UP, TAPS, PADL, TB>;
  if (a.rows <= 0) return TRACE_OK;
  TRACE_REQUIRE(a.cout % 64 == 0 && a.chunk % 16 == 0 && a.cin % a.chunk == 0 &&
                    a.nkb == TAPS * a.cin / 16,
                "conv: bad channel geometry cin=%d chunk=%d cout=%d nkb=%d", a.cin, a.chunk,
                a.cout, a.nkb);
  TRACE_REQUIRE(!a.gn || (a.cg > 0 && 64 % a.cg == 0), "conv: group size %d", a.cg);
  size_t lds_bytes = G::lds_floats(a.chunk) * sizeof(float);
  // two workgroups per CU must fit in the 160 KiB LDS
  TRACE_REQUIRE(lds_bytes <= 80 * 1024, "conv: LDS %zu too large", lds_bytes);
  static bool attr_set = false;
  if (!attr_set) {
    TRACE_CHECK_HIP(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&conv_gemm_kernel<L_OUT, IN_STRIDE, UP, TAPS, PADL, TB>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    attr_set = true;
  }
  dim3 grid((a.rows + TB - 1) / TB, a.cout / 64);
  hipLaunchKernelGGL((conv_gemm_kernel<L_OUT, IN_STRIDE, UP, TAPS, PADL, TB>), grid, dim3(256),
                     lds_bytes, st, a);
  return launch_status("conv_gemm_kernel");
}

}  // namespace trace
