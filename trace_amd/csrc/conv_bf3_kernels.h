// Implicit-GEMM Conv1d on the bf16 matrix pipe with fp32 results: every fp32 operand is split
// exactly into three bf16 terms, a = h + m + l (round-to-nearest at each step, so the three terms
// carry all 24 significand bits), and a product a*b is accumulated as the six terms
//     l*h' + h*l' + m*m' + m*h' + h*m' + h*h'
// on v_mfma_f32_16x16x32_bf16 (fp32 accumulate; the dropped m*l', l*m', l*l' are below 2^-25 of
// the product).  Measured against an fp64 sum at K = 1280 (scripts/probe/bf16x3.hip): rms relative
// error 5.3e-7, the native fp32 MFMA 5.8e-7; rate 405 vs 154 fp32-equivalent TFLOP/s.
//
// Tile: 208 rows x 64 output channels per workgroup, wave w owns channels 16w .. 16w+15 and all 13 row tiles;
// launches of at most one such tile per CU take HALF tiles (104 rows, 7 row tiles, the last half phantom: Bf3Geom;
// same bits).  The fused epilogues (GroupNorm + Mish forward / backward, residuals) are
// conv_kernels.h's conv_epilogue.
//
// A operand: activations stay fp32 in HBM.  A chunk of 32 channels is fetched global -> VGPR
//            (8 channels x 1 position per thread and pass, dword buffer loads coalesced along the
//            position axis), split in registers, and written as three bf16 planes of four arrays
//            [plane][channel octet][position] x 16 B with one ds_write_b128 per plane; a lane's MFMA
//            fragment (8 consecutive channels of one position) is one ds_read_b128 per plane.
//            The arrays are multiples of 256 B, so the 16-lane groups the LDS serves a b128 read in
//            (rows 0-3, 12-15 of octet kq with rows 4-11 of octet kq+1, MI355X_MICROARCH.md) fall on
//            16 consecutive 16-byte slots: conflict-free (the [position][channel] image of the first
//            version spent 62 % of its LDS cycles on bank conflicts, LDS 66 % busy).
// B operand: weights split and packed on the host in fragment order [col tile][slice][plane][lane],
//            streamed L2 -> VGPR WPF slices (one slice = one tap x 32 channels) ahead: one on full tiles, three on
//            half tiles (a lone workgroup per CU has no co-resident wave to cover the round trip).
// rows     : row tile rt = positions PPT rt .. of ALL trajectories of the workgroup (PPT = L / 13), so
//            one base address serves every tile (+ immediates) and the 16 rows of a read are 16
//            trajectories x positions at distinct slots (Bf3Geom::lp_pad); the epilogue's ROWMAP = 1
// K loop   : fully unrolled over the taps of a chunk; the A fragments of row tile n+3 are requested
//            while tile n multiplies (4 register stages), 6 MFMAs per (row tile, slice).
#pragma once
#include <cstdint>
#include <type_traits>

#include "conv_kernels.h"

namespace trace {

#ifndef TRACE_BF3_NW
#define TRACE_BF3_NW 4            // waves per workgroup of the full-chip launches (small launches: launch_conv_bf3)
#endif
// IS = input stride (2: the down-sampling convs and the transposed convs' dgrad), UP = zero-stuffing factor
template <int L_OUT, int UP, int TAPS, int PADL, int TB, int NW = TRACE_BF3_NW, int CHN = 32, int IS = 1>
struct Bf3Geom {
  static_assert(IS == 1 || (IS == 2 && UP == 1), "stride 2 without zero-stuffing");
  static constexpr int NTH = NW * 64;                      // threads per workgroup
  static_assert(NW == 4 || NW == 8 || NW == 16, "4, 8 or 16 waves per workgroup");
  // wave w owns column tile w & 3 and part w >> 2 of the row tiles: the NRT tiles are dealt to NW / 4 parts,
  // the first NRT % NPART parts take one more
  static constexpr int NPART = NW / 4;
  static_assert(CHN == 32 || CHN == 64, "chunks of 32 or 64 channels");
  static constexpr int CH = CHN;                           // channels per chunk = CH / 32 MFMA k-slices per tap
  static constexpr int NOCT = CH / 8;                      // channel octets = (plane, octet) arrays per plane
  static constexpr int NPLMAX = CHN == 32 ? 3 : 2;         // 64-channel chunks exist for the two-plane scheme only
  static constexpr int R = TB * L_OUT;                     // output rows of the workgroup (the epilogue's tile)
  static constexpr int PPT = 16 / TB;                      // positions per row tile (8, 4, 2, 1)
  static constexpr int NRT = (L_OUT + PPT - 1) / PPT;      // row tiles
  // Full tiles (TB L_OUT = 208: L = 13 / 26 / 52 with TB = 16 / 8 / 4) have NRT PPT = L_OUT.  HALF tiles (TB = 8 / 4 /
  // 2: 104 rows, launches of at most one workgroup per CU -- twice as many workgroups, each with half the serial
  // chain) have 7 row tiles whose last one is half phantom: positions L_OUT .. LPOS - 1 read zero halo cells and
  // their accumulators are never written (conv_epilogue E0).
  static constexpr int LPOS = NRT * PPT;
  // Row tile rt = positions PPT rt .. + PPT - 1 of all TB trajectories (MFMA row i = trajectory * PPT +
  // position offset), so the 16 rows of an LDS read sit at cells trajectory * LP + const: LP is the
  // staged positions per trajectory, padded until those 16 cells fall into 16 different 16-byte
  // slots modulo 16 (TB = 2: LP = 8 mod 16; TB = 4: LP = 4 or 12 mod 16; TB = 8: LP / 2 odd; TB = 16: LP odd).
  static constexpr int lp_pad(int lp) {
    while (!(TB == 2 ? lp % 16 == 8 : TB == 4 ? (lp % 16 == 4 || lp % 16 == 12)
                                    : TB == 8 ? (lp % 2 == 0 && (lp / 2) % 2 == 1) : lp % 2 == 1)) ++lp;
    return lp;
  }
  static constexpr int LP = lp_pad((LPOS - 1) * IS + TAPS);
  static constexpr int S = TB * LP;
  static constexpr int L_IN = L_OUT * IS / UP;
  static constexpr int SUB = (S * 16 + 255) / 256 * 256;   // bytes of one (plane, channel octet) array
  static constexpr int PLANE = NOCT * SUB;                 // bytes per plane
  static constexpr int NPOS = TB * L_IN;                   // data positions per chunk
  static constexpr int NU = NPOS * (CH / 8);               // staging units (position, channel octet)
  static constexpr int NUT = (NU + NTH - 1) / NTH;
  static constexpr int NACC = (NRT + NPART - 1) / NPART;       // accumulator tiles per wave (the larger parts)
  static constexpr int NREM = NRT % NPART;                     // parts that have NACC tiles (0: all of them)
  // the output tile (conv_epilogue)
  static constexpr size_t EPI_BYTES = (size_t)R * 64 * sizeof(float);
  static constexpr size_t LDS_BYTES = (size_t)NPLMAX * PLANE > EPI_BYTES ? (size_t)NPLMAX * PLANE : EPI_BYTES;
  static_assert(L_OUT % UP == 0 && (TB == 2 || TB == 4 || TB == 8 || TB == 16) && PPT * TB == 16 && (R * 64) % 4 == 0, "bad tile");
  static_assert(PLANE + NOCT * SUB < 131072, "plane offsets stay small");
};

#ifndef TRACE_BF3_WPF
#define TRACE_BF3_WPF 1           // weight slices in flight ahead of the one being multiplied (full-chip launches)
#endif
#ifndef TRACE_BF3_WGS
#define TRACE_BF3_WGS 2           // resident workgroups per CU the register budget is sized for
#endif
// The input is staged from the fp32 tensor through registers, with the split in the kernel.
// MINW = waves per SIMD the register budget is sized for (2: up to 256 VGPRs)
// SCH = 0: three bf16 planes, six MFMAs per product (header of this file).
// SCH = 1: TWO fp16 planes, a s = h + l with s a power of two that brings the operand into fp16's range (the
//          weights' on the host, the activations' from the amax their producer recorded per 16 trajectories), and
//          the three terms  l*h' + h*l' + h*h'  on v_mfma_f32_16x16x32_f16 (the dropped l*l' is below 2^-22 of the
//          product; MFMA honours fp16 subnormal inputs, so small low terms degrade gracefully).  Measured against
//          an fp64 sum at K = 1280 (scripts/probe/f16x2.hip): rms relative error 3.6e-7 (bf16 x 3: 5.3e-7, the fp32
//          MFMA 5.8e-7 -- half the fp32 accumulations), at twice the rate.  Same LDS image, tile and epilogue.
// CHN = channels per chunk (LDS image / barrier pair): 32, or 64 for the fp16 scheme (half the chunk switches; the
//       two k-slices of a tap are multiplied one after the other, the next chunk's loads issued half in each).
template <int L_OUT, int UP, int TAPS, int PADL, int TB, int SCH, int NW = TRACE_BF3_NW,
          int MINW = TRACE_BF3_WGS * NW / 4, int CHN = 32, int IS = 1, int WPF = TRACE_BF3_WPF>
__device__ __forceinline__ void conv_bf3_body(const ConvArgs& a, const int rb, const int ct) {
  using G = Bf3Geom<L_OUT, UP, TAPS, PADL, TB, NW, CHN, IS>;
  static_assert(SCH == 0 || SCH == 1, "two operand schemes");
  static_assert(CHN == 32 || SCH == 1, "64-channel chunks: fp16 scheme only");
  constexpr int NH = CHN / 32;                   // k-slices per tap and chunk
  constexpr int NPL = SCH == 1 ? 2 : 3;
  constexpr int LP = G::LP, L_IN = G::L_IN, SUB = G::SUB, PLANE = G::PLANE;
  constexpr int NUT = G::NUT, NU = G::NU, NPOS = G::NPOS, CH = G::CH, NTH = G::NTH, NACC = G::NACC;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const ldsb = reinterpret_cast<char*>(lds);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  // wave -> (column tile, row tiles).  A workgroup's waves go to the SIMDs cyclically, so the waves of one
  // column tile (w, w + 4, ...) share a SIMD and every SIMD carries the same MFMA load.
  const int cwave = wave & 3;
  const int rpart = wave >> 2;
  constexpr int NREM = G::NREM;
  const int nrtw = (NREM == 0 || rpart < NREM) ? NACC : NACC - 1;
  const int rt0 = NREM == 0 ? rpart * NACC : (rpart < NREM ? rpart * NACC : NREM * NACC + (rpart - NREM) * (NACC - 1));
  if (rb * TB >= a.rows) return;
  const int traj0 = rb * TB;
  const int co0 = ct * 64;
  TRACE_STAMP(0);
#ifdef TRACE_CONV_STAMPS
  if (a.dbg && threadIdx.x == 0) {
    a.dbg[(size_t)blockIdx.x * 16 + 12] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_ID
    a.dbg[(size_t)blockIdx.x * 16 + 13] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // XCC_ID
  }
#endif
  // fp16 scheme: the activations of this workgroup's trajectories are multiplied by 2^(14 - E), E = exponent of
  // the largest |value| of their group of 16 (both K sources share the accumulators, hence one scale), so the
  // scaled maximum lies in [2^14, 2^15); the accumulators are scaled back by 2^-(14 - E) 2^-wexp
  float ascale = 1.f, oscale = 1.f;
  if constexpr (SCH == 1) {
    unsigned am = __hip_atomic_load(a.in_amax + (traj0 >> 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.cin2) {
      const unsigned am2 = __hip_atomic_load(a.in2_amax + (traj0 >> 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      am = am > am2 ? am : am2;
    }
    int ef = (int)(__builtin_amdgcn_readfirstlane(am) >> 23);
    // |max| clamped to [2^-71, 2^126]: with wexp in [-10, 40] both scales below stay normal floats (exponent fields
    // 268 - ef in [15, 212], ef - 14 - wexp in [2, 249]), i.e. every group maximum an fp32 activation can take short
    // of overflowing is scaled exactly (a clamp at 2^40 turned residual-path activations of 1e12 and more -- heavy-tailed
    // weights, tests/test_gpu_parity.py -- into fp16 infinities)
    ef = ef < 56 ? 56 : (ef > 253 ? 253 : ef);
    const int sf = 268 - ef;                               // exponent field of 2^(14 - E)
    ascale = __builtin_bit_cast(float, (unsigned)sf << 23);
    oscale = __builtin_bit_cast(float, (unsigned)(254 - sf - a.wexp) << 23);
  }

  // ---- staging ------------------------------------------------------------------------------
  const int nch1 = a.cin / CH;
  const int nch = nch1 + a.cin2 / CH;
  // the fp32 tensor goes through registers
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in), 0, (unsigned)((size_t)a.rows * a.in_tstride * sizeof(float)),
      0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in2 ? a.in2 : a.in), 0,
      a.in2 ? (unsigned)((size_t)a.rows * a.in2_tstride * sizeof(float)) : 0u, 0x00020000);
  constexpr int NUTR = NUT;
  // unit q = tid + NTH j: octet o = q / NPOS, position pp = q % NPOS (lanes run along positions)
  unsigned spk[NUTR];           // (trajectory in tile) << 20 | float offset inside the chunk
  unsigned wofs[NUTR];          // LDS byte offset of the unit inside a plane
  {
#pragma unroll
    for (int j = 0; j < NUT; ++j) {
      const int q = tid + NTH * j;
      const int o = q / NPOS;
      const int pp = q - o * NPOS;
      const int tr = pp / L_IN;
      const int i = pp - tr * L_IN;
      spk[j] = ((unsigned)tr << 20) | (unsigned)(o * 8 * L_IN + i);
      wofs[j] = (unsigned)((tr * LP + i * UP + PADL) * 16 + o * SUB);
    }
  }
  float stg[NUTR][8];
  // the 8 channel loads of unit j of chunk ch (cbase / descriptor choice are wave-uniform)
  // ch < 0 (no next chunk): the loads are still issued, at an offset outside the descriptor's range (zeros, no
  // memory access) -- a conditional load would make the compiler copy all staging registers at every chunk
  auto stage_load_unit = [&](int ch, int j) {
    {
      const bool none = ch < 0;
      const bool second = ch >= nch1;
      const int tstride = second ? a.in2_tstride : a.in_tstride;
      const int cbase = ((second ? a.in2_coff + (ch - nch1) * CH : a.in_coff + ch * CH) * L_IN) +
                        traj0 * tstride;
      // units past NU (last pass) read a valid address and are never written to LDS; trajectories
      // past `rows` fall outside the descriptor's range and read as zeros
      const bool live = (NU % NTH == 0) || (tid + NTH * j < NU);
      const unsigned sp = spk[j];
      const unsigned off = none ? 0xffffe000u
                                : (live ? ((sp >> 20) * (unsigned)tstride + (sp & 0xfffffu) + (unsigned)cbase) * 4u : 0u);
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        stg[j][jj] = second ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc2, off, jj * L_IN * 4, 0))
                            : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, jj * L_IN * 4, 0));
      }
    }
  };
#ifndef TRACE_BF3_SPLIT_IN_K
#define TRACE_BF3_SPLIT_IN_K 0    // split the next chunk into bf16 planes between the MFMAs of the current one
#endif
  constexpr bool SPLITK = TRACE_BF3_SPLIT_IN_K && TAPS > 1;     // the 1x1 kernels have no MFMA shadow to hide it in
  constexpr int NUTS = SCH == 1 ? 1 : NUTR;      // the fp16 scheme splits inside stage_write
  u32x4v sph[NUTS], spm[NUTS], spl[NUTS];
  auto split_unit = [&](int j) {
    if constexpr (SCH == 0) split8_bf16(stg[j], sph[j], spm[j], spl[j]);
  };
  // the bf16 planes of the staged chunk go to LDS from sph / spm / spl
  auto stage_write = [&]() {
    if constexpr (SCH == 1) {                    // split and write unit by unit: no second register set
#pragma unroll
      for (int j = 0; j < NUT; ++j) {
        u32x4v h, l;
        split8_f16(stg[j], ascale, h, l);
        if ((NU % NTH == 0) || (tid + NTH * j < NU)) {
          char* p = ldsb + wofs[j];
          *reinterpret_cast<u32x4v*>(p) = h;
          *reinterpret_cast<u32x4v*>(p + PLANE) = l;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NUT; ++j) {
        if ((NU % NTH == 0) || (tid + NTH * j < NU)) {
          char* p = ldsb + wofs[j];
          *reinterpret_cast<u32x4v*>(p) = sph[j];
          *reinterpret_cast<u32x4v*>(p + PLANE) = spm[j];
          *reinterpret_cast<u32x4v*>(p + 2 * PLANE) = spl[j];
        }
      }
    }
  };

  {
#pragma unroll
    for (int j = 0; j < NUT; ++j) stage_load_unit(0, j);
    // halo / zero-stuffing positions are written once; data positions are rewritten by every chunk
    if constexpr (UP == 1 && TAPS == 1 && IS == 1) {
    } else if constexpr (UP == 1) {
      constexpr int HALO = LP - L_IN;                      // PADL cells in front of the data, the rest behind
      for (int i = tid; i < G::NOCT * NPL * TB * HALO; i += NTH) {    // (plane, octet) arrays x trajectories x halo cells
        const int arr = i / (TB * HALO);
        const int r0 = i - arr * (TB * HALO);
        const int tr = r0 / HALO;
        const int h = r0 - tr * HALO;
        const int pos = tr * LP + (h < PADL ? h : L_IN + h);
        *reinterpret_cast<u32x4v*>(ldsb + arr * SUB + pos * 16) = u32x4v{0u, 0u, 0u, 0u};
      }
    } else {
#pragma unroll 2
      for (int i = tid; i < NPL * PLANE / 16; i += NTH)
        reinterpret_cast<u32x4v*>(ldsb)[i] = u32x4v{0u, 0u, 0u, 0u};
    }
  }

  // lane (MFMA row i = lane & 15, kq = lane >> 4) reads channels 8 kq .. 8 kq + 7 of the input position of
  // its output row (trajectory i / PPT, position PPT rt + i % PPT): one ds_read_b128 per plane, and
  // the row tiles differ by a constant (PPT cells)
  constexpr int PPT = G::PPT;
  const int rowoff0 = (((lane & 15) / PPT) * LP + ((lane & 15) % PPT) * IS) * 16 + (lane >> 4) * SUB;
  // one base per tap, made opaque: tile rt at tap t and tile rt - 1 at tap t + PPT read the same cell,
  // and if the compiler can see that it merges the reads and keeps the fragments live (spills)
  int tapoff[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    tapoff[t] = rowoff0 + t * 16 + rt0 * PPT * IS * 16;
    asm volatile("" : "+v"(tapoff[t]));
  }
  f32x4 acc[NACC];
#pragma unroll
  for (int rt = 0; rt < NACC; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // weight stream: [col tile][slice][plane][lane] fragments of 8 bf16
  constexpr int WSL = 64 * NPL;                  // fragments per slice
  const bf16x8* wp = reinterpret_cast<const bf16x8*>(SCH == 1 ? a.wh : a.wb) + ((size_t)(ct * 4 + cwave) * a.nsl) * WSL + lane;
#ifndef TRACE_BF3_NST
#define TRACE_BF3_NST 4           // A-fragment register stages (prefetch distance + 1)
#endif
  // WPF weight slices in flight ahead of the one being multiplied: bn[0] is the next slice, bn[WPF - 1] the newest
  // request.  Full-chip launches keep one (the co-resident wave covers an L2 round trip and the registers are
  // needed elsewhere); a lone workgroup per CU waits a full L2 / Infinity-Cache round trip per slice with one,
  // so half tiles (at most one workgroup per CU, <= 181 VGPRs) keep more
  constexpr int WPFN = WPF < 1 ? 1 : WPF;
  bf16x8 bn[WPFN][NPL];
#pragma unroll
  for (int i = 0; i < WPFN; ++i) {
#pragma unroll
    for (int p = 0; p < NPL; ++p) bn[i][p] = (i == 0 || i < a.nsl) ? wp[i * WSL + 64 * p] : bn[0][p];
  }
  int sdone = 0;

  // one chunk's MFMAs over taps [T0, T1); the next chunk's activations (chunk `next`, -1 = none) are
  // requested one staging unit at a time between row tiles, after the slice's weight loads, so that
  // neither the issue of 32 loads in a row nor a vmcnt wait for weights stalls on them
#ifdef TRACE_STAMP_SLICES       // diagnostic (scripts/stamps_slices.py): per-slice / per-tile stamps of chunks 1 and 2, wave 0
  int stamp_ch = -1;
#define TRACE_SLICE_STAMP(k) do { if (a.dbg && threadIdx.x == 0) a.dbg[131072 + (size_t)blockIdx.x * 64 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
  auto chunk_mma = [&](auto t0c, auto t1c, auto nrc, const int next, auto halfc) {
    constexpr int T0 = decltype(t0c)::value, T1 = decltype(t1c)::value, NR = decltype(nrc)::value;
    constexpr int HALF = decltype(halfc)::value;                               // k-slice of the chunk: octets 4 HALF ..
    constexpr int J0 = (HALF * NUT) / NH, J1 = ((HALF + 1) * NUT) / NH;        // staging units loaded beside it
    constexpr int NT = (T1 - T0) * NR;
    constexpr int NST = TAPS == 1 ? (TRACE_BF3_NST < 3 ? TRACE_BF3_NST : 3) : TRACE_BF3_NST, DIST = NST - 1;   // A register stages / prefetch distance in tiles
    bf16x8 av[NST][NPL];
    auto load_a = [&](int n, int stage) {
      const int tap = T0 + n / NR, rt = n % NR;
      const char* p = ldsb + tapoff[tap] + rt * PPT * IS * 16 + HALF * 4 * SUB;
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) av[stage][pl] = *reinterpret_cast<const bf16x8*>(p + pl * PLANE);
    };
#pragma unroll
    for (int n = 0; n < DIST && n < NT; ++n) load_a(n, n);
    bf16x8 b[NPL];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
#ifdef TRACE_STAMP_SLICES
      if (n % NR == 0) {
        if (stamp_ch == 1) TRACE_SLICE_STAMP(HALF * (T1 - T0) + n / NR);
        else if (stamp_ch == 2) TRACE_SLICE_STAMP(16 + HALF * (T1 - T0) + n / NR);
      }
      if (stamp_ch == 1 && HALF == 0 && n / NR == 2) TRACE_SLICE_STAMP(32 + n % NR);
#endif
      if (n % NR == 0) {
#pragma unroll
        for (int p = 0; p < NPL; ++p) b[p] = bn[0][p];
#pragma unroll
        for (int i = 0; i + 1 < WPFN; ++i) {
#pragma unroll
          for (int p = 0; p < NPL; ++p) bn[i][p] = bn[i + 1][p];
        }
        ++sdone;
        wp += WSL;
#ifndef TRACE_EXP_NOWLOAD
        if (sdone + WPFN - 1 < a.nsl) {
#pragma unroll
          for (int p = 0; p < NPL; ++p) bn[WPFN - 1][p] = wp[(WPFN - 1) * WSL + 64 * p];
        }
#endif
      }
#ifdef TRACE_EXP_NOLDSREAD         // timing-only build: the A fragments of the first tiles are reused
      if (n + DIST < NT && n + DIST < NST) load_a(n + DIST, (n + DIST) % NST);
#else
      if (n + DIST < NT) load_a(n + DIST, (n + DIST) % NST);
#endif
#ifndef TRACE_BF3_PIN
#define TRACE_BF3_PIN 1           // pin the LDS reads of tile n + DIST ahead of tile n's MFMAs (the scheduler sinks them otherwise)
#endif
      if constexpr (TRACE_BF3_PIN) __builtin_amdgcn_sched_barrier(0);
#ifndef TRACE_BF3_LOAD_SPREAD
#define TRACE_BF3_LOAD_SPREAD 0   // row tiles between the global loads of two staging units of the next chunk (0: all at once)
#endif
      if constexpr (TRACE_BF3_LOAD_SPREAD == 0) {
#ifdef TRACE_EXP_NOSTAGELOAD      // timing-only build: the next chunk's loads are never issued (its data is garbage)
        if (false) {
#else
        if (n == 1) {
#endif
#pragma unroll
          for (int j = J0; j < J1; ++j) stage_load_unit(next, j);
        }
      } else {
        constexpr int SP = (NT - 2) / NUT < TRACE_BF3_LOAD_SPREAD ? ((NT - 2) / NUT > 0 ? (NT - 2) / NUT : 1) : TRACE_BF3_LOAD_SPREAD;
        if (next >= 0 && n >= 1 && (n - 1) % SP == 0 && (n - 1) / SP < NUT) stage_load_unit(next, (n - 1) / SP);
        if (next >= 0 && n == NT - 1) {
#pragma unroll
          for (int j = 0; j < NUT; ++j)
            if (1 + j * SP > NT - 1) stage_load_unit(next, j);
        }
      }
      if constexpr (SPLITK) {
        // the loads issued at n == 1 have landed by the last tiles of the chunk: split one staging unit per
        // tile there, in the shadow of the MFMAs, so that the chunk switch is only barrier - write - barrier
        constexpr int FIRST = NT - NUT > 2 ? NT - NUT : 2;
        if (n >= FIRST && n - FIRST < NUT) split_unit(n - FIRST);
      }
      const int s = n % NST, rt = n % NR;
      f32x4 c = acc[rt];
      if constexpr (SCH == 1) {
        const f16x8 ah = __builtin_bit_cast(f16x8, av[s][0]), al = __builtin_bit_cast(f16x8, av[s][1]);
        const f16x8 bh = __builtin_bit_cast(f16x8, b[0]), bl = __builtin_bit_cast(f16x8, b[1]);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
      } else {
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[s][2], b[0], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[s][0], b[2], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[s][1], b[1], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[s][1], b[0], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[s][0], b[1], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[s][0], b[0], c, 0, 0, 0);
      }
      acc[rt] = c;
      if constexpr (TRACE_BF3_PIN) __builtin_amdgcn_sched_barrier(0);
    }
  };

#ifndef TRACE_BF3_PRIO
#define TRACE_BF3_PRIO 0          // 1: K-loop waves at issue priority 3; 2: 3 / 1 by bit 8 of the workgroup id
#endif
  if constexpr (TRACE_BF3_PRIO == 1) {
    __builtin_amdgcn_s_setprio(3);
  } else if constexpr (TRACE_BF3_PRIO == 2) {
    // workgroups i and i + 256 share a CU (scripts/probe/wg_map.hip): unequal priorities take the two
    // out of lock-step, so that one's K loop runs beside the other's epilogue / prologue
    if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(3);
  }
  TRACE_STAMP(14);
  for (int ch = 0; ch < nch; ++ch) {
    __syncthreads();            // previous chunk fully consumed (and zero fill done)
#ifdef TRACE_STAMP_SWITCH       // diagnostic: stamps 4 / 5 = chunk 1's switch (all waves arrived / new image ready)
    if (ch == 1) TRACE_STAMP(4);
#endif
    {
#ifdef TRACE_EXP_NOSTAGE
      if (ch == 0)
#endif
      {
      if (!SPLITK || ch == 0) {
#pragma unroll
        for (int j = 0; j < NUT; ++j) split_unit(j);
      }
      stage_write();
      }
    }
    if (ch == 0) TRACE_STAMP(15);
    __syncthreads();
    if (ch == 0) TRACE_STAMP(1);
#ifdef TRACE_STAMP_SWITCH
    if (ch == 1) TRACE_STAMP(5);
#endif
    // with the in-K split the body is branch-free: the last chunk re-stages itself (never written)
#ifdef TRACE_EXP_NOSTAGE
    const int next = -1;
#else
    const int next = ch + 1 < nch ? ch + 1 : (SPLITK ? ch : -1);
#endif
#ifdef TRACE_STAMP_SLICES
    stamp_ch = ch;
#endif
    using IC0 = std::integral_constant<int, 0>;
    using ICT = std::integral_constant<int, TAPS>;
    using ICP = std::integral_constant<int, PADL>;
    using ICP1 = std::integral_constant<int, PADL + 1>;
    {
      using NRC = std::integral_constant<int, NACC>;
      using NRC2 = std::integral_constant<int, NREM != 0 ? NACC - 1 : NACC>;
      using IC1 = std::integral_constant<int, 1>;
      if (NREM == 0 || rpart < NREM) {                     // wave-uniform
        if (ch < nch1) {
          chunk_mma(IC0{}, ICT{}, NRC{}, next, IC0{});
          if constexpr (NH == 2) chunk_mma(IC0{}, ICT{}, NRC{}, next, IC1{});
        } else {
          chunk_mma(ICP{}, ICP1{}, NRC{}, next, IC0{});
          if constexpr (NH == 2) chunk_mma(ICP{}, ICP1{}, NRC{}, next, IC1{});
        }
      } else if constexpr (NREM != 0) {
        if (ch < nch1) chunk_mma(IC0{}, ICT{}, NRC2{}, next, IC0{});
        else chunk_mma(ICP{}, ICP1{}, NRC2{}, next, IC0{});
      }
    }
#ifdef TRACE_STAMP_SLICES
    if (ch == 1) TRACE_SLICE_STAMP(10);
    else if (ch == 2) TRACE_SLICE_STAMP(26);
#endif
    // stamps 2, 3: first two chunks; 4, 5: last two (scripts/stamps.py takes the last as the K end)
    if (ch < 2) TRACE_STAMP(2 + ch);
#ifndef TRACE_STAMP_SWITCH
    else if (ch == nch - 2) TRACE_STAMP(4);
    else if (ch == nch - 1) TRACE_STAMP(5);
#endif
  }

  if constexpr (TRACE_BF3_PRIO != 0) __builtin_amdgcn_s_setprio(0);
#ifdef TRACE_EXP_NOEPI
  if (a.rows >= 0) {        // timing-only build: the accumulators stay live, the epilogue never runs
    if (a.rows == 123456789) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < NACC; ++i) t += (acc[i][0] + acc[i][1]) + (acc[i][2] + acc[i][3]);
      a.out[tid] = t;
    }
    return;
  }
#endif
  conv_epilogue<L_OUT, TB, 1, NW, NACC>(a, acc, lds, traj0, co0, rt0, nrtw, oscale);
}

// workgroup id -> (row block, column tile): the ncol column tiles of one row block get ids of the same residue
// mod 8, 8 apart (same XCD, same time: scripts/probe/wg_map.hip measured id -> XCD id % 8, CU (id / 8) % 32)
template <int L_OUT, int UP, int TAPS, int PADL, int TB, int SCH, int NW = TRACE_BF3_NW,
          int MINW = TRACE_BF3_WGS * NW / 4, int CHN = 32, int IS = 1, int WPF = TRACE_BF3_WPF>
__global__ __launch_bounds__(NW * 64, MINW) void conv_bf3_kernel(ConvArgs a) {
  const int ncol = a.cout >> 6;
  const int bid = blockIdx.x;
  conv_bf3_body<L_OUT, UP, TAPS, PADL, TB, SCH, NW, MINW, CHN, IS, WPF>(
      a, (bid / (8 * ncol)) * 8 + (bid & 7), (bid >> 3) % ncol);
}

// A residual block's first k5 conv (a) and its 1x1 residual conv (b) read the same input: one launch, the column
// tiles of b behind those of a in every row block, so the 1x1 workgroups run beside the k5 workgroups of their
// row block (same XCD: the input comes from L2) and fill matrix-pipe time the memory-bound 1x1 launch left idle.
template <int L_OUT, int TB, int CHN, int WPF = TRACE_BF3_WPF>
__global__ __launch_bounds__(256, 2) void conv_pair_kernel(ConvArgs a, ConvArgs b) {
  const int ncol_a = a.cout >> 6;
  const int ncol = ncol_a + (b.cout >> 6);
  const int bid = blockIdx.x;
  const int rb = (bid / (8 * ncol)) * 8 + (bid & 7);
  const int ct = (bid >> 3) % ncol;
  if (ct < ncol_a) conv_bf3_body<L_OUT, 1, 5, 2, TB, 1, 4, 2, CHN, 1, WPF>(a, rb, ct);
  else conv_bf3_body<L_OUT, 1, 1, 0, TB, 1, 4, 2, 32, 1, WPF>(b, rb, ct - ncol_a);
}

#ifndef TRACE_HALF_TILES
#define TRACE_HALF_TILES 1        // launches of at most SMALL_WGS full tiles run on half tiles (Bf3Geom)
#endif
#ifndef TRACE_HALF_WPF
#define TRACE_HALF_WPF 3          // weight slices in flight on half tiles (cfg1: 1 -> 74.6, 2 -> 66.2, 3 -> 65.7, 4 -> 65.8 ms per call)
#endif
#ifndef TRACE_SMALL_WGS
#define TRACE_SMALL_WGS 256       // full tiles of a launch up to which it runs on half tiles (256 = one workgroup per CU)
#endif
constexpr long SMALL_WGS = TRACE_SMALL_WGS;
template <int L_OUT, int TB, bool MAY_HALVE = true>
int launch_conv_pair(const ConvArgs& a, const ConvArgs& b, hipStream_t st) {
  if constexpr (MAY_HALVE && TRACE_HALF_TILES && TB >= 4) {
    if ((long)((a.rows + TB - 1) / TB) * ((a.cout + b.cout) / 64) <= SMALL_WGS)
      return launch_conv_pair<L_OUT, TB / 2, false>(a, b, st);
  }
  constexpr int CHN = 64;
  using GA = Bf3Geom<L_OUT, 1, 5, 2, TB, 4, CHN, 1>;
  using GB = Bf3Geom<L_OUT, 1, 1, 0, TB, 4, 32, 1>;
  if (a.rows <= 0) return TRACE_OK;
  TRACE_REQUIRE(a.rows == b.rows && a.in == b.in && a.in_coff == b.in_coff && a.in_tstride == b.in_tstride,
                "conv pair: the two convs must read the same input");
  TRACE_REQUIRE(a.wh && b.wh && a.in_amax && b.in_amax && a.cin2 == 0 && b.cin2 == 0 && a.cin % 64 == 0 &&
                    b.cin % 32 == 0 && a.cout % 64 == 0 && b.cout % 64 == 0 &&
                    a.nsl == 5 * a.cin / 32 && b.nsl == b.cin / 32 && !a.gnb && !b.gnb && !b.gn && !b.pc,
                "conv pair: bad geometry cin=%d/%d cout=%d/%d", a.cin, b.cin, a.cout, b.cout);
  auto ok128 = [&](const ConvArgs& c, const float* p, int tstride, int coff) {
    return p == nullptr || ((reinterpret_cast<uintptr_t>(p) & 15) == 0 && tstride % 4 == 0 && (coff * L_OUT) % 4 == 0 &&
                            (size_t)c.rows * tstride * sizeof(float) < 0xffffffffull);
  };
  TRACE_REQUIRE((reinterpret_cast<uintptr_t>(a.in) & 3) == 0 &&
                    (size_t)(a.rows + TB) * a.in_tstride * sizeof(float) < 0xffffffffull &&
                    ok128(a, a.out, a.out_tstride, a.out_coff) && ok128(b, b.out, b.out_tstride, b.out_coff) &&
                    ok128(a, a.res, a.res_tstride, a.res_coff) && ok128(a, a.ysave, a.cout * L_OUT, 0) && !b.res &&
                    !b.ysave && !a.res2 && !b.res2 && (!a.pc || a.pt),
                "conv pair: tensors must be 16-byte aligned per trajectory tile and below 4 GiB");
  constexpr size_t la = (size_t)2 * GA::PLANE > GA::EPI_BYTES ? (size_t)2 * GA::PLANE : GA::EPI_BYTES;
  constexpr size_t lb = (size_t)2 * GB::PLANE > GB::EPI_BYTES ? (size_t)2 * GB::PLANE : GB::EPI_BYTES;
  constexpr size_t lds_bytes = la > lb ? la : lb;
  static_assert(lds_bytes <= 80 * 1024, "conv pair: LDS per workgroup too large for two per CU");
  static unsigned long long attr_done = 0;
  TRACE_CHECK_HIP(ensure_dynamic_lds(&attr_done, reinterpret_cast<const void*>(&conv_pair_kernel<L_OUT, TB, CHN, (MAY_HALVE ? TRACE_BF3_WPF : TRACE_HALF_WPF)>), 100 * 1024));
  const int nrb = (a.rows + TB - 1) / TB;
  dim3 grid(((nrb + 7) / 8) * 8 * ((a.cout + b.cout) / 64));
  const bool prof = prof_active();
  if (prof) prof_pre(st);
  hipLaunchKernelGGL((conv_pair_kernel<L_OUT, TB, CHN, (MAY_HALVE ? TRACE_BF3_WPF : TRACE_HALF_WPF)>), grid, dim3(256), lds_bytes, st, a, b);
  if (prof) {
    const double fl = 2.0 * (a.alg_macs_per_pos + b.alg_macs_per_pos) * (double)a.rows * L_OUT;
    prof_post(st, fl, fl * 3.0 / PEAK_MFMA16_FLOPS);
  }
  return launch_status("conv_pair_kernel");
}

template <int L_OUT, int UP, int TAPS, int PADL, int TB, int SCH = 0, int NW = TRACE_BF3_NW,
          int MINW = TRACE_BF3_WGS * NW / 4, int CHN = 32, int IS = 1, int WPF = TRACE_BF3_WPF>
int launch_conv_bf3_impl(const ConvArgs& a_in, hipStream_t st) {
  using G = Bf3Geom<L_OUT, UP, TAPS, PADL, TB, NW, CHN, IS>;
  static_assert(CHN == 32 || NW == 4, "64-channel chunks: 4-wave workgroups");
  ConvArgs a = a_in;
  if (a.rows <= 0) return TRACE_OK;
  TRACE_REQUIRE(SCH == 0 || (a.wh && a.in_amax && (a.cin2 == 0 || a.in2_amax) && a.wexp >= -10 && a.wexp <= 40),
                "conv(fp16x2): weight stream, weight exponent or input amax missing");
  TRACE_REQUIRE((SCH == 1 || a.wb) && a.cout % 64 == 0 && a.cin % CHN == 0 && a.cin2 % CHN == 0 &&
                    a.nsl == TAPS * a.cin / 32 + a.cin2 / 32,
                "conv(bf16x3): bad channel geometry cin=%d cin2=%d cout=%d nsl=%d", a.cin, a.cin2, a.cout,
                a.nsl);
  TRACE_REQUIRE(a.cin2 == 0 || (a.in2 && UP == 1 && IS == 1 && a.in2_tstride % 4 == 0 &&
                                (size_t)(a.rows + TB) * a.in2_tstride * sizeof(float) < 0xffffffffull),
                "conv(bf16x3): second K source needs a stride-1 conv and a tensor below 4 GiB");
  TRACE_REQUIRE(!a.gnb || (!a.gn && !a.pc && !a.ysave && a.gb_y && a.gb_stats && a.gb_gamma && a.gb_beta &&
                           a.gb_cg > 0 && (a.gb_cg & (a.gb_cg - 1)) == 0 && a.gb_cg <= 64 &&
                           a.gb_cmax % 64 == 0 && a.gb_cmax <= a.gb_c &&
                           (reinterpret_cast<uintptr_t>(a.gb_y) & 15) == 0),
                "conv(bf16x3): bad GroupNorm-backward epilogue arguments");
  TRACE_REQUIRE(!a.gn || (a.cg > 0 && (a.cg & (a.cg - 1)) == 0 && a.cg <= 64), "conv: group size %d",
                a.cg);
  TRACE_REQUIRE((reinterpret_cast<uintptr_t>(a.in) & 3) == 0 &&
                    (size_t)(a.rows + TB) * a.in_tstride * sizeof(float) < 0xffffffffull,
                "conv(bf16x3): input of %d rows exceeds the 4 GiB buffer-descriptor range; split the rows",
                a.rows);
  auto ok128 = [&](const float* p, int tstride, int coff) {
    return p == nullptr || ((reinterpret_cast<uintptr_t>(p) & 15) == 0 && tstride % 4 == 0 &&
                            (coff * L_OUT) % 4 == 0 &&
                            (size_t)a.rows * tstride * sizeof(float) < 0xffffffffull);
  };
  TRACE_REQUIRE(ok128(a.out, a.out_tstride, a.out_coff) && ok128(a.res, a.res_tstride, a.res_coff) &&
                    ok128(a.res2, a.res2_tstride, a.res2_coff) && ok128(a.ysave, a.cout * L_OUT, 0) &&
                    ok128(a.out_raw, a.raw_tstride, a.raw_coff),
                "conv(bf16x3): output / residual tensors must be 16-byte aligned per trajectory tile and "
                "below 4 GiB (%d rows)", a.rows);
  TRACE_REQUIRE(!a.pc || a.pt, "conv: cond table without step table");
  constexpr size_t LDS_NEED = (size_t)(SCH == 1 ? 2 : 3) * G::PLANE > G::EPI_BYTES ? (size_t)(SCH == 1 ? 2 : 3) * G::PLANE
                                                                                   : G::EPI_BYTES;
  // two workgroups per CU (the stride-2 bf16-triple fallback, 83 KB, runs one per CU)
  static_assert(LDS_NEED <= (IS == 2 && SCH == 0 ? 100 : 80) * 1024, "conv(bf16x3): LDS per workgroup too large");
  constexpr size_t lds_bytes = LDS_NEED;
  static unsigned long long attr_done = 0;      // one bit per device
  TRACE_CHECK_HIP(ensure_dynamic_lds(
      &attr_done, reinterpret_cast<const void*>(&conv_bf3_kernel<L_OUT, UP, TAPS, PADL, TB, SCH, NW, MINW, CHN, IS, WPF>),
      100 * 1024));
  const int nrb = (a.rows + TB - 1) / TB;
  dim3 grid(((nrb + 7) / 8) * 8 * (a.cout / 64));
#ifdef TRACE_CONV_STAMPS
  a.dbg = stamp_buffer_for_next_launch();
#endif
  const bool prof = prof_active();
  if (prof) prof_pre(st);
  hipLaunchKernelGGL((conv_bf3_kernel<L_OUT, UP, TAPS, PADL, TB, SCH, NW, MINW, CHN, IS, WPF>), grid, dim3(G::NTH), lds_bytes, st, a);
  if (prof) {
    const double fl = 2.0 * a.alg_macs_per_pos * (double)a.rows * L_OUT;
    prof_post(st, fl, fl * (SCH == 1 ? 3.0 : 6.0) / PEAK_MFMA16_FLOPS);
  }
  return launch_status(SCH == 1 ? "conv_bf3_kernel(fp16x2)" : "conv_bf3_kernel");
}

// Which K loop a launch takes.  Every operand with its scale (weights: wh / wexp, activations: the amax their
// producer recorded) -> the fp16-pair scheme, 64-channel chunks where the channel counts allow; otherwise the
// bf16 triple.  trace_conv_ws_mode(1) sends the full-chip multi-tap bf16-triple launches to the persistent
// wave-specialised kernel (conv_ws_kernels.h; same bits, tests/test_gpu_full_size.py).
template <int L_OUT, int UP, int TAPS, int PADL, int TB>
int launch_conv_ws(const ConvArgs& a_in, hipStream_t st);      // conv_ws_kernels.h
int conv_ws_mode(int mode);                                    // unet.hip (trace_conv_ws_mode)
int conv_scheme(int mode);                                     // unet.hip (trace_conv_scheme): 0 bf16 x 3, 1 fp16 x 2 where possible

// stride-2 convs (32-channel chunks: a chunk already stages 2 L_OUT positions per trajectory)
template <int L_OUT, int TAPS, int PADL, int TB, bool MAY_HALVE = true>
int launch_conv_bf3_s2(const ConvArgs& a, hipStream_t st) {
  if constexpr (MAY_HALVE && TRACE_HALF_TILES && TB >= 4) {
    if ((long)((a.rows + TB - 1) / TB) * (a.cout / 64) <= SMALL_WGS)
      return launch_conv_bf3_s2<L_OUT, TAPS, PADL, TB / 2, false>(a, st);
  }
  constexpr int MW = TRACE_BF3_WGS * TRACE_BF3_NW / 4;
  if (a.wh && a.in_amax)
    return launch_conv_bf3_impl<L_OUT, 1, TAPS, PADL, TB, 1, TRACE_BF3_NW, MW, 32, 2>(a, st);
  return launch_conv_bf3_impl<L_OUT, 1, TAPS, PADL, TB, 0, TRACE_BF3_NW, MW, 32, 2>(a, st);
}

template <int L_OUT, int UP, int TAPS, int PADL, int TB, bool MAY_HALVE = true>
int launch_conv_bf3(const ConvArgs& a, hipStream_t st) {
  constexpr int MW = TRACE_BF3_WGS * TRACE_BF3_NW / 4;
  const long wgs = (long)((a.rows + TB - 1) / TB) * (a.cout / 64);
  if constexpr (MAY_HALVE && TRACE_HALF_TILES && TB >= 4) {
    if (wgs <= SMALL_WGS) return launch_conv_bf3<L_OUT, UP, TAPS, PADL, TB / 2, false>(a, st);
  }
  if (a.wh && a.in_amax && (a.cin2 == 0 || a.in2_amax)) {          // every operand has its scale: the fp16 pipe
#ifndef TRACE_F16_CH1
#define TRACE_F16_CH1 64          // ... and of the 1x1 launches (32: 10.09 ms per pass, 64: 10.02; same bits)
#endif
#ifndef TRACE_F16_CH
#define TRACE_F16_CH 64           // channels per chunk of the fp16 launches with more than one tap (32 / 64; build switch)
#endif
    if constexpr ((TAPS > 1 || TRACE_F16_CH1 == 64) && TRACE_F16_CH == 64) {
      if (a.cin % 64 == 0 && a.cin2 % 64 == 0)
        return launch_conv_bf3_impl<L_OUT, UP, TAPS, PADL, TB, 1, TRACE_BF3_NW, MW, 64, 1, (MAY_HALVE ? TRACE_BF3_WPF : TRACE_HALF_WPF)>(a, st);
    }
    return launch_conv_bf3_impl<L_OUT, UP, TAPS, PADL, TB, 1, TRACE_BF3_NW, MW, 32, 1, (MAY_HALVE ? TRACE_BF3_WPF : TRACE_HALF_WPF)>(a, st);
  }
  if constexpr (TAPS > 1) {
    if constexpr (MAY_HALVE) {             // full tiles only
      if (conv_ws_mode(-1) && !conv_scheme(-1) && wgs > 256 && (a.cin / 32 + a.cin2 / 32) % 2 == 0 && !a.head_pred)
        return launch_conv_ws<L_OUT, UP, TAPS, PADL, TB>(a, st);
    }
  }
  return launch_conv_bf3_impl<L_OUT, UP, TAPS, PADL, TB, 0>(a, st);
}

}  // namespace trace
