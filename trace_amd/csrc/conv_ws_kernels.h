// conv_ws_kernel: the split-bf16 implicit-GEMM Conv1d of conv_bf3_kernels.h as a PERSISTENT, WAVE-SPECIALISED
// kernel (round-2 experiment, opt-in with TRACE_CONV_WS=1; bit-identical to conv_bf3_kernel, measured at parity with
// it -- DESIGN.md section 5.1 has the numbers and what they say about the bound).  One workgroup of 8 waves per CU
// walks a list of (row block, column tile) tiles;
//   waves 0..3 ("K waves", one per SIMD) do nothing but the MFMA stream: A fragments from the staged LDS image,
//              weights L2 -> VGPR, 13 accumulator tiles;  at the end of a tile they drop the accumulators into
//              an LDS output tile and go straight on to the next tile's K loop;
//   waves 4..7 ("helper waves", the other wave of each SIMD) stage the activations two chunks ahead (global ->
//              VGPR, two register sets -> exact 3-way bf16 split -> LDS image, double-buffered) and run the fused
//              epilogue of the PREVIOUS tile (GroupNorm + Mish forward or backward, condition / residual adds,
//              128-bit stores) out of the LDS output tile, one piece per half chunk.
// Synchronisation is the workgroup barrier only, on a static schedule: every chunk has a MID and an END barrier
// that all 8 waves pass.  K waves: MFMAs of the first taps | mid | MFMAs of the rest | end.  Helper waves: a piece
// without stores, the loads of chunk + 2 | mid | split + write of chunk + 1, a piece with stores | end.
// LDS: two images (2 x 3 planes) + the 208 x 64 fp32 output tile = 150..158 KB: one workgroup per CU.
// Results are bit-identical to conv_bf3_kernel (same split, same MFMA order per accumulator, same epilogue
// arithmetic in the same order).
#pragma once
#include <type_traits>

#include "conv_bf3_kernels.h"

namespace trace {

#ifndef TRACE_WS_WPF
#define TRACE_WS_WPF 2            // weight slices in flight ahead of the one being multiplied (K waves)
#endif
#ifndef TRACE_WS_NST
#define TRACE_WS_NST 4            // A-fragment register stages of the K waves (prefetch distance + 1)
#endif

template <int L_OUT, int UP, int TAPS, int PADL, int TB>
struct WsGeom {
  using B = Bf3Geom<L_OUT, UP, TAPS, PADL, TB, 4>;
  static constexpr int IMG = 3 * B::PLANE;                             // one staged chunk
  static constexpr size_t LDS_BYTES = (size_t)2 * IMG + (size_t)B::R * 64 * sizeof(float);
  static_assert(LDS_BYTES <= 160 * 1024, "conv(ws): LDS over 160 KiB");
  static_assert(2 * IMG < 65536 * 2, "image offsets");
};

// GNB: the GroupNorm + Mish BACKWARD epilogue (ConvArgs::gnb), a compile-time choice here: the forward epilogue
// keeps the residual operands in registers to the end, the backward one only up to its first piece, and the
// register allocator has to see that
template <int L_OUT, int UP, int TAPS, int PADL, int TB, bool GNB>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_ws_kernel(ConvArgs a, int ntv) {
#pragma clang fp contract(off)      // explicit FMAs only, as conv_epilogue (conv_kernels.h): the two must agree bit for bit
  using G = Bf3Geom<L_OUT, UP, TAPS, PADL, TB, 4>;
  constexpr int LP = G::LP, L_IN = G::L_IN, SUB = G::SUB, PLANE = G::PLANE;
  constexpr int NUT = G::NUT, NU = G::NU, NPOS = G::NPOS, CH = G::CH, PPT = G::PPT;
  constexpr int NRT = G::NRT, R = G::R;
  constexpr int IMG = 3 * PLANE;
  constexpr int TILE = 64 * L_OUT;
  constexpr int NQ = R * 64 / 4 / 256;
  static_assert((R * 64 / 4) % 256 == 0 && NRT == 13, "output tile must split evenly over the 256 E threads");
  typedef unsigned u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const ldsb = reinterpret_cast<char*>(lds);
  float* const tile = reinterpret_cast<float*>(ldsb + 2 * IMG);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int ncol = a.cout >> 6;
  const int nch1 = a.cin / CH;
  const int nch = nch1 + a.cin2 / CH;
  // tiles of this workgroup: virtual ids t = blockIdx.x + it * gridDim.x, decoded as in conv_bf3_kernel
  // (row block (t / (8 ncol)) * 8 + (t & 7), column tile (t >> 3) % ncol).  gridDim.x is a multiple of 8 ncol,
  // so the column tile -- and with it the weight stream and every per-channel parameter -- is the same for
  // all tiles of a workgroup.
  const int G8 = 8 * ncol;
  const int ct = ((int)blockIdx.x >> 3) % ncol;
  const int co0 = ct * 64;
  auto traj_of = [&](int it) __attribute__((always_inline)) -> int {
    const int t = (int)blockIdx.x + it * (int)gridDim.x;
    return ((t / G8) * 8 + (t & 7)) * TB;
  };
  int nt = 0;
  while ((int)blockIdx.x + nt * (int)gridDim.x < ntv && traj_of(nt) < a.rows) ++nt;
  if (nt == 0) return;                       // workgroup-uniform, before any barrier

  using IC0 = std::integral_constant<int, 0>;
  using ICT = std::integral_constant<int, TAPS>;
  using ICP = std::integral_constant<int, PADL>;
  using ICP1 = std::integral_constant<int, PADL + 1>;

  if (wave < 4) {
    // =========================================================== K waves ==============================
    const int cwave = wave, col = lane & 15, q = lane >> 4;
    const int co_l = cwave * 16 + col;
    const float bv = a.bias ? a.bias[co0 + co_l] : 0.f;
    float* const tcol = tile + co_l * L_OUT;
    const int rowoff0 = ((col / PPT) * LP + col % PPT) * 16 + q * SUB;
    int tapoff[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) tapoff[t] = rowoff0 + t * 16;
    f32x4 acc[NRT];

    const bf16x8* const wp0 = reinterpret_cast<const bf16x8*>(a.wb) + ((size_t)(ct * 4 + cwave) * a.nsl) * 192 + lane;
    constexpr int WPF = TRACE_WS_WPF;
    bf16x8 wq[WPF][3];
    int snext = 0;                           // next slice to request; the stream wraps from tile to tile
    auto wfetch = [&](bf16x8 (&dst)[3]) __attribute__((always_inline)) {
      const bf16x8* p = wp0 + (size_t)snext * 192;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) dst[pl] = p[64 * pl];
      snext = (snext + 1 == a.nsl) ? 0 : snext + 1;
    };
#pragma unroll
    for (int i = 0; i < WPF; ++i) wfetch(wq[i]);

#ifdef TRACE_CONV_STAMPS
    unsigned long long kt_mid = 0, kt_end = 0, kt_tile = 0;
    const unsigned long long kt_start = __builtin_amdgcn_s_memtime();
#define WS_KBAR(accum)                                                     \
  do {                                                                     \
    const unsigned long long _t0 = __builtin_amdgcn_s_memtime();          \
    __builtin_amdgcn_s_barrier();                                          \
    accum += __builtin_amdgcn_s_memtime() - _t0;                           \
  } while (0)
#else
#define WS_KBAR(accum) __builtin_amdgcn_s_barrier()
#endif
    // one chunk's MFMAs over taps [T0, T1) from the image at byte offset imgoff; the MID barrier sits after
    // the first half of the taps (after all of them when there is only one)
    auto k_chunk = [&](auto t0c, auto t1c, const int imgoff) __attribute__((always_inline)) {
      constexpr int T0 = decltype(t0c)::value, T1 = decltype(t1c)::value;
      constexpr int NT = (T1 - T0) * NRT;
      constexpr int NMID = ((T1 - T0) / 2 > 0 ? (T1 - T0) / 2 : 1) * NRT;    // odd phases (second half) are the helpers' busy ones
      constexpr int NST = TRACE_WS_NST, DIST = NST - 1;
      int tbase[TAPS];
#pragma unroll
      for (int t = T0; t < T1; ++t) {
        tbase[t] = tapoff[t] + imgoff;
        asm volatile("" : "+v"(tbase[t]));   // opaque: overlapping windows of neighbouring taps must not be merged
      }
      bf16x8 av[NST][3];
      auto load_a = [&](int n, int stage) __attribute__((always_inline)) {
        const int tap = T0 + n / NRT, rt = n % NRT;
        const char* p = ldsb + tbase[tap] + rt * PPT * 16;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) av[stage][pl] = *reinterpret_cast<const bf16x8*>(p + pl * PLANE);
      };
#pragma unroll
      for (int n = 0; n < DIST && n < NT; ++n) load_a(n, n);
      bf16x8 b[3];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        if (n % NRT == 0) {
#pragma unroll
          for (int p = 0; p < 3; ++p) b[p] = wq[0][p];
#pragma unroll
          for (int i = 0; i + 1 < WPF; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) wq[i][p] = wq[i + 1][p];
          wfetch(wq[WPF - 1]);
        }
        if (n + DIST < NT) load_a(n + DIST, (n + DIST) % NST);
        __builtin_amdgcn_sched_barrier(0);
        const int s = n % NST, rt = n % NRT;
        f32x4 c = acc[rt];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[s][2], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[s][0], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[s][1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[s][1], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[s][0], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[s][0], b[0], c, 0, 0, 0);
        acc[rt] = c;
        __builtin_amdgcn_sched_barrier(0);
        if (n + 1 == NMID) WS_KBAR(kt_mid);                        // MID (exactly once: 0 < NMID <= NT)
      }
    };

    __builtin_amdgcn_s_barrier();            // P1: images zero-filled
    __builtin_amdgcn_s_barrier();            // P2: chunk 0 of the first tile staged
#ifdef TRACE_CONV_STAMPS
    const unsigned long long kt_loop = __builtin_amdgcn_s_memtime();
#endif
#ifdef TRACE_WS_PRIO
    __builtin_amdgcn_s_setprio(TRACE_WS_PRIO);
#endif
    int g = 0;
    for (int it = 0; it <= nt; ++it) {
      if (it > 0) {
#ifdef TRACE_CONV_STAMPS
        const unsigned long long _tt = __builtin_amdgcn_s_memtime();
#endif
        // the finished accumulators (+ conv bias) -> output tile, laid out like the global tensor
        // ([trajectory][channel][position]); the E waves are done with the tile before that (their last piece
        // of the tile before ran at least one barrier ago)
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
          f32x4 v = acc[rt];
          v[0] += bv; v[1] += bv; v[2] += bv; v[3] += bv;
          if constexpr (PPT == 4) {
            *reinterpret_cast<f32x4*>(tcol + q * TILE + 4 * rt) = v;
          } else if constexpr (PPT == 2) {
            *reinterpret_cast<float2*>(tcol + (2 * q) * TILE + 2 * rt) = float2{v[0], v[1]};
            *reinterpret_cast<float2*>(tcol + (2 * q + 1) * TILE + 2 * rt) = float2{v[2], v[3]};
          } else {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) tcol[(4 * q + reg) * TILE + rt] = v[reg];
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef TRACE_CONV_STAMPS
        kt_tile += __builtin_amdgcn_s_memtime() - _tt;
#endif
      }
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const bool cur = it < nt;
      for (int c = 0; c < nch; ++c, ++g) {
        const int imgoff = (g & 1) * IMG;
        if (cur) {
          if (c < nch1) k_chunk(IC0{}, ICT{}, imgoff);
          else k_chunk(ICP{}, ICP1{}, imgoff);
        } else {
          __builtin_amdgcn_s_barrier();      // MID of the drain iteration
        }
        if (cur) WS_KBAR(kt_end); else __builtin_amdgcn_s_barrier();        // END
      }
    }
#ifdef TRACE_CONV_STAMPS
    if (a.dbg && tid == 0) {
      unsigned long long* d = a.dbg + (size_t)blockIdx.x * 16;
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      d[0] = kt_start; d[1] = kt_loop - kt_start; d[2] = now - kt_loop; d[3] = kt_mid; d[4] = kt_end; d[5] = kt_tile;
      d[6] = (unsigned long long)nt; d[7] = (unsigned long long)nch;
    }
#endif
    return;
  }

  // ======================================================= H waves: staging + epilogue ================
  const int etid = tid - 256;
  const int ewave = wave - 4;
  // ---- staging (the register path of conv_bf3_kernel on 256 threads) ---------------------------------
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in), 0, (unsigned)((size_t)a.rows * a.in_tstride * sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in2 ? a.in2 : a.in), 0,
      a.in2 ? (unsigned)((size_t)a.rows * a.in2_tstride * sizeof(float)) : 0u, 0x00020000);
  unsigned spk[NUT];            // (trajectory in tile) << 20 | float offset inside the chunk
  unsigned wofs[NUT];           // LDS byte offset of the unit inside a plane
#pragma unroll
  for (int j = 0; j < NUT; ++j) {
    const int qq = etid + 256 * j;
    const int o = qq / NPOS;
    const int pp = qq - o * NPOS;
    const int tr = pp / L_IN;
    const int i = pp - tr * L_IN;
    spk[j] = ((unsigned)tr << 20) | (unsigned)(o * 8 * L_IN + i);
    wofs[j] = (unsigned)((tr * LP + i * UP + PADL) * 16 + o * SUB);
  }
  // two register sets: the loads of chunk g + 2 go out while chunk g multiplies and are split + written while
  // chunk g + 1 does, so a load has a whole chunk (~3 us) to arrive and the split's vmcnt wait never sits right
  // behind the epilogue's stores
  float stgA[NUT][8], stgB[NUT][8];
  auto stage_load = [&](float (&stg)[NUT][8], const int traj0s, const int ch) __attribute__((always_inline)) {
    const bool second = ch >= nch1;
    const int tstride = second ? a.in2_tstride : a.in_tstride;
    const int cbase = ((second ? a.in2_coff + (ch - nch1) * CH : a.in_coff + ch * CH) * L_IN) + traj0s * tstride;
#pragma unroll
    for (int j = 0; j < NUT; ++j) {
      // units past NU (last pass) read a valid address and are never written to LDS; trajectories past `rows`
      // fall outside the descriptor's range and read as zeros
      const bool live = (NU % 256 == 0) || (etid + 256 * j < NU);
      const unsigned off = live ? ((spk[j] >> 20) * (unsigned)tstride + (spk[j] & 0xfffffu) + (unsigned)cbase) * 4u : 0u;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        stg[j][jj] = second ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc2, off, jj * L_IN * 4, 0))
                            : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, jj * L_IN * 4, 0));
      }
    }
  };
  auto stage_write = [&](float (&stg)[NUT][8], const int imgoff) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NUT; ++j) {
      if ((NU % 256 == 0) || (etid + 256 * j < NU)) {
        u32x4v h, m, l;
#ifdef TRACE_WS_EXP_NOSPLIT          // timing-only: no conversion work
        h = u32x4v{__builtin_bit_cast(unsigned, stg[j][0]), __builtin_bit_cast(unsigned, stg[j][1]),
                   __builtin_bit_cast(unsigned, stg[j][2]), __builtin_bit_cast(unsigned, stg[j][3])};
        m = u32x4v{__builtin_bit_cast(unsigned, stg[j][4]), __builtin_bit_cast(unsigned, stg[j][5]),
                   __builtin_bit_cast(unsigned, stg[j][6]), __builtin_bit_cast(unsigned, stg[j][7])};
        l = h;
#else
        split8_bf16(stg[j], h, m, l);
#endif
        char* p = ldsb + imgoff + wofs[j];
        *reinterpret_cast<u32x4v*>(p) = h;
        *reinterpret_cast<u32x4v*>(p + PLANE) = m;
        *reinterpret_cast<u32x4v*>(p + 2 * PLANE) = l;
      }
    }
  };

  stage_load(stgA, traj_of(0), 0);
  if (nch > 1 || nt > 1) stage_load(stgB, nch > 1 ? traj_of(0) : traj_of(1), nch > 1 ? 1 : 0);
  // halo / zero-stuffing cells of both images are written once; data cells are rewritten by every chunk
  if constexpr (UP == 1 && TAPS == 1) {
  } else if constexpr (UP == 1) {
    constexpr int HALO = LP - L_IN;
    for (int i = etid; i < 2 * 12 * TB * HALO; i += 256) {
      const int arr = i / (TB * HALO);                     // (image, plane, octet) array
      const int r0 = i - arr * (TB * HALO);
      const int tr = r0 / HALO;
      const int h = r0 - tr * HALO;
      const int pos = tr * LP + (h < PADL ? h : L_IN + h);
      *reinterpret_cast<u32x4v*>(ldsb + arr * SUB + pos * 16) = u32x4v{0u, 0u, 0u, 0u};
    }
  } else {
#pragma unroll 2
    for (int i = etid; i < 2 * IMG / 16; i += 256)
      reinterpret_cast<u32x4v*>(ldsb)[i] = u32x4v{0u, 0u, 0u, 0u};
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();              // P1
  stage_write(stgA, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();              // P2
#ifdef TRACE_WS_EXP_NOSTORE          // timing-only: every epilogue store is range-checked away
#define WS_STORE_BYTES(x) 0u
#else
#define WS_STORE_BYTES(x) (x)
#endif
  // ---- epilogue pieces (conv_epilogue of conv_kernels.h on 256 threads, cut at its barriers) ---------
  constexpr int NP = TB / 4;                 // (trajectory, channel) rows per thread: trajectory ewave + 4 j
  const int c = lane, co = co0 + c;
  const bool has_tab = a.gn || a.pc;
  const bool normed = GNB && co0 < a.gb_cmax;
  float gam, bet, ptv = 0.f;
  int cg;
  if (GNB) {
    gam = a.gb_gamma[co]; bet = a.gb_beta[co]; cg = a.gb_cg;
  } else {
    gam = a.gn ? a.gamma[co] : 1.f; bet = a.gn ? a.beta[co] : 0.f; cg = a.gn ? a.cg : 1;
    ptv = a.pc ? a.pt[co] : 0.f;
  }
  const int cgs = __builtin_ctz(cg);
  // Flat walk over the 16-byte cells of the output tile: cell u = etid + 256 k is floats 4 u .. 4 u + 3 of the tile,
  // trajectory tr = 4 u / TILE, offset rem inside the trajectory's [channel][position] block.  The walker starts
  // from an opaque copy of etid inside every piece, so that the compiler does not hoist 13 offsets per tensor
  // out of the tile loop (they cost ~70 VGPRs and were spilled).
  struct FlatIx { int tr, rem; };
  auto flat_begin = [&]() __attribute__((always_inline)) -> FlatIx {
    int e = etid;
    asm volatile("" : "+v"(e));
    const int f = e * 4;
    FlatIx x;
    x.tr = f / TILE;
    x.rem = f - x.tr * TILE;
    return x;
  };
  auto flat_next = [&](FlatIx& x) __attribute__((always_inline)) {
    x.tr += 1024 / TILE;
    x.rem += 1024 % TILE;
    if (x.rem >= TILE) { x.rem -= TILE; ++x.tr; }
  };
  // unsigned 32-bit byte offset (the launcher guarantees rows * tstride * 4 < 4 GiB); trajectories past `rows`
  // fall outside every descriptor's range
  auto flat_off = [&](const FlatIx& x, int traj0p, int tstride, int coff) __attribute__((always_inline)) -> unsigned {
    const int tg = traj0p + x.tr;
    return tg < a.rows ? ((unsigned)tg * (unsigned)tstride + (unsigned)((coff + co0) * L_OUT + x.rem)) * 4u : 0xffffffffu;
  };
  f32x4 rv[NQ];
  float addv[NP];
  auto sel_add = [&](int j) __attribute__((always_inline)) -> float {
    float r = addv[0];
#pragma unroll
    for (int i = 1; i < NP; ++i) r = (j == i) ? addv[i] : r;
    return r;
  };

  // piece 0: residual operands and condition adds of the tile into registers (their latency hides here)
  auto piece_prefetch = [&](const int traj0p) __attribute__((always_inline)) {
    if (a.res) {
      const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(a.res), 0, (unsigned)((size_t)a.rows * a.res_tstride * sizeof(float)), 0x00020000);
      FlatIx x = flat_begin();
#pragma unroll
      for (int k = 0; k < NQ; ++k) {
        rv[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                              rr, flat_off(x, traj0p, a.res_tstride, a.res_coff), 0, 0));
        flat_next(x);
      }
    } else {
#pragma unroll
      for (int k = 0; k < NQ; ++k) rv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (a.res2) {
      const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(a.res2), 0, (unsigned)((size_t)a.rows * a.res2_tstride * sizeof(float)), 0x00020000);
      FlatIx x = flat_begin();
#pragma unroll
      for (int k = 0; k < NQ; ++k) {
        const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                      rr, flat_off(x, traj0p, a.res2_tstride, a.res2_coff), 0, 0));
        rv[k] += t;
        flat_next(x);
      }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int tg = min(traj0p + ewave + 4 * j, a.rows - 1);
      addv[j] = (!GNB && a.pc) ? a.pc[(size_t)(tg / a.rows_per_cond) * a.pc_stride + a.pc_off + co] + ptv : 0.f;
    }
  };

  // piece 1, forward: raw conv outputs for the backward (flat copy of the tile before it is normalised)
  auto piece_ysave = [&](const int traj0p) __attribute__((always_inline)) {
    if (!a.ysave) return;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        a.ysave, 0, WS_STORE_BYTES((unsigned)((size_t)a.rows * a.cout * L_OUT * sizeof(float))), 0x00020000);
    FlatIx x = flat_begin();
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      const f32x4 y = *reinterpret_cast<const f32x4*>(tile + (etid + 256 * k) * 4);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), ry, flat_off(x, traj0p, a.cout * L_OUT, 0), 0, 0);
      flat_next(x);
    }
  };
  // piece 1, backward (G1): v = tile + residuals; raw copy out; v back into the tile
  auto piece_g1 = [&](const int traj0p) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        a.out_raw ? a.out_raw : a.out, 0,
        a.out_raw ? WS_STORE_BYTES((unsigned)((size_t)a.rows * a.raw_tstride * sizeof(float))) : 0u, 0x00020000);
    FlatIx x = flat_begin();
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      float* cell = tile + (etid + 256 * k) * 4;
      f32x4 y = *reinterpret_cast<const f32x4*>(cell);
      y += rv[k];
      if (a.out_raw)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), rw, flat_off(x, traj0p, a.raw_tstride, a.raw_coff), 0, 0);
      *reinterpret_cast<f32x4*>(cell) = y;
      flat_next(x);
    }
  };

  auto read_row = [&](const float* row, float (&v)[L_OUT]) __attribute__((always_inline)) {
    if constexpr (L_OUT % 4 == 0) {
#pragma unroll
      for (int i = 0; i < L_OUT / 4; ++i) {
        const f32x4 t = reinterpret_cast<const f32x4*>(row)[i];
        v[4 * i] = t[0]; v[4 * i + 1] = t[1]; v[4 * i + 2] = t[2]; v[4 * i + 3] = t[3];
      }
    } else if constexpr (L_OUT % 2 == 0) {
#pragma unroll
      for (int i = 0; i < L_OUT / 2; ++i) {
        const float2 t = reinterpret_cast<const float2*>(row)[i];
        v[2 * i] = t.x; v[2 * i + 1] = t.y;
      }
    } else {
#pragma unroll
      for (int i = 0; i < L_OUT; ++i) v[i] = row[i];
    }
  };
  auto write_row = [&](float* row, const float (&v)[L_OUT]) __attribute__((always_inline)) {
    if constexpr (L_OUT % 4 == 0) {
#pragma unroll
      for (int i = 0; i < L_OUT / 4; ++i)
        reinterpret_cast<f32x4*>(row)[i] = f32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
    } else if constexpr (L_OUT % 2 == 0) {
#pragma unroll
      for (int i = 0; i < L_OUT / 2; ++i) reinterpret_cast<float2*>(row)[i] = float2{v[2 * i], v[2 * i + 1]};
    } else {
#pragma unroll
      for (int i = 0; i < L_OUT; ++i) row[i] = v[i];
    }
  };

  // piece 2.., forward (E1): one (trajectory, channel) row per thread; lanes = the 64 channels, so the channels
  // of a group are neighbouring lanes.  Exact two-level moments (row mean / centred sum of squares, Chan's
  // combination over the group), normalise, Mish, + condition, row back into the tile.
  auto piece_rows_fwd = [&](const int traj0p, const int j0, const int j1) __attribute__((always_inline)) {
    if (!has_tab) return;
#pragma unroll 1
    for (int j = j0; j < j1; ++j) {
      const int tr = ewave + 4 * j;
      float* row = tile + (tr * 64 + c) * L_OUT;
      float v[L_OUT];
      read_row(row, v);
      float scale = 1.f, shift = 0.f;
      if (a.gn) {
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < L_OUT; ++i) sacc += v[i];
        float rm = sacc * (1.f / (float)L_OUT);
        float qq = 0.f;
#pragma unroll
        for (int i = 0; i < L_OUT; ++i) {
          const float d = v[i] - rm;
          qq = fmaf(d, d, qq);
        }
        asm volatile("" : "+v"(rm), "+v"(qq));         // the guard of conv_epilogue E1
        float gs[1] = {rm};
        group_sum<1>(gs, cg);
        gs[0] = gs[0] / (float)cg;                     // group mean
        const float dm = rm - gs[0];
        float gv[1] = {fmaf(dm * dm, (float)L_OUT, qq)};
        group_sum<1>(gv, cg);
        const float mean = gs[0];
        const float var = gv[0] / (float)(cg * L_OUT);
        const float rstd = 1.f / sqrtf(var + 1e-5f);
        scale = rstd * gam;
        shift = bet - mean * scale;
#ifdef TRACE_WS_EXP_NOSTORE
        if (false) {
#else
        if (a.stats && (c & (cg - 1)) == 0 && traj0p + tr < a.rows) {
#endif
          const size_t si = ((size_t)(traj0p + tr) * (a.cout >> cgs) + (co >> cgs)) * 2;
          a.stats[si] = mean;
          a.stats[si + 1] = rstd;
        }
      }
      const float ad = sel_add(j);
#pragma unroll
      for (int i = 0; i < L_OUT; ++i) {
        float z = fmaf(v[i], scale, shift);
#ifndef TRACE_WS_EXP_NOMISH
        if (a.gn) z = mish_fast_f(z);
#endif
        v[i] = z + ad;
      }
      write_row(row, v);
    }
  };
  // piece 2.., backward (G2): GroupNorm + Mish backward on the rows (conv_epilogue G2)
  auto piece_rows_bwd = [&](const int traj0p, const int j0, const int j1) __attribute__((always_inline)) {
    if (!normed) return;
    const float inv_n = 1.f / (float)(cg * L_OUT);
#pragma unroll 1
    for (int j = j0; j < j1; ++j) {
      const int tr = ewave + 4 * j;
      const int tg = min(traj0p + tr, a.rows - 1);
      float* row = tile + (tr * 64 + c) * L_OUT;
      const float* yrow = a.gb_y + ((size_t)tg * a.gb_c + co) * L_OUT;
      const float* stp = a.gb_stats + ((size_t)tg * (a.gb_c >> cgs) + (co >> cgs)) * 2;
      const float mean = stp[0], rstd = stp[1];
      float gz[L_OUT], xh[L_OUT];
      read_row(row, gz);
      read_row(yrow, xh);
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int i = 0; i < L_OUT; ++i) {
        const float x = (xh[i] - mean) * rstd;
        const float z = fmaf(x, gam, bet);
        const float d = gz[i] * mish_grad_fast_f(z) * gam;
        gz[i] = d;
        xh[i] = x;
        t1 += d;
        t2 = fmaf(d, x, t2);
      }
      asm volatile("" : "+v"(t1), "+v"(t2));
      float s1[1] = {t1}, s2[1] = {t2};
      group_sum<1>(s1, cg);
      group_sum<1>(s2, cg);
      const float m1 = s1[0] * inv_n, m2 = s2[0] * inv_n;
#pragma unroll
      for (int i = 0; i < L_OUT; ++i) gz[i] = rstd * (gz[i] - m1 - xh[i] * m2);
      write_row(row, gz);
    }
  };
  // last piece: flat pass -- (+ residuals on the forward path), 128-bit stores
  auto piece_store = [&](const int traj0p) __attribute__((always_inline)) {
    if (GNB && !normed) return;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        a.out, 0, WS_STORE_BYTES((unsigned)((size_t)a.rows * a.out_tstride * sizeof(float))), 0x00020000);
    FlatIx x = flat_begin();
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      f32x4 y = *reinterpret_cast<const f32x4*>(tile + (etid + 256 * k) * 4);
      if (!GNB) y += rv[k];
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), ro, flat_off(x, traj0p, a.out_tstride, a.out_coff), 0, 0);
      flat_next(x);
    }
  };
  // the pieces of a tile over the 2 nch half-chunk phases of the NEXT tile's K loop: prefetch | flat 1 |
  // rows (one row set per phase when there are enough phases, else all at once) | store
  // The pieces of a tile over the 2 nch phases of the NEXT tile's K loop.  Everything that stores runs in ODD
  // phases, after that phase's split + write of the staged chunk: phase 0 prefetch, phase 1 the first flat
  // pass, the row sets over the odd phases 3, 5, ... (rpp row sets each), then the store pass.  With only two
  // chunks per tile the rows run in phase 2 and the store pass in phase 3.
  const int rpp = nch >= 3 ? (NP + (nch - 2) - 1) / (nch - 2) : NP;        // row sets per piece
  const int nrp = (NP + rpp - 1) / rpp;                                     // row pieces
  const int ph_store = nch >= 3 ? 3 + 2 * nrp : 3;
  auto run_piece = [&](const int ph, const int traj0p) __attribute__((always_inline)) {
    if (ph == 0) {
      piece_prefetch(traj0p);
    } else if (ph == 1) {
      if (GNB) piece_g1(traj0p); else piece_ysave(traj0p);
    } else if (ph == ph_store) {
      piece_store(traj0p);
    } else if (nch >= 3 ? ((ph & 1) && ph >= 3 && ph < ph_store) : ph == 2) {
      const int i = nch >= 3 ? (ph - 3) >> 1 : 0;
      const int j0 = i * rpp, j1 = min(NP, j0 + rpp);
      if (GNB) piece_rows_bwd(traj0p, j0, j1); else piece_rows_fwd(traj0p, j0, j1);
    }
  };

#ifdef TRACE_CONV_STAMPS
  // helper-wave time per activity: 0 even piece + load issue, 1 MID wait, 2 split + write (incl. its load wait),
  // 3 odd piece, 4 END wait
  unsigned long long ht[5] = {0, 0, 0, 0, 0}, ht_last = 0;
#define WS_HT(i)                                                            \
  do {                                                                      \
    const unsigned long long _n = __builtin_amdgcn_s_memtime();            \
    if ((i) > 0) ht[(i) - 1] += _n - ht_last;                               \
    ht_last = _n;                                                           \
  } while (0)
#else
#define WS_HT(i) do { } while (0)
#endif
  // chunk (it, cc) -> the chunk two further on in this workgroup's stream (nch is even, so the parity of the
  // global chunk counter is the parity of cc and the register set of a chunk is a compile-time choice)
  auto chunk_after = [&](int it, int cc, int ahead, int& it2, int& cc2) __attribute__((always_inline)) -> bool {
    it2 = it;
    cc2 = cc + ahead;
    while (cc2 >= nch) { cc2 -= nch; ++it2; }
    return it2 < nt;
  };
  auto phase_pair = [&](auto parc, const int it, const int cc, const bool prev, const int traj0p) __attribute__((always_inline)) {
    constexpr int PAR = decltype(parc)::value;          // parity of the chunk the K waves multiply now
    int it1, cc1, it2, cc2;
    bool have1 = chunk_after(it, cc, 1, it1, cc1);      // staged in the other register set, written in this chunk's odd phase
    bool have2 = chunk_after(it, cc, 2, it2, cc2);      // requested now, into this chunk's (free) register set
#ifdef TRACE_WS_EXP_NOSTAGE
    have1 = have2 = false;
#endif
#ifdef TRACE_WS_EXP_NOPIECE
    const bool prevp = false;
#else
    const bool prevp = prev;
#endif
    // even phase: a piece without stores (its loads go out ahead of the staging loads), then the loads of chunk + 2
    WS_HT(0);
    if (prevp) run_piece(2 * cc, traj0p);
    if (have2) {
      if constexpr (PAR == 0) stage_load(stgA, traj_of(it2), cc2); else stage_load(stgB, traj_of(it2), cc2);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    WS_HT(1);
    __builtin_amdgcn_s_barrier();          // MID
    WS_HT(2);
    // odd phase: chunk + 1 into the other image, then the piece (everything that stores)
    if (have1) {
      if constexpr (PAR == 0) stage_write(stgB, IMG); else stage_write(stgA, 0);
    }
    WS_HT(3);
    if (prevp) run_piece(2 * cc + 1, traj0p);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    WS_HT(4);
    __builtin_amdgcn_s_barrier();          // END
    WS_HT(5);
  };
  for (int it = 0; it <= nt; ++it) {
    const bool prev = it > 0;
    const int traj0p = prev ? traj_of(it - 1) : 0;
    for (int cc = 0; cc < nch; cc += 2) {
      phase_pair(std::integral_constant<int, 0>{}, it, cc, prev, traj0p);
      phase_pair(std::integral_constant<int, 1>{}, it, cc + 1, prev, traj0p);
    }
  }
#ifdef TRACE_CONV_STAMPS
  if (a.dbg && tid == 256) {
    unsigned long long* d = a.dbg + (size_t)blockIdx.x * 16;
    for (int i = 0; i < 5; ++i) d[8 + i] = ht[i];
  }
#endif
}

int conv_ws_num_cus();                       // unet.hip: multiprocessor count of the current device (cached)

template <int L_OUT, int UP, int TAPS, int PADL, int TB>
int launch_conv_ws(const ConvArgs& a_in, hipStream_t st) {
  using W = WsGeom<L_OUT, UP, TAPS, PADL, TB>;
  ConvArgs a = a_in;
  if (a.rows <= 0) return TRACE_OK;
  TRACE_REQUIRE(a.wb && a.cout % 64 == 0 && a.cin % 32 == 0 && a.cin2 % 32 == 0 &&
                    a.nsl == TAPS * a.cin / 32 + a.cin2 / 32 && a.cin / 32 + a.cin2 / 32 >= 2 &&
                    (a.cin / 32 + a.cin2 / 32) % 2 == 0,
                "conv(ws): bad channel geometry cin=%d cin2=%d cout=%d nsl=%d", a.cin, a.cin2, a.cout, a.nsl);
  TRACE_REQUIRE(a.cin2 == 0 || (a.in2 && UP == 1 && a.in2_tstride % 4 == 0 &&
                                (size_t)(a.rows + TB) * a.in2_tstride * sizeof(float) < 0xffffffffull),
                "conv(ws): second K source needs a stride-1 conv and a tensor below 4 GiB");
  TRACE_REQUIRE(!a.gnb || (!a.gn && !a.pc && !a.ysave && a.gb_y && a.gb_stats && a.gb_gamma && a.gb_beta &&
                           a.gb_cg > 0 && (a.gb_cg & (a.gb_cg - 1)) == 0 && a.gb_cg <= 64 &&
                           a.gb_cmax % 64 == 0 && a.gb_cmax <= a.gb_c &&
                           (reinterpret_cast<uintptr_t>(a.gb_y) & 15) == 0),
                "conv(ws): bad GroupNorm-backward epilogue arguments");
  TRACE_REQUIRE(!a.gn || (a.cg > 0 && (a.cg & (a.cg - 1)) == 0 && a.cg <= 64), "conv: group size %d", a.cg);
  TRACE_REQUIRE((reinterpret_cast<uintptr_t>(a.in) & 3) == 0 &&
                    (size_t)(a.rows + TB) * a.in_tstride * sizeof(float) < 0xffffffffull,
                "conv(ws): input of %d rows exceeds the 4 GiB buffer-descriptor range; split the rows", a.rows);
  auto ok128 = [&](const float* p, int tstride, int coff) {
    return p == nullptr || ((reinterpret_cast<uintptr_t>(p) & 15) == 0 && tstride % 4 == 0 &&
                            (coff * L_OUT) % 4 == 0 && (size_t)a.rows * tstride * sizeof(float) < 0xffffffffull);
  };
  TRACE_REQUIRE(ok128(a.out, a.out_tstride, a.out_coff) && ok128(a.res, a.res_tstride, a.res_coff) &&
                    ok128(a.res2, a.res2_tstride, a.res2_coff) && ok128(a.ysave, a.cout * L_OUT, 0) &&
                    ok128(a.out_raw, a.raw_tstride, a.raw_coff),
                "conv(ws): output / residual tensors must be 16-byte aligned per trajectory tile and below 4 GiB "
                "(%d rows)", a.rows);
  TRACE_REQUIRE(!a.pc || a.pt, "conv: cond table without step table");
  static unsigned long long attr_done = 0;      // one bit per device
  static unsigned long long attr_done_gnb = 0;
  TRACE_CHECK_HIP(ensure_dynamic_lds(&attr_done, reinterpret_cast<const void*>(&conv_ws_kernel<L_OUT, UP, TAPS, PADL, TB, false>),
                                     (int)W::LDS_BYTES));
  TRACE_CHECK_HIP(ensure_dynamic_lds(&attr_done_gnb, reinterpret_cast<const void*>(&conv_ws_kernel<L_OUT, UP, TAPS, PADL, TB, true>),
                                     (int)W::LDS_BYTES));
  const int ncol = a.cout / 64;
  const int nrb = (a.rows + TB - 1) / TB;
  const int ntv = ((nrb + 7) / 8) * 8 * ncol;            // virtual tile ids (the last group of 8 row blocks may be ragged)
  int grid = (conv_ws_num_cus() / (8 * ncol)) * (8 * ncol);      // one workgroup per CU, a multiple of 8 ncol
  if (grid <= 0) grid = 8 * ncol;
  if (grid > ntv) grid = ntv;
#ifdef TRACE_CONV_STAMPS
  a.dbg = stamp_buffer_for_next_launch();
#endif
  const bool prof = prof_active();
  if (prof) prof_pre(st);
  if (a.gnb)
    hipLaunchKernelGGL((conv_ws_kernel<L_OUT, UP, TAPS, PADL, TB, true>), dim3(grid), dim3(512), W::LDS_BYTES, st, a, ntv);
  else
    hipLaunchKernelGGL((conv_ws_kernel<L_OUT, UP, TAPS, PADL, TB, false>), dim3(grid), dim3(512), W::LDS_BYTES, st, a, ntv);
  if (prof) {
    const double fl = 2.0 * a.alg_macs_per_pos * (double)a.rows * L_OUT;
    prof_post(st, fl, fl * 6.0 / PEAK_MFMA16_FLOPS);
  }
  return launch_status("conv_ws_kernel");
}

}  // namespace trace
