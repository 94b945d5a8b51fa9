// 2-D convolutions and the small layers of TRACE's map encoder (ResNet-18 trunk + MapGridDecoder) on gfx950.
// Reference behaviour: tbsim/models/base_models.py:169-260 (DoubleConv / Up / ConvBlock), :402-491
// (RasterizedMapEncoder: torchvision ResNet-18 with a C-channel 7x7 stem), tbsim/models/trace_helpers.py:176-268
// (MapEncoder, MapGridDecoder incl. the LayerNorm over (C, H, W)).  Paths relative to the reference root.
//
// conv2d_bf3_kernel: implicit GEMM on the bf16 matrix pipe with fp32 results, the operand scheme of
// conv_bf3_kernels.h (exact 3-way bf16 split of both operands, 6 MFMA terms per product, fp32 accumulate).
//   layout   : activations NHWC fp32 (a pixel's channels are contiguous: one staging unit = 8 channels = two 16-byte
//              loads; a channel window [coff, coff + C) of a wider buffer is addressable, so skip concatenations are
//              written in place by their producers)
//   tile     : one workgroup = TR output rows x the full output width of ONE image (<= 128 pixels = up to 8 MFMA row
//              tiles) x 64 output channels; wave w owns output channels 16w .. 16w+15 and every row tile
//   A operand: per 32-channel chunk the input REGION of the tile -- ((TR-1) S + KH) x ((WO-1) S + KW) pixels incl. the
//              zero halo (S: stride) -- is fetched once, split into three bf16 planes in registers and written to LDS as
//              [plane][channel octet][cell] x 16 B; every filter tap then reads the same image at a constant offset
//              ((ky RW + kx) cells), so an input pixel comes from L2 once per chunk and is reused KH KW times from LDS
//   B operand: weights split and packed on the host in MFMA fragment order [col tile][chunk][tap][plane][lane],
//              streamed L2 -> VGPR one slice ahead
//   epilogue : per-channel scale / shift (evaluation-mode BatchNorm and the conv bias folded), optional residual,
//              optional ReLU, straight from the accumulators (16 lanes = 16 consecutive channels of a pixel = 64 B)
#include "conv_kernels.h"

namespace trace {

struct Conv2dArgs {
  const float* in;      // (B, H, W, in_ctot)
  int in_ctot, in_coff;
  int H, W, cin;        // cin: multiple of 32
  const void* wb;       // packed split-bf16 weights
  int nsl;              // KH * KW * cin / 32
  float* out;           // (B, Ho, Wo, out_ctot)
  int out_ctot, out_coff;
  int Ho, cout;         // cout: multiple of 16
  const float* scale;   // [cout] or null (1)
  const float* shift;   // [cout] or null (0)
  const float* res;     // (B, Ho, Wo, res_ctot) or null
  int res_ctot, res_coff;
  int relu;
  int B;
  int tiles_per_image;  // ceil(Ho / TR)
};

template <int KH, int KW, int S, int PAD, int TR, int WO>
struct Conv2dGeom {
  static constexpr int MT = TR * WO;                       // output pixels per tile
  static constexpr int NRT = (MT + 15) / 16;               // MFMA row tiles
  // a 1x1 filter never touches the pixels between its strided taps: its region holds one cell per output pixel
  // (cell step CS input pixels, LDS stride SL = 1); wider filters hold every input pixel of the window (CS = 1, SL = S)
  static constexpr int CS = (KH == 1 && KW == 1) ? S : 1;
  static constexpr int SL = S / CS;
  static constexpr int RH = (TR - 1) * SL + KH;            // input region incl. halo, in cells
  static constexpr int RW = (WO - 1) * SL + KW;
  static constexpr int NCELL = RH * RW;
  static constexpr int SUB = (NCELL * 16 + 255) / 256 * 256;   // bytes of one (plane, octet) array
  static constexpr int PLANE = 4 * SUB;
  static constexpr int NU = NCELL * 4;                     // staging units (cell, octet)
  static constexpr int NUT = (NU + 255) / 256;
  static constexpr int TAPS = KH * KW;
  static constexpr size_t LDS_BYTES = (size_t)3 * PLANE;
  static_assert(NRT <= 8 && LDS_BYTES <= 80 * 1024 && 3 * PLANE < 65536, "tile too large");
};

template <int KH, int KW, int S, int PAD, int TR, int WO>
__global__ __launch_bounds__(256, 2) void conv2d_bf3_kernel(Conv2dArgs a) {
  using G = Conv2dGeom<KH, KW, S, PAD, TR, WO>;
  constexpr int NRT = G::NRT, RW = G::RW, NCELL = G::NCELL, SUB = G::SUB, PLANE = G::PLANE;
  constexpr int NUT = G::NUT, NU = G::NU, TAPS = G::TAPS, MT = G::MT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const ldsb = reinterpret_cast<char*>(lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // workgroup id -> (image, row tile of the image, 64-channel column group); the column groups of one tile are
  // neighbours in the grid: they stage the same region and share it through L2
  const int ncolwg = (a.cout + 63) >> 6;
  int bid = blockIdx.x;
  const int cg = bid % ncolwg;
  bid /= ncolwg;
  const int tile = bid % a.tiles_per_image;
  const int b = bid / a.tiles_per_image;
  const int y0 = tile * TR;                                 // first output row
  const int co0 = cg * 64 + wave * 16;
  const bool wave_on = co0 < a.cout;                        // waves past the last 16-channel column tile only stage

  // ---- staging plan: unit q = tid + 256 j -> octet o = q / NCELL, cell = q % NCELL (lanes run along cells)
  const float* img = a.in + (size_t)b * a.H * a.W * a.in_ctot + a.in_coff;
  int goff[NUT];          // float offset of the unit's 8 channels inside the image (chunk 0), -1 = zeros
  unsigned wofs[NUT];     // LDS byte offset inside a plane
#pragma unroll
  for (int j = 0; j < NUT; ++j) {
    const int q = tid + 256 * j;
    const int o = q / NCELL;
    const int cell = q - o * NCELL;
    const int r = cell / RW, c = cell - r * RW;
    const int yin = y0 * S - PAD + r * G::CS, xin = c * G::CS - PAD;
    const bool ok = q < NU && yin >= 0 && yin < a.H && xin >= 0 && xin < a.W;
    goff[j] = ok ? (yin * a.W + xin) * a.in_ctot + o * 8 : -1;
    wofs[j] = (unsigned)(cell * 16 + o * SUB);
  }
  f32x4 stg[NUT][2];
  auto stage_load = [&](int ch) {
#pragma unroll
    for (int j = 0; j < NUT; ++j) {
      if (goff[j] >= 0) {
        const f32x4* p = reinterpret_cast<const f32x4*>(img + goff[j] + ch * 32);
        stg[j][0] = p[0];
        stg[j][1] = p[1];
      } else {
        stg[j][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        stg[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto stage_write = [&]() {
#pragma unroll
    for (int j = 0; j < NUT; ++j) {
      if ((NU % 256 == 0) || (tid + 256 * j < NU)) {
        const float v[8] = {stg[j][0][0], stg[j][0][1], stg[j][0][2], stg[j][0][3],
                            stg[j][1][0], stg[j][1][1], stg[j][1][2], stg[j][1][3]};
        u32x4v h, m, l;
        split8_bf16(v, h, m, l);
        char* p = ldsb + wofs[j];
        *reinterpret_cast<u32x4v*>(p) = h;
        *reinterpret_cast<u32x4v*>(p + PLANE) = m;
        *reinterpret_cast<u32x4v*>(p + 2 * PLANE) = l;
      }
    }
  };

  // ---- A fragment addresses: MFMA row i = lane & 15 of row tile rt is output pixel p = 16 rt + i of the tile
  int rowoff[NRT];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) {
    int p = rt * 16 + (lane & 15);
    if (p >= MT) p = 0;                                     // padding rows of the last row tile: any valid cell
    const int yl = p / WO, x = p - yl * WO;
    rowoff[rt] = (yl * G::SL * RW + x * G::SL) * 16 + (lane >> 4) * SUB;
  }
  f32x4 acc[NRT];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // weight stream: [16-channel column tile][slice = chunk * TAPS + tap][plane][lane] x 8 bf16
  const bf16x8* wp = reinterpret_cast<const bf16x8*>(a.wb) + ((size_t)(wave_on ? co0 >> 4 : 0) * a.nsl) * 192 + lane;
  bf16x8 bn[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) bn[p] = wp[64 * p];
  int sdone = 0;

  const int nch = a.cin >> 5;
  stage_load(0);
  for (int ch = 0; ch < nch; ++ch) {
    __syncthreads();              // previous chunk fully consumed
    stage_write();
    __syncthreads();
    if (ch + 1 < nch) stage_load(ch + 1);                   // in flight during this chunk's MFMAs
    if (wave_on) {
      constexpr int NT = TAPS * NRT;
      constexpr int NST = 3, DIST = NST - 1;
      bf16x8 av[NST][3];
      auto load_a = [&](int n, int stage) {
        const int tap = n / NRT, rt = n % NRT;
        const int ky = tap / KW, kx = tap - ky * KW;
        const char* p = ldsb + rowoff[rt] + (ky * RW + kx) * 16;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) av[stage][pl] = *reinterpret_cast<const bf16x8*>(p + pl * PLANE);
      };
#pragma unroll
      for (int n = 0; n < DIST && n < NT; ++n) load_a(n, n);
      bf16x8 bw[3];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        if (n % NRT == 0) {
#pragma unroll
          for (int p = 0; p < 3; ++p) bw[p] = bn[p];
          ++sdone;
          wp += 192;
          if (sdone < a.nsl) {
#pragma unroll
            for (int p = 0; p < 3; ++p) bn[p] = wp[64 * p];
          }
        }
        if (n + DIST < NT) load_a(n + DIST, (n + DIST) % NST);
        __builtin_amdgcn_sched_barrier(0);
        const int s = n % NST, rt = n % NRT;
        f32x4 c = acc[rt];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[s][2], bw[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[s][0], bw[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[s][1], bw[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[s][1], bw[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[s][0], bw[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[s][0], bw[0], c, 0, 0, 0);
        acc[rt] = c;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (!wave_on) return;

  // ---- epilogue: MFMA C layout -- column (channel) = lane & 15, rows (pixels) 4 (lane >> 4) + reg
  const int co = co0 + (lane & 15);
  const float sc = a.scale ? a.scale[co] : 1.f;
  const float sh = a.shift ? a.shift[co] : 0.f;
  const size_t obase = (size_t)b * a.Ho * WO;
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int p = rt * 16 + (lane >> 4) * 4 + reg;
      const int yl = p / WO, x = p - yl * WO;
      if (p < MT && y0 + yl < a.Ho) {
        const size_t pix = obase + (size_t)(y0 + yl) * WO + x;
        float v = fmaf(acc[rt][reg], sc, sh);
        if (a.res) v += a.res[pix * a.res_ctot + a.res_coff + co];
        if (a.relu) v = fmaxf(v, 0.f);
        a.out[pix * a.out_ctot + a.out_coff + co] = v;
      }
    }
  }
}

template <int KH, int KW, int S, int PAD, int TR, int WO>
int launch_conv2d(const Conv2dArgs& a_in, hipStream_t st) {
  using G = Conv2dGeom<KH, KW, S, PAD, TR, WO>;
  Conv2dArgs a = a_in;
  a.tiles_per_image = (a.Ho + TR - 1) / TR;
  static unsigned long long attr_done = 0;
  TRACE_CHECK_HIP(ensure_dynamic_lds(&attr_done, reinterpret_cast<const void*>(&conv2d_bf3_kernel<KH, KW, S, PAD, TR, WO>),
                                     80 * 1024));
  const long wgs = (long)a.B * a.tiles_per_image * ((a.cout + 63) / 64);
  TRACE_REQUIRE(wgs > 0 && wgs < 0x7fffffffL, "conv2d: grid of %ld workgroups", wgs);
  hipLaunchKernelGGL((conv2d_bf3_kernel<KH, KW, S, PAD, TR, WO>), dim3((unsigned)wgs), dim3(256), G::LDS_BYTES, st, a);
  return launch_status("conv2d_bf3_kernel");
}

// ---- 7x7 stride-2 stem (C <= 8 input channels), BatchNorm + ReLU fused; NCHW image in, NHWC out ---------------
// One workgroup = one output row segment of 16 pixels x 64 channels: the 7 x 37 x C input patch and the weights
// ([tap][ci][co], co contiguous) sit in LDS; thread = (channel, 4 pixels).
__global__ __launch_bounds__(256) void stem7x7_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      float* __restrict__ out, int C, int H, int W, int Ho, int Wo) {
  extern __shared__ float sm[];
  float* sw = sm;                               // [49 * C][64]
  float* sp = sm + 49 * C * 64;                 // [C][7][37]
  const int tid = threadIdx.x;
  const int xt = blockIdx.x, y = blockIdx.y, b = blockIdx.z;
  for (int i = tid; i < 49 * C * 64; i += 256) sw[i] = w[i];
  const int x0 = xt * 16;
  for (int i = tid; i < C * 7 * 37; i += 256) {
    const int ci = i / (7 * 37), r = (i / 37) % 7, c = i % 37;
    const int yin = y * 2 - 3 + r, xin = x0 * 2 - 3 + c;
    sp[i] = (yin >= 0 && yin < H && xin >= 0 && xin < W) ? img[(((size_t)b * C + ci) * H + yin) * W + xin] : 0.f;
  }
  __syncthreads();
  const int co = tid & 63, pg = tid >> 6;       // pixels pg*4 .. pg*4+3
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int ci = 0; ci < C; ++ci)
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        const float wv = sw[((ky * 7 + kx) * C + ci) * 64 + co];
        const float* row = sp + (ci * 7 + ky) * 37 + kx + pg * 8;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = fmaf(row[2 * k], wv, acc[k]);
      }
  const float sc = scale[co], sh = shift[co];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int x = x0 + pg * 4 + k;
    if (x < Wo) out[(((size_t)b * Ho + y) * Wo + x) * 64 + co] = fmaxf(fmaf(acc[k], sc, sh), 0.f);
  }
}

// ---- max-pool 3x3 stride 2 pad 1, NHWC; thread = (output pixel, 4 channels) ----------------------------------
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           int H, int W, int C, int Ho, int Wo, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c4 = C >> 2;
  const int c = (int)(i % c4);
  long p = i / c4;
  const int x = (int)(p % Wo);
  p /= Wo;
  const int y = (int)(p % Ho);
  const long b = p / Ho;
  f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int yy = 2 * y - 1 + dy, xx = 2 * x - 1 + dx;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
        const f32x4 v = reinterpret_cast<const f32x4*>(in + ((b * H + yy) * W + xx) * C)[c];
        m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
      }
    }
  reinterpret_cast<f32x4*>(out + ((b * Ho + y) * Wo + x) * C)[c] = m;
}

// ---- bilinear x2 up-sampling, align_corners=True (base_models.py:190-227), written into the channel window
// [coff, coff + C) of the concatenation buffer; thread = (output pixel, 4 channels) -------------------------------
__global__ __launch_bounds__(256) void upsample2x_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                         int H, int W, int C, int out_ctot, int out_coff,
                                                         long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int Ho = 2 * H, Wo = 2 * W;
  const int c4 = C >> 2;
  const int c = (int)(i % c4);
  long p = i / c4;
  const int x = (int)(p % Wo);
  p /= Wo;
  const int y = (int)(p % Ho);
  const long b = p / Ho;
  // src = dst * (in - 1) / (out - 1), as ATen's area_pixel_compute_source_index for align_corners
  const float ry = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
  const float rx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const float fy = ry * (float)y, fx = rx * (float)x;
  const int yi = (int)fy, xi = (int)fx;
  const int y1 = yi + (yi < H - 1 ? 1 : 0), x1 = xi + (xi < W - 1 ? 1 : 0);
  const float ly = fy - (float)yi, lx = fx - (float)xi;
  const float hy = 1.f - ly, hx = 1.f - lx;
  auto px = [&](int yy, int xx) { return reinterpret_cast<const f32x4*>(in + ((b * H + yy) * W + xx) * C)[c]; };
  const f32x4 v00 = px(yi, xi), v01 = px(yi, x1), v10 = px(y1, xi), v11 = px(y1, x1);
  f32x4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) r[k] = hy * (hx * v00[k] + lx * v01[k]) + ly * (hx * v10[k] + lx * v11[k]);
  *reinterpret_cast<f32x4*>(out + ((b * Ho + y) * Wo + x) * out_ctot + out_coff + 4 * c) = r;
}

// ---- LayerNorm over all N = H W C elements of an image with an elementwise affine (trace_helpers.py:258,267);
// one workgroup of 1024 threads per image, exact two-pass moments -------------------------------------------------
__global__ __launch_bounds__(1024) void layernorm_image_kernel(const float* __restrict__ in,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               float* __restrict__ out, int n, float eps) {
  __shared__ float red[16];
  __shared__ float stat[2];
  const int tid = threadIdx.x;
  const float* x = in + (size_t)blockIdx.x * n;
  float* o = out + (size_t)blockIdx.x * n;
  const int n4 = n >> 2;
  auto block_sum = [&](float v) -> float {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k];
    return t;
  };
  float s = 0.f;
  for (int i = tid; i < n4; i += 1024) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    s += (v[0] + v[1]) + (v[2] + v[3]);
  }
  const float mean = block_sum(s) / (float)n;
  float q = 0.f;
  for (int i = tid; i < n4; i += 1024) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float d = v[k] - mean; q = fmaf(d, d, q); }
  }
  const float rstd = 1.f / sqrtf(block_sum(q) / (float)n + eps);
  (void)stat;
  for (int i = tid; i < n4; i += 1024) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    const f32x4 g = reinterpret_cast<const f32x4*>(gamma)[i];
    const f32x4 bb = reinterpret_cast<const f32x4*>(beta)[i];
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = fmaf((v[k] - mean) * rstd, g[k], bb[k]);
    reinterpret_cast<f32x4*>(o)[i] = r;
  }
}

}  // namespace trace

using namespace trace;

extern "C" int trace_conv2d(const float* in, int B, int H, int W, int in_ctot, int in_coff, int cin,
                            const void* packed_w, int kh, int kw, int stride, int pad, float* out, int out_ctot,
                            int out_coff, int cout, const float* scale, const float* shift, const float* res,
                            int res_ctot, int res_coff, int relu, void* stream) {
  TRACE_REQUIRE(in && packed_w && out && B >= 0, "trace_conv2d: null argument");
  if (B == 0) return TRACE_OK;
  TRACE_REQUIRE(cin > 0 && cin % 32 == 0 && cout > 0 && cout % 16 == 0, "trace_conv2d: cin %d must be a multiple of "
                "32 and cout %d of 16", cin, cout);
  TRACE_REQUIRE(in_ctot % 4 == 0 && in_coff % 4 == 0 && in_coff + cin <= in_ctot && out_coff + cout <= out_ctot &&
                    (reinterpret_cast<uintptr_t>(in) & 15) == 0,
                "trace_conv2d: channel windows must lie inside their buffers and be 16-byte aligned");
  TRACE_REQUIRE(!res || res_coff + cout <= res_ctot, "trace_conv2d: residual window outside its buffer");
  TRACE_REQUIRE((size_t)H * W * in_ctot < 0x7fffffffull, "trace_conv2d: image of %d x %d x %d too large", H, W, in_ctot);
  const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  Conv2dArgs a{};
  a.in = in; a.in_ctot = in_ctot; a.in_coff = in_coff; a.H = H; a.W = W; a.cin = cin;
  a.wb = packed_w; a.nsl = kh * kw * cin / 32;
  a.out = out; a.out_ctot = out_ctot; a.out_coff = out_coff; a.Ho = Ho; a.cout = cout;
  a.scale = scale; a.shift = shift; a.res = res; a.res_ctot = res_ctot; a.res_coff = res_coff;
  a.relu = relu; a.B = B;
  hipStream_t st = (hipStream_t)stream;
#define C2D(KH_, KW_, S_, P_, TR_, WO_) \
  if (kh == KH_ && kw == KW_ && stride == S_ && pad == P_ && Wo == WO_) return launch_conv2d<KH_, KW_, S_, P_, TR_, WO_>(a, st)
  C2D(3, 3, 1, 1, 2, 56);
  C2D(3, 3, 1, 1, 4, 28);
  C2D(3, 3, 1, 1, 8, 14);
  C2D(3, 3, 1, 1, 7, 7);
  C2D(3, 3, 2, 1, 2, 28);
  C2D(3, 3, 2, 1, 4, 14);
  C2D(3, 3, 2, 1, 7, 7);
  C2D(1, 1, 1, 0, 2, 56);
  C2D(1, 1, 2, 0, 4, 28);
  C2D(1, 1, 2, 0, 8, 14);
  C2D(1, 1, 2, 0, 7, 7);
#undef C2D
  set_error("trace_conv2d: no kernel for %dx%d stride %d pad %d at output width %d (built: the ResNet-18 / "
            "MapGridDecoder shapes of a 224 x 224 raster)", kh, kw, stride, pad, Wo);
  return TRACE_EINVAL;
}

extern "C" int trace_map_stem(const float* image, const float* w_tap_ci_co, const float* scale, const float* shift,
                              float* out, int B, int C, int H, int W, void* stream) {
  TRACE_REQUIRE(image && w_tap_ci_co && scale && shift && out && B >= 0, "trace_map_stem: null argument");
  TRACE_REQUIRE(C >= 1 && C <= 8, "trace_map_stem: %d input channels unsupported (1..8)", C);
  if (B == 0) return TRACE_OK;
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  TRACE_REQUIRE(Ho <= 65535 && B <= 65535, "trace_map_stem: grid too large (B = %d)", B);
  const size_t smem = (size_t)(49 * C * 64 + C * 7 * 37) * sizeof(float);
  static unsigned long long attr_done = 0;
  TRACE_CHECK_HIP(ensure_dynamic_lds(&attr_done, reinterpret_cast<const void*>(&stem7x7_kernel), 120 * 1024));
  hipLaunchKernelGGL(stem7x7_kernel, dim3((Wo + 15) / 16, Ho, B), dim3(256), smem, (hipStream_t)stream, image,
                     w_tap_ci_co, scale, shift, out, C, H, W, Ho, Wo);
  return launch_status("stem7x7_kernel");
}

extern "C" int trace_maxpool3x3s2(const float* in, float* out, int B, int H, int W, int C, void* stream) {
  TRACE_REQUIRE(in && out && B >= 0 && C % 4 == 0, "trace_maxpool3x3s2: bad argument");
  if (B == 0) return TRACE_OK;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total = (long)B * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     in, out, H, W, C, Ho, Wo, total);
  return launch_status("maxpool3x3s2_kernel");
}

extern "C" int trace_upsample2x(const float* in, float* out, int B, int H, int W, int C, int out_ctot, int out_coff,
                                void* stream) {
  TRACE_REQUIRE(in && out && B >= 0 && C % 4 == 0 && out_ctot % 4 == 0 && out_coff % 4 == 0 &&
                    out_coff + C <= out_ctot, "trace_upsample2x: bad argument");
  if (B == 0) return TRACE_OK;
  const long total = (long)B * (2 * H) * (2 * W) * (C / 4);
  hipLaunchKernelGGL(upsample2x_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in,
                     out, H, W, C, out_ctot, out_coff, total);
  return launch_status("upsample2x_kernel");
}

extern "C" int trace_layernorm_image(const float* in, const float* gamma, const float* beta, float* out, int B, int n,
                                     float eps, void* stream) {
  TRACE_REQUIRE(in && gamma && beta && out && B >= 0 && n > 0 && n % 4 == 0, "trace_layernorm_image: bad argument");
  if (B == 0) return TRACE_OK;
  hipLaunchKernelGGL(layernorm_image_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, in, gamma, beta, out, n,
                     eps);
  return launch_status("layernorm_image_kernel");
}
