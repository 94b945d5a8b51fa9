// Probe: an fp32 GEMM emulated by a 2-way fp16 split (a = h + l, 3 product terms per k on
// v_mfma_f32_16x16x32_f16: l*h' + h*l' + h*h') against the 3-way bf16 split (6 terms) and the native fp32 MFMA.
//   subnormals: are fp16 subnormal INPUTS of the MFMA honoured (the low terms of small activations are)?
//   accuracy  : one wave, 16 x 16 tile, K = 1280, vs an fp64 host sum; B pre-scaled by a power of two so that
//               its low terms stay normal (the weights' scale is known on the host), A unscaled
//   rate      : register-resident loops, 1 and 2 waves per SIMD on every CU
// build: hipcc --offload-arch=gfx950 -O3 -o f16x2 f16x2.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// a - (float)half(h, sel): one mixed-precision FMA
__device__ __forceinline__ float sub_lo(float a, unsigned h) {
  float r;
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(h));
  return r;
}
__device__ __forceinline__ float sub_hi(float a, unsigned h) {
  float r;
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(h));
  return r;
}
__device__ __forceinline__ void split8_f16(const float* v, f16x8& h, f16x8& l) {
  unsigned hh[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned ph = cvt_pk_f16(v[2 * i], v[2 * i + 1]);
    hh[i] = ph;
    ll[i] = cvt_pk_f16(sub_lo(v[2 * i], ph), sub_hi(v[2 * i + 1], ph));
  }
  h = __builtin_bit_cast(f16x8, (u32x4){hh[0], hh[1], hh[2], hh[3]});
  l = __builtin_bit_cast(f16x8, (u32x4){ll[0], ll[1], ll[2], ll[3]});
}
__device__ __forceinline__ unsigned cvt_pk_bf(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ void split8_bf(const float* v, bf16x8& h, bf16x8& m, bf16x8& l) {
  unsigned hh[4], mm[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a0 = v[2 * i], a1 = v[2 * i + 1];
    const unsigned ph = cvt_pk_bf(a0, a1);
    const float r0 = a0 - __builtin_bit_cast(float, ph << 16);
    const float r1 = a1 - __builtin_bit_cast(float, ph & 0xffff0000u);
    const unsigned pm = cvt_pk_bf(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, pm << 16);
    const float s1 = r1 - __builtin_bit_cast(float, pm & 0xffff0000u);
    hh[i] = ph; mm[i] = pm; ll[i] = cvt_pk_bf(s0, s1);
  }
  h = __builtin_bit_cast(bf16x8, (u32x4){hh[0], hh[1], hh[2], hh[3]});
  m = __builtin_bit_cast(bf16x8, (u32x4){mm[0], mm[1], mm[2], mm[3]});
  l = __builtin_bit_cast(bf16x8, (u32x4){ll[0], ll[1], ll[2], ll[3]});
}

// out[0]: A = 2^-20 (fp16 subnormal), B = 2^10, K = 32: honoured -> 32 * 2^-10 = 2^-5; flushed -> 0
// out[1..8]: the split of 8 probe values (h | l bit patterns)
__global__ void subnormals(float* out, unsigned* bits) {
  f16x8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)9.5367431640625e-07f; b[j] = (_Float16)1024.f; }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
  const float v[8] = {1.0f + 1.0f / 4096 + 1.0f / 8388608, 0.1f, 3.0e-3f, 1.0e-5f, 7.7e-8f, -0.3333333f, 65000.f, 1e-9f};
  f16x8 h, l;
  split8_f16(v, h, l);
  if (threadIdx.x == 0) {
    const u32x4 hb = __builtin_bit_cast(u32x4, h), lb = __builtin_bit_cast(u32x4, l);
    for (int i = 0; i < 4; ++i) { bits[i] = hb[i]; bits[4 + i] = lb[i]; }
  }
}

// A [16][K] row-major, B [K][16] row-major (Bs = B * bscale), out [3][16][16]: fp32 MFMA, bf16 x 3, f16 x 2
__global__ void accuracy(const float* A, const float* B, const float* Bs, float* out, int K, float inv_bscale) {
  const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  f32x4 c0 = {0, 0, 0, 0};
  for (int k = 0; k < K; k += 4)
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + k + q], B[(k + q) * 16 + r], c0, 0, 0, 0);
  f32x4 c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0};
  for (int k = 0; k < K; k += 32) {
    float av[8], bv[8], bs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      av[j] = A[r * K + k + 8 * q + j];
      bv[j] = B[(k + 8 * q + j) * 16 + r];
      bs[j] = Bs[(k + 8 * q + j) * 16 + r];
    }
    bf16x8 ah, am, al, bh, bm, bl;
    split8_bf(av, ah, am, al);
    split8_bf(bv, bh, bm, bl);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c1, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c1, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c1, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c1, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c1, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c1, 0, 0, 0);
    f16x8 xh, xl, yh, yl;
    split8_f16(av, xh, xl);
    split8_f16(bs, yh, yl);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, yh, c2, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, yl, c2, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, yh, c2, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    out[(q * 4 + i) * 16 + r] = c0[i];
    out[256 + (q * 4 + i) * 16 + r] = c1[i];
    out[512 + (q * 4 + i) * 16 + r] = c2[i] * inv_bscale;
  }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void rate(float* out, int iters) {
  f32x4 acc[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) acc[i] = f32x4{0, 0, 0, 0};
  if (MODE == 0) {
    bf16x8 a[3], b[3];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int j = 0; j < 8; ++j) { a[p][j] = (short)(threadIdx.x + p + j); b[p][j] = (short)(threadIdx.x * 3 + p - j); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 13; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc[i], 0, 0, 0);
      }
    }
  } else {
    f16x8 a[2], b[2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int j = 0; j < 8; ++j) { a[p][j] = (_Float16)(0.001f * (threadIdx.x + p + j)); b[p][j] = (_Float16)(0.002f * (threadIdx.x + p - j)); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 13; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[0], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[1], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], acc[i], 0, 0, 0);
      }
    }
  }
  float t = 0;
#pragma unroll
  for (int i = 0; i < 13; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}

static double frand() { return (double)rand() / RAND_MAX; }
static double gauss() { return sqrt(-2.0 * log(frand() + 1e-12)) * cos(6.283185307179586 * frand()); }

int main() {
  float* dS; unsigned* dBits;
  hipMalloc(&dS, 64); hipMalloc(&dBits, 64);
  hipLaunchKernelGGL(subnormals, dim3(1), dim3(64), 0, 0, dS, dBits);
  float s0; unsigned bits[8];
  hipMemcpy(&s0, dS, 4, hipMemcpyDeviceToHost);
  hipMemcpy(bits, dBits, 32, hipMemcpyDeviceToHost);
  printf("MFMA f16 subnormal inputs: 32 * 2^-20 * 2^10 = %.9g (honoured: 0.03125, flushed: 0)\n", s0);
  printf("split h: %08x %08x %08x %08x   l: %08x %08x %08x %08x\n", bits[0], bits[1], bits[2], bits[3], bits[4], bits[5], bits[6], bits[7]);

  const int K = 1280;
  srand(1);
  std::vector<float> A(16 * K), B(K * 16), Bs(K * 16);
  float *dA, *dB, *dBs, *dO;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dBs, B.size() * 4); hipMalloc(&dO, 768 * 4);
  for (int scale_case = 0; scale_case < 5; ++scale_case) {
    const double sa = scale_case == 0 ? 1.0 : scale_case == 1 ? 1e-4 : scale_case == 2 ? 30.0 : scale_case == 3 ? 1e-2 : 1.0;
    const double sb = scale_case == 0 ? 0.05 : scale_case == 1 ? 1e-3 : scale_case == 2 ? 2.0 : 0.05;
    for (auto& v : A) v = (float)(gauss() * sa);
    if (scale_case == 4) for (size_t i = 0; i < A.size(); ++i) if (i % 7) A[i] *= 1e-3f;     // wide range inside a row
    float bmax = 0;
    for (auto& v : B) { v = (float)(gauss() * sb); bmax = fmaxf(bmax, fabsf(v)); }
    int e; frexpf(bmax, &e);                       // bmax in [2^(e-1), 2^e)
    const float bscale = ldexpf(1.f, 14 - e);      // scaled max in [2^13, 2^14)
    for (size_t i = 0; i < B.size(); ++i) Bs[i] = B[i] * bscale;
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dBs, Bs.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(accuracy, dim3(1), dim3(64), 0, 0, dA, dB, dBs, dO, K, 1.f / bscale);
    std::vector<float> o(768);
    hipMemcpy(o.data(), dO, 768 * 4, hipMemcpyDeviceToHost);
    double err[3] = {0, 0, 0}, mx[3] = {0, 0, 0}, ref2 = 0, mref = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double s = 0;
        for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * (double)B[k * 16 + j];
        ref2 += s * s; mref = fmax(mref, fabs(s));
        for (int m = 0; m < 3; ++m) {
          const double d = o[256 * m + i * 16 + j] - s;
          err[m] += d * d; mx[m] = fmax(mx[m], fabs(d));
        }
      }
    printf("scale case %d (A %.0e, B %.0e, B scale 2^%d): rms rel err fp32-mfma %.3e  bf16x3 %.3e  f16x2 %.3e   max abs / max|ref| %.3e  %.3e  %.3e\n",
           scale_case, sa, sb, 14 - e, sqrt(err[0] / ref2), sqrt(err[1] / ref2), sqrt(err[2] / ref2), mx[0] / mref, mx[1] / mref, mx[2] / mref);
  }
  float* dR;
  hipMalloc(&dR, 1024 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs : {256, 512}) {
    for (int mode = 0; mode < 2; ++mode) {
      const int iters = 2000;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(wgs), dim3(256), 0, 0, dR, iters);
        else hipLaunchKernelGGL(rate<1>, dim3(wgs), dim3(256), 0, 0, dR, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double macs = 13.0 * 16 * 16 * 32 * iters * 4.0 * wgs;
      printf("rate wgs %d mode %s: %.3f ms  %.1f fp32-equivalent TFLOP/s\n", wgs, mode ? "f16x2(3)" : "bf16x3(6)", ms,
             2 * macs / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
