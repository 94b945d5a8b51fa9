// Probe: accuracy and rate of an fp32 GEMM emulated by a 3-way bf16 split (a = h + m + l, 6 product
// terms per k on v_mfma_f32_16x16x32_bf16) against the native v_mfma_f32_16x16x4_f32.
//   accuracy: one wave, 16 x 16 tile, K = 1280 (a 256-channel k5 conv row), both vs an fp64 host sum
//   rate    : register-resident loops, one and two waves per SIMD on every CU
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// 8 consecutive-k fp32 values -> three bf16 planes
__device__ __forceinline__ void split8(const float* v, bf16x8& h, bf16x8& m, bf16x8& l) {
  unsigned hh[4], mm[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a0 = v[2 * i], a1 = v[2 * i + 1];
    const unsigned ph = cvt_pk(a0, a1);
    const float r0 = a0 - __builtin_bit_cast(float, ph << 16);
    const float r1 = a1 - __builtin_bit_cast(float, ph & 0xffff0000u);
    const unsigned pm = cvt_pk(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, pm << 16);
    const float s1 = r1 - __builtin_bit_cast(float, pm & 0xffff0000u);
    const unsigned pl = cvt_pk(s0, s1);
    hh[i] = ph; mm[i] = pm; ll[i] = pl;
  }
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  h = __builtin_bit_cast(bf16x8, (u32x4){hh[0], hh[1], hh[2], hh[3]});
  m = __builtin_bit_cast(bf16x8, (u32x4){mm[0], mm[1], mm[2], mm[3]});
  l = __builtin_bit_cast(bf16x8, (u32x4){ll[0], ll[1], ll[2], ll[3]});
}

// A [16][K] row-major, B [K][16] row-major, out [2][16][16]: 0 = fp32 MFMA, 1 = bf16 x 3
__global__ void accuracy(const float* A, const float* B, float* out, int K, int nterms) {
  const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  f32x4 c0 = {0, 0, 0, 0};
  for (int k = 0; k < K; k += 4)
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + k + q], B[(k + q) * 16 + r], c0, 0, 0, 0);
  f32x4 c1 = {0, 0, 0, 0};
  for (int k = 0; k < K; k += 32) {
    float av[8], bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      av[j] = A[r * K + k + 8 * q + j];
      bv[j] = B[(k + 8 * q + j) * 16 + r];
    }
    bf16x8 ah, am, al, bh, bm, bl;
    split8(av, ah, am, al);
    split8(bv, bh, bm, bl);
    // small terms first
    if (nterms >= 6) {
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c1, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c1, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c1, 0, 0, 0);
    }
    if (nterms >= 3) {
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c1, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c1, 0, 0, 0);
    }
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c1, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    out[(q * 4 + i) * 16 + r] = c0[i];
    out[256 + (q * 4 + i) * 16 + r] = c1[i];
  }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void rate(float* out, int iters) {
  f32x4 acc[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) acc[i] = f32x4{0, 0, 0, 0};
  const float s = (float)threadIdx.x * 1e-3f;
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 13; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(s + u, s - i, acc[i], 0, 0, 0);
    }
  } else {
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
  }
  float t = 0;
#pragma unroll
  for (int i = 0; i < 13; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}

static double frand() { return (double)rand() / RAND_MAX; }
static double gauss() { return sqrt(-2.0 * log(frand() + 1e-12)) * cos(6.283185307179586 * frand()); }

int main() {
  const int K = 1280;
  srand(1);
  std::vector<float> A(16 * K), B(K * 16);
  float *dA, *dB, *dO;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dO, 512 * 4);
  for (int scale_case = 0; scale_case < 3; ++scale_case) {
    const double sa = scale_case == 0 ? 1.0 : scale_case == 1 ? 1e-4 : 30.0;
    const double sb = scale_case == 0 ? 0.05 : scale_case == 1 ? 1e-3 : 2.0;
    for (auto& v : A) v = (float)(gauss() * sa);
    for (auto& v : B) v = (float)(gauss() * sb);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    for (int nterms : {6, 3, 1}) {
      hipLaunchKernelGGL(accuracy, dim3(1), dim3(64), 0, 0, dA, dB, dO, K, nterms);
      std::vector<float> o(512);
      hipMemcpy(o.data(), dO, 512 * 4, hipMemcpyDeviceToHost);
      double e0 = 0, e1 = 0, ref2 = 0, m0 = 0, m1 = 0, mref = 0;
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          double s = 0;
          for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * (double)B[k * 16 + j];
          const double d0 = o[i * 16 + j] - s, d1 = o[256 + i * 16 + j] - s;
          e0 += d0 * d0; e1 += d1 * d1; ref2 += s * s;
          m0 = fmax(m0, fabs(d0)); m1 = fmax(m1, fabs(d1)); mref = fmax(mref, fabs(s));
        }
      printf("scale %d terms %d: rms rel err fp32-mfma %.3e  bf16-split %.3e   max abs / max|ref| %.3e  %.3e\n",
             scale_case, nterms, sqrt(e0 / ref2), sqrt(e1 / ref2), m0 / mref, m1 / mref);
    }
  }
  // rate: 512 and 1024 workgroups of 4 waves (1 and 2 waves per SIMD on 256 CUs)
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
      // fp32-equivalent MACs: mode 0: 8 x 13 MFMAs of 16x16x4 per iter; mode 1: 13 x (16x16x32)
      const double macs = (mode == 0 ? 8.0 * 13 * 16 * 16 * 4 : 13.0 * 16 * 16 * 32) * iters * 4.0 * wgs;
      printf("rate wgs %d mode %s: %.3f ms  %.1f fp32-equivalent TFLOP/s\n", wgs, mode ? "bf16x3(6)" : "fp32", ms,
             2 * macs / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
