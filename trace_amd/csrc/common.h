// Shared helpers for libtrace_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "trace_hip.h"

namespace trace {

void set_error(const char* fmt, ...);

#define TRACE_CHECK_HIP(expr)                                                              \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      ::trace::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                         __LINE__);                                                        \
      return TRACE_EHIP;                                                                   \
    }                                                                                      \
  } while (0)

#define TRACE_REQUIRE(cond, ...)          \
  do {                                    \
    if (!(cond)) {                        \
      ::trace::set_error(__VA_ARGS__);    \
      return TRACE_EINVAL;                \
    }                                     \
  } while (0)

// Optional per-launch timing of the conv kernels (bench.py roofline): HIP events recorded on
// the launch stream around every conv_gemm launch while a profile is open.
bool prof_active();
void prof_pre(hipStream_t st);
// `pipe_seconds`: the launch's matrix-pipe time at the chip's dense peak (its MFMA FLOPs -- split terms included --
// over the peak rate of the MFMA type it uses): what the roofline divides by
void prof_post(hipStream_t st, double flops, double pipe_seconds);
constexpr double PEAK_MFMA16_FLOPS = 2500e12;    // dense bf16 / fp16 MFMA (MI355X_MICROARCH.md)
constexpr double PEAK_MFMA32_FLOPS = 157.3e12;   // fp32 MFMA

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) belongs to a (kernel, device) pair: `done` is one bit per device
// ordinal, owned by the launcher of one kernel instantiation.  The bit is set only AFTER the attribute call has
// succeeded, so a second host thread can never launch with more than 64 KB of LDS ahead of it (two threads that race
// both make the call, which is harmless).  Returns a HIP error code.
inline hipError_t ensure_dynamic_lds(unsigned long long* done, const void* func, int bytes) {
  int dev = 0;
  const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev <= 63;
  const unsigned long long bit = known ? 1ull << dev : 0ull;
  if (known && (__atomic_load_n(done, __ATOMIC_ACQUIRE) & bit)) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess && known) __atomic_fetch_or(done, bit, __ATOMIC_ACQ_REL);
  return e;
}

inline int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("launch of %s failed: %s", what, hipGetErrorString(e));
    return TRACE_EHIP;
  }
  return TRACE_OK;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// mish(x) = x * tanh(softplus(x)); tanh(log(1+e)) = n/(n+2), n = e*(e+2)  (no cancellation).
// softplus threshold 20 as torch.
__device__ __forceinline__ float mish_f(float x) {
  if (x > 20.f) return x;
  float e = expf(x);
  float n = e * (e + 2.f);
  return x * (n / (n + 2.f));
}

// Same function on the hardware transcendental units (v_exp_f32 / v_rcp_f32, ~1 ulp each): used in
// the conv epilogue, where the library expf + IEEE division cost more VALU cycles than the MFMAs of
// the small-K layers they follow.  A bare v_exp_f32(x*log2e) loses |x|*6e-8 relative to the
// rounding of the product (the noisy-guidance loop amplifies that beyond the parity bar), so the
// product's rounding error and the low part of log2(e) are fed back through e^d ~ 1 + d.
// Every fused multiply-add of these two functions is spelled out and the compiler's own contraction is off:
// left to itself it fused a different subset of the candidates per element and per kernel instantiation, so the
// same activation came out 1 ulp apart in the full-tile and the half-tile conv kernels (round 3).
__device__ __forceinline__ float mish_fast_f(float x) {
#pragma clang fp contract(off)
  const float L2E_HI = 1.44269502162933349609375f;   // float(log2 e)
  const float L2E_LO = 1.92596299112661746e-8f;      // log2 e - L2E_HI
  float xc = fminf(x, 20.f);
  float p = xc * L2E_HI;
  float d = __builtin_fmaf(xc, L2E_LO, __builtin_fmaf(xc, L2E_HI, -p));      // exact remainder of the product + low part
  float e = __builtin_amdgcn_exp2f(p);
  e = __builtin_fmaf(e, d * 0.693147182464599609375f, e);
  float n = e * (e + 2.f);
  float w = n + 2.f;
  float r = __builtin_amdgcn_rcpf(w);
  r = __builtin_fmaf(__builtin_fmaf(-w, r, 1.f), r, r);          // one Newton step: <= 0.5 ulp
  // x > 20: the clamped exponent gives n r = 1 to within an ulp, i.e. x itself -- no select needed
  return x * (n * r);
}

// d/dx mish(x) = tanh(sp) + x * sigmoid(x) * (1 - tanh(sp)^2),  1 - tanh^2 = 4(n+1)/(n+2)^2
__device__ __forceinline__ float mish_grad_f(float x) {
  if (x > 20.f) return 1.f;
  float e = expf(x);
  float n = e * (e + 2.f);
  float w = n + 2.f;
  float tsp = n / w;
  float sig = e / (1.f + e);
  return tsp + x * sig * (4.f * (n + 1.f) / (w * w));
}

// mish'(x) on the hardware transcendental units, same compensated exponent and Newton-refined
// reciprocals as mish_fast_f (fused GroupNorm/Mish backward epilogue of the dgrad convs).
__device__ __forceinline__ float mish_grad_fast_f(float x) {
#pragma clang fp contract(off)
  const float L2E_HI = 1.44269502162933349609375f;
  const float L2E_LO = 1.92596299112661746e-8f;
  float xc = fminf(x, 20.f);
  float p = xc * L2E_HI;
  float d = __builtin_fmaf(xc, L2E_LO, __builtin_fmaf(xc, L2E_HI, -p));
  float e = __builtin_amdgcn_exp2f(p);
  e = __builtin_fmaf(e, d * 0.693147182464599609375f, e);
  float n = e * (e + 2.f);
  float w = n + 2.f;
  float r = __builtin_amdgcn_rcpf(w);
  r = __builtin_fmaf(__builtin_fmaf(-w, r, 1.f), r, r);
  float e1 = 1.f + e;
  float q = __builtin_amdgcn_rcpf(e1);
  q = __builtin_fmaf(__builtin_fmaf(-e1, q, 1.f), q, q);
  float tsp = n * r;
  float sig = e * q;
  // x > 20: tsp is 1 to within an ulp and, with x clamped here too, the second term stays below 2^-24 for every x
  // up to +inf (the reference returns exactly 1 there)
  return tsp + xc * sig * (4.f * (n + 1.f) * (r * r));
}

}  // namespace trace
