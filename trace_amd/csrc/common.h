// Shared helpers for libtrace_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

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
void prof_post(hipStream_t st, double flops);

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

}  // namespace trace
