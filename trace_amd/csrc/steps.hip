// Per-diffusion-step kernels of the TRACE sampler: map-feature query, classifier-free-guidance
// combine, posterior step and the unicycle rollout with its hand-written adjoint.
//
// This file is compiled with -ffp-contract=off: the reference evaluates these chains as separate
// ATen elementwise ops (one rounding per op), and the CFG detour x0 -> eps -> x0 is ill-conditioned
// at t=99, so fused multiply-adds would break the 1e-4 per-step bar.
//
// Work decomposition: one 64-lane wavefront per trajectory, lane t = horizon step t (T <= 64);
// the prefix sums of the rollout are wave scans (DPP/shuffle), no LDS, no barriers.
#include "common.h"

namespace trace {

// Inclusive prefix sum over lanes 0..T-1 in strict left-to-right order, starting from `init`:
// out[k] = ((init + v[0]) + v[1]) + ... + v[k].  The reference takes these sums as products with
// a lower-triangular ones matrix (trace_helpers.py:421-438,463,472), which ATen evaluates as a
// k-ordered FMA chain, i.e. exactly this sequential order; a tree scan would round differently
// and could flip the clip decisions that follow.
__device__ __forceinline__ float scan_seq(float init, float v, int lane, int T) {
  float acc = init, out = 0.f;
  for (int j = 0; j < T; ++j) {
    float vj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
    acc = acc + vj;
    if (lane == j) out = acc;
  }
  return out;
}

// suffix (reverse inclusive) sum over lanes [lane, 63]
__device__ __forceinline__ float scan_suffix(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float u = __shfl_down(v, o, 64);
    if (lane + o < 64) v = v + u;
  }
  return v;
}

__device__ __forceinline__ float clipf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

struct Roll {
  float x, y, v, yaw;          // state at this lane's step
  float v_raw, v_prev, v_avg;  // pre-clip speed, previous clipped speed, midpoint speed
  float yaw_prev, bound;       // heading before the step, yaw-rate bound
  float inc_ok, yv_ok;         // acc / yaw-rate clip masks (1 = gradient passes)
};

// tbsim/models/trace_helpers.py:431-475 for one trajectory; a0/a1 are the DESCALED actions of
// this lane (0 for lanes >= T).
__device__ __forceinline__ Roll rollout_lane(float a0, float a1, const float* __restrict__ s0,
                                             int lane, int T, const trace_dyn_params& p) {
  Roll r;
  const bool act = lane < T;
  float acc = clipf(a0, p.acc_lo, p.acc_hi);
  float inc = act ? acc * p.dt : 0.f;
  r.inc_ok = (a0 >= p.acc_lo && a0 <= p.acc_hi) ? 1.f : 0.f;
  r.v_raw = scan_seq(s0[2], inc, lane, T);
  r.v = clipf(r.v_raw, p.v_lo, p.v_hi);
  float vp = __shfl_up(r.v, 1, 64);
  r.v_prev = lane == 0 ? clipf(s0[2], p.v_lo, p.v_hi) : vp;
  r.v_avg = fmaf(0.5f, r.v, 0.5f * r.v_prev);
  float av = fabsf(r.v_prev);
  float b = fminf(p.max_steer * av, p.max_yawvel / fmaxf(av, 0.1f));
  r.bound = fmaxf(b, 0.1f);
  float yv = clipf(a1, -r.bound, r.bound);
  r.yv_ok = (a1 >= -r.bound && a1 <= r.bound) ? 1.f : 0.f;
  float yinc = act ? yv * p.dt : 0.f;
  r.yaw = scan_seq(s0[3], yinc, lane, T);
  float yp = __shfl_up(r.yaw, 1, 64);
  r.yaw_prev = lane == 0 ? s0[3] : yp;
  float vx = r.v_avg * cosf(r.yaw_prev);
  float vy = r.v_avg * sinf(r.yaw_prev);
  r.x = scan_seq(s0[0], act ? vx * p.dt : 0.f, lane, T);
  r.y = scan_seq(s0[1], act ? vy * p.dt : 0.f, lane, T);
  return r;
}

__device__ __forceinline__ void store_traj(float* __restrict__ o, const Roll& r, float a0, float a1,
                                           int scaled, const trace_dyn_params& p) {
  float v[6] = {r.x, r.y, r.v, r.yaw, a0, a1};
#pragma unroll
  for (int j = 0; j < 6; ++j) o[j] = scaled ? (v[j] + p.norm_add[j]) / p.norm_div[j] : v[j];
}

__global__ __launch_bounds__(256) void dyn_rollout_kernel(const float* __restrict__ actions,
                                                          const float* __restrict__ curr,
                                                          float* __restrict__ out, int M,
                                                          int num_samp, int T, int scaled_in,
                                                          int scaled_out, trace_dyn_params p) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  float a0 = 0.f, a1 = 0.f;
  if (lane < T) {
    a0 = actions[((size_t)m * T + lane) * 2];
    a1 = actions[((size_t)m * T + lane) * 2 + 1];
    if (scaled_in) {
      a0 = a0 * p.norm_div[4] - p.norm_add[4];
      a1 = a1 * p.norm_div[5] - p.norm_add[5];
    }
  }
  Roll r = rollout_lane(a0, a1, curr + (size_t)(m / num_samp) * 4, lane, T, p);
  if (lane < T) store_traj(out + ((size_t)m * T + lane) * 6, r, a0, a1, scaled_out, p);
}

// Reverse-mode of rollout_lane.
__global__ __launch_bounds__(256) void dyn_adjoint_kernel(const float* __restrict__ actions,
                                                          const float* __restrict__ curr,
                                                          const float* __restrict__ g,
                                                          float* __restrict__ ga, int M,
                                                          int num_samp, int T,
                                                          trace_dyn_params p) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  float a0 = 0.f, a1 = 0.f;
  float gx = 0.f, gy = 0.f, gv = 0.f, gyaw = 0.f, gacc = 0.f, gyv = 0.f;
  if (lane < T) {
    a0 = actions[((size_t)m * T + lane) * 2] * p.norm_div[4] - p.norm_add[4];
    a1 = actions[((size_t)m * T + lane) * 2 + 1] * p.norm_div[5] - p.norm_add[5];
    const float* gp = g + ((size_t)m * T + lane) * 6;
    gx = gp[0]; gy = gp[1]; gv = gp[2]; gyaw = gp[3]; gacc = gp[4]; gyv = gp[5];
  }
  Roll r = rollout_lane(a0, a1, curr + (size_t)(m / num_samp) * 4, lane, T, p);
  float sx = scan_suffix(gx, lane) * p.dt;   // dL/d vx_j
  float sy = scan_suffix(gy, lane) * p.dt;
  float c = cosf(r.yaw_prev), s = sinf(r.yaw_prev);
  float g_avg = sx * c + sy * s;
  float g_yawp = r.v_avg * (sy * c - sx * s);
  if (lane >= T) { g_avg = 0.f; g_yawp = 0.f; }
  // v_avg_j = .5 v_{j-1} + .5 v_j
  float g_avg_next = __shfl_down(g_avg, 1, 64);
  if (lane + 1 >= T) g_avg_next = 0.f;
  float g_v = gv + 0.5f * g_avg + 0.5f * g_avg_next;
  float v_ok = (r.v_raw >= p.v_lo && r.v_raw <= p.v_hi) ? 1.f : 0.f;
  float g_vraw = lane < T ? g_v * v_ok : 0.f;
  float g_inc = scan_suffix(g_vraw, lane);
  float out0 = (g_inc * p.dt * r.inc_ok + gacc) * p.norm_div[4];
  // yaw_k feeds yaw_prev_{k+1}
  float g_yawp_next = __shfl_down(g_yawp, 1, 64);
  if (lane + 1 >= T) g_yawp_next = 0.f;
  float g_yaw_tot = lane < T ? gyaw + g_yawp_next : 0.f;
  float g_yinc = scan_suffix(g_yaw_tot, lane);
  float out1 = (g_yinc * p.dt * r.yv_ok + gyv) * p.norm_div[5];
  if (lane < T) {
    ga[((size_t)m * T + lane) * 2] = out0;
    ga[((size_t)m * T + lane) * 2 + 1] = out1;
  }
}

// tbsim/models/trace.py:591-598, one thread per (trajectory, step, channel)
__global__ void cfg_combine_kernel(const float* __restrict__ pc, const float* __restrict__ pu,
                                   const float* __restrict__ x, float* __restrict__ x0, int n,
                                   float w1, float w, trace_step_coefs c) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float xt = x[(size_t)(i >> 1) * 6 + 4 + (i & 1)];
  float a = c.sqrt_recip_one_minus_alphas_cumprod, b = c.sqrt_alphas_over_one_minus_alphas_cumprod;
  float eps_c = a * xt - b * pc[i];
  float eps_u = a * xt - b * pu[i];
  float eps = w1 * eps_c - w * eps_u;
  x0[i] = c.sqrt_recip_alphas_cumprod * xt - c.sqrt_recipm1_alphas_cumprod * eps;
}

__global__ void cfg_adjoint_kernel(const float* __restrict__ g, float* __restrict__ gc,
                                   float* __restrict__ gu, int n, float kc, float ku) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = g[i];
  gc[i] = v * kc;
  if (gu) gu[i] = v * ku;
}

__global__ void posterior_mean_kernel(const float* __restrict__ x0, const float* __restrict__ x,
                                      float* __restrict__ mean, int n, trace_step_coefs c) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float xt = x[(size_t)(i >> 1) * 6 + 4 + (i & 1)];
  mean[i] = c.posterior_mean_coef1 * x0[i] + c.posterior_mean_coef2 * xt;
}

// tbsim/models/trace.py:653-704
__global__ __launch_bounds__(256) void posterior_step_kernel(
    const float* __restrict__ x0, const float* __restrict__ grad, const float* __restrict__ grad2,
    const float* __restrict__ x, const float* __restrict__ noise, const float* __restrict__ curr,
    float* __restrict__ x_next, float* __restrict__ a_out, int M, int num_samp, int T, int clean,
    trace_step_coefs c, trace_dyn_params p) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const float nz = c.t == 0 ? 0.f : 1.f;
  const float sigma = expf(0.5f * c.posterior_log_variance_clipped);
  float a[2] = {0.f, 0.f};
  float ad[2] = {0.f, 0.f};
  if (lane < T) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      size_t i = ((size_t)m * T + lane) * 2 + j;
      float xt = x[((size_t)m * T + lane) * 6 + 4 + j];
      // grad2: the classifier-free half of the input gradient, summed first as `gx[:M] + gx[M:]` does
      float gm = grad ? nz * (grad2 ? grad[i] + grad2[i] : grad[i]) : 0.f;
      float nt = (nz * sigma) * noise[i];
      float v;
      if (clean) {
        float xc = x0[i] - gm;
        float mean = c.posterior_mean_coef1 * xc + c.posterior_mean_coef2 * xt;
        v = mean + nt;
      } else {
        float mean = c.posterior_mean_coef1 * x0[i] + c.posterior_mean_coef2 * xt;
        v = (mean - gm) + nt;
      }
      a[j] = v;
      if (a_out) a_out[i] = v;
      ad[j] = v * p.norm_div[4 + j] - p.norm_add[4 + j];
    }
  }
  Roll r = rollout_lane(ad[0], ad[1], curr + (size_t)(m / num_samp) * 4, lane, T, p);
  if (lane < T) store_traj(x_next + ((size_t)m * T + lane) * 6, r, ad[0], ad[1], 1, p);
}

// query_map_feats + concat + transpose, one workgroup per trajectory.
// grid is channels-last (n_agents, Hg, Wg, C): the C features of a pixel are one 128-B line.
__global__ __launch_bounds__(256) void map_query_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ grid,
                                                        const float* __restrict__ grid2,
                                                        const float* __restrict__ rfa,
                                                        float* __restrict__ x_in, int M,
                                                        int num_samp, int T, int C, int Hg, int Wg,
                                                        int c_pad, float xs, float ys,
                                                        trace_dyn_params p) {
  extern __shared__ float sm[];
  float* wts = sm;                               // [T][4]
  int* base = reinterpret_cast<int*>(sm + 4 * T);  // [T] pixel index of (y0, x0)
  float* tile = sm + 5 * T;                      // [C][T]
  const int m = blockIdx.x;
  const int tid = threadIdx.x;
  const int ag = m / num_samp;
  if (tid < T) {
    const float* xp = x + ((size_t)m * T + tid) * 6;
    float px = xp[0] * p.norm_div[0] - p.norm_add[0];
    float py = xp[1] * p.norm_div[1] - p.norm_add[1];
    const float* A = rfa + (size_t)ag * 9;
    // points @ M^T: ATen's matmul is a k-ordered FMA chain
    float rx = fmaf(py, A[1], px * A[0]) + A[2];
    float ry = fmaf(py, A[4], px * A[3]) + A[5];
    rx = rx * xs;
    ry = ry * ys;
    rx = fminf(fmaxf(rx, 0.f), (float)(Wg - 1) - 1e-3f);
    ry = fminf(fmaxf(ry, 0.f), (float)(Hg - 1) - 1e-3f);
    float x0f = floorf(rx), y0f = floorf(ry);
    float x1f = x0f + 1.f, y1f = y0f + 1.f;
    wts[tid * 4 + 0] = (x1f - rx) * (y1f - ry);   // (y0, x0)
    wts[tid * 4 + 1] = (x1f - rx) * (ry - y0f);   // (y1, x0)
    wts[tid * 4 + 2] = (rx - x0f) * (y1f - ry);   // (y0, x1)
    wts[tid * 4 + 3] = (rx - x0f) * (ry - y0f);   // (y1, x1)
    base[tid] = (int)y0f * Wg + (int)x0f;
  }
  __syncthreads();
  // blockIdx.y = 1: the unconditioned half of a classifier-free call (rows M .. 2M - 1 of x_in, features of grid2)
  const float* gb = (blockIdx.y ? grid2 : grid) + (size_t)ag * Hg * Wg * C;
  for (int e = tid; e < T * C; e += 256) {
    int t = e / C, ch = e - t * C;
    const float* g0 = gb + (size_t)base[t] * C + ch;
    float fa = g0[0], fb = g0[(size_t)Wg * C], fc = g0[C], fd = g0[(size_t)(Wg + 1) * C];
    float v = wts[t * 4] * fa + wts[t * 4 + 1] * fb + wts[t * 4 + 2] * fc + wts[t * 4 + 3] * fd;
    tile[ch * T + t] = v;
  }
  __syncthreads();
  float* ob = x_in + ((size_t)blockIdx.y * M + m) * c_pad * T;
  for (int e = tid; e < 6 * T; e += 256) {
    int j = e / T, t = e - j * T;
    ob[e] = x[((size_t)m * T + t) * 6 + j];
  }
  for (int e = tid; e < C * T; e += 256) ob[6 * T + e] = tile[e];
}

}  // namespace trace

using namespace trace;

static int map_query_impl(const float* x, const float* grid, const float* grid2, const float* raster_from_agent,
                          float* x_in, int M, int num_samp, int T, int C, int Hg, int Wg, int c_pad, float img_w,
                          float img_h, const trace_dyn_params* p, void* stream) {
  size_t sm = (size_t)(5 * T + C * T) * sizeof(float);
  hipLaunchKernelGGL(map_query_kernel, dim3(M, grid2 ? 2 : 1), dim3(256), sm, (hipStream_t)stream, x, grid, grid2,
                     raster_from_agent, x_in, M, num_samp, T, C, Hg, Wg, c_pad, (float)Wg / img_w,
                     (float)Hg / img_h, *p);
  return launch_status("map_query_kernel");
}

extern "C" int trace_map_query(const float* x, const float* grid, const float* raster_from_agent,
                               float* x_in, int M, int num_samp, int T, int C, int Hg, int Wg,
                               int c_pad, float img_w, float img_h, const trace_dyn_params* p,
                               void* stream) {
  TRACE_REQUIRE(x && grid && raster_from_agent && x_in && p, "trace_map_query: null argument");
  TRACE_REQUIRE(M >= 0 && num_samp > 0 && T > 0 && T <= 64 && C > 0 && 6 + C <= c_pad,
                "trace_map_query: bad shape M=%d T=%d C=%d c_pad=%d", M, T, C, c_pad);
  if (M == 0) return TRACE_OK;
  return map_query_impl(x, grid, nullptr, raster_from_agent, x_in, M, num_samp, T, C, Hg, Wg, c_pad, img_w, img_h, p,
                        stream);
}

extern "C" int trace_map_query_cfg(const float* x, const float* grid, const float* grid_non_cond,
                                   const float* raster_from_agent, float* x_in, int M, int num_samp, int T,
                                   int C, int Hg, int Wg, int c_pad, float img_w, float img_h,
                                   const trace_dyn_params* p, void* stream) {
  TRACE_REQUIRE(x && grid && grid_non_cond && raster_from_agent && x_in && p, "trace_map_query_cfg: null argument");
  TRACE_REQUIRE(M >= 0 && num_samp > 0 && T > 0 && T <= 64 && C > 0 && 6 + C <= c_pad,
                "trace_map_query_cfg: bad shape M=%d T=%d C=%d c_pad=%d", M, T, C, c_pad);
  if (M == 0) return TRACE_OK;
  return map_query_impl(x, grid, grid_non_cond, raster_from_agent, x_in, M, num_samp, T, C, Hg, Wg, c_pad, img_w,
                        img_h, p, stream);
}

extern "C" int trace_dyn_rollout(const float* actions, const float* curr_states, float* out, int M,
                                 int num_samp, int T, int scaled_input, int scaled_output,
                                 const trace_dyn_params* p, void* stream) {
  TRACE_REQUIRE(actions && curr_states && out && p, "trace_dyn_rollout: null argument");
  TRACE_REQUIRE(M >= 0 && num_samp > 0 && T > 0 && T <= 64, "trace_dyn_rollout: bad shape");
  if (M == 0) return TRACE_OK;
  hipLaunchKernelGGL(dyn_rollout_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     actions, curr_states, out, M, num_samp, T, scaled_input, scaled_output, *p);
  return launch_status("dyn_rollout_kernel");
}

extern "C" int trace_dyn_adjoint(const float* actions, const float* curr_states,
                                 const float* grad_traj, float* grad_actions, int M, int num_samp,
                                 int T, const trace_dyn_params* p, void* stream) {
  TRACE_REQUIRE(actions && curr_states && grad_traj && grad_actions && p,
                "trace_dyn_adjoint: null argument");
  TRACE_REQUIRE(M >= 0 && num_samp > 0 && T > 0 && T <= 64, "trace_dyn_adjoint: bad shape");
  if (M == 0) return TRACE_OK;
  hipLaunchKernelGGL(dyn_adjoint_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     actions, curr_states, grad_traj, grad_actions, M, num_samp, T, *p);
  return launch_status("dyn_adjoint_kernel");
}

extern "C" int trace_cfg_combine(const float* pred_c, const float* pred_u, const float* x,
                                 float* x0, int M, int T, float w, const trace_step_coefs* c,
                                 void* stream) {
  TRACE_REQUIRE(pred_c && x && x0 && c, "trace_cfg_combine: null argument");
  int n = M * T * 2;
  if (n == 0) return TRACE_OK;
  if (w == 0.f) {
    TRACE_CHECK_HIP(hipMemcpyAsync(x0, pred_c, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice,
                                   (hipStream_t)stream));
    return TRACE_OK;
  }
  TRACE_REQUIRE(pred_u, "trace_cfg_combine: pred_u is required when w != 0");
  float w1 = (float)(1.0 + (double)w);
  hipLaunchKernelGGL(cfg_combine_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     pred_c, pred_u, x, x0, n, w1, w, *c);
  return launch_status("cfg_combine_kernel");
}

extern "C" int trace_cfg_combine_adjoint(const float* grad_x0, float* grad_pred_c,
                                         float* grad_pred_u, int M, int T, float w,
                                         const trace_step_coefs* c, void* stream) {
  TRACE_REQUIRE(grad_x0 && grad_pred_c && c, "trace_cfg_combine_adjoint: null argument");
  int n = M * T * 2;
  if (n == 0) return TRACE_OK;
  float kc = 1.f, ku = 0.f;
  if (w != 0.f) {
    TRACE_REQUIRE(grad_pred_u, "trace_cfg_combine_adjoint: grad_pred_u required when w != 0");
    float bb = c->sqrt_recipm1_alphas_cumprod * c->sqrt_alphas_over_one_minus_alphas_cumprod;
    kc = bb * (float)(1.0 + (double)w);
    ku = -bb * w;
  }
  hipLaunchKernelGGL(cfg_adjoint_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     grad_x0, grad_pred_c, w != 0.f ? grad_pred_u : nullptr, n, kc, ku);
  return launch_status("cfg_adjoint_kernel");
}

extern "C" int trace_posterior_mean(const float* x0, const float* x, float* mean, int M, int T,
                                    const trace_step_coefs* c, void* stream) {
  TRACE_REQUIRE(x0 && x && mean && c, "trace_posterior_mean: null argument");
  int n = M * T * 2;
  if (n == 0) return TRACE_OK;
  hipLaunchKernelGGL(posterior_mean_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, x0, x, mean, n, *c);
  return launch_status("posterior_mean_kernel");
}

extern "C" int trace_posterior_step(const float* x0, const float* grad, const float* x,
                                    const float* noise, const float* curr_states, float* x_next,
                                    float* a_out, int M, int num_samp, int T, int clean,
                                    const trace_step_coefs* c, const trace_dyn_params* p,
                                    void* stream) {
  TRACE_REQUIRE(x0 && x && noise && curr_states && x_next && c && p,
                "trace_posterior_step: null argument");
  TRACE_REQUIRE(M >= 0 && num_samp > 0 && T > 0 && T <= 64, "trace_posterior_step: bad shape");
  TRACE_REQUIRE(x != x_next, "trace_posterior_step: x_next must not alias x");
  if (M == 0) return TRACE_OK;
  hipLaunchKernelGGL(posterior_step_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     x0, grad, nullptr, x, noise, curr_states, x_next, a_out, M, num_samp, T, clean, *c, *p);
  return launch_status("posterior_step_kernel");
}

extern "C" int trace_posterior_step_cfg(const float* x0, const float* grad_cond, const float* grad_non_cond,
                                        const float* x, const float* noise, const float* curr_states,
                                        float* x_next, float* a_out, int M, int num_samp, int T, int clean,
                                        const trace_step_coefs* c, const trace_dyn_params* p, void* stream) {
  TRACE_REQUIRE(x0 && grad_cond && grad_non_cond && x && noise && curr_states && x_next && c && p,
                "trace_posterior_step_cfg: null argument");
  TRACE_REQUIRE(M >= 0 && num_samp > 0 && T > 0 && T <= 64, "trace_posterior_step_cfg: bad shape");
  TRACE_REQUIRE(x != x_next, "trace_posterior_step_cfg: x_next must not alias x");
  if (M == 0) return TRACE_OK;
  hipLaunchKernelGGL(posterior_step_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x0, grad_cond,
                     grad_non_cond, x, noise, curr_states, x_next, a_out, M, num_samp, T, clean, *c, *p);
  return launch_status("posterior_step_kernel");
}

// ---- environment side: agent-centric drivable raster --------------------
// One thread = 4 consecutive pixels of one agent's H x W raster: pixel centre -> agent frame
// (inverse of raster_from_agent, trajdata_utils.py:196-200) -> world (the agent's pose) -> inside test
// against the axis-aligned obstacle boxes of the agent's scene.  Pure HBM write stream:
// (4 C + 1) bytes per pixel out, a few hundred bytes per agent in.
namespace trace {
__global__ __launch_bounds__(256) void env_rasterize_kernel(
    const float* __restrict__ pose, const int* __restrict__ scene_of_agent,
    const float* __restrict__ boxes, int n_boxes, float r00, float r02, float r11, float r12,
    float* __restrict__ image, int channels, uint8_t* __restrict__ drivable, int H, int W) {
  const int b = blockIdx.y;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;        // quad of pixels
  const int npix = H * W;
  if (q * 4 >= npix) return;
  const int py = (q * 4) / W, px0 = (q * 4) - py * W;
  const float x0 = pose[b * 3], y0 = pose[b * 3 + 1], yaw = pose[b * 3 + 2];
  const float c = cosf(yaw), s = sinf(yaw);
  const float* bx = boxes + (size_t)scene_of_agent[b] * n_boxes * 4;
  const float ay = ((float)py - r12) / r11;
  float drv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float ax = ((float)(px0 + e) - r02) / r00;
    const float wx = (c * ax - s * ay) + x0;
    const float wy = (s * ax + c * ay) + y0;
    bool free_cell = true;
    for (int k = 0; k < n_boxes; ++k)
      free_cell = free_cell && !(fabsf(wx - bx[4 * k]) < bx[4 * k + 2] && fabsf(wy - bx[4 * k + 1]) < bx[4 * k + 3]);
    drv[e] = free_cell ? 1.f : 0.f;
  }
  if (drivable) {
    uchar4 d;
    d.x = drv[0] != 0.f; d.y = drv[1] != 0.f; d.z = drv[2] != 0.f; d.w = drv[3] != 0.f;
    reinterpret_cast<uchar4*>(drivable + (size_t)b * npix)[q] = d;
  }
  if (image) {
    float4* img = reinterpret_cast<float4*>(image + (size_t)b * channels * npix);
    img[q] = float4{drv[0], drv[1], drv[2], drv[3]};
    if (channels > 1) img[npix / 4 + q] = float4{1.f - drv[0], 1.f - drv[1], 1.f - drv[2], 1.f - drv[3]};
    for (int ch = 2; ch < channels; ++ch) img[(size_t)ch * (npix / 4) + q] = float4{0.f, 0.f, 0.f, 0.f};
  }
}
}  // namespace trace

extern "C" int trace_env_rasterize(const float* pose, const int* scene_of_agent, const float* boxes,
                                   int n_boxes, const float* raster_from_agent_host, float* image,
                                   int channels, uint8_t* drivable, int n_agents, int H, int W,
                                   void* stream) {
  TRACE_REQUIRE(pose && scene_of_agent && raster_from_agent_host && (image || drivable) && (boxes || n_boxes == 0),
                "trace_env_rasterize: null argument");
  TRACE_REQUIRE(n_agents >= 0 && H > 0 && W > 0 && W % 4 == 0 && n_boxes >= 0 && channels >= 1 && n_agents <= 65535,
                "trace_env_rasterize: bad shape agents=%d H=%d W=%d channels=%d (W must be a multiple of 4)",
                n_agents, H, W, channels);
  if (n_agents == 0) return TRACE_OK;
  const float* r = raster_from_agent_host;
  TRACE_REQUIRE(r[0] != 0.f && r[4] != 0.f && r[1] == 0.f && r[3] == 0.f,
                "trace_env_rasterize: raster_from_agent must be an axis-aligned scale + offset");
  const int quads = H * W / 4;
  dim3 grid((quads + 255) / 256, n_agents), block(256);
  hipLaunchKernelGGL(trace::env_rasterize_kernel, grid, block, 0, (hipStream_t)stream, pose, scene_of_agent, boxes,
                     n_boxes, r[0], r[2], r[4], r[5], image, channels, drivable, H, W);
  return trace::launch_status("env_rasterize_kernel");
}
