// Test-time guidance objectives of TRACE with analytic gradients (no autograd), gfx950.
// Reference behaviour: tbsim/utils/guidance_loss.py (lines cited per kernel).
//
// Compiled with -ffp-contract=off so threshold decisions (collision hinge, raster truncation)
// see the same roundings as the reference's separate ATen ops.
//
// Decomposition: one 64-lane wavefront per trajectory, lane t = horizon step; reductions over
// time are wave reductions.  Every kernel ADDS scale[agent] * dLoss/dtraj into grad_traj
// (owner-computes: the lane that owns (trajectory, t) is the only writer, kernels run in stream
// order, so no atomics) and writes the unweighted per-trajectory loss.
#include "common.h"

namespace trace {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

#define TRAJ_PROLOGUE()                                   \
  const int lane = threadIdx.x & 63;                      \
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);      \
  if (m >= M) return;                                     \
  const int ag = m / num_samp;                            \
  const float sc = scale[ag];                             \
  if (sc < 0.f) return;

// ---- TargetPosLoss, guidance_loss.py:523-540 ------------------------------------------
__global__ __launch_bounds__(256) void target_pos_kernel(const float* __restrict__ traj,
                                                         float* __restrict__ grad,
                                                         float* __restrict__ loss,
                                                         const float* __restrict__ target,
                                                         const int* __restrict__ t_min,
                                                         const float* __restrict__ scale, int M,
                                                         int num_samp, int T) {
  TRAJ_PROLOGUE();
  const int t0 = t_min[ag];
  const bool on = lane >= t0 && lane < T;
  float dx = 0.f, dy = 0.f;
  if (on) {
    dx = traj[((size_t)m * T + lane) * 6] - target[ag * 2];
    dy = traj[((size_t)m * T + lane) * 6 + 1] - target[ag * 2 + 1];
  }
  float q = dx * dx + dy * dy;
  float r = sqrtf(q);
  // softmin over time = softmax(-r)
  float mx = wave_max(on ? -r : -INFINITY);
  float e = on ? expf(-r - mx) : 0.f;
  float w = e / wave_sum(e);
  const float inv_n = 1.f / (float)(T - t0);
  float qbar = wave_sum(w * q);
  if (lane == 0) loss[m] = qbar * inv_n;
  if (grad && on) {
    float k = w * inv_n * (2.f - (r > 0.f ? (q - qbar) / r : 0.f));
    float* gp = grad + ((size_t)m * T + lane) * 6;
    gp[0] += sc * k * dx;
    gp[1] += sc * k * dy;
  }
}

// ---- TargetPosAtTimeLoss, guidance_loss.py:488-501 ---------------------------------------
__global__ __launch_bounds__(256) void target_pos_at_time_kernel(
    const float* __restrict__ traj, float* __restrict__ grad, float* __restrict__ loss,
    const float* __restrict__ target, const int* __restrict__ t_at,
    const float* __restrict__ scale, int M, int num_samp, int T) {
  TRAJ_PROLOGUE();
  const int ta = t_at[ag];
  if (lane != ta || ta >= T) return;
  float dx = traj[((size_t)m * T + lane) * 6] - target[ag * 2];
  float dy = traj[((size_t)m * T + lane) * 6 + 1] - target[ag * 2 + 1];
  float r = sqrtf(dx * dx + dy * dy);
  loss[m] = r;
  if (grad && r > 0.f) {
    float* gp = grad + ((size_t)m * T + lane) * 6;
    gp[0] += sc * dx / r;
    gp[1] += sc * dy / r;
  }
}

// ---- compute_progress_loss, guidance_loss.py:542-590 -------------------------------------
__global__ __launch_bounds__(256) void progress_kernel(const float* __restrict__ traj,
                                                       float* __restrict__ grad,
                                                       float* __restrict__ loss,
                                                       const float* __restrict__ target,
                                                       const float* __restrict__ goal,
                                                       const int* __restrict__ mode,
                                                       const float* __restrict__ scale, int M,
                                                       int num_samp, int T) {
  TRAJ_PROLOGUE();
  if (lane != 0) return;
  const float tx = target[ag * 2], ty = target[ag * 2 + 1];
  const float* pl = traj + ((size_t)m * T + (T - 1)) * 6;
  float fx = pl[0] - tx, fy = pl[1] - ty;
  float fd = sqrtf(fx * fx + fy * fy);
  float* gl = grad ? grad + ((size_t)m * T + (T - 1)) * 6 : nullptr;
  if (mode[ag] == 0) {
    float v = fd - goal[ag];
    loss[m] = v > 0.f ? v : 0.f;
    if (gl && v > 0.f && fd > 0.f) {
      gl[0] += sc * fx / fd;
      gl[1] += sc * fy / fd;
    }
  } else {
    const float* p0 = traj + (size_t)m * T * 6;
    float ix = p0[0] - tx, iy = p0[1] - ty;
    float id = sqrtf(ix * ix + iy * iy);
    float v = goal[ag] - (id - fd);
    loss[m] = v > 0.f ? v : 0.f;
    if (gl && v > 0.f) {
      if (fd > 0.f) {
        gl[0] += sc * fx / fd;
        gl[1] += sc * fy / fd;
      }
      if (id > 0.f) {
        float* g0 = grad + (size_t)m * T * 6;
        g0[0] -= sc * ix / id;
        g0[1] -= sc * iy / id;
      }
    }
  }
}

// ---- TargetSpeedLoss / MinSpeedLoss, guidance_loss.py:148-196 ------------------------------
__global__ __launch_bounds__(256) void speed_kernel(const float* __restrict__ traj,
                                                    float* __restrict__ grad,
                                                    float* __restrict__ loss,
                                                    const float* __restrict__ speed,
                                                    const float* __restrict__ scale, int kind,
                                                    float dt, int M, int num_samp, int T) {
  TRAJ_PROLOGUE();
  const float s = speed[ag];
  if (kind == 0) {
    float d = lane < T ? traj[((size_t)m * T + lane) * 6 + 2] - s : 0.f;
    float tot = wave_sum(d * d) / (float)T;
    if (lane == 0) loss[m] = tot;
    if (grad && lane < T) grad[((size_t)m * T + lane) * 6 + 2] += sc * 2.f * d / (float)T;
    return;
  }
  // finite-difference speed between step lane and lane+1
  const bool on = lane < T - 1;
  float vx = 0.f, vy = 0.f;
  if (on) {
    const float* p = traj + ((size_t)m * T + lane) * 6;
    vx = (p[6] - p[0]) / dt;
    vy = (p[7] - p[1]) / dt;
  }
  float vel = sqrtf(vx * vx + vy * vy);
  float l = 0.f, dl = 0.f;  // per-step loss and d loss / d vel
  if (on) {
    if (kind == 1) {
      float d = vel - s;
      l = d * d;
      dl = 2.f * d;
    } else {
      float u = s - vel;
      float cdf = 0.5f * (1.f + erff(u * 0.70710678118654752440f));
      float pdf = 0.39894228040143267794f * expf(-0.5f * u * u);
      float gl = u * cdf;
      l = gl * gl;
      dl = -2.f * gl * (cdf + u * pdf);
    }
  }
  const float inv_n = 1.f / (float)(T - 1);
  float tot = wave_sum(l) * inv_n;
  if (lane == 0) loss[m] = tot;
  if (grad) {
    // d vel / d p_{lane+1} = (v/|v|)/dt, d vel / d p_lane = -(v/|v|)/dt
    float kx = (on && vel > 0.f) ? sc * dl * inv_n * (vx / vel) / dt : 0.f;
    float ky = (on && vel > 0.f) ? sc * dl * inv_n * (vy / vel) / dt : 0.f;
    float px = __shfl_up(kx, 1, 64), py = __shfl_up(ky, 1, 64);
    if (lane == 0) { px = 0.f; py = 0.f; }
    if (lane < T) {
      float* gp = grad + ((size_t)m * T + lane) * 6;
      gp[0] += px - kx;
      gp[1] += py - ky;
    }
  }
}

// ---- AgentCollisionLoss, guidance_loss.py:264-338 ----------------------------------------
// pass 1: agent frame -> world pose (tbsim/models/trace_helpers.py:343-364)
//   ws[(m*T+t)*4] = (world x, world y, world yaw, d world_yaw / d yaw)
__global__ void world_pose_kernel(const float* __restrict__ traj, const float* __restrict__ wfa,
                                  float* __restrict__ ws, int n, int num_samp, int T) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int ag = (i / T) / num_samp;
  const float* A = wfa + (size_t)ag * 9;
  const float* p = traj + (size_t)i * 6;
  float px = p[0], py = p[1], yaw = p[3];
  // matmuls as k-ordered FMA chains (what ATen's bmm does for these shapes)
  float wx = fmaf(py, A[1], px * A[0]) + A[2];
  float wy = fmaf(py, A[4], px * A[3]) + A[5];
  float c = cosf(yaw), s = sinf(yaw);
  // the reference adds the translation and subtracts it again (trace_helpers.py:359-361)
  float hx = (fmaf(s, A[1], c * A[0]) + A[2]) - A[2];
  float hy = (fmaf(s, A[4], c * A[3]) + A[5]) - A[5];
  float dhx = -s * A[0] + c * A[1];
  float dhy = -s * A[3] + c * A[4];
  float n2 = hx * hx + hy * hy;
  float4 o;
  o.x = wx;
  o.y = wy;
  o.z = atan2f(hy, hx);
  o.w = n2 > 0.f ? (hx * dhy - hy * dhx) / n2 : 0.f;
  reinterpret_cast<float4*>(ws)[i] = o;
}

template <int D>
__global__ __launch_bounds__(256) void agent_collision_kernel(
    const float* __restrict__ ws, float* __restrict__ grad, float* __restrict__ loss,
    const float* __restrict__ wfa, const float* __restrict__ radius,
    const float* __restrict__ disk_x, float buffer_dist, const int* __restrict__ scene_start,
    const int* __restrict__ scene_size, const int* __restrict__ in_mask,
    const float* __restrict__ scale, int M, int num_samp, int T) {
  TRAJ_PROLOGUE();
  if (!in_mask[ag]) return;
  const int n = m - ag * num_samp;
  const bool on = lane < T;
  const int tl = on ? lane : T - 1;
  const float4 me = reinterpret_cast<const float4*>(ws)[(size_t)m * T + tl];
  const float ci = cosf(me.z), si = sinf(me.z);
  float cxi[D], cyi[D];
#pragma unroll
  for (int k = 0; k < D; ++k) {
    float dxk = disk_x[ag * D + k];
    cxi[k] = dxk * ci + me.x;
    cyi[k] = dxk * si + me.y;
  }
  const float ri = radius[ag];
  float tot = 0.f, gwx = 0.f, gwy = 0.f, gwyaw = 0.f;
  const int j0 = scene_start[ag], j1 = j0 + scene_size[ag];
  for (int j = j0; j < j1; ++j) {
    if (j == ag) continue;
    const float4 ot = reinterpret_cast<const float4*>(ws)[((size_t)j * num_samp + n) * T + tl];
    const float cj = cosf(ot.z), sj = sinf(ot.z);
    float best = INFINITY, bdx = 0.f, bdy = 0.f;
    int bk = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
#pragma unroll
      for (int q = 0; q < D; ++q) {
        float dxq = disk_x[j * D + q];
        float ex = cxi[k] - (dxq * cj + ot.x);
        float ey = cyi[k] - (dxq * sj + ot.y);
        float d = sqrtf(ex * ex + ey * ey);
        if (d < best) { best = d; bdx = ex; bdy = ey; bk = k; }
      }
    }
    const float pd = (ri + radius[j]) + buffer_dist;
    if (on && best <= pd) {
      tot += 1.f - best / pd;
      if (best > 0.f) {
        // row (ag, j) plus, when j is guided too, the symmetric row (j, ag)
        float wgt = in_mask[j] ? 2.f : 1.f;
        float k = -wgt / (best * pd);
        float gx = k * bdx, gy = k * bdy;
        gwx += gx;
        gwy += gy;
        float dxk = disk_x[ag * D + bk];
        gwyaw += dxk * (-si * gx + ci * gy);
      }
    }
  }
  float l = wave_sum(tot);
  if (lane == 0) loss[m] = l;
  if (grad && on) {
    const float* A = wfa + (size_t)ag * 9;
    float* gp = grad + ((size_t)m * T + lane) * 6;
    gp[0] += sc * (A[0] * gwx + A[3] * gwy);
    gp[1] += sc * (A[1] * gwx + A[4] * gwy);
    gp[3] += sc * gwyaw * me.w;
  }
}

// ---- MapCollisionLoss, guidance_loss.py:382-464 ------------------------------------------
// one wavefront per (trajectory, step); lane handles sample points lane and lane + 64.
// Distances use the same  |a|^2 + |b|^2 - 2ab  form torch.cdist takes for >25 points
// (SURVEY.md appendix A), evaluated as the k-ordered fma chain of its matmul.
__global__ __launch_bounds__(256) void map_collision_kernel(
    const float* __restrict__ traj, float* __restrict__ grad, float* __restrict__ loss_mt,
    const uint8_t* __restrict__ drivable, const float* __restrict__ extent,
    const float* __restrict__ rfa, const float* __restrict__ local_pts, int P,
    const float* __restrict__ scale, int H, int W, int M, int num_samp, int T, int* __restrict__ dbg_argmin) {
  __shared__ float spx[4][128], spy[4][128], spn[4][128];
  __shared__ unsigned char soff[4][128];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int mt = blockIdx.x * 4 + wv;
  if (mt >= M * T) return;
  const int m = mt / T;
  const int ag = m / num_samp;
  const float sc = scale[ag];
  // diagnostic output (tests only, null in production): per sample point two ints, the on-road point(s) the loss
  // paired it with -- {-1, -1} for on-road points and inactive steps, else {first arg-min row | number of tied rows
  // << 8, the 2nd .. 4th tied row one per byte (0xff = none)}
  if (dbg_argmin) {
    for (int i = lane; i < 2 * P; i += 64) dbg_argmin[(size_t)mt * 2 * P + i] = -1;
  }
  if (sc < 0.f) return;
  const float* p = traj + (size_t)mt * 6;
  const float px = p[0], py = p[1], yaw = p[3];
  const float el = extent[ag * 3], ew = extent[ag * 3 + 1];
  const float diag = sqrtf(el * el + ew * ew);
  const float c = cosf(yaw), s = sinf(yaw);
  const float* A = rfa + (size_t)ag * 9;
  const uint8_t* dm = drivable + (size_t)ag * H * W;
  float lx[2], ly[2], qx[2], qy[2];
  bool off[2] = {false, false}, valid[2];
  int cnt = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    int i = lane + h * 64;
    valid[h] = i < P;
    lx[h] = ly[h] = qx[h] = qy[h] = 0.f;
    if (valid[h]) {
      lx[h] = local_pts[i * 2] * el;
      ly[h] = local_pts[i * 2 + 1] * ew;
      qx[h] = fmaf(ly[h], -s, lx[h] * c) + px;
      qy[h] = fmaf(ly[h], c, lx[h] * s) + py;
      float rx = fmaf(qy[h], A[1], qx[h] * A[0]) + A[2];
      float ry = fmaf(qy[h], A[4], qx[h] * A[3]) + A[5];
      rx = fminf(fmaxf(rx, -1e6f), 1e6f);
      ry = fminf(fmaxf(ry, -1e6f), 1e6f);
      int ix = (int)rx, iy = (int)ry;   // .long(): truncation toward zero
      ix = min(max(ix, 0), W - 1);
      iy = min(max(iy, 0), H - 1);
      off[h] = dm[iy * W + ix] == 0;
      spx[wv][i] = qx[h];
      spy[wv][i] = qy[h];
      spn[wv][i] = qx[h] * qx[h] + qy[h] * qy[h];
      soff[wv][i] = off[h] ? 1 : 0;
    }
    cnt += __popcll(__ballot(valid[h] && off[h]));
  }
  if (cnt == 0 || cnt == P) {
    if (lane == 0) loss_mt[mt] = 0.f;
    return;
  }
  __builtin_amdgcn_wave_barrier();
  float tot = 0.f, gpx = 0.f, gpy = 0.f, gyaw = 0.f;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (!(valid[h] && off[h])) continue;
    // this lane's off-road point is the detached column j; rows i are the on-road points
    const float bx = qx[h], by = qy[h];
    const float bn = bx * bx + by * by;
    float dmin = INFINITY;
    int nties = 0, imin = -1;
    unsigned more = 0xffffffu;                 // 2nd .. 4th tied row
    for (int i = 0; i < P; ++i) {
      if (soff[wv][i]) continue;
      float acc = (-2.f * spx[wv][i]) * bx;
      acc = fmaf(-2.f * spy[wv][i], by, acc);
      acc = acc + spn[wv][i];
      acc = acc + bn;
      float d = sqrtf(fmaxf(acc, 0.f));
      if (d < dmin) { dmin = d; nties = 1; imin = i; more = 0xffffffu; }
      else if (d == dmin) {
        if (nties < 4) more = (more & ~(0xffu << (8 * (nties - 1)))) | ((unsigned)i << (8 * (nties - 1)));
        ++nties;
      }
    }
    if (dbg_argmin) {
      int* o = dbg_argmin + ((size_t)mt * P + lane + h * 64) * 2;
      o[0] = imin | (nties << 8);
      o[1] = (int)more;
    }
    tot += 1.f - dmin / diag;
    if (grad && dmin > 0.f) {
      // amin shares the gradient evenly between tied rows; cdist backward = (a-b)/dist
      const float k = -1.f / (dmin * diag * (float)nties);
      for (int i = 0; i < P; ++i) {
        if (soff[wv][i]) continue;
        float acc = (-2.f * spx[wv][i]) * bx;
        acc = fmaf(-2.f * spy[wv][i], by, acc);
        acc = acc + spn[wv][i];
        acc = acc + bn;
        float d = sqrtf(fmaxf(acc, 0.f));
        if (d == dmin) {
          float gx = k * (spx[wv][i] - bx), gy = k * (spy[wv][i] - by);
          float lxi = local_pts[i * 2] * el, lyi = local_pts[i * 2 + 1] * ew;
          gpx += gx;
          gpy += gy;
          gyaw += gx * (-lxi * s - lyi * c) + gy * (lxi * c - lyi * s);
        }
      }
    }
  }
  tot = wave_sum(tot);
  gpx = wave_sum(gpx);
  gpy = wave_sum(gpy);
  gyaw = wave_sum(gyaw);
  if (lane == 0) {
    loss_mt[mt] = tot;
    if (grad) {
      float* gp = grad + (size_t)mt * 6;
      gp[0] += sc * gpx;
      gp[1] += sc * gpy;
      gp[3] += sc * gyaw;
    }
  }
}

// ---- SocialGroupLoss, guidance_loss.py:762-814 -------------------------------------------
// Every member of a group keeps `social_dist` to ONE assigned neighbour per (sample, step): its
// nearest group member, or (the caller's random draw, `pick`) another member.  The leader's position is
// detached: it is pulled by nobody and pulls its followers.
//   members[ag * G + s]  global agent index of the s-th member of ag's group (-1 = padding)
//   pick[m * T + t]      >= 0: global index of the randomly assigned neighbour; -1: nearest
// pass 1: nbr[m*T+t] (global agent index) and dist[m*T+t].  The distance takes the form torch.cdist takes for the
// group's size (guidance_loss.py:786): the exact one up to TRACE_SOCIAL_GROUP_WIDTH = 25
// members, above that ATen's matmul form |a|^2 + |b|^2 - 2ab, evaluated as the k-ordered fma chain of its K = 4
// matmul (row = this member, column = the other one; the chain of map_collision_kernel) and clamped at 0 before the
// root -- the form decides which member is the nearest when two candidates are within its rounding noise.
// G = width of the members table (>= the largest group).
__global__ __launch_bounds__(256) void social_neighbor_kernel(
    const float* __restrict__ ws, const int* __restrict__ members, int G, const int* __restrict__ pick,
    int* __restrict__ nbr, float* __restrict__ dist, const float* __restrict__ scale, int M,
    int num_samp, int T) {
  TRAJ_PROLOGUE();
  if (lane >= T) return;
  const int n = m - ag * num_samp;
  const size_t i = (size_t)m * T + lane;
  const float4 me = reinterpret_cast<const float4*>(ws)[i];
  const int forced = pick ? pick[i] : -1;
  int cnt = 0;
#pragma unroll 1
  for (int s = 0; s < G; ++s) cnt += members[(size_t)ag * G + s] >= 0 ? 1 : 0;
  const bool matmul_form = cnt > TRACE_SOCIAL_GROUP_WIDTH;
  const float an = me.x * me.x + me.y * me.y;
  float best = INFINITY;
  int bj = -1;
#pragma unroll 1
  for (int s = 0; s < G; ++s) {
    const int j = members[(size_t)ag * G + s];
    if (j < 0 || j == ag) continue;
    const float4 ot = reinterpret_cast<const float4*>(ws)[((size_t)j * num_samp + n) * T + lane];
    float d;
    if (matmul_form) {
      const float bn = ot.x * ot.x + ot.y * ot.y;
      float acc = (-2.f * me.x) * ot.x;
      acc = fmaf(-2.f * me.y, ot.y, acc);
      acc = acc + an;
      acc = acc + bn;
      d = sqrtf(fmaxf(acc, 0.f));
    } else {
      const float ex = me.x - ot.x, ey = me.y - ot.y;
      d = sqrtf(ex * ex + ey * ey);
    }
    if (forced >= 0 ? j == forced : d < best) { best = d; bj = j; }
  }
  nbr[i] = bj;
  dist[i] = best;
}

// pass 2: loss and owner-computes gradient: lane (ag, n, t) collects its own row's pull towards its
// neighbour and the pull of every member whose assigned neighbour it is.
__global__ __launch_bounds__(256) void social_group_kernel(
    const float* __restrict__ ws, float* __restrict__ grad, float* __restrict__ loss,
    const float* __restrict__ wfa, const int* __restrict__ members, int G, const int* __restrict__ is_leader,
    const int* __restrict__ nbr, const float* __restrict__ dist, float social_dist,
    const float* __restrict__ scale, int M, int num_samp, int T) {
  TRAJ_PROLOGUE();
  const int n = m - ag * num_samp;
  const bool on = lane < T;
  const int tl = on ? lane : T - 1;
  const size_t i = (size_t)m * T + tl;
  const float4 me = reinterpret_cast<const float4*>(ws)[i];
  const float inv_t = 1.f / (float)T;
  const float d = dist[i];
  const int j = nbr[i];
  const float diff = d - social_dist;
  const float l = wave_sum(on && j >= 0 ? diff * diff : 0.f);
  if (lane == 0) loss[m] = l * inv_t;
  if (!grad || !on || is_leader[ag]) return;
  float gwx = 0.f, gwy = 0.f;
  if (j >= 0 && d > 0.f) {
    const float4 ot = reinterpret_cast<const float4*>(ws)[((size_t)j * num_samp + n) * T + tl];
    const float k = sc * (2.f * diff * inv_t) / d;
    gwx += k * (me.x - ot.x);
    gwy += k * (me.y - ot.y);
  }
#pragma unroll 1
  for (int s = 0; s < G; ++s) {
    const int o = members[(size_t)ag * G + s];
    if (o < 0 || o == ag) continue;
    const size_t io = ((size_t)o * num_samp + n) * T + tl;
    if (nbr[io] != ag) continue;
    const float dd = dist[io];
    if (!(dd > 0.f)) continue;
    const float4 ot = reinterpret_cast<const float4*>(ws)[io];
    const float k = scale[o] * (2.f * (dd - social_dist) * inv_t) / dd;
    gwx += k * (me.x - ot.x);
    gwy += k * (me.y - ot.y);
  }
  const float* A = wfa + (size_t)ag * 9;
  float* gp = grad + ((size_t)m * T + lane) * 6;
  gp[0] += A[0] * gwx + A[3] * gwy;
  gp[1] += A[1] * gwx + A[4] * gwy;
}

// loss[m] = sum_t loss_mt[m*T + t] for active agents
__global__ __launch_bounds__(256) void sum_over_time_kernel(const float* __restrict__ loss_mt,
                                                            float* __restrict__ loss,
                                                            const float* __restrict__ scale, int M,
                                                            int num_samp, int T) {
  TRAJ_PROLOGUE();
  (void)sc;
  float v = lane < T ? loss_mt[(size_t)m * T + lane] : 0.f;
  v = wave_sum(v);
  if (lane == 0) loss[m] = v;
}

}  // namespace trace

using namespace trace;

#define GUIDE_COMMON_CHECK(name)                                                          \
  TRACE_REQUIRE(traj && loss && scale, name ": null argument");                           \
  TRACE_REQUIRE(M >= 0 && num_samp > 0 && T > 0 && T <= 64, name ": bad shape M=%d T=%d", M, T); \
  if (M == 0) return TRACE_OK;                                                            \
  hipStream_t st = (hipStream_t)stream;                                                   \
  dim3 grid((M + 3) / 4), block(256);

extern "C" int trace_guide_target_pos(const float* traj, float* grad_traj, float* loss,
                                      const float* target, const int* t_min, const float* scale,
                                      int M, int num_samp, int T, void* stream) {
  GUIDE_COMMON_CHECK("trace_guide_target_pos");
  TRACE_REQUIRE(target && t_min, "trace_guide_target_pos: null table");
  hipLaunchKernelGGL(target_pos_kernel, grid, block, 0, st, traj, grad_traj, loss, target, t_min,
                     scale, M, num_samp, T);
  return launch_status("target_pos_kernel");
}

extern "C" int trace_guide_target_pos_at_time(const float* traj, float* grad_traj, float* loss,
                                              const float* target, const int* t_at,
                                              const float* scale, int M, int num_samp, int T,
                                              void* stream) {
  GUIDE_COMMON_CHECK("trace_guide_target_pos_at_time");
  TRACE_REQUIRE(target && t_at, "trace_guide_target_pos_at_time: null table");
  hipLaunchKernelGGL(target_pos_at_time_kernel, grid, block, 0, st, traj, grad_traj, loss, target,
                     t_at, scale, M, num_samp, T);
  return launch_status("target_pos_at_time_kernel");
}

extern "C" int trace_guide_progress(const float* traj, float* grad_traj, float* loss,
                                    const float* target, const float* goal, const int* mode,
                                    const float* scale, int M, int num_samp, int T, void* stream) {
  GUIDE_COMMON_CHECK("trace_guide_progress");
  TRACE_REQUIRE(target && goal && mode, "trace_guide_progress: null table");
  hipLaunchKernelGGL(progress_kernel, grid, block, 0, st, traj, grad_traj, loss, target, goal, mode,
                     scale, M, num_samp, T);
  return launch_status("progress_kernel");
}

extern "C" int trace_guide_speed(const float* traj, float* grad_traj, float* loss,
                                 const float* speed, const float* scale, int kind, float dt, int M,
                                 int num_samp, int T, void* stream) {
  GUIDE_COMMON_CHECK("trace_guide_speed");
  TRACE_REQUIRE(speed && kind >= 0 && kind <= 2 && dt > 0.f, "trace_guide_speed: bad argument");
  hipLaunchKernelGGL(speed_kernel, grid, block, 0, st, traj, grad_traj, loss, speed, scale, kind, dt,
                     M, num_samp, T);
  return launch_status("speed_kernel");
}

extern "C" int trace_guide_agent_collision(const float* traj, float* grad_traj, float* loss,
                                           const float* world_from_agent, const float* radius,
                                           const float* disk_x, int num_disks, float buffer_dist,
                                           const int* scene_start, const int* scene_size,
                                           const int* in_mask, const float* scale, int n_agents,
                                           int num_samp, int T, float* workspace, void* stream) {
  const int M = n_agents * num_samp;
  GUIDE_COMMON_CHECK("trace_guide_agent_collision");
  TRACE_REQUIRE(world_from_agent && radius && disk_x && scene_start && scene_size && in_mask &&
                    workspace,
                "trace_guide_agent_collision: null table");
  TRACE_REQUIRE(num_disks >= 1 && num_disks <= 5, "num_disks %d unsupported (1..5)", num_disks);
  int n = M * T;
  hipLaunchKernelGGL(world_pose_kernel, dim3((n + 255) / 256), dim3(256), 0, st, traj,
                     world_from_agent, workspace, n, num_samp, T);
  int rc = launch_status("world_pose_kernel");
  if (rc) return rc;
#define AC(DD)                                                                                  \
  hipLaunchKernelGGL(agent_collision_kernel<DD>, grid, block, 0, st, workspace, grad_traj, loss,    \
                     world_from_agent, radius, disk_x, buffer_dist, scene_start, scene_size,    \
                     in_mask, scale, M, num_samp, T)
  switch (num_disks) {
    case 1: AC(1); break;
    case 2: AC(2); break;
    case 3: AC(3); break;
    case 4: AC(4); break;
    default: AC(5); break;
  }
#undef AC
  return launch_status("agent_collision_kernel");
}

// ---- the entry point that allocates nothing -------------------------------------------------
// World poses of trace_guide_agent_collision (its `workspace` argument) kept in a file-scope device
// array instead, for callers that cannot allocate.  `scratch` is a host-side stand-in here: the
// device address is looked up once per device with hipGetSymbolAddress and only that pointer is
// ever handed to kernels.
namespace {
constexpr int kStaticScratchFloats = 2 * 1024 * 1024;             // 8 MiB per GPU
static __device__ __align__(256) float scratch[kStaticScratchFloats];
constexpr int kMaxDevices = 64;
float* g_scratch_ptr[kMaxDevices] = {};                          // per device, resolved on first use

int static_scratch(float** out) {
  int dev = -1;
  TRACE_CHECK_HIP(hipGetDevice(&dev));
  TRACE_REQUIRE(dev >= 0 && dev < kMaxDevices, "device ordinal %d outside 0..%d", dev, kMaxDevices - 1);
  float* p = __atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE);
  if (!p) {
    void* sym = nullptr;
    TRACE_CHECK_HIP(hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch)));
    TRACE_REQUIRE(sym != nullptr, "hipGetSymbolAddress returned a null address for the static scratch");
    p = reinterpret_cast<float*>(sym);
    __atomic_store_n(&g_scratch_ptr[dev], p, __ATOMIC_RELEASE);  // every thread resolves the same address
  }
  *out = p;
  return TRACE_OK;
}
}  // namespace

extern "C" int trace_static_scratch_init(void) {
  float* p = nullptr;
  return static_scratch(&p);
}

extern "C" int64_t trace_static_scratch_bytes(void) { return (int64_t)kStaticScratchFloats * sizeof(float); }

extern "C" int trace_guide_agent_collision_static(const float* traj, float* grad_traj, float* loss,
                                                  const float* world_from_agent, const float* radius,
                                                  const float* disk_x, int num_disks,
                                                  float buffer_dist, const int* scene_start,
                                                  const int* scene_size, const int* in_mask,
                                                  const float* scale, int n_agents, int num_samp,
                                                  int T, void* stream) {
  TRACE_REQUIRE(n_agents >= 0 && num_samp > 0 && T > 0 && T <= 64,
                "trace_guide_agent_collision_static: bad shape n_agents=%d num_samp=%d T=%d", n_agents,
                num_samp, T);
  const long long need = (long long)n_agents * num_samp * T * 4;
  TRACE_REQUIRE(need <= (long long)kStaticScratchFloats,
                "trace_guide_agent_collision_static: %d agents x %d samples x %d steps need %lld scratch "
                "floats, the static scratch holds %d (use trace_guide_agent_collision with a workspace)",
                n_agents, num_samp, T, need, kStaticScratchFloats);
  float* ws = nullptr;
  int rc = static_scratch(&ws);
  if (rc) return rc;
  // every element the second kernel reads is written by the first one of the same call: nothing
  // of the previous call's contents has to be cleared
  return trace_guide_agent_collision(traj, grad_traj, loss, world_from_agent, radius, disk_x, num_disks,
                                     buffer_dist, scene_start, scene_size, in_mask, scale, n_agents,
                                     num_samp, T, ws, stream);
}

static int map_collision_impl(const float* traj, float* grad_traj, float* loss, const uint8_t* drivable,
                              const float* extent, const float* raster_from_agent, const float* local_pts,
                              int n_pts, const float* scale, int H, int W, int M, int num_samp, int T,
                              float* workspace, int* argmin, void* stream) {
  GUIDE_COMMON_CHECK("trace_guide_map_collision");
  TRACE_REQUIRE(drivable && extent && raster_from_agent && local_pts && workspace,
                "trace_guide_map_collision: null table");
  TRACE_REQUIRE(n_pts >= 1 && n_pts <= 128, "map collision: %d sample points unsupported (<=128)",
                n_pts);
  int n = M * T;
  hipLaunchKernelGGL(map_collision_kernel, dim3((n + 3) / 4), block, 0, st, traj, grad_traj,
                     workspace, drivable, extent, raster_from_agent, local_pts, n_pts, scale, H, W, M,
                     num_samp, T, argmin);
  int rc = launch_status("map_collision_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(sum_over_time_kernel, grid, block, 0, st, workspace, loss, scale, M, num_samp,
                     T);
  return launch_status("sum_over_time_kernel");
}

extern "C" int trace_guide_map_collision(const float* traj, float* grad_traj, float* loss,
                                         const uint8_t* drivable, const float* extent,
                                         const float* raster_from_agent, const float* local_pts,
                                         int n_pts, const float* scale, int H, int W, int M,
                                         int num_samp, int T, float* workspace, void* stream) {
  return map_collision_impl(traj, grad_traj, loss, drivable, extent, raster_from_agent, local_pts, n_pts, scale, H,
                            W, M, num_samp, T, workspace, nullptr, stream);
}

extern "C" int trace_guide_map_collision_argmin(const float* traj, float* grad_traj, float* loss,
                                                const uint8_t* drivable, const float* extent,
                                                const float* raster_from_agent, const float* local_pts,
                                                int n_pts, const float* scale, int H, int W, int M,
                                                int num_samp, int T, float* workspace, int* argmin,
                                                void* stream) {
  TRACE_REQUIRE(argmin, "trace_guide_map_collision_argmin: null argmin");
  return map_collision_impl(traj, grad_traj, loss, drivable, extent, raster_from_agent, local_pts, n_pts, scale, H,
                            W, M, num_samp, T, workspace, argmin, stream);
}

extern "C" int trace_guide_social_group(const float* traj, float* grad_traj, float* loss,
                                        const float* world_from_agent, const int* members,
                                        int group_width, const int* is_leader, const int* pick,
                                        float social_dist, const float* scale, int n_agents,
                                        int num_samp, int T, float* workspace, int* nbr_ws,
                                        float* dist_ws, void* stream) {
  const int M = n_agents * num_samp;
  GUIDE_COMMON_CHECK("trace_guide_social_group");
  TRACE_REQUIRE(world_from_agent && members && is_leader && workspace && nbr_ws && dist_ws,
                "trace_guide_social_group: null table");
  TRACE_REQUIRE(group_width >= 2 && group_width <= n_agents + TRACE_SOCIAL_GROUP_WIDTH,
                "trace_guide_social_group: members table of width %d for %d agents", group_width, n_agents);
  int n = M * T;
  hipLaunchKernelGGL(world_pose_kernel, dim3((n + 255) / 256), dim3(256), 0, st, traj,
                     world_from_agent, workspace, n, num_samp, T);
  int rc = launch_status("world_pose_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(social_neighbor_kernel, grid, block, 0, st, workspace, members, group_width, pick, nbr_ws,
                     dist_ws, scale, M, num_samp, T);
  rc = launch_status("social_neighbor_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(social_group_kernel, grid, block, 0, st, workspace, grad_traj, loss, world_from_agent, members,
                     group_width, is_leader, nbr_ws, dist_ws, social_dist, scale, M, num_samp, T);
  return launch_status("social_group_kernel");
}
