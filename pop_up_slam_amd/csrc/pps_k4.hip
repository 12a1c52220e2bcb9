// pps_k4.hip -- K4: retraction (Slam::self_exmap / apply_exmap, Slam.cpp:216-234) and the chi^2 sweep whose last block
// writes the pinned result record (Slam::weighted_errors / chi2, Slam.cpp:254-268); patch scatter of the difference upload.
#include <algorithm>
#include <cstdlib>

#include <hip/hip_ext.h>

#include "pps_geom.h"
#include "pps_kcommon.h"
#include "pps_k1_body.h"

namespace pps {

// ------------------------------------------------------------------------------------------
// K4: retraction and chi^2
// ------------------------------------------------------------------------------------------
// pose_lin / pose_est / plane_lin / plane_est are passed explicitly: the batched form swaps them per graph
template <bool TRIAL>
__device__ __forceinline__ void body_retract(const DevGraph& d, double* __restrict__ pose_lin, double* __restrict__ pose_est,
                                             double* __restrict__ plane_lin, double* __restrict__ plane_est, int bx, double* red) {
  const int i = bx * blockDim.x + threadIdx.x;
  double dn = 0.0;
  if (i < d.n_pose) {
    double p[7], o[7], dl[6];
    load_pose(pose_lin, d.pose_ld, i, p);
    const int off = d.pose_voff[i];
#pragma unroll
    for (int k = 0; k < 6; k++) { dl[k] = d.delta[off + k]; dn += dl[k] * dl[k]; }
    pose_exmap(p, dl, o);
    if (TRIAL) {
#pragma unroll
      for (int k = 0; k < 7; k++) { pose_est[(size_t)k * d.pose_ld + i] = p[k]; pose_lin[(size_t)k * d.pose_ld + i] = o[k]; }
    } else {
#pragma unroll
      for (int k = 0; k < 7; k++) pose_est[(size_t)k * d.pose_ld + i] = o[k];
    }
  } else if (i < d.n_pose + d.n_plane) {
    const int l = i - d.n_pose;
    double p[4], o[4], dl[3];
    load_plane(plane_lin, d.plane_ld, l, p);
    const int off = d.plane_voff[l];
#pragma unroll
    for (int k = 0; k < 3; k++) { dl[k] = d.delta[off + k]; dn += dl[k] * dl[k]; }
    plane_exmap(p, dl, o);
    if (TRIAL) {
#pragma unroll
      for (int k = 0; k < 4; k++) { plane_est[(size_t)k * d.plane_ld + l] = p[k]; plane_lin[(size_t)k * d.plane_ld + l] = o[k]; }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) plane_est[(size_t)k * d.plane_ld + l] = o[k];
    }
  }
  // |delta|^2 partial of this block (summed by the last block of the following k_chi2)
#pragma unroll
  for (int o2 = 32; o2 > 0; o2 >>= 1) dn += __shfl_down(dn, o2, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dn;
  __syncthreads();
  if (threadIdx.x == 0) d.dn_partials[bx] = red[0] + red[1] + red[2] + red[3];
}

template <bool TRIAL>
__global__ __launch_bounds__(256) void k_retract(DevGraph d) {
  __shared__ double red[4];
  body_retract<TRIAL>(d, d.pose_lin, d.pose_est, d.plane_lin, d.plane_est, blockIdx.x, red);
}

hipError_t launch_retract_trial(const DevGraph& d, hipStream_t st) {
  const int n = d.n_pose + d.n_plane;
  if (n == 0) return hipSuccess;
  PPS_LAUNCH(k_retract<true>, dim3(cdiv(n, 256)), dim3(256), 0, st, d);
  return hipGetLastError();
}
hipError_t launch_retract_apply(const DevGraph& d, hipStream_t st) {
  const int n = d.n_pose + d.n_plane;
  if (n == 0) return hipSuccess;
  PPS_LAUNCH(k_retract<false>, dim3(cdiv(n, 256)), dim3(256), 0, st, d);
  return hipGetLastError();
}

constexpr int kChiBlock = 256;

// sum of the nb block partials and of the n_dn |delta|^2 partials -> the 32-byte result record (one 256-thread block)
__device__ __forceinline__ void chi2_finish(const DevGraph& d, int nb, int n_dn, double* __restrict__ out, double seq) {
  double cs = 0.0, dn = 0.0;
  for (int i = threadIdx.x; i < nb; i += kChiBlock) cs += __hip_atomic_load(&d.chi2_partials[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int i = threadIdx.x; i < n_dn; i += kChiBlock) dn += __hip_atomic_load(&d.dn_partials[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { cs += __shfl_down(cs, o, 64); dn += __shfl_down(dn, o, 64); }
  __shared__ double red2[2][kChiBlock / 64];
  if ((threadIdx.x & 63) == 0) { red2[0][threadIdx.x >> 6] = cs; red2[1][threadIdx.x >> 6] = dn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b2 = 0.0;
    for (int k = 0; k < kChiBlock / 64; k++) { a += red2[0][k]; b2 += red2[1][k]; }
    const double npd = __hip_atomic_load(&d.result_dev[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    d.result_dev[0] = a; d.result_dev[1] = b2; d.result_dev[2] = 0.0;   // the flag belongs to the solve before this record
    out[0] = a; out[1] = b2; out[2] = npd;                 // `out` is pinned host memory: no copy kernel
    // the sequence number goes last, with system-scope release: the host polls it instead of paying a stream sync
    __hip_atomic_store(&out[3], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// bx: block within the graph, nb: blocks of the graph (TICKET: the one that draws the last ticket reduces)
// TOTAL > 0 (fused trial): the ticket counts TOTAL blocks -- the chi2 blocks and the retraction blocks of the same launch
template <bool TICKET = true, bool APPLY = false>
__device__ __forceinline__ void body_chi2(const DevGraph& d, const double* __restrict__ pose,
                                          const double* __restrict__ plane, int nb_obs, int nb_odo, int nb_pp,
                                          int n_dn, double* __restrict__ out, double seq, int bx, int nb, int total_blocks = 0) {
  __shared__ double red[kChiBlock / 64];
  int b = bx;
  double s = 0.0;
  if (b < nb_obs) {
    const int i = b * kChiBlock + threadIdx.x;
    if (i < d.n_obs) {
      double pz[7], pl[4], ms[4], w[6], e[3], r[3];
      fetch_pose<APPLY>(d, pose, d.obs_pose[i], pz);
      fetch_plane<APPLY>(d, plane, d.obs_plane[i], pl);
      if (i < d.n_obs_fixed) load_soa<4>(d.obs_meas, d.obs_ld, i, ms);
      else {                                  // Pose3d_Plane3d_Factor2: re-pop the measurement at this pose
        double ray[6];
        load_soa<6>(d.obs_ray, d.n_obs - d.n_obs_fixed, i - d.n_obs_fixed, ray);
        repop_wall_plane(pz, ray, ms);
      }
      load_soa<6>(d.obs_w, d.obs_ld, i, w);
      res_plane_obs(pz, pl, ms, e);
      whiten<3>(w, e, r);
      s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    }
  } else if ((b -= nb_obs) < nb_odo) {
    const int i = b * kChiBlock + threadIdx.x;
    if (i < d.n_odo) {
      double p1[7], p2[7], ms[6], w[21], e[6], r[6];
      fetch_pose<APPLY>(d, pose, d.odo_a[i], p1);
      fetch_pose<APPLY>(d, pose, d.odo_b[i], p2);
      load_soa<6>(d.odo_meas, d.odo_ld, i, ms);
      load_soa<21>(d.odo_w, d.odo_ld, i, w);
      res_odometry(p1, p2, ms, e);
      whiten<6>(w, e, r);
#pragma unroll
      for (int k = 0; k < 6; k++) s += r[k] * r[k];
    }
  } else if ((b -= nb_odo) < nb_pp) {
    const int i = b * kChiBlock + threadIdx.x;
    if (i < d.n_pp) {
      double pz[7], ms[6], w[21], e[6], r[6];
      fetch_pose<APPLY>(d, pose, d.pp_pose[i], pz);
      load_soa<6>(d.pp_meas, d.pp_ld, i, ms);
      load_soa<21>(d.pp_w, d.pp_ld, i, w);
      res_pose_prior(pz, ms, e);
      whiten<6>(w, e, r);
#pragma unroll
      for (int k = 0; k < 6; k++) s += r[k] * r[k];
    }
  } else {
    b -= nb_pp;
    const int i = b * kChiBlock + threadIdx.x;
    if (i < d.n_lp) {
      double pl[4], ms[4], w[6], e[3], r[3];
      fetch_plane<APPLY>(d, plane, d.lp_plane[i], pl);
      load_soa<4>(d.lp_meas, d.lp_ld, i, ms);
      load_soa<6>(d.lp_w, d.lp_ld, i, w);
      res_plane_prior(pl, ms, e);
      whiten<3>(w, e, r);
      s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    }
  }
  // wave reduction (64 lanes), then across the 4 waves
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (!TICKET) {                                          // a batch: the partials are summed by kb_chi2_finish, a kernel boundary later
    if (threadIdx.x == 0) { double t = 0.0; for (int k = 0; k < kChiBlock / 64; k++) t += red[k]; d.chi2_partials[bx] = t; }
    return;
  }
  __shared__ bool last;
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < kChiBlock / 64; k++) t += red[k];
    d.chi2_partials[bx] = t;
    // publish, then take a ticket: the block that draws the last one reduces everything.  (An agent-scope release writes the
    // XCD's L2 back on this chip -- microseconds; fine for the few dozen blocks of one graph, not for the thousands of a batch.)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    last = atomicAdd(d.ticket, 1u) == (unsigned int)((total_blocks > 0 ? total_blocks : nb) - 1);
  }
  __syncthreads();
  if (!last) return;
  if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  __syncthreads();
  chi2_finish(d, nb, n_dn, out, seq);
  if (threadIdx.x == 0) *d.ticket = 0u;
}

__global__ __launch_bounds__(kChiBlock) void k_chi2(DevGraph d, const double* __restrict__ pose,
                                                    const double* __restrict__ plane, int nb_obs, int nb_odo, int nb_pp,
                                                    int n_dn, double* __restrict__ out, double seq) {
  body_chi2(d, pose, plane, nb_obs, nb_odo, nb_pp, n_dn, out, seq, blockIdx.x, gridDim.x);
}

// out <- base (+) delta for one block of nodes; the block's |delta|^2 partial goes to d.dn_partials[bx]
__device__ __forceinline__ void body_retract_to(const DevGraph& d, const double* __restrict__ base_pose, const double* __restrict__ base_plane,
                                                double* __restrict__ out_pose, double* __restrict__ out_plane, int bx, double* red) {
  const int i = bx * blockDim.x + threadIdx.x;
  double dn = 0.0;
  if (i < d.n_pose) {
    double p[7], o[7], dl[6];
    load_pose(base_pose, d.pose_ld, i, p);
    const int off = d.pose_voff[i];
#pragma unroll
    for (int k = 0; k < 6; k++) { dl[k] = d.delta[off + k]; dn += dl[k] * dl[k]; }
    pose_exmap(p, dl, o);
#pragma unroll
    for (int k = 0; k < 7; k++) out_pose[(size_t)k * d.pose_ld + i] = o[k];
  } else if (i < d.n_pose + d.n_plane) {
    const int l = i - d.n_pose;
    double p[4], o[4], dl[3];
    load_plane(base_plane, d.plane_ld, l, p);
    const int off = d.plane_voff[l];
#pragma unroll
    for (int k = 0; k < 3; k++) { dl[k] = d.delta[off + k]; dn += dl[k] * dl[k]; }
    plane_exmap(p, dl, o);
#pragma unroll
    for (int k = 0; k < 4; k++) out_plane[(size_t)k * d.plane_ld + l] = o[k];
  }
#pragma unroll
  for (int o2 = 32; o2 > 0; o2 >>= 1) dn += __shfl_down(dn, o2, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dn;
  __syncthreads();
  if (threadIdx.x == 0) d.dn_partials[bx] = red[0] + red[1] + red[2] + red[3];
}

// Both trials of a dual solve in ONE launch (grid y = trial): blocks [0, nb_ret) write out_k <- base (+) delta_k and their
// |delta|^2 partials, the others reduce chi2 at base (+) delta_k computed on the fly -- they do not wait for the retraction,
// which only the next linearisation reads.  All of them take a ticket; the last one writes the result record.
__global__ __launch_bounds__(kChiBlock) void k_trial_dual(DevGraph d, DualAlt alt, const double* __restrict__ base_pose, const double* __restrict__ base_plane,
                                                          double* __restrict__ out_pose, double* __restrict__ out_plane, double* __restrict__ out_pose1,
                                                          double* __restrict__ out_plane1, int nb_ret, int nb_obs, int nb_odo, int nb_pp, int nb_chi,
                                                          double* __restrict__ out, double seq, double* __restrict__ out1, double seq1) {
  if (blockIdx.y) {
    d.delta = alt.delta; d.chi2_partials = alt.chi2_partials; d.dn_partials = alt.dn_partials; d.ticket = alt.ticket; d.result_dev = alt.result_dev;
    out_pose = out_pose1; out_plane = out_plane1; out = out1; seq = seq1;
  }
  const int total = nb_ret + nb_chi;
  if ((int)blockIdx.x >= nb_ret) {
    body_chi2<true, true>(d, base_pose, base_plane, nb_obs, nb_odo, nb_pp, nb_ret, out, seq, (int)blockIdx.x - nb_ret, nb_chi, total);
    return;
  }
  __shared__ double red[4];
  body_retract_to(d, base_pose, base_plane, out_pose, out_plane, blockIdx.x, red);
  // the same publish / ticket protocol as a chi2 block (body_retract_to has written this block's |delta|^2 partial)
  __shared__ bool last;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    last = atomicAdd(d.ticket, 1u) == (unsigned int)(total - 1);
  }
  __syncthreads();
  if (!last) return;
  if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  __syncthreads();
  chi2_finish(d, nb_chi, nb_ret, out, seq);
  if (threadIdx.x == 0) *d.ticket = 0u;
}

hipError_t launch_trial_dual(const DevGraph& d, const DualAlt& alt, const double* base_pose, const double* base_plane, double* out_pose0,
                             double* out_plane0, double* out_pose1, double* out_plane1, double* host_result0, double seq0, double* host_result1,
                             double seq1, hipStream_t st, int n_trials) {
  const int n = d.n_pose + d.n_plane;
  const int nb_obs = cdiv(d.n_obs, kChiBlock), nb_odo = cdiv(d.n_odo, kChiBlock), nb_pp = cdiv(d.n_pp, kChiBlock), nb_lp = cdiv(d.n_lp, kChiBlock);
  const int nb = nb_obs + nb_odo + nb_pp + nb_lp;
  if (nb == 0 || n == 0) return hipErrorInvalidValue;
  const int nb_ret = cdiv(n, 256);
  PPS_LAUNCH(k_trial_dual, dim3(nb_ret + nb, n_trials), dim3(kChiBlock), 0, st, d, alt, base_pose, base_plane, out_pose0, out_plane0, out_pose1, out_plane1, nb_ret,
             nb_obs, nb_odo, nb_pp, nb, host_result0, seq0, host_result1, seq1);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Round 6: the trials AND the next linearisation in one launch.  Grid y = trial (damping value); blocks [0, nb_ret + nb_chi) of a row are
// k_trial_dual's -- retraction into the trial's copy of the state, chi2 at x (+) delta on the fly, ticket, result record --; row sl.which
// has more blocks behind them: K1 in its lane form at that trial's point x (+) delta, also evaluated on the fly, writing Jacobians, product
// records and direct H blocks into the spare set (SpecLin).  Nothing in the launch waits for anything else in it: the sweep that used to
// start after the trial kernel had ended -- and after its verdict had been tested by a guard -- runs beside it.  Whether that linearisation
// is the one LM wants the host decides from the records as before (Optimizer.cpp:425-458).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kChiBlock) void k_trial_lin(DevGraph d, DualAlt alt, SpecLin sl, const double* __restrict__ base_pose, const double* __restrict__ base_plane,
                                                         double* __restrict__ out_pose, double* __restrict__ out_plane, double* __restrict__ out_pose1,
                                                         double* __restrict__ out_plane1, int nb_ret, int nb_obs, int nb_odo, int nb_pp, int nb_chi,
                                                         double* __restrict__ out, double seq, double* __restrict__ out1, double seq1, int lb_obs, int lb_odo, int lb_pp, int apply) {
  static_assert(kChiBlock == kLanesPerBlock, "one block size for the trial blocks and the lane-form sweep");
  // blocks [0, total) trial 0 | [total, 2 total) trial 1 | the rest: the sweep at trial sl.which's point
  const int total = nb_ret + nb_chi;
  const bool sweep = (int)blockIdx.x >= 2 * total;
  const int y = sweep ? sl.which : ((int)blockIdx.x >= total ? 1 : 0);
  const int bx0 = sweep ? (int)blockIdx.x - 2 * total : (int)blockIdx.x - y * total;
  if (y) {
    d.delta = alt.delta; d.chi2_partials = alt.chi2_partials; d.dn_partials = alt.dn_partials; d.ticket = alt.ticket; d.result_dev = alt.result_dev;
    out_pose = out_pose1; out_plane = out_plane1; out = out1; seq = seq1;
  }
  if (sweep) {
    d.J = sl.J; d.P = sl.P; d.H = sl.H; d.Hf = sl.Hf;
    body_linearize_lanes(d, base_pose, base_plane, lb_obs, lb_odo, lb_pp, bx0, apply != 0);      // (1: see k_linearize_lanes)
    return;
  }
  if (bx0 >= nb_ret) {
    body_chi2<true, true>(d, base_pose, base_plane, nb_obs, nb_odo, nb_pp, nb_ret, out, seq, bx0 - nb_ret, nb_chi, total);
    return;
  }
  __shared__ double red[4];
  body_retract_to(d, base_pose, base_plane, out_pose, out_plane, bx0, red);
  __shared__ bool last;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    last = atomicAdd(d.ticket, 1u) == (unsigned int)(total - 1);
  }
  __syncthreads();
  if (!last) return;
  if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  __syncthreads();
  chi2_finish(d, nb_chi, nb_ret, out, seq);
  if (threadIdx.x == 0) *d.ticket = 0u;
}

int trial_lin_waves(const DevGraph& d) {
  const int nb = cdiv(d.n_obs, kChiBlock) + cdiv(d.n_odo, kChiBlock) + cdiv(d.n_pp, kChiBlock) + cdiv(d.n_lp, kChiBlock) + cdiv(d.n_pose + d.n_plane, 256);
  const int lb = cdiv(d.n_obs_fixed, kObsPerBlock) + cdiv(d.n_odo, kFactorsPerBlock) + cdiv(d.n_pp, kFactorsPerBlock) + cdiv(d.n_lp, kFactorsPerBlock);
  return (2 * nb + lb) * (kChiBlock / 64);
}
bool trial_lin_ok(const DevGraph& d, int mode) { return k1_lane_form(d, mode) && d.n_obs == d.n_obs_fixed && d.P != nullptr; }

hipError_t launch_trial_lin(const DevGraph& d, const DualAlt& alt, const SpecLin& sl, const double* base_pose, const double* base_plane, double* out_pose0,
                            double* out_plane0, double* out_pose1, double* out_plane1, double* host_result0, double seq0, double* host_result1,
                            double seq1, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
  const int n = d.n_pose + d.n_plane;
  const int nb_obs = cdiv(d.n_obs, kChiBlock), nb_odo = cdiv(d.n_odo, kChiBlock), nb_pp = cdiv(d.n_pp, kChiBlock), nb_lp = cdiv(d.n_lp, kChiBlock);
  const int nb = nb_obs + nb_odo + nb_pp + nb_lp;
  if (nb == 0 || n == 0) return hipErrorInvalidValue;
  const int nb_ret = cdiv(n, 256);
  const int lb_obs = cdiv(d.n_obs_fixed, kObsPerBlock), lb_odo = cdiv(d.n_odo, kFactorsPerBlock), lb_pp = cdiv(d.n_pp, kFactorsPerBlock),
            lb_lp = cdiv(d.n_lp, kFactorsPerBlock);
  const int lb_all = lb_obs + lb_odo + lb_pp + lb_lp;
  PPS_LAUNCH_EV(ev0, ev1, k_trial_lin, dim3(2 * (nb_ret + nb) + lb_all), dim3(kChiBlock), 0, st, d, alt, sl, base_pose, base_plane, out_pose0,
                out_plane0, out_pose1, out_plane1, nb_ret, nb_obs, nb_odo, nb_pp, nb, host_result0, seq0, host_result1, seq1, lb_obs, lb_odo, lb_pp, 1);
  return hipGetLastError();
}

hipError_t launch_chi2(const DevGraph& d, bool at_estimate, double* host_result, double seq, hipStream_t st) {
  const int nb_obs = cdiv(d.n_obs, kChiBlock), nb_odo = cdiv(d.n_odo, kChiBlock), nb_pp = cdiv(d.n_pp, kChiBlock),
            nb_lp = cdiv(d.n_lp, kChiBlock);
  const int nb = nb_obs + nb_odo + nb_pp + nb_lp;
  const double* pose = at_estimate ? d.pose_est : d.pose_lin;
  const double* plane = at_estimate ? d.plane_est : d.plane_lin;
  if (nb == 0) return hipErrorInvalidValue;
  const int n_dn = cdiv(d.n_pose + d.n_plane, 256);
  PPS_LAUNCH(k_chi2, dim3(nb), dim3(kChiBlock), 0, st, d, pose, plane, nb_obs, nb_odo, nb_pp, n_dn, host_result, seq);
  return hipGetLastError();
}

// chi2 of the trial of the one-step loop: after launch_retract_trial est holds the point x the step started from and lin its
// result; the residuals are evaluated at est (+) delta computed on the spot -- the same arithmetic, in the same kernel body, as
// the fused trial of the dual loop and the batch, so that every form of the LM loop reports the same bits
__global__ __launch_bounds__(kChiBlock) void k_chi2_trial(DevGraph d, int nb_obs, int nb_odo, int nb_pp, int n_dn, double* __restrict__ out, double seq) {
  body_chi2<true, true>(d, d.pose_est, d.plane_est, nb_obs, nb_odo, nb_pp, n_dn, out, seq, blockIdx.x, gridDim.x);
}

hipError_t launch_chi2_trial(const DevGraph& d, double* host_result, double seq, hipStream_t st) {
  const int nb_obs = cdiv(d.n_obs, kChiBlock), nb_odo = cdiv(d.n_odo, kChiBlock), nb_pp = cdiv(d.n_pp, kChiBlock),
            nb_lp = cdiv(d.n_lp, kChiBlock);
  const int nb = nb_obs + nb_odo + nb_pp + nb_lp;
  if (nb == 0) return hipErrorInvalidValue;
  PPS_LAUNCH(k_chi2_trial, dim3(nb), dim3(kChiBlock), 0, st, d, nb_obs, nb_odo, nb_pp, cdiv(d.n_pose + d.n_plane, 256), host_result, seq);
  return hipGetLastError();
}

// ---- batched forms ----
__global__ __launch_bounds__(kChiBlock) void kb_chi2(BatchArgs a, int slot) {
  PPS_BATCH_PROLOGUE(BF_ACTIVE)
  const int nb_obs = dcdiv(d.n_obs, kChiBlock), nb_odo = dcdiv(d.n_odo, kChiBlock), nb_pp = dcdiv(d.n_pp, kChiBlock),
            nb_lp = dcdiv(d.n_lp, kChiBlock);
  const int nb = nb_obs + nb_odo + nb_pp + nb_lp;
  if ((int)blockIdx.x >= nb) return;
  body_chi2<false>(d, pose_lin, plane_lin, nb_obs, nb_odo, nb_pp, dcdiv(d.n_pose + d.n_plane, 256),
                   a.results + (size_t)(a.alt ? 12 : 8) * (size_t)(a.b0 + b) + 4 * slot, a.seq, blockIdx.x, nb);
}

// the result records of a batch: one block per graph (and per trial: grid z) sums the partials of the sweep before it
__global__ __launch_bounds__(kChiBlock) void kb_chi2_finish(BatchArgs a, int slot, int dual) {
  PPS_BATCH_PROLOGUE(BF_ACTIVE)
  const int nb = dcdiv(d.n_obs, kChiBlock) + dcdiv(d.n_odo, kChiBlock) + dcdiv(d.n_pp, kChiBlock) + dcdiv(d.n_lp, kChiBlock);
  const int n_dn = dcdiv(d.n_pose + d.n_plane, 256);
  if (!dual) { chi2_finish(d, nb, n_dn, a.results + (size_t)(a.alt ? 12 : 8) * (size_t)(a.b0 + b) + 4 * slot, a.seq); return; }
  DevGraph d2 = d;
  if (blockIdx.z) {
    const BatchAlt al = load_alt(a.alt + a.b0 + b);
    d2.chi2_partials = al.chi2_partials; d2.dn_partials = al.dn_partials; d2.result_dev = al.result_dev;
  }
  chi2_finish(d2, nb, n_dn, a.results + 12 * (size_t)(a.b0 + b) + 4 * (1 + blockIdx.z), a.seq);
}

hipError_t launch_batch_chi2(const BatchArgs& a, const BatchGeom& g, int slot, hipStream_t st) {
  if (g.chi2 <= 0) return hipErrorInvalidValue;
  PPS_LAUNCH(kb_chi2, dim3(g.chi2, a.n), dim3(kChiBlock), 0, st, a, slot);
  PPS_LAUNCH(kb_chi2_finish, dim3(1, a.n), dim3(kChiBlock), 0, st, a, slot, 0);
  return hipGetLastError();
}

// ---- dual-lambda batch (BatchAlt): both trials of a graph in one launch, grid z = 0 / 1 ----
__global__ __launch_bounds__(64) void kb_begin_dual(BatchArgs a) {
  PPS_BATCH_PROLOGUE(BF_ACTIVE)
  const BatchAlt al = load_alt(a.alt + a.b0 + b);
  if (threadIdx.x < 4) { d.result_dev[threadIdx.x] = 0.0; al.result_dev[threadIdx.x] = 0.0; }
}

// both trials of every graph of the chunk in one launch (grid z = trial): blocks [0, nb_ret) of a graph's row write the trial's copy of the
// state, x (+) delta_z, and their |delta|^2 partials; the rest reduce chi2 at x (+) delta_z computed on the spot, like k_trial_dual (the stored
// copy is for the next K1).  kb_chi2_finish sums the partials a kernel boundary later.  (Two launches up to round 5: kb_retract_dual, kb_chi2_dual.)
__global__ __launch_bounds__(kChiBlock) void kb_trial_dual(BatchArgs a, int nb_ret) {
  __shared__ double red[4];
  PPS_BATCH_PROLOGUE(BF_ACTIVE)
  const BatchAlt al = load_alt(a.alt + a.b0 + b);
  const int xs = a.xsel[b], z = blockIdx.z;
  const int ts = (xs + 1 + z) % 3;
  DevGraph d2 = d;
  if (z) { d2.delta = al.delta; d2.chi2_partials = al.chi2_partials; d2.dn_partials = al.dn_partials; d2.ticket = al.ticket; d2.result_dev = al.result_dev; }
  if ((int)blockIdx.x < nb_ret) {
    if ((int)blockIdx.x * 256 >= d.n_pose + d.n_plane) return;
    body_retract_to(d2, pose_lin, plane_lin, sel3(al.pose, ts), sel3(al.plane, ts), blockIdx.x, red);
    return;
  }
  const int nb_obs = dcdiv(d.n_obs, kChiBlock), nb_odo = dcdiv(d.n_odo, kChiBlock), nb_pp = dcdiv(d.n_pp, kChiBlock),
            nb_lp = dcdiv(d.n_lp, kChiBlock);
  const int nb = nb_obs + nb_odo + nb_pp + nb_lp;
  const int bx = (int)blockIdx.x - nb_ret;
  if (bx >= nb) return;
  body_chi2<false, true>(d2, pose_lin, plane_lin, nb_obs, nb_odo, nb_pp, dcdiv(d.n_pose + d.n_plane, 256),
                         a.results + 12 * (size_t)(a.b0 + b) + 4 * (1 + z), a.seq, bx, nb);
}

// pps_multi_restore_state: the snapshots of n graphs back into their estimates, one launch (tab: n records in pinned host memory)
__global__ __launch_bounds__(256) void kb_restore(const RestoreRec* __restrict__ tab) {
  const RestoreRec r = tab[blockIdx.y];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < r.n; i += 256LL * gridDim.x) r.dst[i] = r.src[i];
}

hipError_t launch_batch_restore(const RestoreRec* tab, int n, hipStream_t st) {
  PPS_LAUNCH(kb_restore, dim3(8, n), dim3(256), 0, st, tab);
  return hipGetLastError();
}

hipError_t launch_batch_begin_dual(const BatchArgs& a, const BatchGeom& g, hipStream_t st) {
  (void)g;
  PPS_LAUNCH(kb_begin_dual, dim3(1, a.n), dim3(64), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_batch_trial_dual(const BatchArgs& a, const BatchGeom& g, hipStream_t st) {
  if (g.chi2 <= 0) return hipErrorInvalidValue;
  static_assert(kChiBlock == 256, "the retraction blocks of kb_trial_dual are 256 nodes wide");
  PPS_LAUNCH(kb_trial_dual, dim3(g.retract + g.chi2, a.n, 2), dim3(kChiBlock), 0, st, a, g.retract);
  PPS_LAUNCH(kb_chi2_finish, dim3(1, a.n, 2), dim3(kChiBlock), 0, st, a, 0, 1);
  return hipGetLastError();
}

// patch upload of a re-uploaded topology (pps_upload.cpp: flush_uploads): piece i of `src` (the pinned mirror of the arena, read over the bus)
// -> its place in the arena; the table is in pinned host memory as well
__global__ __launch_bounds__(256) void k_scatter_patches(const char* __restrict__ table, const char* __restrict__ src_base, char* __restrict__ arena) {
  const long long* tab = reinterpret_cast<const long long*>(table) + 4 * (size_t)blockIdx.x;
  const long long dst = tab[0], src = tab[1], len = tab[2];
  if (tab[3]) {                  // an exact piece: 8-byte units (refreshed measurements sit right next to it)
    const long long* s8 = reinterpret_cast<const long long*>(src_base + src);
    long long* d8 = reinterpret_cast<long long*>(arena + dst);
    for (long long i = (long long)blockIdx.y * 256 + threadIdx.x; i < len / 8; i += (long long)gridDim.y * 256) d8[i] = s8[i];
    return;
  }
  const int4* s4 = reinterpret_cast<const int4*>(src_base + src);
  int4* d4 = reinterpret_cast<int4*>(arena + dst);
  for (long long i = (long long)blockIdx.y * 256 + threadIdx.x; i < len / 16; i += (long long)gridDim.y * 256) d4[i] = s4[i];
}

hipError_t launch_scatter_patches(const char* table, const char* src_base, int n_patches, char* arena, hipStream_t st) {
  if (n_patches <= 0) return hipSuccess;
  PPS_LAUNCH(k_scatter_patches, dim3(n_patches, 8), dim3(256), 0, st, table, src_base, arena);
  return hipGetLastError();
}

hipError_t launch_clear_status(const DevGraph& d, hipStream_t st) {
  return hipMemsetAsync(d.result_dev, 0, 4 * sizeof(double), st);
}

// ---- diagnostic: the retraction of ONE node outside any graph (pps_debug_exmap) ----
// n nodes of one kind through the functions k_retract / k_trial_dual call: pose_exmap (Pose3d::exmap, Pose3d.h:131-136) or
// plane_exmap (Plane3d::exmap_3dof, isam_plane3d.h:101-127)
__global__ void k_debug_exmap(int kind, int n, const double* __restrict__ x, const double* __restrict__ dl, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (kind == 0) {
    double p[7], d[6], o[7];
    for (int k = 0; k < 7; k++) p[k] = x[i * 7 + k];
    for (int k = 0; k < 6; k++) d[k] = dl[i * 6 + k];
    pose_exmap(p, d, o);
    for (int k = 0; k < 7; k++) out[i * 7 + k] = o[k];
  } else {
    double p[4], d[3], o[4];
    for (int k = 0; k < 4; k++) p[k] = x[i * 4 + k];
    for (int k = 0; k < 3; k++) d[k] = dl[i * 3 + k];
    plane_exmap(p, d, o);
    for (int k = 0; k < 4; k++) out[i * 4 + k] = o[k];
  }
}

// returns a hipError_t (0 = ok), -1 for bad arguments
int debug_exmap(int kind, int n, const double* x_host, const double* delta_host, double* out_host) {
  if ((kind != 0 && kind != 1) || n < 1) return -1;
  const size_t nx = (size_t)n * (kind == 0 ? 7 : 4), nd = (size_t)n * (kind == 0 ? 6 : 3);
  double *x = nullptr, *dl = nullptr, *o = nullptr;
  hipError_t e = hipMalloc(&x, nx * 8);
  if (e == hipSuccess) e = hipMalloc(&dl, nd * 8);
  if (e == hipSuccess) e = hipMalloc(&o, nx * 8);
  if (e == hipSuccess) e = hipMemcpy(x, x_host, nx * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dl, delta_host, nd * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_debug_exmap, dim3(cdiv(n, 64)), dim3(64), 0, 0, kind, n, x, dl, o);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out_host, o, nx * 8, hipMemcpyDeviceToHost);
  (void)hipFree(x); (void)hipFree(dl); (void)hipFree(o);
  return (int)e;
}
}  // namespace pps
