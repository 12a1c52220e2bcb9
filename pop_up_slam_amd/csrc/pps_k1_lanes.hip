// pps_k1_lanes.hip -- K1, the latency form: k_linearize_lanes (19 lanes per plane observation), its batched and replicated-edge variants.
// Bodies: pps_k1_body.h.
//
// THIS FILE AND pps_k4.hip ARE COMPILED WITHOUT MULTIPLY-ADD CONTRACTION (csrc/Makefile).  Several kernels evaluate the same factor -- the plain
// sweep, the batched one, the sweep inside the fused trial + linearisation launch (k_trial_lin, pps_k4.hip) -- and an LM run
// must not depend on which of them produced a Jacobian: pps_multi is compared bit for bit with single handles, the loop forms with each other.
// With -ffp-contract=fast the compiler decides per inlined copy which multiply-adds to fuse; round 6 saw that decision change twice (a second
// template instantiation of the sweep; new kernels added to the translation unit) and move chi2 in the 13th digit.  Without contraction every
// operation is an IEEE operation wherever it is inlined.  (A form with three / four lanes per observation, built in the same round for C3,
// produced the lane form's LM trace bit for bit under this regime -- and was not faster: DESIGN.md section 8.)  The lane form is bound by
// memory round trips and dependent-operation latency, not by instruction issue, so the unfused multiply-adds cost nothing measurable (C2 K1
// 9.3 us either way); the throughput form (pps_k1.hip: one thread per factor, issue bound) keeps contraction.  H products (PPS_MAC) are explicit fused multiply-adds in every file.
#include <hip/hip_ext.h>

#include "pps_k1_body.h"

namespace pps {

// apply: always 0 here, and a kernel ARGUMENT on purpose -- see body_linearize_lanes: the compiler must not know it, so that this kernel, the
// batched one and the fused trial + linearisation launch (k_trial_lin, pps_k4.hip: apply = 1) all compile the same sweep
__global__ __launch_bounds__(kLanesPerBlock) void k_linearize_lanes(DevGraph d, const double* __restrict__ pose,
                                                                    const double* __restrict__ plane, int nb_obs, int nb_odo,
                                                                    int nb_pp, LinGuard gd, int apply) {
  if (!lin_guard(gd, pose, plane)) return;
  // (one block of work per workgroup: a loop over several -- fewer, longer-lived workgroups -- was measured in round 6: the compiler hoists the
  // sweep's invariants out of it, 291 registers and one wave per SIMD instead of 124 and four)
  body_linearize_lanes(d, pose, plane, nb_obs, nb_odo, nb_pp, blockIdx.x, apply != 0);
}


hipError_t launch_linearize_lanes(const DevGraph& d, const double* pose, const double* plane, const LinGuard& gd, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
  const int lb_obs = cdiv(d.n_obs_fixed, kObsPerBlock), lb_odo = cdiv(d.n_odo, kFactorsPerBlock), lb_pp = cdiv(d.n_pp, kFactorsPerBlock),
            lb_lp = cdiv(d.n_lp, kFactorsPerBlock);
  PPS_LAUNCH_EV(ev0, ev1, k_linearize_lanes, dim3(lb_obs + lb_odo + lb_pp + lb_lp), dim3(kLanesPerBlock), 0, st, d, pose, plane, lb_obs, lb_odo, lb_pp, gd, 0);
  return hipGetLastError();
}

// the lane-parallel numeric form over the replicated edges (mode 2 of the sweep benchmark): PART 0 plane observations (19 lanes
// each), PART 1 odometry edges (32 lanes each)
template <int PART>
__global__ __launch_bounds__(kLanesPerBlock) void k_sweep_bench_lanes(DevGraph d, double* __restrict__ Jbig, int lb_obs_per, int lb_odo_per, int apply) {
  const int per = PART == 0 ? lb_obs_per : lb_odo_per;
  const int rep = blockIdx.x / per;
  const int b = blockIdx.x % per + (PART == 0 ? 0 : lb_obs_per);
  const size_t slab = (size_t)d.n_obs * 30 + (size_t)d.n_odo * 78;
  DevGraph r = d;                       // replica `rep` reads shifted copies of the edge arrays and writes its own J slab
  r.J = Jbig + (size_t)rep * slab; r.joff_obs = 0; r.joff_odo = (int64_t)d.n_obs * 30;
  r.obs_meas = d.obs_meas + (size_t)rep * 4 * d.obs_ld; r.obs_w = d.obs_w + (size_t)rep * 6 * d.obs_ld;
  r.obs_pose = d.obs_pose + (size_t)rep * d.n_obs; r.obs_plane = d.obs_plane + (size_t)rep * d.n_obs;
  r.odo_meas = d.odo_meas + (size_t)rep * 6 * d.odo_ld; r.odo_w = d.odo_w + (size_t)rep * 21 * d.odo_ld;
  r.odo_a = d.odo_a + (size_t)rep * d.n_odo; r.odo_b = d.odo_b + (size_t)rep * d.n_odo;
  r.n_obs_fixed = d.n_obs;
  r.obs_dir = nullptr; r.P = nullptr;
  body_linearize_lanes(r, d.pose_lin, d.plane_lin, lb_obs_per, lb_odo_per, 0, b, apply != 0);
}


hipError_t launch_sweep_bench_lanes(const DevGraph& d, int replicas, double* Jbig, int part, hipStream_t st) {
  const int lb_obs = cdiv(d.n_obs, kObsPerBlock), lb_odo = cdiv(d.n_odo, kFactorsPerBlock);
  if (lb_obs && part != 1) PPS_LAUNCH(k_sweep_bench_lanes<0>, dim3(lb_obs * replicas), dim3(kLanesPerBlock), 0, st, d, Jbig, lb_obs, lb_odo, 0);
  if (lb_odo && part != 0) PPS_LAUNCH(k_sweep_bench_lanes<1>, dim3(lb_odo * replicas), dim3(kLanesPerBlock), 0, st, d, Jbig, lb_obs, lb_odo, 0);
  return hipGetLastError();
}

// ---- batched form ----
__global__ __launch_bounds__(kLanesPerBlock) void kb_linearize_lanes(BatchArgs a) {
  PPS_BATCH_PROLOGUE(BF_ACTIVE | BF_RELIN)
  const int nb_obs = dcdiv(d.n_obs_fixed, kObsPerBlock), nb_odo = dcdiv(d.n_odo, kFactorsPerBlock),
            nb_pp = dcdiv(d.n_pp, kFactorsPerBlock), nb_lp = dcdiv(d.n_lp, kFactorsPerBlock);
  if ((int)blockIdx.x >= nb_obs + nb_odo + nb_pp + nb_lp) return;
  body_linearize_lanes(d, pose_lin, plane_lin, nb_obs, nb_odo, nb_pp, blockIdx.x, a.lin_apply != 0);      // (always 0: see k_linearize_lanes)
}


hipError_t launch_batch_linearize_lanes(const BatchArgs& a, const BatchGeom& g, hipStream_t st) {
  PPS_LAUNCH(kb_linearize_lanes, dim3(g.lin_blocks, a.n), dim3(kLanesPerBlock), 0, st, a);
  return hipGetLastError();
}

}  // namespace pps
