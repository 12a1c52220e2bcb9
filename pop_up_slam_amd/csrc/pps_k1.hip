// pps_k1.hip -- K1 kernels: k_linearize<MODE,PART>, k_linearize_lanes, k_linearize_repop, the batched forms (blockIdx.y =
// graph) and the replicated-edge sweep of the roofline micro-benchmark (k_sweep_bench).  Bodies: pps_k1_body.h.
#include <atomic>
#include <cstdlib>
#include <mutex>

#include <hip/hip_ext.h>

#include "pps_k1_body.h"

namespace pps {

// the lane form lives in pps_k1_lanes.hip (compiled WITHOUT multiply-add contraction: see there)
hipError_t launch_linearize_lanes(const DevGraph& d, const double* pose, const double* plane, const LinGuard& gd, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);
hipError_t launch_sweep_bench_lanes(const DevGraph& d, int replicas, double* Jbig, int part, hipStream_t st);
hipError_t launch_batch_linearize_lanes(const BatchArgs& a, const BatchGeom& g, hipStream_t st);

constexpr int kObsNumericWaves = 2;      // waves per SIMD of the numeric plane-observation launch of the thread form (see k_linearize_obs_numeric)

template <int MODE, int PART>
__global__ __launch_bounds__(kLinBlock) void k_linearize(DevGraph d, const double* __restrict__ pose,
                                                          const double* __restrict__ plane, int nb_obs, int nb_odo,
                                                          int nb_pp, LinGuard gd) {
  extern __shared__ double lin_lds[];
  if (!lin_guard(gd, pose, plane)) return;
  body_linearize<MODE, PART, PART == 0>(d, pose, plane, nb_obs, nb_odo, nb_pp, blockIdx.x, lin_lds);      // (the plane-observation launch writes the direct H blocks)
}

// The numeric plane-observation launch of the thread-per-factor form on its own: left to itself the compiler gives it 309 vector
// registers + spill moves into the accumulator half (one wave per SIMD); held to two waves per SIMD (256 registers) it is a fifth
// faster on the batched sweep (540 000 edges: 160 -> 126 us; three waves: 134, four: 153).  The odometry launch is fastest alone
// on its SIMD (103 us; 154 with two waves) and keeps the default.
__global__ __launch_bounds__(kLinBlock) __attribute__((amdgpu_waves_per_eu(kObsNumericWaves, kObsNumericWaves)))
void k_linearize_obs_numeric(DevGraph d, const double* __restrict__ pose, const double* __restrict__ plane, int nb_obs, int nb_odo, int nb_pp, LinGuard gd) {
  extern __shared__ double lin_lds[];
  if (!lin_guard(gd, pose, plane)) return;
  body_linearize<0, 0, true>(d, pose, plane, nb_obs, nb_odo, nb_pp, blockIdx.x, lin_lds);
}

// rot_exp / plane_exp of the one step size a numerical Jacobian uses, by the device's own arithmetic, once per device
// (eps arrives as a kernel argument: with a literal the compiler folds sqrt / sincos at build time, with the HOST's libm, which may
// round the last bit differently from the device's own evaluation that a step computed in a lane would see)
__global__ void k_step_constants(double* out, double eps) {
  double ac[2];
  rot_step_quat(ac, eps); out[0] = ac[0]; out[1] = ac[1];
  plane_step_quat(ac, eps); out[2] = ac[0]; out[3] = ac[1];
}
static std::atomic<int> g_step_state[64];          // per device ordinal: 0 unset, 2 ready
static double g_step_ac[64][4];
static std::mutex g_step_mutex;
hipError_t step_constants(double ac[4]) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev &= 63;
  if (g_step_state[dev].load(std::memory_order_acquire) != 2) {
    std::lock_guard<std::mutex> lk(g_step_mutex);
    if (g_step_state[dev].load(std::memory_order_acquire) != 2) {
      double* o = nullptr;
      hipError_t e = hipMalloc(&o, 4 * sizeof(double));
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(k_step_constants, dim3(1), dim3(1), 0, 0, o, kNumDiffEps);
      e = hipGetLastError();
      if (e == hipSuccess) e = hipMemcpy(g_step_ac[dev], o, 4 * sizeof(double), hipMemcpyDeviceToHost);
      (void)hipFree(o);
      if (e != hipSuccess) return e;
      g_step_state[dev].store(2, std::memory_order_release);
    }
  }
  for (int k = 0; k < 4; k++) ac[k] = g_step_ac[dev][k];
  return hipSuccess;
}

static thread_local unsigned long long t_launches = 0;
unsigned long long launch_count() { return t_launches; }
void count_launch() { ++t_launches; }

__global__ __launch_bounds__(64) void k_linearize_repop(DevGraph d, const double* __restrict__ pose,
                                                        const double* __restrict__ plane, LinGuard gd) {
  __shared__ double repop_lds[64 * 31];
  if (!lin_guard(gd, pose, plane)) return;
  body_linearize_repop(d, pose, plane, blockIdx.x, repop_lds);
}

// PPS_K1_THREAD_FORM=1 forces the thread-per-factor kernels on small graphs (parity tests of that form)
bool k1_lane_form(const DevGraph& d, int mode) {
  return mode == 0 && d.n_obs + d.n_odo + d.n_pp + d.n_lp <= kLaneParallelMaxFactors && !(d.sw & SW_K1_THREAD_FORM);
}

// does K1 write the plane observations' product records (and K2 sum them)?  Every form does, except under the test switch above
bool k1_products(const DevGraph& d, int mode) { (void)mode; return !(d.sw & SW_K1_THREAD_FORM); }

// ev0 / ev1 (profiling level 1): the launch is made with hipExtLaunchKernelGGL, whose start / stop events take the DISPATCH's own
// begin / end timestamps -- what rocprofv3 reports as the kernel's duration -- instead of two event records around the launch,
// which also time the event packets themselves (13.6 us against 10.6 us for the C2 sweep inside a solve).
hipError_t launch_linearize(const DevGraph& d, int mode, bool at_estimate, hipStream_t st, const LinGuard* guard, hipEvent_t ev0, hipEvent_t ev1) {
  const LinGuard gd = guard ? *guard : LinGuard{};
  if (d.n_obs > d.n_obs_fixed) {
    PPS_LAUNCH(k_linearize_repop, dim3(cdiv(d.n_obs - d.n_obs_fixed, 64)), dim3(64), 0, st, d,
                       at_estimate ? d.pose_est : d.pose_lin, at_estimate ? d.plane_est : d.plane_lin, gd);
  }
  const int nb_obs = cdiv(d.n_obs_fixed, kLinBlock), nb_odo = cdiv(d.n_odo, kLinBlock), nb_pp = cdiv(d.n_pp, kLinBlock),
            nb_lp = cdiv(d.n_lp, kLinBlock);
  const int nb = nb_obs + nb_odo + nb_pp + nb_lp;
  if (nb == 0) return hipSuccess;
  const double* pose = at_estimate ? d.pose_est : d.pose_lin;
  const double* plane = at_estimate ? d.plane_est : d.plane_lin;
  if (k1_lane_form(d, mode)) return launch_linearize_lanes(d, pose, plane, gd, st, ev0, ev1);
  if (ev0) { const hipError_t e = hipEventRecord(ev0, st); if (e != hipSuccess) return e; }      // (two launches: the pair goes around both)
  const size_t lds0 = (size_t)(kLinBlock / 64) * 64 * 31 * sizeof(double), lds1 = (size_t)(kLinBlock / 64) * 64 * 79 * sizeof(double);
  const int nb_rest = nb - nb_obs;
  // PPS_K1_THREAD_FORM=1 (the parity test of the batched throughput forms) also switches the product records off: K2 then
  // multiplies the Jacobians of every contribution, as the throughput form of a large batch does
  DevGraph dn = d;
  if (!k1_products(d, mode)) dn.P = nullptr;
  if (mode == 1) {
    if (nb_obs) PPS_LAUNCH((k_linearize<1, 0>), dim3(nb_obs), dim3(kLinBlock), lds0, st, dn, pose, plane, nb_obs, nb_odo, nb_pp, gd);
    if (nb_rest) PPS_LAUNCH((k_linearize<1, 1>), dim3(nb_rest), dim3(kLinBlock), lds1, st, dn, pose, plane, nb_obs, nb_odo, nb_pp, gd);
  } else {
    if (nb_obs) PPS_LAUNCH(k_linearize_obs_numeric, dim3(nb_obs), dim3(kLinBlock), lds0, st, dn, pose, plane, nb_obs, nb_odo, nb_pp, gd);
    if (nb_rest) PPS_LAUNCH((k_linearize<0, 1>), dim3(nb_rest), dim3(kLinBlock), lds1, st, dn, pose, plane, nb_obs, nb_odo, nb_pp, gd);
  }
  if (ev1) { const hipError_t e = hipEventRecord(ev1, st); if (e != hipSuccess) return e; }
  return hipGetLastError();
}

// K1 over replicated plane/odometry edges (roofline micro-benchmark): replica r writes its own J slab.
template <int MODE, int PART>
__device__ __forceinline__ void body_sweep_bench(const DevGraph& d, double* __restrict__ Jbig, int nb_obs_per, int nb_odo_per, int replicas) {
  const int per = PART == 0 ? nb_obs_per : nb_odo_per;
  const int rep = blockIdx.x / per;
  int b = blockIdx.x % per + (PART == 0 ? 0 : nb_obs_per);
  const size_t slab = (size_t)d.n_obs * 30 + (size_t)d.n_odo * 78;
  double* Jr = Jbig + (size_t)rep * slab;
  // replicas read shifted copies of the edge arrays so that no two replicas share cache lines
  const double* obs_meas = d.obs_meas + (size_t)rep * 4 * d.obs_ld;
  const double* obs_w = d.obs_w + (size_t)rep * 6 * d.obs_ld;
  const int* obs_pose = d.obs_pose + (size_t)rep * d.n_obs;
  const int* obs_plane = d.obs_plane + (size_t)rep * d.n_obs;
  const double* odo_meas = d.odo_meas + (size_t)rep * 6 * d.odo_ld;
  const double* odo_w = d.odo_w + (size_t)rep * 21 * d.odo_ld;
  const int* odo_a = d.odo_a + (size_t)rep * d.n_odo;
  const int* odo_b = d.odo_b + (size_t)rep * d.n_odo;
  extern __shared__ double lin_lds[];
  double* lds_wave = lin_lds + (size_t)(threadIdx.x >> 6) * 64 * (PART == 0 ? 31 : 79);
  if (b < nb_obs_per) {
    const int i0 = b * kLinBlock + (threadIdx.x & ~63);
    const int i = min(b * kLinBlock + (int)threadIdx.x, d.n_obs - 1);
    double pz[7], pl[4], ms[4], w[6], out[30];
    load_pose(d.pose_lin, d.pose_ld, obs_pose[i], pz);
    load_plane(d.plane_lin, d.plane_ld, obs_plane[i], pl);
    load_soa<4>(obs_meas, d.obs_ld, i, ms);
    load_soa<6>(obs_w, d.obs_ld, i, w);
    lin_plane_obs<MODE>(pz, pl, ms, w, out);
    if (i0 < d.n_obs) store_records_coalesced<30>(out, Jr + (size_t)i0 * 30, min(64, d.n_obs - i0), lds_wave);
    return;
  }
  if (PART == 0) return;                // (a PART 0 launch holds plane-observation blocks only: without this the odometry path is compiled in
                                        // and its register needs -- 66 spilled registers under the two-waves cap -- are the kernel's)
  b -= nb_obs_per;
  const int i0 = b * kLinBlock + (threadIdx.x & ~63);
  const int i = min(b * kLinBlock + (int)threadIdx.x, d.n_odo - 1);
  double p1[7], p2[7], ms[6], w[21];
  load_pose(d.pose_lin, d.pose_ld, odo_a[i], p1);
  load_pose(d.pose_lin, d.pose_ld, odo_b[i], p2);
  load_soa<6>(odo_meas, d.odo_ld, i, ms);
  load_soa<21>(odo_w, d.odo_ld, i, w);
  if (MODE == 1) {
    double out[78];
    lin_odometry<MODE>(p1, p2, ms, w, out);
    if (i0 < d.n_odo) store_records_coalesced<78>(out, Jr + (size_t)d.n_obs * 30 + (size_t)i0 * 78, min(64, d.n_odo - i0), lds_wave);
  } else {
    lin_odometry<MODE>(p1, p2, ms, w, lds_wave + (threadIdx.x & 63) * 79);
    if (i0 < d.n_odo) flush_records_coalesced<78>(Jr + (size_t)d.n_obs * 30 + (size_t)i0 * 78, min(64, d.n_odo - i0), lds_wave);
  }
}

template <int MODE, int PART>
__global__ __launch_bounds__(kLinBlock) void k_sweep_bench(DevGraph d, double* __restrict__ Jbig, int nb_obs_per, int nb_odo_per, int replicas) {
  body_sweep_bench<MODE, PART>(d, Jbig, nb_obs_per, nb_odo_per, replicas);
}
__global__ __launch_bounds__(kLinBlock) __attribute__((amdgpu_waves_per_eu(kObsNumericWaves, kObsNumericWaves)))      // (see k_linearize_obs_numeric)
void k_sweep_bench_obs_numeric(DevGraph d, double* __restrict__ Jbig, int nb_obs_per, int nb_odo_per, int replicas) {
  body_sweep_bench<0, 0>(d, Jbig, nb_obs_per, nb_odo_per, replicas);
}

hipError_t launch_sweep_bench(const DevGraph& d, int mode, int replicas, double* Jbig, int part, hipStream_t st) {
  if (mode == 2) return launch_sweep_bench_lanes(d, replicas, Jbig, part, st);
  const int nb_obs = cdiv(d.n_obs, kLinBlock), nb_odo = cdiv(d.n_odo, kLinBlock);
  const size_t lds0 = (size_t)(kLinBlock / 64) * 64 * 31 * sizeof(double), lds1 = (size_t)(kLinBlock / 64) * 64 * 79 * sizeof(double);
  if (nb_obs + nb_odo == 0) return hipSuccess;
  if (mode == 1) {
    if (nb_obs && part != 1) PPS_LAUNCH((k_sweep_bench<1, 0>), dim3(nb_obs * replicas), dim3(kLinBlock), lds0, st, d, Jbig, nb_obs, nb_odo, replicas);
    if (nb_odo && part != 0) PPS_LAUNCH((k_sweep_bench<1, 1>), dim3(nb_odo * replicas), dim3(kLinBlock), lds1, st, d, Jbig, nb_obs, nb_odo, replicas);
  } else {
    if (nb_obs && part != 1) PPS_LAUNCH(k_sweep_bench_obs_numeric, dim3(nb_obs * replicas), dim3(kLinBlock), lds0, st, d, Jbig, nb_obs, nb_odo, replicas);
    if (nb_odo && part != 0) PPS_LAUNCH((k_sweep_bench<0, 1>), dim3(nb_odo * replicas), dim3(kLinBlock), lds1, st, d, Jbig, nb_obs, nb_odo, replicas);
  }
  return hipGetLastError();
}

// ---- batched forms ----
template <int MODE, int PART, bool DIRECT>
__global__ __launch_bounds__(kLinBlock) void kb_linearize(BatchArgs a) {
  extern __shared__ double lin_lds[];
  PPS_BATCH_PROLOGUE(BF_ACTIVE | BF_RELIN)
  const int nb_obs = dcdiv(d.n_obs_fixed, kLinBlock), nb_odo = dcdiv(d.n_odo, kLinBlock), nb_pp = dcdiv(d.n_pp, kLinBlock),
            nb_lp = dcdiv(d.n_lp, kLinBlock);
  if ((int)blockIdx.x >= (PART == 0 ? nb_obs : nb_odo + nb_pp + nb_lp)) return;
  DevGraph dn = d;
  if (a.no_products) dn.P = nullptr;    // a chunk of > 200 000 factors: its K2 (kb_hblocks_t) multiplies the Jacobians itself
  body_linearize<MODE, PART, DIRECT>(dn, pose_lin, plane_lin, nb_obs, nb_odo, nb_pp, blockIdx.x, lin_lds);
}

__global__ __launch_bounds__(64) void kb_linearize_repop(BatchArgs a) {
  PPS_BATCH_PROLOGUE(BF_ACTIVE | BF_RELIN)
  __shared__ double repop_lds[64 * 31];
  if ((int)blockIdx.x * 64 >= d.n_obs - d.n_obs_fixed) return;
  body_linearize_repop(d, pose_lin, plane_lin, blockIdx.x, repop_lds);
}

hipError_t launch_batch_linearize(const BatchArgs& a, const BatchGeom& g, int mode, hipStream_t st) {
  if (g.repop_blocks > 0) PPS_LAUNCH(kb_linearize_repop, dim3(g.repop_blocks, a.n), dim3(64), 0, st, a);
  const size_t lds0 = (size_t)(kLinBlock / 64) * 64 * 31 * sizeof(double), lds1 = (size_t)(kLinBlock / 64) * 64 * 79 * sizeof(double);
  if (mode == 0) {
    if (g.lin_blocks > 0) { const hipError_t e = launch_batch_linearize_lanes(a, g, st); if (e != hipSuccess) return e; }
  } else if (mode == 2) {          // numeric, one thread per factor
    if (g.lin_obs_blocks > 0) {
      PPS_LAUNCH((kb_linearize<0, 0, true>), dim3(g.lin_obs_blocks, a.n), dim3(kLinBlock), lds0, st, a);
    }
    if (g.lin_rest_blocks > 0) PPS_LAUNCH((kb_linearize<0, 1, false>), dim3(g.lin_rest_blocks, a.n), dim3(kLinBlock), lds1, st, a);
  } else {
    if (g.lin_obs_blocks > 0) {
      PPS_LAUNCH((kb_linearize<1, 0, true>), dim3(g.lin_obs_blocks, a.n), dim3(kLinBlock), lds0, st, a);
    }
    if (g.lin_rest_blocks > 0) PPS_LAUNCH((kb_linearize<1, 1, false>), dim3(g.lin_rest_blocks, a.n), dim3(kLinBlock), lds1, st, a);
  }
  return hipGetLastError();
}

}  // namespace pps
