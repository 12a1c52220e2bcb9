// pps_kernels.hip -- gfx950 kernels of the plane-SLAM graph solve.
//
//   K1  k_linearize      per-edge residual + Jacobian sweep (reference: Slam::jacobian_partial,
//                        isamlib/Slam.cpp:395-432 + numericalDiff.cpp:41-87)             HBM-bound
//   K2  k_hblocks        block-sparse J'J / J'b reduction (cholmod_ssmult/sdmult,
//                        isamlib/Cholesky.cpp:87-89,120)
//   K3  k_front_factor   multifrontal partial Cholesky, one workgroup per front, front in LDS
//       k_front_solve    back-substitution, root to leaves (cholmod_factorize/solve, Cholesky.cpp:100-128)
//   K4  k_retract_*      exmap per node (Slam::self_exmap/apply_exmap, Slam.cpp:216-234)
//       k_chi2, k_finalize  residual-only sweep + chi^2 reduction (Slam::weighted_errors/chi2, Slam.cpp:254-268)
#include "pps_device.h"
#include "pps_geom.h"

namespace pps {

// ------------------------------------------------------------------------------------------
// K1: one thread per factor; SoA loads (coalesced across the wave), state gathered by index.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_pose(const double* __restrict__ base, int ld, int i, double p[7]) {
#pragma unroll
  for (int k = 0; k < 7; k++) p[k] = base[(size_t)k * ld + i];
}
__device__ __forceinline__ void load_plane(const double* __restrict__ base, int ld, int i, double p[4]) {
#pragma unroll
  for (int k = 0; k < 4; k++) p[k] = base[(size_t)k * ld + i];
}
template <int K>
__device__ __forceinline__ void load_soa(const double* __restrict__ base, int ld, int i, double* o) {
#pragma unroll
  for (int k = 0; k < K; k++) o[k] = base[(size_t)k * ld + i];
}

template <int MODE>
__device__ __forceinline__ void lin_plane_obs(const double pz[7], const double pl[4], const double ms[4],
                                              const double w[6], double* __restrict__ out) {
  double Jp[18], Jl[9], r[3];
  if (MODE == 1) {
    double e[3];
    jac_plane_obs(pz, pl, ms, e, Jp, Jl);
    whiten<3>(w, e, r);
    whiten_rows<3, 6>(w, Jp);
    whiten_rows<3, 3>(w, Jl);
  } else {
    double e[3];
    res_plane_obs(pz, pl, ms, e);
    whiten<3>(w, e, r);
    const double inv2e = 1.0 / (kNumDiffEps + kNumDiffEps);
#pragma unroll
    for (int j = 0; j < 6; j++) {
      double d[6] = {0, 0, 0, 0, 0, 0}, pp[7], yp[3], ym[3];
      d[j] = kNumDiffEps;
      pose_exmap(pz, d, pp); res_plane_obs(pp, pl, ms, e); whiten<3>(w, e, yp);
      d[j] = -kNumDiffEps;
      pose_exmap(pz, d, pp); res_plane_obs(pp, pl, ms, e); whiten<3>(w, e, ym);
#pragma unroll
      for (int i = 0; i < 3; i++) Jp[i * 6 + j] = (yp[i] - ym[i]) * inv2e;
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double d[3] = {0, 0, 0}, pp[4], yp[3], ym[3];
      d[j] = kNumDiffEps;
      plane_exmap(pl, d, pp); res_plane_obs(pz, pp, ms, e); whiten<3>(w, e, yp);
      d[j] = -kNumDiffEps;
      plane_exmap(pl, d, pp); res_plane_obs(pz, pp, ms, e); whiten<3>(w, e, ym);
#pragma unroll
      for (int i = 0; i < 3; i++) Jl[i * 3 + j] = (yp[i] - ym[i]) * inv2e;
    }
  }
#pragma unroll
  for (int k = 0; k < 18; k++) out[k] = Jp[k];
#pragma unroll
  for (int k = 0; k < 9; k++) out[18 + k] = Jl[k];
#pragma unroll
  for (int k = 0; k < 3; k++) out[27 + k] = r[k];
}

template <int MODE>
__device__ __forceinline__ void lin_odometry(const double p1[7], const double p2[7], const double ms[6],
                                             const double* w, double* __restrict__ out) {
  double e[6], r[6];
  if (MODE == 1) {
    double J1[36], J2[36];
    jac_odometry(p1, p2, ms, e, J1, J2);
    whiten<6>(w, e, r);
    whiten_rows<6, 6>(w, J1);
    whiten_rows<6, 6>(w, J2);
#pragma unroll
    for (int k = 0; k < 36; k++) out[k] = J1[k];
#pragma unroll
    for (int k = 0; k < 36; k++) out[36 + k] = J2[k];
  } else {
    const double inv2e = 1.0 / (kNumDiffEps + kNumDiffEps);
    for (int n = 0; n < 2; n++) {
      for (int j = 0; j < 6; j++) {
        double d[6] = {0, 0, 0, 0, 0, 0}, pp[7], yp[6], ym[6];
        d[j] = kNumDiffEps;
        pose_exmap(n == 0 ? p1 : p2, d, pp);
        if (n == 0) res_odometry(pp, p2, ms, e); else res_odometry(p1, pp, ms, e);
        whiten<6>(w, e, yp);
        d[j] = -kNumDiffEps;
        pose_exmap(n == 0 ? p1 : p2, d, pp);
        if (n == 0) res_odometry(pp, p2, ms, e); else res_odometry(p1, pp, ms, e);
        whiten<6>(w, e, ym);
#pragma unroll
        for (int i = 0; i < 6; i++) out[n * 36 + i * 6 + j] = (yp[i] - ym[i]) * inv2e;
      }
    }
    res_odometry(p1, p2, ms, e);
    whiten<6>(w, e, r);
  }
#pragma unroll
  for (int k = 0; k < 6; k++) out[72 + k] = r[k];
}

template <int MODE>
__device__ __forceinline__ void lin_pose_prior(const double pz[7], const double ms[6], const double* w,
                                               double* __restrict__ out) {
  double e[6], r[6];
  if (MODE == 1) {
    double J[36];
    jac_pose_prior(pz, ms, e, J);
    whiten<6>(w, e, r);
    whiten_rows<6, 6>(w, J);
#pragma unroll
    for (int k = 0; k < 36; k++) out[k] = J[k];
  } else {
    const double inv2e = 1.0 / (kNumDiffEps + kNumDiffEps);
    for (int j = 0; j < 6; j++) {
      double d[6] = {0, 0, 0, 0, 0, 0}, pp[7], yp[6], ym[6];
      d[j] = kNumDiffEps;
      pose_exmap(pz, d, pp); res_pose_prior(pp, ms, e); whiten<6>(w, e, yp);
      d[j] = -kNumDiffEps;
      pose_exmap(pz, d, pp); res_pose_prior(pp, ms, e); whiten<6>(w, e, ym);
#pragma unroll
      for (int i = 0; i < 6; i++) out[i * 6 + j] = (yp[i] - ym[i]) * inv2e;
    }
    res_pose_prior(pz, ms, e);
    whiten<6>(w, e, r);
  }
#pragma unroll
  for (int k = 0; k < 6; k++) out[36 + k] = r[k];
}

template <int MODE>
__device__ __forceinline__ void lin_plane_prior(const double pl[4], const double ms[4], const double w[6],
                                                double* __restrict__ out) {
  double e[3], r[3], Jl[9];
  if (MODE == 1) {
    jac_plane_prior(pl, ms, e, Jl);
    whiten<3>(w, e, r);
    whiten_rows<3, 3>(w, Jl);
  } else {
    const double inv2e = 1.0 / (kNumDiffEps + kNumDiffEps);
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double d[3] = {0, 0, 0}, pp[4], yp[3], ym[3];
      d[j] = kNumDiffEps;
      plane_exmap(pl, d, pp); res_plane_prior(pp, ms, e); whiten<3>(w, e, yp);
      d[j] = -kNumDiffEps;
      plane_exmap(pl, d, pp); res_plane_prior(pp, ms, e); whiten<3>(w, e, ym);
#pragma unroll
      for (int i = 0; i < 3; i++) Jl[i * 3 + j] = (yp[i] - ym[i]) * inv2e;
    }
    res_plane_prior(pl, ms, e);
    whiten<3>(w, e, r);
  }
#pragma unroll
  for (int k = 0; k < 9; k++) out[k] = Jl[k];
#pragma unroll
  for (int k = 0; k < 3; k++) out[9 + k] = r[k];
}

constexpr int kLinBlock = 128;

template <int MODE>
__global__ __launch_bounds__(kLinBlock) void k_linearize(DevGraph d, const double* __restrict__ pose,
                                                          const double* __restrict__ plane, int nb_obs, int nb_odo,
                                                          int nb_pp) {
  int b = blockIdx.x;
  if (b < nb_obs) {
    const int i = b * kLinBlock + threadIdx.x;
    if (i >= d.n_obs) return;
    double pz[7], pl[4], ms[4], w[6];
    load_pose(pose, d.pose_ld, d.obs_pose[i], pz);
    load_plane(plane, d.plane_ld, d.obs_plane[i], pl);
    load_soa<4>(d.obs_meas, d.n_obs, i, ms);
    load_soa<6>(d.obs_w, d.n_obs, i, w);
    lin_plane_obs<MODE>(pz, pl, ms, w, d.J + d.joff_obs + (size_t)i * 30);
    return;
  }
  b -= nb_obs;
  if (b < nb_odo) {
    const int i = b * kLinBlock + threadIdx.x;
    if (i >= d.n_odo) return;
    double p1[7], p2[7], ms[6], w[21];
    load_pose(pose, d.pose_ld, d.odo_a[i], p1);
    load_pose(pose, d.pose_ld, d.odo_b[i], p2);
    load_soa<6>(d.odo_meas, d.n_odo, i, ms);
    load_soa<21>(d.odo_w, d.n_odo, i, w);
    lin_odometry<MODE>(p1, p2, ms, w, d.J + d.joff_odo + (size_t)i * 78);
    return;
  }
  b -= nb_odo;
  if (b < nb_pp) {
    const int i = b * kLinBlock + threadIdx.x;
    if (i >= d.n_pp) return;
    double pz[7], ms[6], w[21];
    load_pose(pose, d.pose_ld, d.pp_pose[i], pz);
    load_soa<6>(d.pp_meas, d.n_pp, i, ms);
    load_soa<21>(d.pp_w, d.n_pp, i, w);
    lin_pose_prior<MODE>(pz, ms, w, d.J + d.joff_pp + (size_t)i * 42);
    return;
  }
  b -= nb_pp;
  {
    const int i = b * kLinBlock + threadIdx.x;
    if (i >= d.n_lp) return;
    double pl[4], ms[4], w[6];
    load_plane(plane, d.plane_ld, d.lp_plane[i], pl);
    load_soa<4>(d.lp_meas, d.n_lp, i, ms);
    load_soa<6>(d.lp_w, d.n_lp, i, w);
    lin_plane_prior<MODE>(pl, ms, w, d.J + d.joff_lp + (size_t)i * 12);
  }
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

hipError_t launch_linearize(const DevGraph& d, int mode, bool at_estimate, hipStream_t st) {
  const int nb_obs = cdiv(d.n_obs, kLinBlock), nb_odo = cdiv(d.n_odo, kLinBlock), nb_pp = cdiv(d.n_pp, kLinBlock),
            nb_lp = cdiv(d.n_lp, kLinBlock);
  const int nb = nb_obs + nb_odo + nb_pp + nb_lp;
  if (nb == 0) return hipSuccess;
  const double* pose = at_estimate ? d.pose_est : d.pose_lin;
  const double* plane = at_estimate ? d.plane_est : d.plane_lin;
  if (mode == 1) hipLaunchKernelGGL(k_linearize<1>, dim3(nb), dim3(kLinBlock), 0, st, d, pose, plane, nb_obs, nb_odo, nb_pp);
  else           hipLaunchKernelGGL(k_linearize<0>, dim3(nb), dim3(kLinBlock), 0, st, d, pose, plane, nb_obs, nb_odo, nb_pp);
  return hipGetLastError();
}

// K1 over replicated plane/odometry edges (roofline micro-benchmark): replica r writes its own J slab.
template <int MODE>
__global__ __launch_bounds__(kLinBlock) void k_sweep_bench(DevGraph d, double* __restrict__ Jbig, int nb_obs_per,
                                                            int nb_odo_per, int replicas) {
  const int per = nb_obs_per + nb_odo_per;
  const int rep = blockIdx.x / per;
  int b = blockIdx.x % per;
  const size_t slab = (size_t)d.n_obs * 30 + (size_t)d.n_odo * 78;
  double* Jr = Jbig + (size_t)rep * slab;
  // replicas read shifted copies of the edge arrays so that no two replicas share cache lines
  const double* obs_meas = d.obs_meas + (size_t)rep * 4 * d.n_obs;
  const double* obs_w = d.obs_w + (size_t)rep * 6 * d.n_obs;
  const int* obs_pose = d.obs_pose + (size_t)rep * d.n_obs;
  const int* obs_plane = d.obs_plane + (size_t)rep * d.n_obs;
  const double* odo_meas = d.odo_meas + (size_t)rep * 6 * d.n_odo;
  const double* odo_w = d.odo_w + (size_t)rep * 21 * d.n_odo;
  const int* odo_a = d.odo_a + (size_t)rep * d.n_odo;
  const int* odo_b = d.odo_b + (size_t)rep * d.n_odo;
  if (b < nb_obs_per) {
    const int i = b * kLinBlock + threadIdx.x;
    if (i >= d.n_obs) return;
    double pz[7], pl[4], ms[4], w[6];
    load_pose(d.pose_lin, d.pose_ld, obs_pose[i], pz);
    load_plane(d.plane_lin, d.plane_ld, obs_plane[i], pl);
    load_soa<4>(obs_meas, d.n_obs, i, ms);
    load_soa<6>(obs_w, d.n_obs, i, w);
    lin_plane_obs<MODE>(pz, pl, ms, w, Jr + (size_t)i * 30);
    return;
  }
  b -= nb_obs_per;
  const int i = b * kLinBlock + threadIdx.x;
  if (i >= d.n_odo) return;
  double p1[7], p2[7], ms[6], w[21];
  load_pose(d.pose_lin, d.pose_ld, odo_a[i], p1);
  load_pose(d.pose_lin, d.pose_ld, odo_b[i], p2);
  load_soa<6>(odo_meas, d.n_odo, i, ms);
  load_soa<21>(odo_w, d.n_odo, i, w);
  lin_odometry<MODE>(p1, p2, ms, w, Jr + (size_t)d.n_obs * 30 + (size_t)i * 78);
}

hipError_t launch_sweep_bench(const DevGraph& d, int mode, int replicas, double* Jbig, hipStream_t st) {
  const int nb_obs = cdiv(d.n_obs, kLinBlock), nb_odo = cdiv(d.n_odo, kLinBlock);
  const int nb = (nb_obs + nb_odo) * replicas;
  if (nb == 0) return hipSuccess;
  if (mode == 1) hipLaunchKernelGGL(k_sweep_bench<1>, dim3(nb), dim3(kLinBlock), 0, st, d, Jbig, nb_obs, nb_odo, replicas);
  else           hipLaunchKernelGGL(k_sweep_bench<0>, dim3(nb), dim3(kLinBlock), 0, st, d, Jbig, nb_obs, nb_odo, replicas);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// K2: one wavefront per H-block segment; lane = block entry, loop over <= seg_len contributions.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hblocks(DevGraph d) {
  const int seg = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (seg >= d.n_segs) return;
  const int blk = d.seg_blk[seg];
  const int rows = d.blk_rows[blk], cols = d.blk_cols[blk], size = d.blk_size[blk];
  if (lane >= size) return;
  const int rc = rows * cols;
  const bool is_g = lane >= rc;
  const int i = is_g ? lane - rc : lane / cols;
  const int j = is_g ? 0 : lane % cols;
  const int c0 = d.seg_c0[seg], cnt = d.seg_cnt[seg];
  const int4* __restrict__ ctr = reinterpret_cast<const int4*>(d.contrib);
  const double* __restrict__ J = d.J;
  double acc = 0.0;
  for (int c = c0; c < c0 + cnt; c++) {
    const int4 cc = ctr[c];
    const double* jv = J + cc.x + i;
    if (!is_g) {
      const double* ju = J + cc.y + j;
      for (int k = 0; k < cc.w; k++) acc += jv[k * rows] * ju[k * cols];
    } else {
      const double* r = J + cc.z;
      for (int k = 0; k < cc.w; k++) acc -= jv[k * rows] * r[k];   // b = -r (isam/Jacobian.h:98)
    }
  }
  d.H[d.seg_hoff[seg] + lane] = acc;
}

hipError_t launch_hblocks(const DevGraph& d, hipStream_t st) {
  if (d.n_segs == 0) return hipSuccess;
  hipLaunchKernelGGL(k_hblocks, dim3(cdiv(d.n_segs, 4)), dim3(256), 0, st, d);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// K3: multifrontal partial Cholesky.  One 256-thread workgroup per front; the (f+1)x(f+1) front
// (last row = right-hand side) lives in LDS (row-major, odd leading dimension), or in a global
// workspace when it does not fit.  Steps: gather original H blocks (damped diagonal,
// Cholesky.cpp:94-97) -> extend-add children update matrices -> right-looking elimination of the
// p pivot columns -> store factor panel and update matrix.
// ------------------------------------------------------------------------------------------
constexpr int kLdsLimitBytes = 160 * 1024 - 1024;

int lds_front_limit() {
  int fa = 1;
  while ((size_t)(fa + 1) * ((fa + 1) | 1) * 8 <= (size_t)kLdsLimitBytes) fa++;
  return fa - 1;   // largest f with (f+1) rows
}

template <bool USE_LDS>
__global__ __launch_bounds__(256) void k_front_factor(DevGraph d, int level_begin, double lambda) {
  extern __shared__ double lds[];
  const int s = d.level_fronts[level_begin + blockIdx.x];
  const int p = d.f_p[s], b = d.f_b[s];
  const int f = p + b, fa = f + 1, ld = fa | 1;
  double* F = USE_LDS ? lds : d.gwork + (size_t)blockIdx.x * d.gwork_stride;
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < fa * ld; i += nt) F[i] = 0.0;
  __syncthreads();
  // ---- original entries: one wave per block, lane per entry ----
  {
    const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
    const int a0 = d.f_asm_off[s], a1 = d.f_asm_off[s + 1];
    for (int a = a0 + wave; a < a1; a += nw) {
      const int blk = d.asm_blk[a], lrow = d.asm_lrow[a], lcol = d.asm_lcol[a];
      const int rows = d.blk_rows[blk], cols = d.blk_cols[blk], size = d.blk_size[blk], nseg = d.blk_nseg[blk];
      const double* __restrict__ h = d.H + d.blk_hoff[blk];
      const int rc = rows * cols;
      const bool diag = size > rc;
      if (lane < size) {
        double v = 0.0;
        for (int q = 0; q < nseg; q++) v += h[(size_t)q * size + lane];
        if (lane < rc) {
          const int i = lane / cols, j = lane % cols;
          if (!diag || i >= j) {
            if (diag && i == j) v *= (1.0 + lambda);
            F[(lrow + i) * ld + lcol + j] += v;
          }
        } else {
          F[f * ld + lcol + (lane - rc)] += v;
        }
      }
    }
  }
  __syncthreads();
  // ---- extend-add of the children's update matrices ----
  for (int ci = d.f_child_off[s]; ci < d.f_child_off[s + 1]; ci++) {
    const int c = d.child[ci];
    const int bc1 = d.f_b[c] + 1;
    const double* __restrict__ Uc = d.U + d.f_Uoff[c];
    const int* __restrict__ cm = d.cmap + d.f_cmap_off[c];
    for (int idx = tid; idx < bc1 * bc1; idx += nt) {
      const int i = idx / bc1, j = idx - i * bc1;
      if (j <= i) F[cm[i] * ld + cm[j]] += Uc[idx];
    }
    __syncthreads();
  }
  // ---- eliminate the p pivot columns (right-looking) ----
  const int tx = tid & 15, ty = tid >> 4;
  for (int k = 0; k < p; k++) {
    const double dkk = F[k * ld + k];
    double dinv;
    if (!(dkk > 0.0)) {
      if (tid == 0) d.result_dev[2] = 1.0;   // not positive definite
      dinv = 0.0;
    } else {
      dinv = 1.0 / sqrt(dkk);
    }
    for (int i = k + 1 + tid; i < fa; i += nt) F[i * ld + k] *= dinv;
    __syncthreads();
    for (int i = k + 1 + ty; i < fa; i += 16) {
      const double lik = F[i * ld + k];
      for (int j = k + 1 + tx; j <= i; j += 16) F[i * ld + j] -= lik * F[j * ld + k];
    }
    __syncthreads();   // column k+1 (diagonal included) is final before the next iteration reads it
  }
  // ---- store the factor panel ((f+1) x p, row-major) and the update matrix ((b+1) x (b+1)) ----
  double* __restrict__ Lp = d.L + d.f_Loff[s];
  for (int idx = tid; idx < fa * p; idx += nt) {
    const int i = idx / p, j = idx - i * p;
    double v = 0.0;
    if (i == j) { const double x = F[j * ld + j]; v = x > 0.0 ? sqrt(x) : 1.0; }
    else if (i > j) v = F[i * ld + j];
    Lp[idx] = v;
  }
  double* __restrict__ Us = d.U + d.f_Uoff[s];
  const int b1 = b + 1;
  for (int idx = tid; idx < b1 * b1; idx += nt) {
    const int i = idx / b1, j = idx - i * b1;
    Us[idx] = (j <= i) ? F[(p + i) * ld + p + j] : 0.0;
  }
}

static bool g_attr_set = false;

hipError_t launch_factor_level(const DevGraph& d, int level_begin, int level_count, int level_max_front, double lambda,
                               hipStream_t st) {
  if (level_count == 0) return hipSuccess;
  const int fa = level_max_front + 1;
  const size_t bytes = (size_t)fa * (fa | 1) * 8;
  if (bytes <= (size_t)kLdsLimitBytes) {
    if (!g_attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_front_factor<true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
      if (e != hipSuccess) return e;
      g_attr_set = true;
    }
    hipLaunchKernelGGL(k_front_factor<true>, dim3(level_count), dim3(256), bytes, st, d, level_begin, lambda);
  } else {
    hipLaunchKernelGGL(k_front_factor<false>, dim3(level_count), dim3(256), 0, st, d, level_begin, lambda);
  }
  return hipGetLastError();
}

// Back-substitution for one level (parents already solved): x_p = L_A^-T (y - L_B^T x_b).
__global__ __launch_bounds__(64) void k_front_solve(DevGraph d, int level_begin) {
  __shared__ double t[256];
  const int s = d.level_fronts[level_begin + blockIdx.x];
  const int p = d.f_p[s], b = d.f_b[s], f = p + b;
  const double* __restrict__ Lp = d.L + d.f_Loff[s];
  const int* __restrict__ bi = d.bidx + d.f_bidx_off[s];
  const int lane = threadIdx.x;
  for (int k = lane; k < p; k += 64) {
    double acc = Lp[(size_t)f * p + k];                      // y_k (forward-solved rhs row)
    for (int i = 0; i < b; i++) acc -= Lp[(size_t)(p + i) * p + k] * d.delta[bi[i]];
    t[k] = acc;
  }
  __syncthreads();
  for (int k = p - 1; k >= 0; k--) {
    const double xk = t[k] / Lp[(size_t)k * p + k];
    __syncthreads();
    for (int j = lane; j < k; j += 64) t[j] -= Lp[(size_t)k * p + j] * xk;
    if (lane == 0) t[k] = xk;
    __syncthreads();
  }
  for (int k = lane; k < p; k += 64) d.delta[d.f_poff[s] + k] = t[k];
}

hipError_t launch_backsolve_level(const DevGraph& d, int level_begin, int level_count, hipStream_t st) {
  if (level_count == 0) return hipSuccess;
  hipLaunchKernelGGL(k_front_solve, dim3(level_count), dim3(64), 0, st, d, level_begin);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// K4: retraction and chi^2
// ------------------------------------------------------------------------------------------
template <bool TRIAL>
__global__ __launch_bounds__(256) void k_retract(DevGraph d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < d.n_pose) {
    double p[7], o[7], dl[6];
    load_pose(d.pose_lin, d.pose_ld, i, p);
    const int off = d.pose_voff[i];
#pragma unroll
    for (int k = 0; k < 6; k++) dl[k] = d.delta[off + k];
    pose_exmap(p, dl, o);
    if (TRIAL) {
#pragma unroll
      for (int k = 0; k < 7; k++) { d.pose_est[(size_t)k * d.pose_ld + i] = p[k]; d.pose_lin[(size_t)k * d.pose_ld + i] = o[k]; }
    } else {
#pragma unroll
      for (int k = 0; k < 7; k++) d.pose_est[(size_t)k * d.pose_ld + i] = o[k];
    }
  } else if (i < d.n_pose + d.n_plane) {
    const int l = i - d.n_pose;
    double p[4], o[4], dl[3];
    load_plane(d.plane_lin, d.plane_ld, l, p);
    const int off = d.plane_voff[l];
#pragma unroll
    for (int k = 0; k < 3; k++) dl[k] = d.delta[off + k];
    plane_exmap(p, dl, o);
    if (TRIAL) {
#pragma unroll
      for (int k = 0; k < 4; k++) { d.plane_est[(size_t)k * d.plane_ld + l] = p[k]; d.plane_lin[(size_t)k * d.plane_ld + l] = o[k]; }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) d.plane_est[(size_t)k * d.plane_ld + l] = o[k];
    }
  }
}

hipError_t launch_retract_trial(const DevGraph& d, hipStream_t st) {
  const int n = d.n_pose + d.n_plane;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_retract<true>, dim3(cdiv(n, 256)), dim3(256), 0, st, d);
  return hipGetLastError();
}
hipError_t launch_retract_apply(const DevGraph& d, hipStream_t st) {
  const int n = d.n_pose + d.n_plane;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_retract<false>, dim3(cdiv(n, 256)), dim3(256), 0, st, d);
  return hipGetLastError();
}

constexpr int kChiBlock = 256;

__global__ __launch_bounds__(kChiBlock) void k_chi2(DevGraph d, const double* __restrict__ pose,
                                                    const double* __restrict__ plane, int nb_obs, int nb_odo, int nb_pp) {
  __shared__ double red[kChiBlock / 64];
  int b = blockIdx.x;
  double s = 0.0;
  if (b < nb_obs) {
    const int i = b * kChiBlock + threadIdx.x;
    if (i < d.n_obs) {
      double pz[7], pl[4], ms[4], w[6], e[3], r[3];
      load_pose(pose, d.pose_ld, d.obs_pose[i], pz);
      load_plane(plane, d.plane_ld, d.obs_plane[i], pl);
      load_soa<4>(d.obs_meas, d.n_obs, i, ms);
      load_soa<6>(d.obs_w, d.n_obs, i, w);
      res_plane_obs(pz, pl, ms, e);
      whiten<3>(w, e, r);
      s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    }
  } else if ((b -= nb_obs) < nb_odo) {
    const int i = b * kChiBlock + threadIdx.x;
    if (i < d.n_odo) {
      double p1[7], p2[7], ms[6], w[21], e[6], r[6];
      load_pose(pose, d.pose_ld, d.odo_a[i], p1);
      load_pose(pose, d.pose_ld, d.odo_b[i], p2);
      load_soa<6>(d.odo_meas, d.n_odo, i, ms);
      load_soa<21>(d.odo_w, d.n_odo, i, w);
      res_odometry(p1, p2, ms, e);
      whiten<6>(w, e, r);
#pragma unroll
      for (int k = 0; k < 6; k++) s += r[k] * r[k];
    }
  } else if ((b -= nb_odo) < nb_pp) {
    const int i = b * kChiBlock + threadIdx.x;
    if (i < d.n_pp) {
      double pz[7], ms[6], w[21], e[6], r[6];
      load_pose(pose, d.pose_ld, d.pp_pose[i], pz);
      load_soa<6>(d.pp_meas, d.n_pp, i, ms);
      load_soa<21>(d.pp_w, d.n_pp, i, w);
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
      load_plane(plane, d.plane_ld, d.lp_plane[i], pl);
      load_soa<4>(d.lp_meas, d.n_lp, i, ms);
      load_soa<6>(d.lp_w, d.n_lp, i, w);
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
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < kChiBlock / 64; k++) t += red[k];
    d.chi2_partials[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(256) void k_finalize(DevGraph d, int nblocks) {
  __shared__ double red[2][4];
  double s = 0.0, dn = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) s += d.chi2_partials[i];
  for (int i = threadIdx.x; i < d.n_scalars; i += 256) { const double x = d.delta[i]; dn += x * x; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o, 64); dn += __shfl_down(dn, o, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = dn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    d.result_dev[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    d.result_dev[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

hipError_t launch_chi2(const DevGraph& d, bool at_estimate, double* host_result, hipStream_t st) {
  const int nb_obs = cdiv(d.n_obs, kChiBlock), nb_odo = cdiv(d.n_odo, kChiBlock), nb_pp = cdiv(d.n_pp, kChiBlock),
            nb_lp = cdiv(d.n_lp, kChiBlock);
  const int nb = nb_obs + nb_odo + nb_pp + nb_lp;
  const double* pose = at_estimate ? d.pose_est : d.pose_lin;
  const double* plane = at_estimate ? d.plane_est : d.plane_lin;
  if (nb > 0) hipLaunchKernelGGL(k_chi2, dim3(nb), dim3(kChiBlock), 0, st, d, pose, plane, nb_obs, nb_odo, nb_pp);
  hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, st, d, nb);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return hipMemcpyAsync(host_result, d.result_dev, 4 * sizeof(double), hipMemcpyDeviceToHost, st);
}

hipError_t launch_clear_status(const DevGraph& d, hipStream_t st) {
  return hipMemsetAsync(d.result_dev, 0, 4 * sizeof(double), st);
}

}  // namespace pps
