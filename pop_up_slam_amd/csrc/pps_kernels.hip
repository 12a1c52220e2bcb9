// pps_kernels.hip -- gfx950 kernels of the plane-SLAM graph solve.
//
//   K1  k_linearize<MODE,PART>, k_linearize_lanes, k_linearize_repop
//                        per-edge residual + Jacobian sweep (reference: Slam::jacobian_partial,
//                        isamlib/Slam.cpp:395-432 + numericalDiff.cpp:41-87)             HBM-bound (analytic mode)
//   K2  k_hblocks, k_hreduce
//                        block-sparse J'J / J'b reduction (cholmod_ssmult/sdmult, isamlib/Cholesky.cpp:87-89,120)
//   K3  k_band_factor, k_band_solve
//                        multifrontal Cholesky, one wavefront per front, a workgroup walks a sub-tree of a band of
//                        tree levels (cholmod_factorize/solve, Cholesky.cpp:100-128).  Fronts <= 64 rows live in
//                        registers as 16x16 fp64 MFMA tiles (pps_regtile.h), up to 128 rows in LDS tiles.
//       k_front_factor, k_front_solve
//                        level-per-launch fallback (one workgroup per front) when neither the band kernels nor the
//                        dense-front kernels (pps_dense.hip) apply
//   K4  k_retract<TRIAL> exmap per node (Slam::self_exmap/apply_exmap, Slam.cpp:216-234)
//       k_chi2           residual-only sweep + chi^2 reduction, last block writes the pinned result record
//                        (Slam::weighted_errors/chi2, Slam.cpp:254-268)
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "pps_device.h"
#include "pps_geom.h"
#include "pps_regtile.h"

namespace pps {

// H = J'J accumulations are written as explicit multiply-adds: three kernels (k_hblocks, the batched kb_hblocks_t and the
// direct blocks of the thread-per-factor sweep) must produce the same bits for the same block, whatever the compiler would
// have contracted on its own.  (PPS_NO_FMA: the diagnostic build without any fused operation.)
#ifdef PPS_NO_FMA
#define PPS_MAC(acc, a, b) ((acc) + (a) * (b))
#else
#define PPS_MAC(acc, a, b) __builtin_fma((a), (b), (acc))
#endif

// PPS_TRACE=1 instrumentation: lane 0 stamps s_memtime at phase boundaries of a front
#define PPS_TR(k) do { if (d.trace && lane == 0) d.trace[(size_t)s * 8 + (k)] = clock64(); } while (0)

// Values that are wave-uniform by construction (they derive from threadIdx.x >> 6) but that the
// compiler must treat as divergent: pin them into SGPRs so loops, branches and address arithmetic
// built on them are scalar instead of exec-masked "waterfall" code.
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ long long uni64(long long x) {
  const int lo = __builtin_amdgcn_readfirstlane((int)(x & 0xffffffffLL));
  const int hi = __builtin_amdgcn_readfirstlane((int)(x >> 32));
  return ((long long)hi << 32) | (unsigned int)lo;
}


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

// A wave's 64 factor records (N doubles each, contiguous in global memory) are staged through LDS
// (row stride N+1: conflict-free) and written back as one contiguous 64*N-double stream with 16-byte
// stores per lane, instead of 64 scattered N*8-byte records per store instruction.
// second half: the wave's records already sit in LDS (lane l at lds_wave + l * (N + 1))
template <int N>
__device__ __forceinline__ void flush_records_coalesced(double* __restrict__ gbase, int n_valid, double* __restrict__ lds_wave);

template <int N>
__device__ __forceinline__ void store_records_coalesced(const double (&v)[N], double* __restrict__ gbase, int n_valid,
                                                         double* __restrict__ lds_wave) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < N; k++) lds_wave[lane * (N + 1) + k] = v[k];
  flush_records_coalesced<N>(gbase, n_valid, lds_wave);
}

template <int N>
__device__ __forceinline__ void flush_records_coalesced(double* __restrict__ gbase, int n_valid, double* __restrict__ lds_wave) {
  const int lane = threadIdx.x & 63;
  __builtin_amdgcn_wave_barrier();
  const int total = n_valid * N;                    // doubles this wave owns (N even -> total even)
#pragma unroll
  for (int k = 0; k < (N + 1) / 2; k++) {
    const int idx = 2 * (lane + 64 * k);
    if (idx < total) {
      const int r0 = idx / N, c0 = idx - r0 * N;
      const int r1 = (idx + 1) / N, c1 = idx + 1 - r1 * N;
      double2 o;
      o.x = lds_wave[r0 * (N + 1) + c0];
      o.y = lds_wave[r1 * (N + 1) + c1];
      *reinterpret_cast<double2*>(gbase + idx) = o;
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// PART 0: plane-observation edges (16 KB of staging LDS per wave); PART 1: odometry edges and priors
// (40 KB per wave in analytic mode) -- separate launches so that the big edge class keeps its occupancy.
// DIRECT (many-graph batches): a plane observation that is the only contribution of its (pose, plane) block writes that H
// block itself -- the product of its own two Jacobian blocks, summed in the order the H-block kernel uses -- into H and into
// the front-ordered copy; kb_hblocks_t then skips those segments (60 % of the segments of a C2 graph).
template <int MODE, int PART, bool DIRECT = false>
__device__ __forceinline__ void body_linearize(const DevGraph& d, const double* __restrict__ pose,
                                               const double* __restrict__ plane, int nb_obs, int nb_odo, int nb_pp, int bx,
                                               double* __restrict__ lin_lds) {
  double* lds_wave = lin_lds + (size_t)(threadIdx.x >> 6) * 64 * (PART == 0 ? 31 : 79);
  int b = bx + (PART == 0 ? 0 : nb_obs);
  if (b < nb_obs) {
    const int i0 = b * kLinBlock + (threadIdx.x & ~63);            // first factor of this wave
    const int i = min(b * kLinBlock + (int)threadIdx.x, d.n_obs_fixed - 1);   // clamped: every lane stays active for the staged store
    double pz[7], pl[4], ms[4], w[6], out[30];
    load_pose(pose, d.pose_ld, d.obs_pose[i], pz);
    load_plane(plane, d.plane_ld, d.obs_plane[i], pl);
    load_soa<4>(d.obs_meas, d.obs_ld, i, ms);
    load_soa<6>(d.obs_w, d.obs_ld, i, w);
    lin_plane_obs<MODE>(pz, pl, ms, w, out);
    if (DIRECT && b * kLinBlock + (int)threadIdx.x < d.n_obs_fixed) {
      const int hoff = d.obs_dir[3 * (size_t)i], el0 = d.obs_dir[3 * (size_t)i + 1], rows6 = d.obs_dir[3 * (size_t)i + 2];
      if (hoff >= 0) {
        double* __restrict__ h = d.H + hoff;
        double* __restrict__ hf = d.Hf + el0;
        // block (v, u), rows = the node eliminated later: entry (i, j) = sum_k Jv[k][i] * Ju[k][j], k = 0, 1, 2
#pragma unroll
        for (int e = 0; e < 18; e++) {
          const int ri = rows6 ? e / 3 : e / 6, cj = rows6 ? e - 3 * (e / 3) : e - 6 * (e / 6);
          double acc = 0.0;
#pragma unroll
          for (int k = 0; k < 3; k++) {
            const double av = rows6 ? out[k * 6 + ri] : out[18 + k * 3 + ri];
            const double bv = rows6 ? out[18 + k * 3 + cj] : out[k * 6 + cj];
            acc = PPS_MAC(acc, av, bv);
          }
          h[e] = acc;
          if (el0 >= 0) hf[e] = acc;
        }
      }
    }
    if (i0 < d.n_obs_fixed) store_records_coalesced<30>(out, d.J + d.joff_obs + (size_t)i0 * 30, min(64, d.n_obs_fixed - i0), lds_wave);
    return;
  }
  b -= nb_obs;
  if (b < nb_odo) {
    // both Jacobian modes leave through the LDS-staged store: written straight from the lanes, a 624-byte record per lane
    // turns every store instruction into 64 partial 32-byte sectors (PMC: 265 MB written for 67 MB of records)
    const int i0 = b * kLinBlock + (threadIdx.x & ~63);
    const int i = min(b * kLinBlock + (int)threadIdx.x, d.n_odo - 1);
    double p1[7], p2[7], ms[6], w[21];
    load_pose(pose, d.pose_ld, d.odo_a[i], p1);
    load_pose(pose, d.pose_ld, d.odo_b[i], p2);
    load_soa<6>(d.odo_meas, d.odo_ld, i, ms);
    load_soa<21>(d.odo_w, d.odo_ld, i, w);
    if (MODE == 1) {
      double out[78];
      lin_odometry<MODE>(p1, p2, ms, w, out);
      if (i0 < d.n_odo) store_records_coalesced<78>(out, d.J + d.joff_odo + (size_t)i0 * 78, min(64, d.n_odo - i0), lds_wave);
    } else {
      // the central-difference loops stay rolled (24 residual evaluations): the record is built in LDS, not in registers
      lin_odometry<MODE>(p1, p2, ms, w, lds_wave + (threadIdx.x & 63) * 79);
      if (i0 < d.n_odo) flush_records_coalesced<78>(d.J + d.joff_odo + (size_t)i0 * 78, min(64, d.n_odo - i0), lds_wave);
    }
    return;
  }
  b -= nb_odo;
  if (b < nb_pp) {
    const int i = b * kLinBlock + threadIdx.x;
    if (i >= d.n_pp) return;
    double pz[7], ms[6], w[21];
    load_pose(pose, d.pose_ld, d.pp_pose[i], pz);
    load_soa<6>(d.pp_meas, d.pp_ld, i, ms);
    load_soa<21>(d.pp_w, d.pp_ld, i, w);
    lin_pose_prior<MODE>(pz, ms, w, d.J + d.joff_pp + (size_t)i * 42);
    return;
  }
  b -= nb_pp;
  {
    const int i = b * kLinBlock + threadIdx.x;
    if (i >= d.n_lp) return;
    double pl[4], ms[4], w[6];
    load_plane(plane, d.plane_ld, d.lp_plane[i], pl);
    load_soa<4>(d.lp_meas, d.lp_ld, i, ms);
    load_soa<6>(d.lp_w, d.lp_ld, i, w);
    lin_plane_prior<MODE>(pl, ms, w, d.J + d.joff_lp + (size_t)i * 12);
  }
}

// LinGuard (pps_device.h): which state does a speculatively queued K1 linearise at?  false = neither trial was accepted
__device__ __forceinline__ bool lin_guard(const LinGuard& gd, const double* __restrict__& pose, const double* __restrict__& plane) {
  if (!gd.on) return true;
  const double c0 = *gd.chi[0], c1 = *gd.chi[1];
  if (gd.error - c0 > 0.) { pose = gd.pose[0]; plane = gd.plane[0]; return true; }
  if (gd.error - c1 > 0.) { pose = gd.pose[1]; plane = gd.plane[1]; return true; }
  return false;
}
__device__ __forceinline__ bool lin_guard(const LinGuard& gd) {
  if (!gd.on) return true;
  return gd.error - *gd.chi[0] > 0. || gd.error - *gd.chi[1] > 0.;
}

template <int MODE, int PART>
__global__ __launch_bounds__(kLinBlock) void k_linearize(DevGraph d, const double* __restrict__ pose,
                                                          const double* __restrict__ plane, int nb_obs, int nb_odo,
                                                          int nb_pp, LinGuard gd) {
  extern __shared__ double lin_lds[];
  if (!lin_guard(gd, pose, plane)) return;
  body_linearize<MODE, PART>(d, pose, plane, nb_obs, nb_odo, nb_pp, blockIdx.x, lin_lds);
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

static thread_local unsigned long long t_launches = 0;
unsigned long long launch_count() { return t_launches; }
void count_launch() { ++t_launches; }

// ------------------------------------------------------------------------------------------
// K1, lane-parallel central differences (the reference's numericalDiff, one evaluation per lane):
// 32 lanes per factor = 2 factors per wavefront.  Lane 2q evaluates the residual at x (+) eps e_q,
// lane 2q+1 at x (-) eps e_q, lane 2*ncols the nominal residual; a lane pair differences through
// one DPP-style shuffle and lane 2q stores column q.  All lanes of a group read the same edge record
// (a broadcast load), state is gathered by index from the SoA arrays.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void perturb6(const double p[7], int q, double sgn, double o[7]) {
  double dl[6];
#pragma unroll
  for (int k = 0; k < 6; k++) dl[k] = (k == q) ? sgn * kNumDiffEps : 0.0;
  pose_exmap(p, dl, o);
}
__device__ __forceinline__ void perturb3(const double p[4], int q, double sgn, double o[4]) {
  double dl[3];
#pragma unroll
  for (int k = 0; k < 3; k++) dl[k] = (k == q) ? sgn * kNumDiffEps : 0.0;
  plane_exmap(p, dl, o);
}

constexpr int kLaneGroup = 32;
constexpr int kLanesPerBlock = 256;
constexpr int kFactorsPerBlock = kLanesPerBlock / kLaneGroup;

__device__ __forceinline__ void body_linearize_lanes(const DevGraph& d, const double* __restrict__ pose,
                                                     const double* __restrict__ plane, int nb_obs, int nb_odo, int nb_pp, int bx) {
  const int grp = threadIdx.x / kLaneGroup, gl = threadIdx.x % kLaneGroup;
  const int q = gl >> 1;                       // perturbed column
  const double sgn = (gl & 1) ? -1.0 : 1.0;
  const double inv2e = 1.0 / (kNumDiffEps + kNumDiffEps);
  int b = bx;
  if (b < nb_obs) {
    const int i = b * kFactorsPerBlock + grp;
    if (i >= d.n_obs_fixed) return;
    double pz[7], pl[4], ms[4], w[6], e[3], y[3];
    load_pose(pose, d.pose_ld, d.obs_pose[i], pz);
    load_plane(plane, d.plane_ld, d.obs_plane[i], pl);
    load_soa<4>(d.obs_meas, d.obs_ld, i, ms);
    load_soa<6>(d.obs_w, d.obs_ld, i, w);
    {
      // every lane takes the same path: a perturbation that does not apply is the zero step, which is the
      // exact identity for a pose; the plane keeps its stored value unless it is the perturbed node
      double pp[7], lp[4];
      perturb6(pz, q, sgn, pp);                       // q >= 6: zero delta -> pp == pz bit for bit
      perturb3(pl, q - 6, sgn, lp);
      const bool pert_plane = q >= 6 && q < 9;
#pragma unroll
      for (int k = 0; k < 4; k++) lp[k] = pert_plane ? lp[k] : pl[k];
      res_plane_obs(pp, lp, ms, e);
    }
    whiten<3>(w, e, y);
    double* __restrict__ out = d.J + d.joff_obs + (size_t)i * 30;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const double other = __shfl_xor(y[r], 1, 64);
      if (!(gl & 1)) {
        const double dcol = (y[r] - other) * inv2e;
        if (q < 6) out[r * 6 + q] = dcol;
        else if (q < 9) out[18 + r * 3 + (q - 6)] = dcol;
        else if (gl == 18) out[27 + r] = y[r];
      }
    }
    return;
  }
  b -= nb_obs;
  if (b < nb_odo) {
    const int i = b * kFactorsPerBlock + grp;
    if (i >= d.n_odo) return;
    double p1[7], p2[7], ms[6], w[21], e[6], y[6];
    load_pose(pose, d.pose_ld, d.odo_a[i], p1);
    load_pose(pose, d.pose_ld, d.odo_b[i], p2);
    load_soa<6>(d.odo_meas, d.odo_ld, i, ms);
    load_soa<21>(d.odo_w, d.odo_ld, i, w);
    {
      double pa[7], pb[7];
      perturb6(p1, q, sgn, pa);                       // out-of-range q: zero step == identity
      perturb6(p2, q - 6, sgn, pb);
      res_odometry(pa, pb, ms, e);
    }
    whiten<6>(w, e, y);
    double* __restrict__ out = d.J + d.joff_odo + (size_t)i * 78;
#pragma unroll
    for (int r = 0; r < 6; r++) {
      const double other = __shfl_xor(y[r], 1, 64);
      if (!(gl & 1)) {
        const double dcol = (y[r] - other) * inv2e;
        if (q < 6) out[r * 6 + q] = dcol;
        else if (q < 12) out[36 + r * 6 + (q - 6)] = dcol;
        else if (gl == 24) out[72 + r] = y[r];
      }
    }
    return;
  }
  b -= nb_odo;
  if (b < nb_pp) {
    const int i = b * kFactorsPerBlock + grp;
    if (i >= d.n_pp) return;
    double pz[7], ms[6], w[21], e[6], y[6];
    load_pose(pose, d.pose_ld, d.pp_pose[i], pz);
    load_soa<6>(d.pp_meas, d.pp_ld, i, ms);
    load_soa<21>(d.pp_w, d.pp_ld, i, w);
    { double pp[7]; perturb6(pz, q, sgn, pp); res_pose_prior(pp, ms, e); }
    whiten<6>(w, e, y);
    double* __restrict__ out = d.J + d.joff_pp + (size_t)i * 42;
#pragma unroll
    for (int r = 0; r < 6; r++) {
      const double other = __shfl_xor(y[r], 1, 64);
      if (!(gl & 1)) {
        const double dcol = (y[r] - other) * inv2e;
        if (q < 6) out[r * 6 + q] = dcol;
        else if (gl == 12) out[36 + r] = y[r];
      }
    }
    return;
  }
  b -= nb_pp;
  {
    const int i = b * kFactorsPerBlock + grp;
    if (i >= d.n_lp) return;
    double pl[4], ms[4], w[6], e[3], y[3];
    load_plane(plane, d.plane_ld, d.lp_plane[i], pl);
    load_soa<4>(d.lp_meas, d.lp_ld, i, ms);
    load_soa<6>(d.lp_w, d.lp_ld, i, w);
    {
      double lp[4];
      perturb3(pl, q, sgn, lp);
#pragma unroll
      for (int k = 0; k < 4; k++) lp[k] = q < 3 ? lp[k] : pl[k];
      res_plane_prior(lp, ms, e);
    }
    whiten<3>(w, e, y);
    double* __restrict__ out = d.J + d.joff_lp + (size_t)i * 12;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const double other = __shfl_xor(y[r], 1, 64);
      if (!(gl & 1)) {
        const double dcol = (y[r] - other) * inv2e;
        if (q < 3) out[r * 3 + q] = dcol;
        else if (gl == 6) out[9 + r] = y[r];
      }
    }
  }
}

__global__ __launch_bounds__(kLanesPerBlock) void k_linearize_lanes(DevGraph d, const double* __restrict__ pose,
                                                                    const double* __restrict__ plane, int nb_obs, int nb_odo,
                                                                    int nb_pp, LinGuard gd) {
  if (!lin_guard(gd, pose, plane)) return;
  body_linearize_lanes(d, pose, plane, nb_obs, nb_odo, nb_pp, blockIdx.x);
}

// below this many factors the lane-parallel form wins (latency); above it the thread-per-factor
// form has the higher throughput (no idle lanes)
constexpr int kLaneParallelMaxFactors = 200000;

// Pose3d_Plane3d_Factor2 edges (slots [n_obs_fixed, n_obs)): central differences in both Jacobian modes -- the
// measurement moves with the pose perturbation (the reference differentiates it numerically too).
__device__ __forceinline__ void body_linearize_repop(const DevGraph& d, const double* __restrict__ pose,
                                                     const double* __restrict__ plane, int bx) {
  const int n2 = d.n_obs - d.n_obs_fixed;
  const int k = bx * 64 + threadIdx.x;
  if (k >= n2) return;
  const int i = d.n_obs_fixed + k;
  double pz[7], pl[4], ray[6], w[6], e[3], r[3], Jp[18], Jl[9];
  load_pose(pose, d.pose_ld, d.obs_pose[i], pz);
  load_plane(plane, d.plane_ld, d.obs_plane[i], pl);
  load_soa<6>(d.obs_ray, n2, k, ray);
  load_soa<6>(d.obs_w, d.obs_ld, i, w);
  res_plane_obs2(pz, pl, ray, e);
  whiten<3>(w, e, r);
  const double inv2e = 1.0 / (kNumDiffEps + kNumDiffEps);
  for (int j = 0; j < 6; j++) {
    double dl[6] = {0, 0, 0, 0, 0, 0}, pp[7], yp[3], ym[3];
    dl[j] = kNumDiffEps;
    pose_exmap(pz, dl, pp); res_plane_obs2(pp, pl, ray, e); whiten<3>(w, e, yp);
    dl[j] = -kNumDiffEps;
    pose_exmap(pz, dl, pp); res_plane_obs2(pp, pl, ray, e); whiten<3>(w, e, ym);
    for (int q = 0; q < 3; q++) Jp[q * 6 + j] = (yp[q] - ym[q]) * inv2e;
  }
  for (int j = 0; j < 3; j++) {
    double dl[3] = {0, 0, 0}, pp[4], yp[3], ym[3];
    dl[j] = kNumDiffEps;
    plane_exmap(pl, dl, pp); res_plane_obs2(pz, pp, ray, e); whiten<3>(w, e, yp);
    dl[j] = -kNumDiffEps;
    plane_exmap(pl, dl, pp); res_plane_obs2(pz, pp, ray, e); whiten<3>(w, e, ym);
    for (int q = 0; q < 3; q++) Jl[q * 3 + j] = (yp[q] - ym[q]) * inv2e;
  }
  double* __restrict__ out = d.J + d.joff_obs + (size_t)i * 30;
  for (int q = 0; q < 18; q++) out[q] = Jp[q];
  for (int q = 0; q < 9; q++) out[18 + q] = Jl[q];
  for (int q = 0; q < 3; q++) out[27 + q] = r[q];
}

__global__ __launch_bounds__(64) void k_linearize_repop(DevGraph d, const double* __restrict__ pose,
                                                        const double* __restrict__ plane, LinGuard gd) {
  if (!lin_guard(gd, pose, plane)) return;
  body_linearize_repop(d, pose, plane, blockIdx.x);
}

hipError_t launch_linearize(const DevGraph& d, int mode, bool at_estimate, hipStream_t st, const LinGuard* guard) {
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
  // PPS_K1_THREAD_FORM=1 forces the thread-per-factor kernels on small graphs (parity tests of that form)
  if (mode == 0 && d.n_obs + d.n_odo + d.n_pp + d.n_lp <= kLaneParallelMaxFactors && !getenv("PPS_K1_THREAD_FORM")) {
    const int lb_obs = cdiv(d.n_obs_fixed, kFactorsPerBlock), lb_odo = cdiv(d.n_odo, kFactorsPerBlock),
              lb_pp = cdiv(d.n_pp, kFactorsPerBlock), lb_lp = cdiv(d.n_lp, kFactorsPerBlock);
    PPS_LAUNCH(k_linearize_lanes, dim3(lb_obs + lb_odo + lb_pp + lb_lp), dim3(kLanesPerBlock), 0, st, d, pose, plane,
                       lb_obs, lb_odo, lb_pp, gd);
    return hipGetLastError();
  }
  const size_t lds0 = (size_t)(kLinBlock / 64) * 64 * 31 * sizeof(double), lds1 = (size_t)(kLinBlock / 64) * 64 * 79 * sizeof(double);
  const int nb_rest = nb - nb_obs;
  if (mode == 1) {
    if (nb_obs) PPS_LAUNCH((k_linearize<1, 0>), dim3(nb_obs), dim3(kLinBlock), lds0, st, d, pose, plane, nb_obs, nb_odo, nb_pp, gd);
    if (nb_rest) PPS_LAUNCH((k_linearize<1, 1>), dim3(nb_rest), dim3(kLinBlock), lds1, st, d, pose, plane, nb_obs, nb_odo, nb_pp, gd);
  } else {
    if (nb_obs) PPS_LAUNCH((k_linearize<0, 0>), dim3(nb_obs), dim3(kLinBlock), lds0, st, d, pose, plane, nb_obs, nb_odo, nb_pp, gd);
    if (nb_rest) PPS_LAUNCH((k_linearize<0, 1>), dim3(nb_rest), dim3(kLinBlock), lds1, st, d, pose, plane, nb_obs, nb_odo, nb_pp, gd);
  }
  return hipGetLastError();
}

// K1 over replicated plane/odometry edges (roofline micro-benchmark): replica r writes its own J slab.
template <int MODE, int PART>
__global__ __launch_bounds__(kLinBlock) void k_sweep_bench(DevGraph d, double* __restrict__ Jbig, int nb_obs_per,
                                                            int nb_odo_per, int replicas) {
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

hipError_t launch_sweep_bench(const DevGraph& d, int mode, int replicas, double* Jbig, int part, hipStream_t st) {
  const int nb_obs = cdiv(d.n_obs, kLinBlock), nb_odo = cdiv(d.n_odo, kLinBlock);
  const size_t lds0 = (size_t)(kLinBlock / 64) * 64 * 31 * sizeof(double), lds1 = (size_t)(kLinBlock / 64) * 64 * 79 * sizeof(double);
  if (nb_obs + nb_odo == 0) return hipSuccess;
  if (mode == 1) {
    if (nb_obs && part != 1) PPS_LAUNCH((k_sweep_bench<1, 0>), dim3(nb_obs * replicas), dim3(kLinBlock), lds0, st, d, Jbig, nb_obs, nb_odo, replicas);
    if (nb_odo && part != 0) PPS_LAUNCH((k_sweep_bench<1, 1>), dim3(nb_odo * replicas), dim3(kLinBlock), lds1, st, d, Jbig, nb_obs, nb_odo, replicas);
  } else {
    if (nb_obs && part != 1) PPS_LAUNCH((k_sweep_bench<0, 0>), dim3(nb_obs * replicas), dim3(kLinBlock), lds0, st, d, Jbig, nb_obs, nb_odo, replicas);
    if (nb_odo && part != 0) PPS_LAUNCH((k_sweep_bench<0, 1>), dim3(nb_odo * replicas), dim3(kLinBlock), lds1, st, d, Jbig, nb_obs, nb_odo, replicas);
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// K2: one wavefront per H-block segment; lane = block entry, loop over <= seg_len contributions.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void body_hblocks(const DevGraph& d, int bx) {
  const int seg = uni(bx * 4 + (threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  if (seg >= d.n_segs) return;
  // one coalesced load of the packed segment record, fields broadcast with v_readlane
  const int rec = d.srec[(size_t)seg * 8 + (lane & 7)];
  const int rows = __builtin_amdgcn_readlane(rec, 0), cols = __builtin_amdgcn_readlane(rec, 1), size = __builtin_amdgcn_readlane(rec, 2);
  const int c0 = __builtin_amdgcn_readlane(rec, 3), cnt = __builtin_amdgcn_readlane(rec, 4);
  const int hoff = __builtin_amdgcn_readlane(rec, 5), doff = __builtin_amdgcn_readlane(rec, 6), nsegb = __builtin_amdgcn_readlane(rec, 7);
  // one contribution descriptor per lane, fetched in a single coalesced load (cnt <= 64)
  int4 mine = make_int4(0, 0, 0, 0);
  if (lane < cnt) mine = reinterpret_cast<const int4*>(d.contrib)[c0 + lane];
  const int rc = rows * cols;
  const bool act = lane < size;
  // where the finished entry goes in front-gather order: does not depend on the values, so the load is issued now
  const int dst = (act && nsegb == 1) ? d.blk_dst[doff + lane] : -1;
  const bool is_g = lane >= rc;
  const int i = is_g ? lane - rc : lane / cols;
  const int j = is_g ? 0 : lane - (lane / cols) * cols;
  const double* __restrict__ J = d.J;
  double acc = 0.0;
  int c = 0;
  for (; c + 2 <= cnt; c += 2) {                          // two contributions' loads in flight (64 VGPRs: 8 waves per SIMD)
    double a[2][6], bb[2][6];
    int m[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int cc = c + u;
      const int jv = __builtin_amdgcn_readlane(mine.x, cc), ju = __builtin_amdgcn_readlane(mine.y, cc);
      const int ro = __builtin_amdgcn_readlane(mine.z, cc);
      m[u] = __builtin_amdgcn_readlane(mine.w, cc);
      const double* pa = J + jv + i;
      const double* pb = is_g ? J + ro : J + ju + j;
      const int sb = is_g ? 1 : cols;
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const bool ok = act && k < m[u];
        a[u][k] = ok ? pa[k * rows] : 0.0;
        bb[u][k] = ok ? pb[k * sb] : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
      for (int k = 0; k < 6; k++) acc = PPS_MAC(acc, a[u][k], bb[u][k]);
  }
  for (; c < cnt; c++) {                                  // tail, and the single-contribution segments (most pose-plane blocks)
    const int jv = __builtin_amdgcn_readlane(mine.x, c), ju = __builtin_amdgcn_readlane(mine.y, c);
    const int ro = __builtin_amdgcn_readlane(mine.z, c), mm = __builtin_amdgcn_readlane(mine.w, c);
    const double* pa = J + jv + i;
    const double* pb = is_g ? J + ro : J + ju + j;
    const int sb = is_g ? 1 : cols;
    double a[6], bb[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const bool ok = act && k < mm;
      a[k] = ok ? pa[k * rows] : 0.0;
      bb[k] = ok ? pb[k * sb] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 6; k++) acc = PPS_MAC(acc, a[k], bb[k]);
  }
  if (!act) return;
  if (is_g) acc = -acc;                                   // b = -r (isam/Jacobian.h:98)
  d.H[hoff + lane] = acc;
  if (dst >= 0) d.Hf[dst] = acc;                          // final value (single-segment block): also where its front gathers it
}

__global__ __launch_bounds__(256, 2) void k_hblocks(DevGraph d, LinGuard gd) { if (!lin_guard(gd)) return; body_hblocks(d, blockIdx.x); }

// Throughput form (many graphs per launch): a wave takes S consecutive segments.  The three dependent round trips of a
// segment -- record, contribution descriptors, Jacobian slices -- are each issued for all S segments before the first
// answer is needed; the sums run in the order of body_hblocks (contribution by contribution, k ascending), bit for bit.
template <int S>
__device__ __forceinline__ void body_hblocks_t(const DevGraph& d, int bx) {
  const int slot0 = uni((bx * 4 + (threadIdx.x >> 6)) * S);        // position in the list of non-direct segments
  const int lane = threadIdx.x & 63;
  if (slot0 >= d.n_nd_segs) return;
  int rec[S];
  {
    const int sidx = (lane >> 3) < S && slot0 + (lane >> 3) < d.n_nd_segs ? d.nd_segs[slot0 + (lane >> 3)] : -1;   // lanes 8q..8q+7: segment q
#pragma unroll
    for (int q = 0; q < S; q++) {
      const int sg = __builtin_amdgcn_readlane(sidx, 8 * q);
      rec[q] = sg >= 0 ? d.srec[(size_t)sg * 8 + (lane & 7)] : 0;
    }
  }
  int rows[S], cols[S], size[S], cnt[S], hoff[S], dst[S], ii[S], jj[S];
  bool act[S], isg[S];
  int4 mine[S];
#pragma unroll
  for (int q = 0; q < S; q++) {
    rows[q] = __builtin_amdgcn_readlane(rec[q], 0); cols[q] = __builtin_amdgcn_readlane(rec[q], 1); size[q] = __builtin_amdgcn_readlane(rec[q], 2);
    const int c0 = __builtin_amdgcn_readlane(rec[q], 3);
    cnt[q] = __builtin_amdgcn_readlane(rec[q], 4); hoff[q] = __builtin_amdgcn_readlane(rec[q], 5);
    const int doff = __builtin_amdgcn_readlane(rec[q], 6), nsegb = __builtin_amdgcn_readlane(rec[q], 7);
    mine[q] = make_int4(0, 0, 0, 0);
    if (lane < cnt[q]) mine[q] = reinterpret_cast<const int4*>(d.contrib)[c0 + lane];
    act[q] = lane < size[q];                                  // size 0 for a segment past the end
    dst[q] = (act[q] && nsegb == 1) ? d.blk_dst[doff + lane] : -1;
    const int rc = rows[q] * cols[q];
    isg[q] = lane >= rc;
    const int cq = cols[q] > 0 ? cols[q] : 1;
    ii[q] = isg[q] ? lane - rc : lane / cq;
    jj[q] = isg[q] ? 0 : lane - (lane / cq) * cq;
  }
  const double* __restrict__ J = d.J;
  double a[S][6], bb[S][6];
#pragma unroll
  for (int q = 0; q < S; q++) {                               // first contribution of every segment: all loads in flight together
    const int jv = __builtin_amdgcn_readlane(mine[q].x, 0), ju = __builtin_amdgcn_readlane(mine[q].y, 0);
    const int ro = __builtin_amdgcn_readlane(mine[q].z, 0), mm = cnt[q] > 0 ? __builtin_amdgcn_readlane(mine[q].w, 0) : 0;
    const double* pa = J + jv + ii[q];
    const double* pb = isg[q] ? J + ro : J + ju + jj[q];
    const int sb = isg[q] ? 1 : cols[q];
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const bool ok = act[q] && k < mm;
      a[q][k] = ok ? pa[k * rows[q]] : 0.0;
      bb[q][k] = ok ? pb[k * sb] : 0.0;
    }
  }
#pragma unroll
  for (int q = 0; q < S; q++) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) acc = PPS_MAC(acc, a[q][k], bb[q][k]);
    for (int c = 1; c < cnt[q]; c++) {                        // further contributions (diagonal blocks)
      const int jv = __builtin_amdgcn_readlane(mine[q].x, c), ju = __builtin_amdgcn_readlane(mine[q].y, c);
      const int ro = __builtin_amdgcn_readlane(mine[q].z, c), mm = __builtin_amdgcn_readlane(mine[q].w, c);
      const double* pa = J + jv + ii[q];
      const double* pb = isg[q] ? J + ro : J + ju + jj[q];
      const int sb = isg[q] ? 1 : cols[q];
      double a2[6], b2[6];
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const bool ok = act[q] && k < mm;
        a2[k] = ok ? pa[k * rows[q]] : 0.0;
        b2[k] = ok ? pb[k * sb] : 0.0;
      }
#pragma unroll
      for (int k = 0; k < 6; k++) acc = PPS_MAC(acc, a2[k], b2[k]);
    }
    if (act[q]) {
      if (isg[q]) acc = -acc;                                 // b = -r (isam/Jacobian.h:98)
      d.H[hoff[q] + lane] = acc;
      if (dst[q] >= 0) d.Hf[dst[q]] = acc;
    }
  }
}

// fold the partial sums of multi-segment blocks (the ground plane's diagonal) into their first slot
__device__ __forceinline__ void body_hreduce(const DevGraph& d, int bx) {
  const int blk = d.mseg_blk[bx];
  const int size = d.blk_size[blk], nseg = d.blk_nseg[blk];
  double* __restrict__ h = d.H + d.blk_hoff[blk];
  const int lane = threadIdx.x;
  if (lane >= size) return;
  double v = 0.0;
  for (int q = 0; q < nseg; q += 16) {
    double x[16];
#pragma unroll
    for (int u = 0; u < 16; u++) x[u] = (q + u < nseg) ? h[(size_t)(q + u) * size + lane] : 0.0;
#pragma unroll
    for (int u = 0; u < 16; u++) v += x[u];
  }
  h[lane] = v;
  const int dst = d.blk_dst[d.blk_doff[blk] + lane];
  if (dst >= 0) d.Hf[dst] = v;
}

__global__ __launch_bounds__(64) void k_hreduce(DevGraph d, LinGuard gd) { if (!lin_guard(gd)) return; body_hreduce(d, blockIdx.x); }

hipError_t launch_hblocks(const DevGraph& d, hipStream_t st, const LinGuard* guard) {
  if (d.n_segs == 0) return hipSuccess;
  const LinGuard gd = guard ? *guard : LinGuard{};
  PPS_LAUNCH(k_hblocks, dim3(cdiv(d.n_segs, 4)), dim3(256), 0, st, d, gd);
  if (d.n_mseg > 0) PPS_LAUNCH(k_hreduce, dim3(d.n_mseg), dim3(64), 0, st, d, gd);
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
      const int rows = d.blk_rows[blk], cols = d.blk_cols[blk], size = d.blk_size[blk];
      const double* __restrict__ h = d.H + d.blk_hoff[blk];
      const int rc = rows * cols;
      const bool diag = size > rc;
      if (lane < size) {
        double v = h[lane];   // k_hreduce has folded multi-segment blocks into their first slot
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

static std::atomic<bool> g_attr_set[64];   // per device ordinal (idempotent set-up: a race only repeats it)

hipError_t launch_factor_level(const DevGraph& d, int level_begin, int level_count, int level_max_front, double lambda,
                               hipStream_t st) {
  if (level_count == 0) return hipSuccess;
  const int fa = level_max_front + 1;
  const size_t bytes = (size_t)fa * (fa | 1) * 8;
  if (bytes <= (size_t)kLdsLimitBytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!g_attr_set[dev & 63]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_front_factor<true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
      if (e != hipSuccess) return e;
      g_attr_set[dev & 63] = true;
    }
    PPS_LAUNCH(k_front_factor<true>, dim3(level_count), dim3(256), bytes, st, d, level_begin, lambda);
  } else {
    PPS_LAUNCH(k_front_factor<false>, dim3(level_count), dim3(256), 0, st, d, level_begin, lambda);
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// K3, wave-per-front form.  The tree levels are cut into bands; inside a band every connected
// sub-tree ("group") is walked by one workgroup: wave w takes fronts w, w+nw, ... of the current
// local level, a workgroup barrier separates the levels, update matrices travel through global
// memory (same CU, workgroup-scope visibility).  A front is a packed lower triangle in LDS
// (index(i,j) = i(i+1)/2 + j, last row = right-hand side); lane i owns row i (and i+64), so the
// elimination needs no barrier at all: column k is scaled, written back and re-read as LDS
// broadcasts by the same wave, in program order.
// ------------------------------------------------------------------------------------------
constexpr int kBandMaxRows = 128;   // rows per front including the rhs row

// One workgroup per front: row i of its (b+1)-row packed update matrix goes to row cmap[i] of the parent.
__global__ __launch_bounds__(64) void k_expand_ea(DevGraph d) {
  const int s = blockIdx.x;
  const int* __restrict__ m = d.cmap + d.f_cmap_off[s];
  const int len = d.f_cmap_off[s + 1] - d.f_cmap_off[s];
  int* __restrict__ out = d.ea_tgt + d.f_ea_off[s];
  const int n = len * (len + 1) / 2;
  int i = 0;                                  // row of entry e: tri(i) <= e < tri(i+1)
  for (int e = threadIdx.x; e < n; e += 64) {
    while ((i + 1) * (i + 2) / 2 <= e) i++;
    const int j = e - i * (i + 1) / 2;
    out[e] = m[i] * (m[i] + 1) / 2 + m[j];
  }
}

// One thread per assembled H block: its elements in the order the front gathers them (lower triangle of a diagonal
// block, a full off-diagonal block, then the gradient entries of a diagonal block).
__global__ __launch_bounds__(256) void k_expand_el(DevGraph d, int n_asm) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n_asm) return;
  const int blk = d.asm_blk[a], rows = d.blk_rows[blk], cols = d.blk_cols[blk];
  const int lrow = d.asm_lrow[a], lcol = d.asm_lcol[a];
  const bool diag = d.blk_size[blk] != rows * cols;
  int* __restrict__ dst = d.blk_dst + d.blk_doff[blk];
  int e = d.asm_el0[a];
  for (int i = 0; i < rows; i++)
    for (int j = 0; j < cols; j++) {
      if (diag && j > i) continue;
      dst[i * cols + j] = e;
      d.el_tgt[e++] = (((lrow + i) * (lrow + i + 1)) / 2 + lcol + j) | ((diag && i == j) ? (1 << 30) : 0);
    }
  if (diag) {
    const int fsz = d.asm_fsz[a];
    for (int i = 0; i < rows; i++) { dst[rows * cols + i] = e; d.el_tgt[e++] = (fsz * (fsz + 1)) / 2 + lcol + i; }
  }
}

hipError_t launch_expand_el(const DevGraph& d, int n_asm, hipStream_t st) {
  if (n_asm <= 0) return hipSuccess;
  PPS_LAUNCH(k_expand_el, dim3((n_asm + 255) / 256), dim3(256), 0, st, d, n_asm);
  return hipGetLastError();
}

hipError_t launch_expand_ea(const DevGraph& d, int n_fronts, hipStream_t st) {
  if (n_fronts <= 0) return hipSuccess;
  PPS_LAUNCH(k_expand_ea, dim3(n_fronts), dim3(64), 0, st, d);
  return hipGetLastError();
}

int band_front_limit() { return kBandMaxRows - 1; }
int band_reg_rows() { return kRegRows; }
int band_max_rows() { return kBandMaxRows; }
size_t band_lds_bytes(int max_front, bool reg_only_kernel) {       // (the register-only kernels keep a 64-row panel buffer, the others 80 rows: strip)
  const size_t fa = (size_t)max_front + 1;
  return (fa * (fa + 1) / 2 + (reg_only_kernel ? kRegRows : kRegRowsMax) * kPStride) * sizeof(double);
}   // packed triangle + panel buffer

__device__ __forceinline__ int tri(int i) { return (i * (i + 1)) >> 1; }

// One row of 16x16 tiles (I, J = o, o+16, ..., I) of the trailing lower triangle gets its rank-nb
// update C -= P_I P_J^T: all LDS reads are issued unconditionally from clamped (always valid)
// addresses and masked by selects afterwards, so the NT tiles' loads overlap; then NT back-to-back
// v_mfma_f64_16x16x4_f64; then the masked stores.
template <int NT>
__device__ __forceinline__ void trailing_tile_row(double* __restrict__ F, int fa, int o, int I, int K, int nb, int lane) {
  const int l16 = lane & 15, lq = lane >> 4;
  const int kk = K + lq;                                   // this lane's k index of the MFMA operands
  const bool kvalid = lq < nb;
  const int ar = I + l16;
  const bool aok = kvalid && ar < fa;
  const double araw = F[aok ? tri(ar) + kk : 0];
  const double av = aok ? -araw : 0.0;
  int crt[4]; bool rok[4];
#pragma unroll
  for (int r = 0; r < 4; r++) { const int cr = I + lq + 4 * r; rok[r] = cr < fa; crt[r] = rok[r] ? tri(cr) : 0; }   // D layout: row (l/16)+4r, col l%16
  double bv[NT];
  double4_t c[NT];
  bool cok[NT][4];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int br = o + 16 * t + l16;
    const bool bok = kvalid && br < fa;
    const double braw = F[bok ? tri(br) + kk : 0];
    bv[t] = bok ? braw : 0.0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      cok[t][r] = rok[r] && br <= I + lq + 4 * r;
      const double craw = F[cok[t][r] ? crt[r] + br : 0];
      c[t][r] = cok[t][r] ? craw : 0.0;
    }
  }
#pragma unroll
  for (int t = 0; t < NT; t++) c[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[t], c[t], 0, 0, 0);
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int cc = o + 16 * t + l16;
#pragma unroll
    for (int r = 0; r < 4; r++)
      if (cok[t][r]) F[crt[r] + cc] = c[t][r];
  }
}

__device__ __forceinline__ void wave_front_factor(const DevGraph& d, int s_in, double lambda, double* __restrict__ F) {
  const int lane = threadIdx.x & 63;
  const int s = uni(s_in);
  const int p = uni(d.f_p[s]), b = uni(d.f_b[s]);
  const int f = p + b, fa = f + 1;
  const int ntri = tri(fa);
  PPS_TR(0);
  for (int i = lane; i < ntri; i += 64) F[i] = 0.0;
  __builtin_amdgcn_wave_barrier();
  PPS_TR(1);
  // ---- original entries: Hf is in gather order, so value and target index are two independent
  // coalesced streams; 8 elements per lane are fetched before the first LDS update ----
  {
    const int e0 = uni(d.f_el_off[s]), e1 = uni(d.f_el_off[s + 1]);
    const double damp = 1.0 + lambda;
    const int* __restrict__ tgp = d.el_tgt;
    const double* __restrict__ hf = d.Hf;
    for (int e = e0 + lane; e < e1; e += 64 * 8) {
      int tg[8]; double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int x = e + 64 * u; tg[u] = x < e1 ? tgp[x] : -1; v[u] = x < e1 ? hf[x] : 0.0; }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (tg[u] >= 0) F[tg[u] & 0x3fffffff] += (tg[u] & (1 << 30)) ? v[u] * damp : v[u];   // Cholesky.cpp:94-97
    }
  }
  __builtin_amdgcn_wave_barrier();
  PPS_TR(2);
  // ---- extend-add of the children's packed update matrices (same batching) ----
  const int ci0 = uni(d.f_child_off[s]), ci1 = uni(d.f_child_off[s + 1]);
  for (int ci = ci0; ci < ci1; ci++) {
    const int c = uni(d.child[ci]);
    const int bc1 = uni(d.f_b[c]) + 1;
    const int n = tri(bc1);
    const double* __restrict__ Uc = d.U + uni64(d.f_Uoff[c]);
    const int* __restrict__ tgc = d.ea_tgt + uni64(d.f_ea_off[c]);
    for (int e = lane; e < n; e += 64 * 8) {
      int tg[8]; double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int x = e + 64 * u; tg[u] = x < n ? tgc[x] : -1; v[u] = x < n ? Uc[x] : 0.0; }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (tg[u] >= 0) F[tg[u]] += v[u];
    }
    __builtin_amdgcn_wave_barrier();
  }
  PPS_TR(3);
  // ---- eliminate the p pivot columns, four at a time ----
  // Panel: lane i holds rows i and i+64 of the 4 panel columns in registers; the 4x4 diagonal block is
  // broadcast with v_readlane, so the panel factorisation never waits on LDS.  Trailing update: the
  // rank-4 update C -= P_I * P_J^T of every 16x16 tile of the remaining lower triangle is ONE
  // v_mfma_f64_16x16x4_f64 (A = -P_I, B = P_J^T), operands gathered from the packed triangle in LDS.
  const int r0 = lane, r1 = lane + 64;
  const int t0 = tri(r0), t1 = tri(r1);
  long long cyc_panel = 0, cyc_trail = 0;
  for (int K = 0; K < p; K += 4) {
    const long long tk0 = d.trace ? clock64() : 0;
    double a0[4], a1[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const int c = K + m;
      const bool v0 = c < p && r0 >= c && r0 < fa, v1 = c < p && r1 >= c && r1 < fa;
      const double x0 = F[v0 ? t0 + c : 0], x1 = F[v1 ? t1 + c : 0];
      a0[m] = v0 ? x0 : 0.0;
      a1[m] = v1 ? x1 : 0.0;
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const int c = K + m;
      if (c < p) {                                         // wave-uniform
        const double dmm = (c < 64) ? readlane_d(a0[m], c) : readlane_d(a1[m], c - 64);
        double dinv = 0.0;
        if (dmm > 0.0) dinv = rsqrt_nr(dmm);
        else if (lane == 0) d.result_dev[2] = 1.0;         // not positive definite
        if (r0 >= c) a0[m] *= dinv;                        // the diagonal becomes sqrt(dmm)
        if (r1 >= c) a1[m] *= dinv;
#pragma unroll
        for (int n = m + 1; n < 4; n++) {
          const int cn = K + n;
          if (cn < p) {
            const double lnm = (cn < 64) ? readlane_d(a0[m], cn) : readlane_d(a1[m], cn - 64);
            if (r0 >= cn) a0[n] -= a0[m] * lnm;
            if (r1 >= cn) a1[n] -= a1[m] * lnm;
          }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const int c = K + m;
      if (c < p) {
        if (r0 >= c && r0 < fa) F[t0 + c] = a0[m];
        if (r1 >= c && r1 < fa) F[t1 + c] = a1[m];
      }
    }
    __builtin_amdgcn_wave_barrier();
    const long long tk1 = d.trace ? clock64() : 0;
    const int o = K + 4 < p ? K + 4 : p;                   // first trailing column
    const int nb = o - K;                                  // panel width (1..4)
    for (int I = o; I < fa; I += 16) {
      const int nt = ((I - o) >> 4) + 1;                   // tiles (I, J <= I) of this tile row, wave-uniform
      switch (nt) {
        case 1: trailing_tile_row<1>(F, fa, o, I, K, nb, lane); break;
        case 2: trailing_tile_row<2>(F, fa, o, I, K, nb, lane); break;
        case 3: trailing_tile_row<3>(F, fa, o, I, K, nb, lane); break;
        case 4: trailing_tile_row<4>(F, fa, o, I, K, nb, lane); break;
        case 5: trailing_tile_row<5>(F, fa, o, I, K, nb, lane); break;
        case 6: trailing_tile_row<6>(F, fa, o, I, K, nb, lane); break;
        case 7: trailing_tile_row<7>(F, fa, o, I, K, nb, lane); break;
        default: trailing_tile_row<8>(F, fa, o, I, K, nb, lane); break;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (d.trace) { const long long tk2 = clock64(); cyc_panel += tk1 - tk0; cyc_trail += tk2 - tk1; }
  }
  PPS_TR(4);
  if (d.trace && lane == 0) { d.trace[(size_t)s * 8 + 6] = cyc_panel; d.trace[(size_t)s * 8 + 7] = cyc_trail; }
  // ---- factor panel (f+1) x p row-major (diagonal = sqrt) and packed update matrix ----
  double* __restrict__ Lp = d.L + uni64(d.f_Loff[s]);
  if (lane < p) {                                          // p <= 64: lane = column
    for (int i0 = 0; i0 < fa; i0 += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int i = i0 + u; v[u] = F[(i < fa && lane <= i) ? tri(i) + lane : 0]; }
#pragma unroll
      for (int u = 0; u < 8; u++) { const int i = i0 + u; if (i < fa && lane <= i) Lp[(size_t)i * p + lane] = v[u]; }
    }
  }
  double* __restrict__ Us = d.U + uni64(d.f_Uoff[s]);
  for (int i0 = p; i0 < fa; i0 += 8) {
    for (int j = p + lane; j < fa; j += 64) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int i = i0 + u; v[u] = F[(i < fa && j <= i) ? tri(i) + j : 0]; }
#pragma unroll
      for (int u = 0; u < 8; u++) { const int i = i0 + u; if (i < fa && j <= i) Us[tri(i - p) + j - p] = v[u]; }
    }
  }
  PPS_TR(5);
}

// ------------------------------------------------------------------------------------------
// Register-resident variant for fronts of <= 64 rows (all of C2, most of C3).  After assembly in LDS
// the whole front lives in the lanes' registers as the ten 16x16 tiles of the lower triangle, each in
// the MFMA accumulator layout (lane l, reg r <-> row (l/16)+4r, col l%16), so a rank-4 update of a
// tile is a single register-to-register v_mfma_f64_16x16x4_f64.  Per 4-column block only the panel
// moves through LDS: tile columns K..K+3 -> P[row][4] -> one lane per row solves its row against the
// 4x4 diagonal block (broadcast with v_readlane, Cholesky-factored redundantly by every lane) ->
// P feeds the MFMA operands.  Entries left of / above the current block are dead and may hold garbage.
// ------------------------------------------------------------------------------------------
// TR: in-kernel phase trace (PPS_TRACE=1) compiled in
template <int NT, bool TR, bool STRIP = false>
__device__ __forceinline__ void wave_front_factor_reg(const DevGraph& d, int rec, double lambda, double* __restrict__ F,
                                                      double* __restrict__ P) {
  const int lane = threadIdx.x & 63;
  const int s = __builtin_amdgcn_readlane(rec, 0), p = __builtin_amdgcn_readlane(rec, 1), b = __builtin_amdgcn_readlane(rec, 2);
  const int l16 = lane & 15, lq = lane >> 4;
  const int f = p + b, fa = f + 1;
  const int ntri = tri(fa);
  // Fronts of 65 .. 80 rows (full kernel only): rows 0 .. 63 live in the register tiles as usual; rows 64 .. fa-1 -- boundary
  // rows, the pivots are among the first 48 -- stay where the assembly put them, in the packed LDS triangle, and are carried
  // along panel by panel: triangular solve by lanes 0 .. 15, rank-4 update with lane = column.
  const bool strip = STRIP && fa > kRegRows;
  if (TR) PPS_TR(0);
  const int e0 = __builtin_amdgcn_readlane(rec, 3), e1 = __builtin_amdgcn_readlane(rec, 4);
  const int cr0 = __builtin_amdgcn_readlane(rec, 5), nch = __builtin_amdgcn_readlane(rec, 6);
  const double damp = 1.0 + lambda;
  const int* __restrict__ tgp = d.el_tgt;
  const double* __restrict__ hf = d.Hf;
  // first gather batch and the child records are requested before the LDS triangle is cleared, so the
  // clearing hides under their latency
  int tg0[8]; double v0[8];
#pragma unroll
  for (int u = 0; u < 8; u++) { const int x = e0 + lane + 64 * u; tg0[u] = x < e1 ? tgp[x] : -1; v0[u] = x < e1 ? hf[x] : 0.0; }
  const int crv0 = (lane < 8 * nch) ? d.crec[(size_t)cr0 * 8 + lane] : 0;
  for (int i = lane; i < ntri; i += 64) F[i] = 0.0;
  __builtin_amdgcn_wave_barrier();
  if (TR) PPS_TR(1);
#pragma unroll
  for (int u = 0; u < 8; u++)
    if (tg0[u] >= 0) F[tg0[u] & 0x3fffffff] += (tg0[u] & (1 << 30)) ? v0[u] * damp : v0[u];   // Cholesky.cpp:94-97
  for (int e = e0 + lane + 64 * 8; e < e1; e += 64 * 8) {
    int tg[8]; double v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { const int x = e + 64 * u; tg[u] = x < e1 ? tgp[x] : -1; v[u] = x < e1 ? hf[x] : 0.0; }
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (tg[u] >= 0) F[tg[u] & 0x3fffffff] += (tg[u] & (1 << 30)) ? v[u] * damp : v[u];
  }
  __builtin_amdgcn_wave_barrier();
  if (TR) PPS_TR(2);
  for (int cb = 0; cb < nch; cb += 8) {
   // child records of up to 8 children in one coalesced load (the first batch was requested above)
   const int crv = cb == 0 ? crv0 : ((lane < 8 * (nch - cb)) ? d.crec[(size_t)(cr0 + cb) * 8 + lane] : 0);
   for (int cj = 0; cj < 8 && cb + cj < nch; cj++) {
    const int n = __builtin_amdgcn_readlane(crv, 8 * cj);
    const long long uo = ((long long)__builtin_amdgcn_readlane(crv, 8 * cj + 2) << 32) | (unsigned int)__builtin_amdgcn_readlane(crv, 8 * cj + 1);
    const long long eo = ((long long)__builtin_amdgcn_readlane(crv, 8 * cj + 4) << 32) | (unsigned int)__builtin_amdgcn_readlane(crv, 8 * cj + 3);
    const double* __restrict__ Uc = d.U + uo;
    const int* __restrict__ tgc = d.ea_tgt + eo;
    for (int e = lane; e < n; e += 64 * 8) {
      int tg[8]; double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int x = e + 64 * u; tg[u] = x < n ? tgc[x] : -1; v[u] = x < n ? Uc[x] : 0.0; }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (tg[u] >= 0) F[tg[u]] += v[u];
    }
    __builtin_amdgcn_wave_barrier();
   }
  }
  if (TR) PPS_TR(3);
  // ---- packed triangle -> register tiles ----
  double4_t c[NT * (NT + 1) / 2];
#pragma unroll
  for (int ti = 0; ti < NT; ti++)
#pragma unroll
    for (int tj = 0; tj <= ti; tj++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = 16 * ti + lq + 4 * r, col = 16 * tj + l16;
        const bool ok = row < fa && col <= row;
        const double x = F[ok ? tri(row) + col : 0];
        c[tile_id(ti, tj)][r] = ok ? x : 0.0;
      }
  double* __restrict__ Lp = d.L + (((long long)__builtin_amdgcn_readlane(rec, 10) << 32) | (unsigned int)__builtin_amdgcn_readlane(rec, 9));
  long long cyc_panel = 0, cyc_trail = 0;
  for (int K = 0; K < p; K += 4) {
    const long long tk0 = (TR && d.trace) ? clock64() : 0;
    const int nb = p - K < 4 ? p - K : 4;
    const int tjK = K >> 4, c0 = K & 15;
    switch (tjK) {
      case 0: reg_extract_panel<0, NT>(c, P, c0, lane); break;
      case 1: reg_extract_panel<1, NT>(c, P, c0, lane); break;
      case 2: if (NT > 2) reg_extract_panel<(NT > 2 ? 2 : 1), NT>(c, P, c0, lane); break;
      default: if (NT > 3) reg_extract_panel<(NT > 3 ? 3 : NT - 1), NT>(c, P, c0, lane); break;
    }
    const int row2 = kRegRows + (lane & 15);                   // the strip row of this lane (lanes 0 .. 15)
    const bool has2 = STRIP && strip && lane < 16 && row2 < fa;
    if (has2) {
#pragma unroll
      for (int m = 0; m < 4; m++) P[row2 * kPStride + m] = F[tri(row2) + K + m];   // (K + m < 64 <= row2: inside the row)
    }
    __builtin_amdgcn_wave_barrier();
    // ---- panel: lane = row ----
    double r0 = P[lane * kPStride + 0], r1 = P[lane * kPStride + 1], r2 = P[lane * kPStride + 2], r3 = P[lane * kPStride + 3];
    const double d00 = readlane_d(r0, K);
    const double d10 = readlane_d(r0, K + 1), d11 = readlane_d(r1, K + 1);
    const double d20 = readlane_d(r0, K + 2), d21 = readlane_d(r1, K + 2), d22 = readlane_d(r2, K + 2);
    const double d30 = readlane_d(r0, K + 3), d31 = readlane_d(r1, K + 3), d32 = readlane_d(r2, K + 3), d33 = readlane_d(r3, K + 3);
    bool bad = false;
    double i0 = 0, i1 = 0, i2 = 0, i3 = 0, l10 = 0, l20 = 0, l30 = 0, l21 = 0, l31 = 0, l32 = 0;
    double x0, x1, x2, x3;
    {
#ifndef PPS_NO_FMA
#pragma clang fp contract(fast)     // the serial pivot chain: a - b * c is one operation here
#endif
      { bad |= !(d00 > 0.0); i0 = d00 > 0.0 ? rsqrt_nr(d00) : 0.0; l10 = d10 * i0; l20 = d20 * i0; l30 = d30 * i0; }
      if (nb > 1) { const double t = d11 - l10 * l10; bad |= !(t > 0.0); i1 = t > 0.0 ? rsqrt_nr(t) : 0.0; l21 = (d21 - l20 * l10) * i1; l31 = (d31 - l30 * l10) * i1; }
      if (nb > 2) { const double t = d22 - l20 * l20 - l21 * l21; bad |= !(t > 0.0); i2 = t > 0.0 ? rsqrt_nr(t) : 0.0; l32 = (d32 - l30 * l20 - l31 * l21) * i2; }
      if (nb > 3) { const double t = d33 - l30 * l30 - l31 * l31 - l32 * l32; bad |= !(t > 0.0); i3 = t > 0.0 ? rsqrt_nr(t) : 0.0; }
      x0 = r0 * i0;
      x1 = (r1 - x0 * l10) * i1;
      x2 = (r2 - x0 * l20 - x1 * l21) * i2;
      x3 = (r3 - x0 * l30 - x1 * l31 - x2 * l32) * i3;
    }
    if (bad && lane == 0) d.result_dev[2] = 1.0;           // not positive definite
    P[lane * kPStride + 0] = x0; P[lane * kPStride + 1] = x1; P[lane * kPStride + 2] = x2; P[lane * kPStride + 3] = x3;
    if (lane < fa) {
      double* __restrict__ lrow = Lp + (size_t)lane * p + K;
      if (lane >= K) lrow[0] = x0;
      if (nb > 1 && lane >= K + 1) lrow[1] = x1;
      if (nb > 2 && lane >= K + 2) lrow[2] = x2;
      if (nb > 3 && lane >= K + 3) lrow[3] = x3;
    }
    if (STRIP && strip) {
      const double q0 = P[row2 * kPStride + 0], q1 = P[row2 * kPStride + 1], q2 = P[row2 * kPStride + 2], q3 = P[row2 * kPStride + 3];
      double y0, y1, y2, y3;
      {
#ifndef PPS_NO_FMA
#pragma clang fp contract(fast)
#endif
        y0 = q0 * i0;
        y1 = (q1 - y0 * l10) * i1;
        y2 = (q2 - y0 * l20 - y1 * l21) * i2;
        y3 = (q3 - y0 * l30 - y1 * l31 - y2 * l32) * i3;
      }
      if (has2) {
        P[row2 * kPStride + 0] = y0; P[row2 * kPStride + 1] = y1; P[row2 * kPStride + 2] = y2; P[row2 * kPStride + 3] = y3;
        double* __restrict__ lrow = Lp + (size_t)row2 * p + K;
        lrow[0] = y0;
        if (nb > 1) lrow[1] = y1;
        if (nb > 2) lrow[2] = y2;
        if (nb > 3) lrow[3] = y3;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (STRIP && strip) {
      // F[r][c] -= sum_k L[r][k] L[c][k] for the strip rows r and the live columns c >= K + nb: the strip is tile row 4 of the
      // front; its five 16x16 tiles are loaded from the LDS triangle, updated with one MFMA each and written back (only the
      // entries that exist: c <= r < fa).  Tile columns left of the panel are finished and skipped.
      const int cmin = K + nb;
      const bool kvalid = lq < nb;
      const double a4r = P[(kRegRows + l16) * kPStride + lq];
      const double a4 = kvalid ? -a4r : 0.0;
#pragma unroll 1                                    // one tile at a time: eight registers next to the ten resident tiles
      for (int tj = cmin >> 4; tj < 5; tj++) {
        const double br = P[(16 * tj + l16) * kPStride + lq];
        const double bj = kvalid ? br : 0.0;
        const int col = 16 * tj + l16;
        double4_t t;
        bool ok[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = kRegRows + lq + 4 * r;
          ok[r] = row < fa && col <= row && col >= cmin;
          const double x = F[ok[r] ? tri(row) + col : 0];
          t[r] = ok[r] ? x : 0.0;
        }
        t = __builtin_amdgcn_mfma_f64_16x16x4f64(a4, bj, t, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = kRegRows + lq + 4 * r;
          if (ok[r]) F[tri(row) + col] = t[r];
        }
      }
    }
    const long long tk1 = (TR && d.trace) ? clock64() : 0;
    switch (tjK) {
      case 0: reg_trailing<0, NT>(c, P, nb, lane, (K + 4) >> 4); break;
      case 1: reg_trailing<1, NT>(c, P, nb, lane, (K + 4) >> 4); break;
      case 2: if (NT > 2) reg_trailing<(NT > 2 ? 2 : 1), NT>(c, P, nb, lane, (K + 4) >> 4); break;
      default: if (NT > 3) reg_trailing<(NT > 3 ? 3 : NT - 1), NT>(c, P, nb, lane, (K + 4) >> 4); break;
    }
    __builtin_amdgcn_wave_barrier();
    if (TR && d.trace) { const long long tk2 = clock64(); cyc_panel += tk1 - tk0; cyc_trail += tk2 - tk1; }
  }
  if (TR) PPS_TR(4);
  if (TR && d.trace && lane == 0) { d.trace[(size_t)s * 8 + 6] = cyc_panel; d.trace[(size_t)s * 8 + 7] = cyc_trail; }
  // ---- update matrix: live part of the tiles -> packed global ----
  double* __restrict__ Us = d.U + (((long long)__builtin_amdgcn_readlane(rec, 12) << 32) | (unsigned int)__builtin_amdgcn_readlane(rec, 11));
#pragma unroll
  for (int ti = 0; ti < NT; ti++)
#pragma unroll
    for (int tj = 0; tj <= ti; tj++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = 16 * ti + lq + 4 * r, col = 16 * tj + l16;
        if (row < fa && col <= row && col >= p) Us[tri(row - p) + col - p] = c[tile_id(ti, tj)][r];
      }
  if (STRIP && strip) {
    for (int r = kRegRows; r < fa; r++)
      for (int col = p + lane; col <= r; col += 64) Us[tri(r - p) + col - p] = F[tri(r) + col];
  }
  if (TR) PPS_TR(5);
}


// x_p = L_A^-T (y - L_B^T x_b) for one front, one wave (p <= 64).  The factor panel ((f+1) x p, contiguous) is
// copied to LDS in batches of 16 independent coalesced loads per lane -- two round trips for a C2 front instead of one
// per 8 rows -- and everything after that reads LDS; the back-substitution chain runs in registers (lane j holds
// t_j, x_k is broadcast with v_readlane).  scratch: xb[128] + the panel.
__device__ __forceinline__ void wave_front_solve(const DevGraph& d, int rec, double* __restrict__ W, double* __restrict__ X, int slot) {
  const int lane = threadIdx.x & 63;
  const int p = __builtin_amdgcn_readlane(rec, 1), b = __builtin_amdgcn_readlane(rec, 2), f = p + b;
  const double* __restrict__ Lp = d.L + (((long long)__builtin_amdgcn_readlane(rec, 10) << 32) | (unsigned int)__builtin_amdgcn_readlane(rec, 9));
  const int pslot = __builtin_amdgcn_readlane(rec, 14);
  // boundary values: from the parent's local solution vector in LDS (through cmap) when the parent was solved by this
  // workgroup, else gathered from delta.  The index load does not depend on the parent and is issued first.
  const int* __restrict__ ix = pslot >= 0 ? d.cmap + __builtin_amdgcn_readlane(rec, 15) : d.bidx + __builtin_amdgcn_readlane(rec, 8);
  double* xb = W;
  double* PL = W + kBandMaxRows;
  const int ix0 = lane < b ? ix[lane] : 0, ix1 = lane + 64 < b ? ix[lane + 64] : 0;
  double g0 = 0.0, g1 = 0.0;
  if (pslot < 0) { g0 = d.delta[ix0]; g1 = d.delta[ix1]; }          // clamped index 0 when out of range: harmless
  const int n = (f + 1) * p;
  for (int e0 = 0; e0 < n; e0 += 64 * 16) {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) { const int e = e0 + 64 * u + lane; v[u] = Lp[e < n ? e : n - 1]; }
#pragma unroll
    for (int u = 0; u < 16; u++) { const int e = e0 + 64 * u + lane; if (e < n) PL[e] = v[u]; }
  }
  if (pslot >= 0) {
    const double* __restrict__ Xp = X + (size_t)pslot * kBandMaxRows;
    g0 = Xp[ix0]; g1 = Xp[ix1];
  }
  if (lane < b) xb[lane] = g0;
  if (lane + 64 < b) xb[lane + 64] = g1;
  __builtin_amdgcn_wave_barrier();
  double tj = 0.0, dinv = 0.0;
  {
#ifndef PPS_NO_FMA
#pragma clang fp contract(fast)     // dependent chains: a - b * c is one operation here
#endif
    if (lane < p) {
      // y - L_B^T x_b in two interleaved partial sums (half the dependent chain)
      double acc = PL[f * p + lane], acc2 = 0.0;
      const double* __restrict__ lb = PL + p * p + lane;
      int i = 0;
#pragma unroll 2
      for (; i + 2 <= b; i += 2) { acc -= lb[i * p] * xb[i]; acc2 -= lb[(i + 1) * p] * xb[i + 1]; }
      if (i < b) acc -= lb[i * p] * xb[i];
      tj = acc + acc2;
      dinv = 1.0 / PL[lane * p + lane];
    }
#pragma unroll 4
    for (int k = p - 1; k >= 0; k--) {
      const double lkj = (lane < k) ? PL[k * p + lane] : 0.0;      // independent of the chain
      const double xk = readlane_d(tj, k) * readlane_d(dinv, k);
      tj = (lane == k) ? xk : tj - lkj * xk;
    }
  }
  if (lane < p) d.delta[d.pidx[__builtin_amdgcn_readlane(rec, 7) + lane]] = tj;
  // own local solution [x_p | x_b] for the children inside this group
  double* __restrict__ Xs = X + (size_t)slot * kBandMaxRows;
  if (lane < p) Xs[lane] = tj;
  if (lane < b) Xs[p + lane] = g0;
  if (lane + 64 < b) Xs[p + lane + 64] = g1;
}

__device__ __forceinline__ void body_band_solve(const DevGraph& d, int g, int lds_doubles_per_wave, double* __restrict__ lds) {
  const int wave = uni(threadIdx.x >> 6), nw = blockDim.x >> 6;
  double* W = lds + (size_t)wave * lds_doubles_per_wave;
  double* X = lds + (size_t)nw * lds_doubles_per_wave;          // one local solution vector per front of the group
  const int l0 = d.grp_lvl_off[g], l1 = d.grp_lvl_off[g + 1];
  const int g0 = d.glvl_front_off[l0];
  for (int l = l1 - 1; l >= l0; l--) {
    const int i1 = d.glvl_front_off[l + 1];
    for (int i = d.glvl_front_off[l] + wave; i < i1; i += nw) {
      const int rec = d.frec[(size_t)i * 16 + (threadIdx.x & 15)];
      wave_front_solve(d, rec, W, X, i - g0);
    }
    __syncthreads();   // delta of this local level is visible to the children
  }
}

__global__ __launch_bounds__(512) void k_band_solve(DevGraph d, DualAlt alt, int grp_begin, int lds_doubles_per_wave) {
  extern __shared__ double lds[];
  if (blockIdx.y) { d.L = alt.L; d.U = alt.U; d.delta = alt.delta; }
  body_band_solve(d, grp_begin + blockIdx.x, lds_doubles_per_wave, lds);
}

// REG_ONLY: every front of the stage fits the register-resident path (C2: all stages) -- the LDS-tile path and the fused
// root solve are compiled out, which halves the kernel's code (the instruction cache is shared by two CUs)
// REG_STRIP (with REG_ONLY): the stage also holds fronts of 65 .. 80 rows -- ten register tiles + the LDS strip -- and still
// nothing that needs the LDS-tile path, the trace or the fused root solve (frame-loop trees, C3)
template <bool REG_ONLY, bool REG_STRIP = false>
__device__ __forceinline__ void body_band_factor(const DevGraph& d, int g, double lambda, int lds_doubles_per_wave,
                                                 int solve_doubles_per_wave, double* __restrict__ lds) {
  const int wave = uni(threadIdx.x >> 6), nw = blockDim.x >> 6;
  double* F = lds + (size_t)wave * lds_doubles_per_wave;
  const int l0 = d.grp_lvl_off[g], l1 = d.grp_lvl_off[g + 1];
  for (int l = l0; l < l1; l++) {
    const int i1 = d.glvl_front_off[l + 1];
    for (int i = d.glvl_front_off[l] + wave; i < i1; i += nw) {
      const int rec = d.frec[(size_t)i * 16 + (threadIdx.x & 15)];          // packed front record, one coalesced load
      const int s = __builtin_amdgcn_readlane(rec, 0);
      const int fa = __builtin_amdgcn_readlane(rec, 1) + __builtin_amdgcn_readlane(rec, 2) + 1;
      double* const Pn = F + lds_doubles_per_wave - ((REG_ONLY && !REG_STRIP) ? kRegRows : kRegRowsMax) * kPStride;
      if (REG_ONLY && fa <= 32) wave_front_factor_reg<2, false>(d, rec, lambda, F, Pn);
      else if (REG_ONLY && fa <= 48) wave_front_factor_reg<3, false>(d, rec, lambda, F, Pn);
      else if (REG_ONLY && (!REG_STRIP || fa <= kRegRows)) wave_front_factor_reg<4, false>(d, rec, lambda, F, Pn);
      else if (REG_ONLY) wave_front_factor_reg<4, false, true>(d, rec, lambda, F, Pn);                   // 65 .. 80 rows: register tiles + LDS strip
      else if (fa <= kRegRowsMax && !d.no_strip) wave_front_factor_reg<4, true, true>(d, rec, lambda, F, Pn);
      else if (fa <= kRegRows) wave_front_factor_reg<4, true>(d, rec, lambda, F, Pn);
      else wave_front_factor(d, s, lambda, F);
    }
    __syncthreads();   // children of the next local level are complete and visible (same CU)
  }
  if (REG_ONLY) return;
  // root stage: the back-substitution of the same group follows at once (one launch less per solve); the factor's
  // LDS is dead by now and is re-partitioned for the solve
  if (solve_doubles_per_wave > 0) {
    double* W = lds + (size_t)wave * solve_doubles_per_wave;
    double* X = lds + (size_t)nw * solve_doubles_per_wave;
    const int g0 = d.glvl_front_off[l0];
    for (int l = l1 - 1; l >= l0; l--) {
      const int i1 = d.glvl_front_off[l + 1];
      for (int i = d.glvl_front_off[l] + wave; i < i1; i += nw) {
        const int rec = d.frec[(size_t)i * 16 + (threadIdx.x & 15)];
        wave_front_solve(d, rec, W, X, i - g0);
      }
      __syncthreads();
    }
  }
}

template <bool REG_ONLY>
__global__ __launch_bounds__(512) void k_band_factor(DevGraph d, DualAlt alt, int grp_begin, double lambda, int lds_doubles_per_wave,
                                                     int solve_doubles_per_wave) {
  extern __shared__ double lds[];
  if (blockIdx.y) { d.L = alt.L; d.U = alt.U; d.delta = alt.delta; d.result_dev = alt.result_dev; lambda = alt.lambda; }
  body_band_factor<REG_ONLY>(d, grp_begin + blockIdx.x, lambda, lds_doubles_per_wave, solve_doubles_per_wave, lds);
}

__global__ __launch_bounds__(512) void k_band_factor_strip(DevGraph d, DualAlt alt, int grp_begin, double lambda, int lds_doubles_per_wave) {
  extern __shared__ double lds[];
  if (blockIdx.y) { d.L = alt.L; d.U = alt.U; d.delta = alt.delta; d.result_dev = alt.result_dev; lambda = alt.lambda; }
  body_band_factor<true, true>(d, grp_begin + blockIdx.x, lambda, lds_doubles_per_wave, 0, lds);
}

static std::atomic<bool> g_band_attr_set[64];   // per device ordinal

static hipError_t ensure_band_attrs() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!g_band_attr_set[dev & 63]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_factor<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_factor<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_solve), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_factor_strip), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e != hipSuccess) return e;
    g_band_attr_set[dev & 63] = true;
  }
  return hipSuccess;
}

hipError_t launch_band_factor(const DevGraph& d, int grp_begin, int grp_count, int nwaves, int max_front, double lambda, hipStream_t st,
                              int fused_solve_panel, int fused_solve_group_fronts) {
  if (grp_count == 0) return hipSuccess;
  { const hipError_t e = ensure_band_attrs(); if (e != hipSuccess) return e; }
  const bool reg_only = max_front + 1 <= kRegRows && fused_solve_panel <= 0 && d.trace == nullptr;
  const int per_wave = (int)(band_lds_bytes(max_front, reg_only) / sizeof(double));
  size_t bytes = (size_t)per_wave * nwaves * sizeof(double);
  int solve_per_wave = 0;
  if (fused_solve_panel > 0) {       // the stage's back-substitution runs in the same launch (root stage)
    solve_per_wave = (int)(band_solve_lds_bytes(fused_solve_panel) / sizeof(double));
    bytes = std::max(bytes, ((size_t)solve_per_wave * nwaves + (size_t)fused_solve_group_fronts * kBandMaxRows) * sizeof(double));
  }
  if (max_front + 1 <= kRegRows && solve_per_wave == 0 && d.trace == nullptr)   // (the phase trace lives in the full kernel)
    PPS_LAUNCH(k_band_factor<true>, dim3(grp_count), dim3(64 * nwaves), bytes, st, d, DualAlt{}, grp_begin, lambda, per_wave, 0);
  else if (max_front + 1 <= kRegRowsMax && solve_per_wave == 0 && d.trace == nullptr && !d.no_strip)
    PPS_LAUNCH(k_band_factor_strip, dim3(grp_count), dim3(64 * nwaves), bytes, st, d, DualAlt{}, grp_begin, lambda, per_wave);
  else
    PPS_LAUNCH(k_band_factor<false>, dim3(grp_count), dim3(64 * nwaves), bytes, st, d, DualAlt{}, grp_begin, lambda, per_wave, solve_per_wave);
  return hipGetLastError();
}

hipError_t launch_band_factor_dual(const DevGraph& d, const DualAlt& alt, int grp_begin, int grp_count, int nwaves, int max_front, double lambda,
                                   hipStream_t st) {
  if (grp_count == 0) return hipSuccess;
  { const hipError_t e = ensure_band_attrs(); if (e != hipSuccess) return e; }
  const int per_wave = (int)(band_lds_bytes(max_front, max_front + 1 <= kRegRows && d.trace == nullptr) / sizeof(double));
  const size_t bytes = (size_t)per_wave * nwaves * sizeof(double);
  if (max_front + 1 <= kRegRows && d.trace == nullptr)
    PPS_LAUNCH(k_band_factor<true>, dim3(grp_count, 2), dim3(64 * nwaves), bytes, st, d, alt, grp_begin, lambda, per_wave, 0);
  else if (max_front + 1 <= kRegRowsMax && d.trace == nullptr && !d.no_strip)
    PPS_LAUNCH(k_band_factor_strip, dim3(grp_count, 2), dim3(64 * nwaves), bytes, st, d, alt, grp_begin, lambda, per_wave);
  else
    PPS_LAUNCH(k_band_factor<false>, dim3(grp_count, 2), dim3(64 * nwaves), bytes, st, d, alt, grp_begin, lambda, per_wave, 0);
  return hipGetLastError();
}

size_t band_solve_lds_bytes(int max_panel) { return (size_t)(kBandMaxRows + max_panel) * sizeof(double); }   // xb + the factor panel

hipError_t launch_band_solve(const DevGraph& d, int grp_begin, int grp_count, int nwaves, int max_panel, int max_group_fronts, hipStream_t st,
                             const DualAlt* alt) {
  if (grp_count == 0) return hipSuccess;
  // per wave: xb + the largest factor panel of the stage; per workgroup: one local solution vector per front of a group
  const int per_wave = (int)(band_solve_lds_bytes(max_panel) / sizeof(double));
  PPS_LAUNCH(k_band_solve, dim3(grp_count, alt ? 2 : 1), dim3(64 * nwaves),
                     ((size_t)per_wave * nwaves + (size_t)max_group_fronts * kBandMaxRows) * sizeof(double), st, d, alt ? *alt : DualAlt{}, grp_begin,
                     per_wave);
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
  for (int k = lane; k < p; k += 64) d.delta[d.pidx[d.f_poff[s] + k]] = t[k];
}

hipError_t launch_backsolve_level(const DevGraph& d, int level_begin, int level_count, hipStream_t st) {
  if (level_count == 0) return hipSuccess;
  PPS_LAUNCH(k_front_solve, dim3(level_count), dim3(64), 0, st, d, level_begin);
  return hipGetLastError();
}

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

// out <- base (+) delta, nothing else touched: the speculative LM trial (step computed for lambda * factor on the
// second stream) is applied to a third copy of the state; |delta|^2 partials go to d.dn_partials
__global__ __launch_bounds__(256) void k_retract_to(DevGraph d, DualAlt alt, const double* __restrict__ base_pose,
                                                    const double* __restrict__ base_plane, double* __restrict__ out_pose,
                                                    double* __restrict__ out_plane, double* __restrict__ out_pose1, double* __restrict__ out_plane1) {
  __shared__ double red[4];
  if (blockIdx.y) { d.delta = alt.delta; d.dn_partials = alt.dn_partials; out_pose = out_pose1; out_plane = out_plane1; }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
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
  if (threadIdx.x == 0) d.dn_partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

hipError_t launch_retract_to(const DevGraph& d, const double* base_pose, const double* base_plane, double* out_pose, double* out_plane,
                             hipStream_t st) {
  const int n = d.n_pose + d.n_plane;
  if (n == 0) return hipSuccess;
  PPS_LAUNCH(k_retract_to, dim3(cdiv(n, 256)), dim3(256), 0, st, d, DualAlt{}, base_pose, base_plane, out_pose, out_plane, nullptr, nullptr);
  return hipGetLastError();
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

// bx: block within the graph, nb: blocks of the graph (the one that draws the last ticket reduces)
__device__ __forceinline__ void body_chi2(const DevGraph& d, const double* __restrict__ pose,
                                          const double* __restrict__ plane, int nb_obs, int nb_odo, int nb_pp,
                                          int n_dn, double* __restrict__ out, double seq, int bx, int nb) {
  __shared__ double red[kChiBlock / 64];
  int b = bx;
  double s = 0.0;
  if (b < nb_obs) {
    const int i = b * kChiBlock + threadIdx.x;
    if (i < d.n_obs) {
      double pz[7], pl[4], ms[4], w[6], e[3], r[3];
      load_pose(pose, d.pose_ld, d.obs_pose[i], pz);
      load_plane(plane, d.plane_ld, d.obs_plane[i], pl);
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
      load_pose(pose, d.pose_ld, d.odo_a[i], p1);
      load_pose(pose, d.pose_ld, d.odo_b[i], p2);
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
      load_pose(pose, d.pose_ld, d.pp_pose[i], pz);
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
      load_plane(plane, d.plane_ld, d.lp_plane[i], pl);
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
  __shared__ bool last;
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < kChiBlock / 64; k++) t += red[k];
    d.chi2_partials[bx] = t;
    // publish, then take a ticket: the block that draws the last one reduces everything
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    last = atomicAdd(d.ticket, 1u) == (unsigned int)(nb - 1);
  }
  __syncthreads();
  if (!last) return;
  if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  __syncthreads();
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
    *d.ticket = 0u;
  }
}

__global__ __launch_bounds__(kChiBlock) void k_chi2(DevGraph d, const double* __restrict__ pose,
                                                    const double* __restrict__ plane, int nb_obs, int nb_odo, int nb_pp,
                                                    int n_dn, double* __restrict__ out, double seq) {
  body_chi2(d, pose, plane, nb_obs, nb_odo, nb_pp, n_dn, out, seq, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(kChiBlock) void k_chi2_dual(DevGraph d, DualAlt alt, const double* __restrict__ pose, const double* __restrict__ plane,
                                                         const double* __restrict__ pose1, const double* __restrict__ plane1, int nb_obs,
                                                         int nb_odo, int nb_pp, int n_dn, double* __restrict__ out, double seq,
                                                         double* __restrict__ out1, double seq1) {
  if (blockIdx.y) {
    d.chi2_partials = alt.chi2_partials; d.dn_partials = alt.dn_partials; d.ticket = alt.ticket; d.result_dev = alt.result_dev;
    pose = pose1; plane = plane1; out = out1; seq = seq1;
  }
  body_chi2(d, pose, plane, nb_obs, nb_odo, nb_pp, n_dn, out, seq, blockIdx.x, gridDim.x);
}

hipError_t launch_trial_dual(const DevGraph& d, const DualAlt& alt, const double* base_pose, const double* base_plane, double* out_pose0,
                             double* out_plane0, double* out_pose1, double* out_plane1, double* host_result0, double seq0, double* host_result1,
                             double seq1, hipStream_t st) {
  const int n = d.n_pose + d.n_plane;
  if (n > 0)
    PPS_LAUNCH(k_retract_to, dim3(cdiv(n, 256), 2), dim3(256), 0, st, d, alt, base_pose, base_plane, out_pose0, out_plane0, out_pose1, out_plane1);
  const int nb_obs = cdiv(d.n_obs, kChiBlock), nb_odo = cdiv(d.n_odo, kChiBlock), nb_pp = cdiv(d.n_pp, kChiBlock), nb_lp = cdiv(d.n_lp, kChiBlock);
  const int nb = nb_obs + nb_odo + nb_pp + nb_lp;
  if (nb == 0) return hipErrorInvalidValue;
  PPS_LAUNCH(k_chi2_dual, dim3(nb, 2), dim3(kChiBlock), 0, st, d, alt, out_pose0, out_plane0, out_pose1, out_plane1, nb_obs, nb_odo, nb_pp,
                     cdiv(n, 256), host_result0, seq0, host_result1, seq1);
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

// chi2 at an explicit state (d.chi2_partials / d.ticket / d.dn_partials are the caller's: a second reduction may run
// concurrently on another stream with its own set)
hipError_t launch_chi2_at(const DevGraph& d, const double* pose, const double* plane, double* host_result, double seq, hipStream_t st) {
  const int nb_obs = cdiv(d.n_obs, kChiBlock), nb_odo = cdiv(d.n_odo, kChiBlock), nb_pp = cdiv(d.n_pp, kChiBlock),
            nb_lp = cdiv(d.n_lp, kChiBlock);
  const int nb = nb_obs + nb_odo + nb_pp + nb_lp;
  if (nb == 0) return hipErrorInvalidValue;
  const int n_dn = cdiv(d.n_pose + d.n_plane, 256);
  PPS_LAUNCH(k_chi2, dim3(nb), dim3(kChiBlock), 0, st, d, pose, plane, nb_obs, nb_odo, nb_pp, n_dn, host_result, seq);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Batched forms (pps_multi_*): the same bodies, blockIdx.y = graph.  The DevGraph record of the graph is read through
// the constant address space -- it does not change while a kernel runs -- so that its fields arrive by scalar loads
// into SGPRs exactly like the by-value kernel argument of the single-graph kernels.
// ------------------------------------------------------------------------------------------
// a pointer that came out of memory is generic to the compiler; these all point into HBM (or pinned host memory)
template <class T>
__device__ __forceinline__ T* gptr(T* p) {
  // through an integer, so that the address-space round trip is not folded away: on gfx9 a global address and its generic
  // form are the same 64 bits
  return (T*)(T __attribute__((address_space(1)))*)(unsigned long long)p;
}

__device__ __forceinline__ DevGraph load_graph(const DevGraph* gp) {
  DevGraph d = *(const DevGraph*)((const DevGraph __attribute__((address_space(4)))*)gp);   // scalar loads; unused fields drop out
#define PPS_G(f) d.f = gptr(d.f);
  PPS_G(pose_est) PPS_G(pose_lin) PPS_G(plane_est) PPS_G(plane_lin) PPS_G(pose_voff) PPS_G(plane_voff)
  PPS_G(obs_pose) PPS_G(obs_plane) PPS_G(obs_meas) PPS_G(obs_w) PPS_G(obs_ray) PPS_G(odo_a) PPS_G(odo_b) PPS_G(odo_meas) PPS_G(odo_w)
  PPS_G(pp_pose) PPS_G(pp_meas) PPS_G(pp_w) PPS_G(lp_plane) PPS_G(lp_meas) PPS_G(lp_w)
  PPS_G(J) PPS_G(H) PPS_G(L) PPS_G(U) PPS_G(delta)
  PPS_G(f_p) PPS_G(f_b) PPS_G(f_poff) PPS_G(pidx) PPS_G(f_Loff) PPS_G(f_Uoff) PPS_G(f_bidx_off) PPS_G(bidx) PPS_G(f_child_off) PPS_G(child)
  PPS_G(f_cmap_off) PPS_G(cmap) PPS_G(level_fronts) PPS_G(f_asm_off) PPS_G(asm_blk) PPS_G(asm_lrow) PPS_G(asm_lcol) PPS_G(asm_el0) PPS_G(asm_fsz)
  PPS_G(blk_rows) PPS_G(blk_cols) PPS_G(blk_size) PPS_G(blk_nseg) PPS_G(blk_hoff) PPS_G(seg_blk) PPS_G(seg_c0) PPS_G(seg_cnt) PPS_G(seg_hoff)
  PPS_G(contrib) PPS_G(mseg_blk) PPS_G(f_el_off) PPS_G(el_src) PPS_G(el_tgt) PPS_G(blk_doff) PPS_G(blk_dst) PPS_G(Hf) PPS_G(f_ea_off) PPS_G(ea_tgt)
  PPS_G(grp_lvl_off) PPS_G(glvl_front_off) PPS_G(glvl_fronts) PPS_G(frec) PPS_G(crec) PPS_G(srec)
  PPS_G(chi2_partials) PPS_G(dn_partials) PPS_G(ticket) PPS_G(result_dev) PPS_G(trace) PPS_G(gwork)
#undef PPS_G
  return d;
}

__device__ __forceinline__ BatchAlt load_alt(const BatchAlt* ap) {
  BatchAlt t = *(const BatchAlt*)((const BatchAlt __attribute__((address_space(4)))*)ap);
  t.L = gptr(t.L); t.U = gptr(t.U); t.delta = gptr(t.delta); t.result_dev = gptr(t.result_dev);
  t.chi2_partials = gptr(t.chi2_partials); t.dn_partials = gptr(t.dn_partials); t.ticket = gptr(t.ticket);
#pragma unroll
  for (int k = 0; k < 3; k++) { t.pose[k] = gptr(t.pose[k]); t.plane[k] = gptr(t.plane[k]); }
  return t;
}
__device__ __forceinline__ double* sel3(double* const (&p)[3], int k) { return k == 0 ? p[0] : (k == 1 ? p[1] : p[2]); }

// single-lambda form: est / lin exchanged by BF_SWAP.  Dual form (a.alt): the linearisation point is state[xsel] -- it takes the
// place of `lin` for K1 and chi2 -- and nothing is ever written over it.
#define PPS_BATCH_PROLOGUE(NEED)                                                                             \
  const int b = blockIdx.y;                                                                                  \
  const unsigned int fl = a.flags[b];                                                                        \
  if ((fl & (NEED)) != (NEED)) return;                                                                       \
  const DevGraph d = load_graph(a.gs + a.b0 + b);                                                            \
  const bool swp = (fl & BF_SWAP) != 0;                                                                      \
  double* pose_lin = swp ? d.pose_est : d.pose_lin;                                                          \
  double* pose_est = swp ? d.pose_lin : d.pose_est;                                                          \
  double* plane_lin = swp ? d.plane_est : d.plane_lin;                                                       \
  double* plane_est = swp ? d.plane_lin : d.plane_est;                                                       \
  if (a.alt) {                                                                                               \
    const BatchAlt al_ = load_alt(a.alt + a.b0 + b);                                                         \
    pose_lin = sel3(al_.pose, a.xsel[b]); plane_lin = sel3(al_.plane, a.xsel[b]);                            \
  }                                                                                                          \
  (void)pose_lin; (void)pose_est; (void)plane_lin; (void)plane_est;

__device__ __forceinline__ int dcdiv(int a, int b) { return (a + b - 1) / b; }

// lin <- est (estimate_to_linpoint, Optimizer.cpp:376) for the graphs of the chunk
__global__ __launch_bounds__(256) void kb_begin(BatchArgs a) {
  PPS_BATCH_PROLOGUE(BF_ACTIVE)
  const int np = 7 * d.pose_ld, nl = 4 * d.plane_ld;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < np + nl; i += gridDim.x * 256) {
    if (i < np) pose_lin[i] = pose_est[i]; else plane_lin[i - np] = plane_est[i - np];
  }
  if (blockIdx.x == 0 && threadIdx.x < 4) d.result_dev[threadIdx.x] = 0.0;
}

__global__ __launch_bounds__(kLanesPerBlock) void kb_linearize_lanes(BatchArgs a) {
  PPS_BATCH_PROLOGUE(BF_ACTIVE | BF_RELIN)
  const int nb_obs = dcdiv(d.n_obs_fixed, kFactorsPerBlock), nb_odo = dcdiv(d.n_odo, kFactorsPerBlock),
            nb_pp = dcdiv(d.n_pp, kFactorsPerBlock), nb_lp = dcdiv(d.n_lp, kFactorsPerBlock);
  if ((int)blockIdx.x >= nb_obs + nb_odo + nb_pp + nb_lp) return;
  body_linearize_lanes(d, pose_lin, plane_lin, nb_obs, nb_odo, nb_pp, blockIdx.x);
}

template <int MODE, int PART, bool DIRECT>
__global__ __launch_bounds__(kLinBlock) void kb_linearize(BatchArgs a) {
  extern __shared__ double lin_lds[];
  PPS_BATCH_PROLOGUE(BF_ACTIVE | BF_RELIN)
  const int nb_obs = dcdiv(d.n_obs_fixed, kLinBlock), nb_odo = dcdiv(d.n_odo, kLinBlock), nb_pp = dcdiv(d.n_pp, kLinBlock),
            nb_lp = dcdiv(d.n_lp, kLinBlock);
  if ((int)blockIdx.x >= (PART == 0 ? nb_obs : nb_odo + nb_pp + nb_lp)) return;
  body_linearize<MODE, PART, DIRECT>(d, pose_lin, plane_lin, nb_obs, nb_odo, nb_pp, blockIdx.x, lin_lds);   // DIRECT pairs with kb_hblocks_t
}

__global__ __launch_bounds__(64) void kb_linearize_repop(BatchArgs a) {
  PPS_BATCH_PROLOGUE(BF_ACTIVE | BF_RELIN)
  if ((int)blockIdx.x * 64 >= d.n_obs - d.n_obs_fixed) return;
  body_linearize_repop(d, pose_lin, plane_lin, blockIdx.x);
}

__global__ __launch_bounds__(256, 2) void kb_hblocks(BatchArgs a) {
  PPS_BATCH_PROLOGUE(BF_ACTIVE | BF_RELIN)
  if ((int)blockIdx.x * 4 >= d.n_segs) return;
  body_hblocks(d, blockIdx.x);
}

constexpr int kHblocksT = 4;      // segments per wave of the throughput form
__global__ __launch_bounds__(256) void kb_hblocks_t(BatchArgs a) {
  PPS_BATCH_PROLOGUE(BF_ACTIVE | BF_RELIN)
  if ((int)blockIdx.x * 4 * kHblocksT >= d.n_nd_segs) return;
  body_hblocks_t<kHblocksT>(d, blockIdx.x);
}

__global__ __launch_bounds__(64) void kb_hreduce(BatchArgs a) {
  PPS_BATCH_PROLOGUE(BF_ACTIVE | BF_RELIN)
  if ((int)blockIdx.x >= d.n_mseg) return;
  body_hreduce(d, blockIdx.x);
}

template <bool REG_ONLY>
__global__ __launch_bounds__(512) void kb_band_factor(BatchArgs a, int stage, int lds_doubles_per_wave) {
  extern __shared__ double lds[];
  PPS_BATCH_PROLOGUE(BF_ACTIVE)
  const BatchStage sg = a.stage_tab[(size_t)stage * a.n_total + a.b0 + b];
  if ((int)blockIdx.x >= sg.grp_count) return;
  if (blockIdx.z) {
    const BatchAlt al = load_alt(a.alt + a.b0 + b);
    DevGraph d2 = d;
    d2.L = al.L; d2.U = al.U; d2.delta = al.delta; d2.result_dev = al.result_dev;
    body_band_factor<REG_ONLY>(d2, sg.grp_begin + blockIdx.x, a.lambda2[b], lds_doubles_per_wave, 0, lds);
    return;
  }
  body_band_factor<REG_ONLY>(d, sg.grp_begin + blockIdx.x, a.lambda[b], lds_doubles_per_wave, 0, lds);
}

__global__ __launch_bounds__(512) void kb_band_solve(BatchArgs a, int stage, int lds_doubles_per_wave) {
  extern __shared__ double lds[];
  PPS_BATCH_PROLOGUE(BF_ACTIVE)
  const BatchStage sg = a.stage_tab[(size_t)stage * a.n_total + a.b0 + b];
  if ((int)blockIdx.x >= sg.grp_count) return;
  if (blockIdx.z) {
    const BatchAlt al = load_alt(a.alt + a.b0 + b);
    DevGraph d2 = d;
    d2.L = al.L; d2.U = al.U; d2.delta = al.delta;
    body_band_solve(d2, sg.grp_begin + blockIdx.x, lds_doubles_per_wave, lds);
    return;
  }
  body_band_solve(d, sg.grp_begin + blockIdx.x, lds_doubles_per_wave, lds);
}

__global__ __launch_bounds__(256) void kb_retract_trial(BatchArgs a) {
  __shared__ double red[4];
  PPS_BATCH_PROLOGUE(BF_ACTIVE)
  if ((int)blockIdx.x * 256 >= d.n_pose + d.n_plane) return;
  body_retract<true>(d, pose_lin, pose_est, plane_lin, plane_est, blockIdx.x, red);
}

__global__ __launch_bounds__(kChiBlock) void kb_chi2(BatchArgs a, int slot) {
  PPS_BATCH_PROLOGUE(BF_ACTIVE)
  const int nb_obs = dcdiv(d.n_obs, kChiBlock), nb_odo = dcdiv(d.n_odo, kChiBlock), nb_pp = dcdiv(d.n_pp, kChiBlock),
            nb_lp = dcdiv(d.n_lp, kChiBlock);
  const int nb = nb_obs + nb_odo + nb_pp + nb_lp;
  if ((int)blockIdx.x >= nb) return;
  body_chi2(d, pose_lin, plane_lin, nb_obs, nb_odo, nb_pp, dcdiv(d.n_pose + d.n_plane, 256),
            a.results + (size_t)(a.alt ? 12 : 8) * (size_t)(a.b0 + b) + 4 * slot, a.seq, blockIdx.x, nb);
}

hipError_t launch_batch_begin(const BatchArgs& a, const BatchGeom& g, hipStream_t st) {
  PPS_LAUNCH(kb_begin, dim3(std::max(1, std::min(8, g.retract)), a.n), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_batch_linearize(const BatchArgs& a, const BatchGeom& g, int mode, hipStream_t st) {
  if (g.repop_blocks > 0) PPS_LAUNCH(kb_linearize_repop, dim3(g.repop_blocks, a.n), dim3(64), 0, st, a);
  const size_t lds0 = (size_t)(kLinBlock / 64) * 64 * 31 * sizeof(double), lds1 = (size_t)(kLinBlock / 64) * 64 * 79 * sizeof(double);
  if (mode == 0) {
    if (g.lin_blocks > 0) PPS_LAUNCH(kb_linearize_lanes, dim3(g.lin_blocks, a.n), dim3(kLanesPerBlock), 0, st, a);
  } else if (mode == 2) {          // numeric, one thread per factor
    if (g.lin_obs_blocks > 0) {
      if (g.k1_direct) PPS_LAUNCH((kb_linearize<0, 0, true>), dim3(g.lin_obs_blocks, a.n), dim3(kLinBlock), lds0, st, a);
      else PPS_LAUNCH((kb_linearize<0, 0, false>), dim3(g.lin_obs_blocks, a.n), dim3(kLinBlock), lds0, st, a);
    }
    if (g.lin_rest_blocks > 0) PPS_LAUNCH((kb_linearize<0, 1, false>), dim3(g.lin_rest_blocks, a.n), dim3(kLinBlock), lds1, st, a);
  } else {
    if (g.lin_obs_blocks > 0) {
      if (g.k1_direct) PPS_LAUNCH((kb_linearize<1, 0, true>), dim3(g.lin_obs_blocks, a.n), dim3(kLinBlock), lds0, st, a);
      else PPS_LAUNCH((kb_linearize<1, 0, false>), dim3(g.lin_obs_blocks, a.n), dim3(kLinBlock), lds0, st, a);
    }
    if (g.lin_rest_blocks > 0) PPS_LAUNCH((kb_linearize<1, 1, false>), dim3(g.lin_rest_blocks, a.n), dim3(kLinBlock), lds1, st, a);
  }
  return hipGetLastError();
}

hipError_t launch_batch_hblocks(const BatchArgs& a, const BatchGeom& g, hipStream_t st) {
  if (g.hblocks > 0) {
    if (g.k1_direct) PPS_LAUNCH(kb_hblocks_t, dim3(std::max(1, g.hblocks_nd), a.n), dim3(256), 0, st, a);   // many graphs: throughput form, direct blocks done by K1
    else PPS_LAUNCH(kb_hblocks, dim3(g.hblocks, a.n), dim3(256), 0, st, a);
  }
  if (g.hreduce > 0) PPS_LAUNCH(kb_hreduce, dim3(g.hreduce, a.n), dim3(64), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_batch_chi2(const BatchArgs& a, const BatchGeom& g, int slot, hipStream_t st) {
  if (g.chi2 <= 0) return hipErrorInvalidValue;
  PPS_LAUNCH(kb_chi2, dim3(g.chi2, a.n), dim3(kChiBlock), 0, st, a, slot);
  return hipGetLastError();
}

static std::atomic<bool> g_batch_attr_set[64];

hipError_t launch_batch_solve(const BatchArgs& a, const BatchGeom& g, hipStream_t st, hipEvent_t after_factor) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!g_batch_attr_set[dev & 63]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&kb_band_factor<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&kb_band_factor<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&kb_band_solve), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e != hipSuccess) return e;
    g_batch_attr_set[dev & 63] = true;
  }
  for (int stg = 0; stg < g.n_stages; stg++) {
    if (g.stage_groups[stg] <= 0) continue;
    const int per_wave = g.stage_per_wave_factor[stg], nw = g.stage_nw_factor[stg];
    const size_t bytes = (size_t)per_wave * nw * sizeof(double);
    if (g.stage_reg_only[stg])
      PPS_LAUNCH(kb_band_factor<true>, dim3(g.stage_groups[stg], a.n, a.alt ? 2 : 1), dim3(64 * nw), bytes, st, a, stg, per_wave);
    else
      PPS_LAUNCH(kb_band_factor<false>, dim3(g.stage_groups[stg], a.n, a.alt ? 2 : 1), dim3(64 * nw), bytes, st, a, stg, per_wave);
  }
  if (after_factor) (void)hipEventRecord(after_factor, st);
  for (int stg = g.n_stages - 1; stg >= 0; stg--) {
    if (g.stage_groups[stg] <= 0) continue;
    const int per_wave = g.stage_per_wave_solve[stg], nw = g.stage_nw_solve[stg];
    const size_t bytes = ((size_t)per_wave * nw + (size_t)g.stage_grp_fronts[stg] * kBandMaxRows) * sizeof(double);
    PPS_LAUNCH(kb_band_solve, dim3(g.stage_groups[stg], a.n, a.alt ? 2 : 1), dim3(64 * nw), bytes, st, a, stg, per_wave);
  }
  return hipGetLastError();
}

hipError_t launch_batch_trial(const BatchArgs& a, const BatchGeom& g, hipStream_t st) {
  if (g.retract > 0) PPS_LAUNCH(kb_retract_trial, dim3(g.retract, a.n), dim3(256), 0, st, a);
  return launch_batch_chi2(a, g, 1, st);
}

// ---- dual-lambda batch (BatchAlt): both trials of a graph in one launch, grid z = 0 / 1 ----
__global__ __launch_bounds__(64) void kb_begin_dual(BatchArgs a) {
  PPS_BATCH_PROLOGUE(BF_ACTIVE)
  const BatchAlt al = load_alt(a.alt + a.b0 + b);
  if (threadIdx.x < 4) { d.result_dev[threadIdx.x] = 0.0; al.result_dev[threadIdx.x] = 0.0; }
}

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

__global__ __launch_bounds__(256) void kb_retract_dual(BatchArgs a) {
  __shared__ double red[4];
  PPS_BATCH_PROLOGUE(BF_ACTIVE)
  if ((int)blockIdx.x * 256 >= d.n_pose + d.n_plane) return;
  const BatchAlt al = load_alt(a.alt + a.b0 + b);
  const int xs = a.xsel[b], z = blockIdx.z;
  const int ts = (xs + 1 + z) % 3;
  DevGraph d2 = d;
  if (z) { d2.delta = al.delta; d2.dn_partials = al.dn_partials; }
  body_retract_to(d2, pose_lin, plane_lin, sel3(al.pose, ts), sel3(al.plane, ts), blockIdx.x, red);
}

__global__ __launch_bounds__(kChiBlock) void kb_chi2_dual(BatchArgs a) {
  PPS_BATCH_PROLOGUE(BF_ACTIVE)
  const int nb_obs = dcdiv(d.n_obs, kChiBlock), nb_odo = dcdiv(d.n_odo, kChiBlock), nb_pp = dcdiv(d.n_pp, kChiBlock),
            nb_lp = dcdiv(d.n_lp, kChiBlock);
  const int nb = nb_obs + nb_odo + nb_pp + nb_lp;
  if ((int)blockIdx.x >= nb) return;
  const BatchAlt al = load_alt(a.alt + a.b0 + b);
  const int xs = a.xsel[b], z = blockIdx.z;
  const int ts = (xs + 1 + z) % 3;
  DevGraph d2 = d;
  if (z) { d2.chi2_partials = al.chi2_partials; d2.dn_partials = al.dn_partials; d2.ticket = al.ticket; d2.result_dev = al.result_dev; }
  body_chi2(d2, sel3(al.pose, ts), sel3(al.plane, ts), nb_obs, nb_odo, nb_pp, dcdiv(d.n_pose + d.n_plane, 256),
            a.results + 12 * (size_t)(a.b0 + b) + 4 * (1 + z), a.seq, blockIdx.x, nb);
}

hipError_t launch_batch_begin_dual(const BatchArgs& a, const BatchGeom& g, hipStream_t st) {
  (void)g;
  PPS_LAUNCH(kb_begin_dual, dim3(1, a.n), dim3(64), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_batch_trial_dual(const BatchArgs& a, const BatchGeom& g, hipStream_t st) {
  if (g.chi2 <= 0) return hipErrorInvalidValue;
  if (g.retract > 0) PPS_LAUNCH(kb_retract_dual, dim3(g.retract, a.n, 2), dim3(256), 0, st, a);
  PPS_LAUNCH(kb_chi2_dual, dim3(g.chi2, a.n, 2), dim3(kChiBlock), 0, st, a);
  return hipGetLastError();
}

// patch upload of a re-uploaded topology (pps_api.cpp: flush_uploads): piece i of the patch buffer -> its place in the arena
__global__ __launch_bounds__(256) void k_scatter_patches(const char* __restrict__ patch, char* __restrict__ arena) {
  const long long* tab = reinterpret_cast<const long long*>(patch) + 4 * (size_t)blockIdx.x;
  const long long dst = tab[0], src = tab[1], len = tab[2];
  const int4* s4 = reinterpret_cast<const int4*>(patch + src);
  int4* d4 = reinterpret_cast<int4*>(arena + dst);
  for (long long i = (long long)blockIdx.y * 256 + threadIdx.x; i < len / 16; i += (long long)gridDim.y * 256) d4[i] = s4[i];
}

hipError_t launch_scatter_patches(const char* patch, int n_patches, char* arena, hipStream_t st) {
  if (n_patches <= 0) return hipSuccess;
  PPS_LAUNCH(k_scatter_patches, dim3(n_patches, 8), dim3(256), 0, st, patch, arena);
  return hipGetLastError();
}

hipError_t launch_clear_status(const DevGraph& d, hipStream_t st) {
  return hipMemsetAsync(d.result_dev, 0, 4 * sizeof(double), st);
}

}  // namespace pps
