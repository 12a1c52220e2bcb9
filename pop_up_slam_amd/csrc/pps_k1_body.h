// pps_k1_body.h -- K1, the per-edge residual + Jacobian sweep (device bodies).
//   reference: Slam::jacobian_partial (isamlib/Slam.cpp:395-432) + numericalDiff (numericalDiff.cpp:41-87)
// Three forms share these bodies: one thread per factor (analytic mode, large batches), 32 lanes per factor with one central-
// difference evaluation per lane (numeric mode, the reference's arithmetic), and the re-popping Factor2 edges.
#pragma once
#include "pps_geom.h"
#include "pps_lin.h"
#include "pps_kcommon.h"

namespace pps {

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

// NS doubles per lane (a slice of a wider record) staged through the wave's LDS rows (stride 31: the buffer of the Jacobian record)
// and written to rows of STRIDE doubles in global memory: runs of NS contiguous doubles per record.
template <int NS, int STRIDE>
__device__ __forceinline__ void store_slices_staged(const double (&v)[NS], double* __restrict__ gbase, int n_valid, double* __restrict__ lds_wave) {
  static_assert(NS <= 30, "the wave buffer holds 31 doubles per lane");
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < NS; k++) lds_wave[lane * 31 + k] = v[k];
  __builtin_amdgcn_wave_barrier();
  const int total = n_valid * NS;
#pragma unroll
  for (int k = 0; k < NS; k++) {
    const int idx = lane + 64 * k;
    if (idx < total) { const int r0 = idx / NS, c0 = idx - r0 * NS; gbase[(size_t)r0 * STRIDE + c0] = lds_wave[r0 * 31 + c0]; }
  }
  __builtin_amdgcn_wave_barrier();
}

// the product record of a plane observation from its Jacobian record out = [J_p 3x6 | J_l 3x3 | r 3] (see body_linearize_lanes)
__device__ __forceinline__ void obs_products(const double (&out)[30], double (&lo)[27], double (&hi)[27]) {
  double pr[54];
#pragma unroll
  for (int e = 0; e < 54; e++) {
    const bool pose = e < 42;
    const int q = pose ? e : e - 42, n = pose ? 6 : 3, base = pose ? 0 : 18;
    const bool grad = q >= n * n;
    const int ca = grad ? q - n * n : q / n, cb = grad ? 0 : q - n * (q / n);
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 3; k++) acc = PPS_MAC(acc, out[base + k * n + ca], grad ? out[27 + k] : out[base + k * n + cb]);
    pr[e] = grad ? -acc : acc;
  }
#pragma unroll
  for (int e = 0; e < 27; e++) { lo[e] = pr[e]; hi[e] = pr[27 + e]; }
}

// PART 0: plane-observation edges (16 KB of staging LDS per wave); PART 1: odometry edges and priors
// (40 KB per wave in analytic mode) -- separate launches so that the big edge class keeps its occupancy.
// DIRECT (every launch of the solver; off in the sweep benchmark): a plane observation that is the only contribution of its (pose, plane) block writes that H
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
    if (DIRECT) {
      // Round 5: the 18 entries of a direct block leave through the LDS like the Jacobian records.  Written straight from the lanes, every
      // store instruction was 64 pieces of 8 bytes in 64 different lines -- the PMC counters showed 1.2 KB written per observation for 528
      // bytes of records and blocks (profiles/r5_pmc_multi128_hbm.txt: 84 GB against 22 GB read).  Now lane l stages its block in row l of
      // a 64 x 19 buffer and the wave writes the 1 152 entries in order: a store instruction covers three to four whole blocks.
      const int lane = threadIdx.x & 63;
      int hoff = -1, el0 = -1, rows6 = 0;
      if (b * kLinBlock + (int)threadIdx.x < d.n_obs_fixed) { hoff = d.obs_dir[3 * (size_t)i]; el0 = d.obs_dir[3 * (size_t)i + 1]; rows6 = d.obs_dir[3 * (size_t)i + 2]; }
      double* __restrict__ S = lds_wave;                                  // 64 x 19 doubles, then 2 x 64 ints (H offset, Hf offset per lane)
      int* __restrict__ SI = reinterpret_cast<int*>(lds_wave + 64 * 19);
      if (hoff >= 0) {
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
          S[lane * 19 + e] = acc;
        }
      }
      SI[lane] = hoff; SI[64 + lane] = el0;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int u = 0; u < 18; u++) {
        const int it = lane + 64 * u, o = it / 18, e = it - 18 * o;
        const int ho = SI[o], eo = SI[64 + o];
        const double v = S[o * 19 + e];
        if (ho >= 0) { d.H[ho + e] = v; if (eo >= 0) d.Hf[eo + e] = v; }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (i0 < d.n_obs_fixed) store_records_coalesced<30>(out, d.J + d.joff_obs + (size_t)i0 * 30, min(64, d.n_obs_fixed - i0), lds_wave);
    if (DIRECT && d.P && i0 < d.n_obs_fixed) {                  // the product record K2 sums (two staged halves of 27 doubles); P is null where K2 multiplies the Jacobians itself
      double lo[27], hi[27];
      obs_products(out, lo, hi);
      double* __restrict__ Pr = d.P + d.poff_obs + (size_t)i0 * 54;
      store_slices_staged<27, 54>(lo, Pr, min(64, d.n_obs_fixed - i0), lds_wave);
      store_slices_staged<27, 54>(hi, Pr + 27, min(64, d.n_obs_fixed - i0), lds_wave);
    }
    return;
  }
  if (PART == 0) return;          // (a PART 0 launch has nb_obs blocks: without this the other factor types are compiled in, 68 KB of dead code)
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

// ------------------------------------------------------------------------------------------
// K1, lane-parallel central differences (the reference's numericalDiff, one evaluation per lane).
// Plane observations -- 5 of every 6 factors -- take 19 lanes each (the nominal residual + 2 x 9 perturbed ones): 3 edges per
// wavefront, 57 of 64 lanes at work (two edges on 32 lanes each left 38).  Lane 0 of a group evaluates the nominal residual,
// lanes 2q+1 / 2q+2 the residual at x (+) / (-) eps e_q; the pair differences through one shuffle and the odd lane stores
// column q.  The other factor types keep 32 lanes per factor: lane 2q at x (+) eps e_q, lane 2q+1 at x (-) eps e_q, lane
// 2*ncols the nominal residual.  All lanes of a group read the same edge record (a broadcast load), state is gathered by index
// from the SoA arrays.
// ------------------------------------------------------------------------------------------
// Pose3d::exmap / Plane3d::exmap_3dof by the step sgn * eps * e_q (q out of range: the zero step, which is the exact identity for a
// pose).  The step quaternions come from DevGraph::step_ac -- (a, 0, 0, c) = rot_exp / plane_exp of (eps, 0, 0), evaluated once per
// device by those very functions -- instead of a square root, a sine / cosine pair and a division in every lane: S * (sgn eps) =
// sgn (S eps) exactly, theta = eps whichever component carries the step, and the zero step is (0, 0, 0, 1) in both maps.  The
// quaternion product (and the plane's normalisation) are the ones of pose_exmap / plane_exmap: the same bits.
__device__ __forceinline__ void perturb6(const double p[7], int q, double sgn, const double ac[4], double o[7]) {
  PPS_FP_EXACT
  const bool rot = q >= 3 && q < 6;
  const double a = sgn * ac[0];
  const double dq[4] = {q == 3 ? a : 0.0, q == 4 ? a : 0.0, q == 5 ? a : 0.0, rot ? ac[1] : 1.0};
  double qq[4];
  quat_mul(p + 3, dq, qq);
#pragma unroll
  for (int k = 0; k < 3; k++) o[k] = p[k] + ((k == q) ? sgn * kNumDiffEps : 0.0);
  o[3] = qq[0]; o[4] = qq[1]; o[5] = qq[2]; o[6] = qq[3];
}
__device__ __forceinline__ void perturb3(const double p[4], int q, double sgn, const double ac[4], double o[4]) {
  PPS_FP_EXACT
  const bool on = q >= 0 && q < 3;
  const double a = sgn * ac[2];
  const double dq[4] = {q == 0 ? a : 0.0, q == 1 ? a : 0.0, q == 2 ? a : 0.0, on ? ac[3] : 1.0};
  quat_mul(dq, p, o);
  normalize4_r(o);
}

constexpr int kLaneGroup = 32;
constexpr int kLanesPerBlock = 256;
constexpr int kFactorsPerBlock = kLanesPerBlock / kLaneGroup;
constexpr int kObsLanes = 19;                                         // evaluations of one plane observation
constexpr int kObsPerWave = 64 / kObsLanes;                           // 3
constexpr int kObsPerBlock = (kLanesPerBlock / 64) * kObsPerWave;     // 12

// apply (round 6, the fused trial + linearisation launch; wave-uniform): the point of linearisation is pose / plane (+) d.delta, evaluated per
// lane on the spot by the functions the retraction kernel uses (pose_exmap / plane_exmap: compiled without contraction, the same bits as the
// copy that kernel stores) -- the sweep then needs no launch boundary behind the retraction.  A RUN-TIME flag, not a template parameter: every
// kernel that sweeps in the lane form inlines the same source, so that the compiler contracts the same multiply-adds in all of them and a
// factor's Jacobian has the same bits whichever launch produced it (a second instantiation was measured to differ in the 13th digit).
__device__ __forceinline__ void state_pose(const DevGraph& d, const double* __restrict__ pose, int idx, bool apply, double o[7]) {
  load_pose(pose, d.pose_ld, idx, o);
  if (apply) {
    double p[7], dl[6];
#pragma unroll
    for (int k = 0; k < 7; k++) p[k] = o[k];
    const int off = d.pose_voff[idx];
#pragma unroll
    for (int k = 0; k < 6; k++) dl[k] = d.delta[off + k];
    pose_exmap(p, dl, o);
  }
}
__device__ __forceinline__ void state_plane(const DevGraph& d, const double* __restrict__ plane, int idx, bool apply, double o[4]) {
  load_plane(plane, d.plane_ld, idx, o);
  if (apply) {
    double p[4], dl[3];
#pragma unroll
    for (int k = 0; k < 4; k++) p[k] = o[k];
    const int off = d.plane_voff[idx];
#pragma unroll
    for (int k = 0; k < 3; k++) dl[k] = d.delta[off + k];
    plane_exmap(p, dl, o);
  }
}
__device__ __forceinline__ void body_linearize_lanes(const DevGraph& d, const double* __restrict__ pose,
                                                     const double* __restrict__ plane, int nb_obs, int nb_odo, int nb_pp, int bx, bool apply = false) {
  const int grp = threadIdx.x / kLaneGroup, gl = threadIdx.x % kLaneGroup;
  const int q = gl >> 1;                       // perturbed column
  const double sgn = (gl & 1) ? -1.0 : 1.0;
  const double inv2e = 1.0 / (kNumDiffEps + kNumDiffEps);
  int b = bx;
  if (b < nb_obs) {
    const int lane = threadIdx.x & 63, g3 = lane / kObsLanes, l3 = lane - g3 * kObsLanes;
    const int i = b * kObsPerBlock + (threadIdx.x >> 6) * kObsPerWave + g3;
    if (g3 >= kObsPerWave || i >= d.n_obs_fixed) return;
    const int q3 = l3 > 0 ? (l3 - 1) >> 1 : 9;                    // perturbed column (9: none, the nominal residual)
    const double s3 = (l3 & 1) ? 1.0 : -1.0;                      // odd lane: + eps, the even lane after it: - eps
    double pz[7], pl[4], ms[4], w[6], e[3], y[3];
    state_pose(d, pose, d.obs_pose[i], apply, pz);
    state_plane(d, plane, d.obs_plane[i], apply, pl);
    load_soa<4>(d.obs_meas, d.obs_ld, i, ms);
    load_soa<6>(d.obs_w, d.obs_ld, i, w);
    {
      // every lane takes the same path: a perturbation that does not apply is the zero step, which is the
      // exact identity for a pose; the plane keeps its stored value unless it is the perturbed node
      double pp[7], lp[4];
      perturb6(pz, q3, s3, d.step_ac, pp);                       // q3 >= 6: zero delta -> pp == pz bit for bit
      perturb3(pl, q3 - 6, s3, d.step_ac, lp);
      const bool pert_plane = q3 >= 6 && q3 < 9;
#pragma unroll
      for (int k = 0; k < 4; k++) lp[k] = pert_plane ? lp[k] : pl[k];
      res_plane_obs(pp, lp, ms, e);
    }
    whiten<3>(w, e, y);
    double* __restrict__ out = d.J + d.joff_obs + (size_t)i * 30;
    double jc[3];                                                 // odd lanes: column q3 of [Jp | Jl], rows 0 .. 2
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const double other = __shfl_down(y[r], 1, 64);             // the (-) evaluation of the same column sits in the next lane
      jc[r] = (y[r] - other) * inv2e;
      if (l3 & 1) {
        if (q3 < 6) out[r * 6 + q3] = jc[r];
        else out[18 + r * 3 + (q3 - 6)] = jc[r];
      } else if (l3 == 0) out[27 + r] = y[r];
    }
    // The observation's PRODUCT record (pps_symbolic.h: kPSize): what it adds to the diagonal H blocks of its pose and its plane
    // -- J_p' J_p (36, row-major), -J_p' r (6), J_l' J_l (9), -J_l' r (3) -- so that K2 only sums such records, one coalesced load per
    // contribution, instead of gathering Jacobian columns.  The group's nine columns and the residual meet in LDS ([row][column],
    // column 9 = r); every lane multiplies three of the 54 entries, k ascending as explicit multiply-adds.
    if (d.P) {                                                  // (null in the sweep benchmark: Jacobians only)
      __shared__ double prod_lds[kObsPerBlock * 32];
      double* __restrict__ S = prod_lds + ((threadIdx.x >> 6) * kObsPerWave + g3) * 32;
      if (l3 & 1) { S[q3] = jc[0]; S[10 + q3] = jc[1]; S[20 + q3] = jc[2]; }
      else if (l3 == 0) { S[9] = y[0]; S[19] = y[1]; S[29] = y[2]; }
      __builtin_amdgcn_wave_barrier();
      double* __restrict__ Pr = d.P + d.poff_obs + (size_t)i * 54;
#pragma unroll
      for (int t = 0; t < 3; t++) {
        const int e = l3 + kObsLanes * t;                       // 0 .. 56
        int ca, cb;
        if (e < 36) { ca = e / 6; cb = e - 6 * (e / 6); }
        else if (e < 42) { ca = e - 36; cb = 9; }
        else if (e < 51) { ca = 6 + (e - 42) / 3; cb = 6 + (e - 42) - 3 * ((e - 42) / 3); }
        else { ca = 6 + (e < 54 ? e - 51 : 0); cb = 9; }
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) acc = PPS_MAC(acc, S[10 * k + ca], S[10 * k + cb]);
        if (e < 54) Pr[e] = cb == 9 ? -acc : acc;               // b = -J' r (isam/Jacobian.h:98)
      }
    }
    // The (pose, plane) block of H that this observation alone contributes to (Analysis::obs_dir) is the product of its own
    // two Jacobian blocks: lane e < 18 of the group gathers the two columns it needs from the odd lanes and writes entry e --
    // summed in the order the H-block kernel uses -- into H and into the front-ordered copy; K2 only visits the other segments.
    if (d.obs_dir) {                                            // (null in the sweep benchmark: Jacobians only)
      const int hoff = d.obs_dir[3 * (size_t)i], el0 = d.obs_dir[3 * (size_t)i + 1], rows6 = d.obs_dir[3 * (size_t)i + 2];
      const int e = l3 < 18 ? l3 : 17;
      const int ri = rows6 ? e / 3 : e / 6, cj = rows6 ? e - 3 * (e / 3) : e - 6 * (e / 6);
      const int col_a = rows6 ? ri : 6 + ri, col_b = rows6 ? 6 + cj : cj;       // columns of [Jp | Jl]: the row node's block, the column node's
      const int src_a = g3 * kObsLanes + 2 * col_a + 1, src_b = g3 * kObsLanes + 2 * col_b + 1;
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const double av = __shfl(jc[k], src_a, 64), bv = __shfl(jc[k], src_b, 64);
        acc = PPS_MAC(acc, av, bv);
      }
      if (hoff >= 0 && l3 < 18) {
        d.H[hoff + e] = acc;
        if (el0 >= 0) d.Hf[el0 + e] = acc;
      }
    }
    return;
  }
  b -= nb_obs;
  if (b < nb_odo) {
    const int i = b * kFactorsPerBlock + grp;
    if (i >= d.n_odo) return;
    double p1[7], p2[7], ms[6], w[21], e[6], y[6];
    state_pose(d, pose, d.odo_a[i], apply, p1);
    state_pose(d, pose, d.odo_b[i], apply, p2);
    load_soa<6>(d.odo_meas, d.odo_ld, i, ms);
    load_soa<21>(d.odo_w, d.odo_ld, i, w);
    {
      double pa[7], pb[7];
      perturb6(p1, q, sgn, d.step_ac, pa);                       // out-of-range q: zero step == identity
      perturb6(p2, q - 6, sgn, d.step_ac, pb);
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
    state_pose(d, pose, d.pp_pose[i], apply, pz);
    load_soa<6>(d.pp_meas, d.pp_ld, i, ms);
    load_soa<21>(d.pp_w, d.pp_ld, i, w);
    { double pp[7]; perturb6(pz, q, sgn, d.step_ac, pp); res_pose_prior(pp, ms, e); }
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
    state_plane(d, plane, d.lp_plane[i], apply, pl);
    load_soa<4>(d.lp_meas, d.lp_ld, i, ms);
    load_soa<6>(d.lp_w, d.lp_ld, i, w);
    {
      double lp[4];
      perturb3(pl, q, sgn, d.step_ac, lp);
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

// below this many factors the lane-parallel form wins (latency); above it the thread-per-factor
// form has the higher throughput (no idle lanes)
constexpr int kLaneParallelMaxFactors = 200000;

// Pose3d_Plane3d_Factor2 edges (slots [n_obs_fixed, n_obs)): central differences in both Jacobian modes -- the
// measurement moves with the pose perturbation (the reference differentiates it numerically too).
__device__ __forceinline__ void body_linearize_repop(const DevGraph& d, const double* __restrict__ pose,
                                                     const double* __restrict__ plane, int bx, double* __restrict__ lds_wave) {
  const int n2 = d.n_obs - d.n_obs_fixed;
  const int k0 = bx * 64;                                             // first edge of this wave (one wave per workgroup)
  if (k0 >= n2) return;
  const int k = min(k0 + (int)threadIdx.x, n2 - 1);                   // clamped: every lane stays active for the staged store
  const int i = d.n_obs_fixed + k;
  double pz[7], pl[4], ray[6], w[6], e[3], r[3];
  double* __restrict__ out = lds_wave + (threadIdx.x & 63) * 31;      // the record is built in LDS (the loops below stay rolled)
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
    for (int q = 0; q < 3; q++) out[q * 6 + j] = (yp[q] - ym[q]) * inv2e;
  }
  for (int j = 0; j < 3; j++) {
    double dl[3] = {0, 0, 0}, pp[4], yp[3], ym[3];
    dl[j] = kNumDiffEps;
    plane_exmap(pl, dl, pp); res_plane_obs2(pz, pp, ray, e); whiten<3>(w, e, yp);
    dl[j] = -kNumDiffEps;
    plane_exmap(pl, dl, pp); res_plane_obs2(pz, pp, ray, e); whiten<3>(w, e, ym);
    for (int q = 0; q < 3; q++) out[18 + q * 3 + j] = (yp[q] - ym[q]) * inv2e;
  }
  for (int q = 0; q < 3; q++) out[27 + q] = r[q];
  // 64 records of 240 bytes leave as one contiguous stream (a record per lane would be 30 scattered 8-byte stores per lane)
  flush_records_coalesced<30>(d.J + d.joff_obs + (size_t)(d.n_obs_fixed + k0) * 30, min(64, n2 - k0), lds_wave);
}

}  // namespace pps
