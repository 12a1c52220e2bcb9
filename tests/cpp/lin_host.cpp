// lin_host.cpp -- csrc/pps_lin.h on the host (test infrastructure, tests/test_host_lin.py).
// The numeric-mode Jacobians of K1's thread form evaluate once what a perturbation leaves bit-identical (R for translation and
// plane steps, the Euler chain for translation steps of an odometry edge).  This file holds the PLAIN form next to it -- 2 n + 1
// complete residual evaluations through exmap, numericalDiff.cpp:41-87 statement by statement -- and reports the largest
// difference between the two records.  Built with -ffp-contract=off both are the same arithmetic, so the difference must be 0.
#include <cmath>
#include <cstring>

#include "pps_lin.h"

using namespace pps;

template <int M, int NA, int NB, class FA, class FB>
static void plain(const double* a, const double* b, int da, int db, FA exa, FB exb, const double* w,
                  void (*res)(const double*, const double*, const double*, double*), const double* ms, double* out) {
  (void)NA; (void)NB;
  const double inv2e = 1.0 / (kNumDiffEps + kNumDiffEps);
  double e[M], yp[M], ym[M];
  for (int j = 0; j < da; j++) {
    double d[6] = {0, 0, 0, 0, 0, 0}, pa[7];
    d[j] = kNumDiffEps; exa(a, d, pa); res(pa, b, ms, e); whiten<M>(w, e, yp);
    d[j] = -kNumDiffEps; exa(a, d, pa); res(pa, b, ms, e); whiten<M>(w, e, ym);
    for (int i = 0; i < M; i++) out[i * da + j] = (yp[i] - ym[i]) * inv2e;
  }
  for (int j = 0; j < db; j++) {
    double d[6] = {0, 0, 0, 0, 0, 0}, pb[7];
    d[j] = kNumDiffEps; exb(b, d, pb); res(a, pb, ms, e); whiten<M>(w, e, yp);
    d[j] = -kNumDiffEps; exb(b, d, pb); res(a, pb, ms, e); whiten<M>(w, e, ym);
    for (int i = 0; i < M; i++) out[M * da + i * db + j] = (yp[i] - ym[i]) * inv2e;
  }
  res(a, b, ms, e);
  whiten<M>(w, e, out + M * (da + db));
}

static void r_obs(const double* p, const double* l, const double* ms, double* e) { res_plane_obs(p, l, ms, e); }
static void r_odo(const double* p1, const double* p2, const double* ms, double* e) { res_odometry(p1, p2, ms, e); }

// kind 0: plane observation (a = pose 7, b = plane 4, ms 4, w 6, record 30); 1: odometry (a, b = poses, ms 6, w 21, record 78);
// 2: pose prior (a = pose, ms 6, w 21, record 42).  Returns max |structured - plain| over the record; both records are copied out.
extern "C" double lin_host_compare(int kind, const double* a, const double* b, const double* ms, const double* w, double* structured, double* plain_out) {
  int n = 0;
  if (kind == 0) {
    n = 30;
    lin_plane_obs<0>(a, b, ms, w, structured);
    plain<3, 7, 4>(a, b, 6, 3, [](const double* x, const double* d, double* o) { pose_exmap(x, d, o); },
                   [](const double* x, const double* d, double* o) { plane_exmap(x, d, o); }, w, r_obs, ms, plain_out);
  } else if (kind == 1) {
    n = 78;
    lin_odometry<0>(a, b, ms, w, structured);
    plain<6, 7, 7>(a, b, 6, 6, [](const double* x, const double* d, double* o) { pose_exmap(x, d, o); },
                   [](const double* x, const double* d, double* o) { pose_exmap(x, d, o); }, w, r_odo, ms, plain_out);
  } else {
    n = 42;
    lin_pose_prior<0>(a, ms, w, structured);
    const double inv2e = 1.0 / (kNumDiffEps + kNumDiffEps);
    double e[6], yp[6], ym[6];
    for (int j = 0; j < 6; j++) {
      double d[6] = {0, 0, 0, 0, 0, 0}, pa[7];
      d[j] = kNumDiffEps; pose_exmap(a, d, pa); res_pose_prior(pa, ms, e); whiten<6>(w, e, yp);
      d[j] = -kNumDiffEps; pose_exmap(a, d, pa); res_pose_prior(pa, ms, e); whiten<6>(w, e, ym);
      for (int i = 0; i < 6; i++) plain_out[i * 6 + j] = (yp[i] - ym[i]) * inv2e;
    }
    res_pose_prior(a, ms, e);
    whiten<6>(w, e, plain_out + 36);
  }
  double worst = 0.0;
  for (int i = 0; i < n; i++) worst = std::fmax(worst, std::fabs(structured[i] - plain_out[i]));
  return worst;
}
