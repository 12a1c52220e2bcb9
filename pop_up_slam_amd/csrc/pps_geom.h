// pps_geom.h -- SE3 / plane algebra shared by the HIP kernels and the host side.
//
// fp64, quaternions stored (x,y,z,w).  Each function names the reference code whose
// arithmetic it reproduces (paths relative to /root/reference/pop_planar_slam;
// "isam/" = Thirdparty/isam/include/isam).  Written for registers: fixed-size arrays,
// fully unrollable loops, no memory traffic.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PPS_HD __host__ __device__ __forceinline__
#else
#define PPS_HD inline
#endif

// The retraction (exmap) of a node must give the same bits wherever it is inlined -- the fused trial kernel evaluates chi2 at
// x (+) delta computed on the fly while another block writes the same x (+) delta to memory for the next linearisation -- so
// these few functions are compiled without multiply-add contraction whatever the translation unit's -ffp-contract says.
#if defined(__clang__)
#define PPS_FP_EXACT _Pragma("clang fp contract(off)")
#else
#define PPS_FP_EXACT
#endif

namespace pps {

constexpr double kPi = 3.14159265358979323846;
constexpr double kTwoPi = 2.0 * kPi;
constexpr double kNumDiffEps = 0.0001;  // isamlib/numericalDiff.cpp:34

// isam/util.h:101-108: fmod(t + pi, 2 pi) - pi for t >= 0, fmod(t - pi, -2 pi) + pi otherwise.
// fmod is exact, and for an argument within one period of the range it is the argument itself or one exact subtraction
// (Sterbenz): an angle difference (|t| <= 2 pi) never reaches the library call, whose remainder loop is ~60 instructions on
// the GPU; the results are the reference's bit for bit either way.
PPS_HD double standard_rad(double t) {
  if (t >= 0.) {
    const double x = t + kPi;
    return (x < kTwoPi ? x : (x < 2.0 * kTwoPi ? x - kTwoPi : fmod(x, kTwoPi))) - kPi;
  }
  const double x = t - kPi;
  return (x > -kTwoPi ? x : (x > -2.0 * kTwoPi ? x + kTwoPi : fmod(x, -kTwoPi))) + kPi;
}

// Eigen quaternion product a*b
PPS_HD void quat_mul(const double a[4], const double b[4], double o[4]) {
  PPS_FP_EXACT
  const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
  const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
  o[0] = aw * bx + ax * bw + ay * bz - az * by;
  o[1] = aw * by + ay * bw + az * bx - ax * bz;
  o[2] = aw * bz + az * bw + ax * by - ay * bx;
  o[3] = aw * bw - ax * bx - ay * by - az * bz;
}

// Eigen::Matrix3d(quat)  (isam/Rot3d.h:96-98); R row-major
PPS_HD void quat_to_R(const double q[4], double R[9]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// Eigen::Quaterniond(Matrix3d)  (isam/Rot3d.h:92-94)
PPS_HD void R_to_quat(const double R[9], double q[4]) {
  double t = R[0] + R[4] + R[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[7] - R[5]) * t;
    q[1] = (R[2] - R[6]) * t;
    q[2] = (R[3] - R[1]) * t;
  } else {
    // branch on the largest diagonal element; written without dynamic indexing
    const double d0 = R[0], d1 = R[4], d2 = R[8];
    if (d0 >= d1 && d0 >= d2) {          // i=0, j=1, k=2
      t = sqrt(d0 - d1 - d2 + 1.0);
      q[0] = 0.5 * t; t = 0.5 / t;
      q[3] = (R[7] - R[5]) * t; q[1] = (R[3] + R[1]) * t; q[2] = (R[6] + R[2]) * t;
    } else if (d1 > d0 && d1 >= d2) {    // i=1, j=2, k=0
      t = sqrt(d1 - d2 - d0 + 1.0);
      q[1] = 0.5 * t; t = 0.5 / t;
      q[3] = (R[2] - R[6]) * t; q[2] = (R[7] + R[5]) * t; q[0] = (R[1] + R[3]) * t;
    } else {                              // i=2, j=0, k=1
      t = sqrt(d2 - d0 - d1 + 1.0);
      q[2] = 0.5 * t; t = 0.5 / t;
      q[3] = (R[3] - R[1]) * t; q[0] = (R[2] + R[6]) * t; q[1] = (R[5] + R[7]) * t;
    }
  }
}

// isam/Rot3d.h:100-112
PPS_HD void euler_to_quat(double yaw, double pitch, double roll, double q[4]) {
  const double sy = sin(yaw * 0.5), cy = cos(yaw * 0.5);
  const double sp = sin(pitch * 0.5), cp = cos(pitch * 0.5);
  const double sr = sin(roll * 0.5), cr = cos(roll * 0.5);
  q[3] = cr * cp * cy + sr * sp * sy;
  q[0] = sr * cp * cy - cr * sp * sy;
  q[1] = cr * sp * cy + sr * cp * sy;
  q[2] = cr * cp * sy - sr * sp * cy;
}

// isam/Rot3d.h:114-124 ; ypr = (yaw, pitch, roll)
PPS_HD void quat_to_euler(const double q[4], double ypr[3]) {
  const double q0 = q[3], q1 = q[0], q2 = q[1], q3 = q[2];
  ypr[2] = atan2(2.0 * (q0 * q1 + q2 * q3), q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3);
  ypr[1] = asin(2.0 * (q0 * q2 - q3 * q1));
  ypr[0] = atan2(2.0 * (q0 * q3 + q1 * q2), q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3);
}

// Rot3d::delta3_to_quat  (isam/Rot3d.h:126-136)
PPS_HD void rot_exp(const double d[3], double q[4]) {
  PPS_FP_EXACT
  const double theta = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  double sn, cs;
  sincos(0.5 * theta, &sn, &cs);        // (one call: whether the compiler would pair a sin with a cos depends on the inlining context)
  double S;
  if (theta < 0.0001) S = 0.5 + theta * theta / 48.;
  else S = sn / theta;
  q[3] = cs;
  q[0] = S * d[0]; q[1] = S * d[1]; q[2] = S * d[2];
}

// boost::math::sinc_pi as used by Plane3d::delta3_to_quat (src/isam_plane3d.h:89)
// (sin_x = sin(x), evaluated by the caller together with the cosine)
PPS_HD double sinc_pi(double x, double sin_x) {
  PPS_FP_EXACT
  const double eps = 2.220446049250313e-16;
  const double t2 = 1.4901161193847656e-08;   // sqrt(eps)
  const double tn = 1.220703125e-04;          // eps^(1/4)
  const double ax = fabs(x);
  if (ax >= tn) return sin_x / x;
  double r = 1.0;
  if (ax >= eps) {
    const double x2 = x * x;
    r -= x2 / 6.0;
    if (ax >= t2) r += (x2 * x2) / 120.0;
  }
  return r;
}

// Plane3d::delta3_to_quat  (src/isam_plane3d.h:78-93)
PPS_HD void plane_exp(const double d[3], double q[4]) {
  PPS_FP_EXACT
  const double theta = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  double sn, cs;
  sincos(0.5 * theta, &sn, &cs);
  const double S = 0.5 * sinc_pi(0.5 * theta, sn);
  q[3] = cs;
  q[0] = S * d[0]; q[1] = S * d[1]; q[2] = S * d[2];
}

PPS_HD void normalize4(double v[4]) {
  PPS_FP_EXACT
  const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
  v[0] /= n; v[1] /= n; v[2] /= n; v[3] /= n;
}

// 1 / sqrt(x) to fp64 round-off.  On the device: the hardware estimate + two Newton steps in explicit fused multiply-adds (the
// same bits wherever it is inlined) -- 9 vector instructions where sqrt + division are 27; on the host: 1 / sqrt.
PPS_HD double rsqrt_acc(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rsq(x);
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const double t = x * y, hy = 0.5 * y;
    const double e = __builtin_fma(-t, hy, 0.5);       // 0.5 - 0.5 x y^2
    y = __builtin_fma(y, e, y);
  }
  return y;
#else
  return 1.0 / sqrt(x);
#endif
}

// v / |v| as v * (1 / |v|): what the hot loops use (the residual of a plane observation normalises the transformed plane in each of
// the 19 evaluations of a numerical Jacobian; four divisions + a square root were 66 of ~310 vector instructions of one
// evaluation, this is 13).  Eigen's Vector4d::normalize() is either form depending on its version (3.2: coefficient * inverse,
// 3.3: coefficient / norm); the two differ by an ulp, far inside the parity tolerance (chi2 rel 1e-5, measured 1e-13).
PPS_HD void normalize4_r(double v[4]) {
  PPS_FP_EXACT
  const double r = rsqrt_acc(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
  v[0] *= r; v[1] *= r; v[2] *= r; v[3] *= r;
}

// Pose3d::exmap  (isam/Pose3d.h:131-136): t += d[0:3] ; q <- q * Exp(d[3:6])
PPS_HD void pose_exmap(const double p[7], const double d[6], double o[7]) {
  PPS_FP_EXACT
  double dq[4], q[4];
  rot_exp(d + 3, dq);
  quat_mul(p + 3, dq, q);
  o[0] = p[0] + d[0]; o[1] = p[1] + d[1]; o[2] = p[2] + d[2];
  o[3] = q[0]; o[4] = q[1]; o[5] = q[2]; o[6] = q[3];
}

// Pose3d::exmap for the rotation steps of a numerical Jacobian: d = +-eps e_k (k = 0, 1, 2 of the rotation part).  Rot3d::exmap's
// quaternion of such a step is (+-a e_k, c) with (a, 0, 0, c) = rot_exp((eps, 0, 0)) -- theta = sqrt(eps^2) = eps whichever
// component carries it, S * 0 = 0 -- so the square root, the sine / cosine pair and the division are evaluated once per factor
// (rot_step_quat) instead of once per step; the product is quat_mul itself: the same bits as pose_exmap(p, d, o).
PPS_HD void rot_step_quat(double ac[2], double eps = kNumDiffEps) {      // (eps as a run-time value: no compile-time folding of the sine)
  const double d1[3] = {eps, 0.0, 0.0};
  double q[4];
  rot_exp(d1, q);
  ac[0] = q[0]; ac[1] = q[3];
}
PPS_HD void pose_exmap_rot_step(const double p[7], int k, bool minus, const double ac[2], double o[7]) {
  PPS_FP_EXACT
  const double a = minus ? -ac[0] : ac[0];
  const double dq[4] = {k == 0 ? a : 0.0, k == 1 ? a : 0.0, k == 2 ? a : 0.0, ac[1]};
  double q[4];
  quat_mul(p + 3, dq, q);
  o[0] = p[0] + 0.0; o[1] = p[1] + 0.0; o[2] = p[2] + 0.0;
  o[3] = q[0]; o[4] = q[1]; o[5] = q[2]; o[6] = q[3];
}

// Plane3d::exmap_3dof for the steps d = +-eps e_k of a numerical Jacobian: the step quaternion is (+-a e_k, c) with
// (a, 0, 0, c) = plane_exp((eps, 0, 0)), evaluated once per factor; the same bits as plane_exmap(pl, d, o).
PPS_HD void plane_step_quat(double ac[2], double eps = kNumDiffEps) {
  const double d1[3] = {eps, 0.0, 0.0};
  double q[4];
  plane_exp(d1, q);
  ac[0] = q[0]; ac[1] = q[3];
}
PPS_HD void plane_exmap_step(const double pl[4], int k, bool minus, const double ac[2], double o[4]) {
  PPS_FP_EXACT
  const double a = minus ? -ac[0] : ac[0];
  const double dq[4] = {k == 0 ? a : 0.0, k == 1 ? a : 0.0, k == 2 ? a : 0.0, ac[1]};
  quat_mul(dq, pl, o);
  normalize4_r(o);
}

// Plane3d::exmap_3dof  (src/isam_plane3d.h:101-127), plane_type == -1
PPS_HD void plane_exmap(const double pl[4], const double d[3], double o[4]) {
  PPS_FP_EXACT
  double dq[4];
  plane_exp(d, dq);
  quat_mul(dq, pl, o);
  normalize4_r(o);
}

// Plane3d::transform_to(wTo) = normalise(wTo^T pi)  (src/isam_plane3d.h:180-182); un-normalised u also returned
PPS_HD void plane_transform_to_raw(const double pl[4], const double pose[7], const double R[9], double u[4]) {
  u[0] = R[0] * pl[0] + R[3] * pl[1] + R[6] * pl[2];
  u[1] = R[1] * pl[0] + R[4] * pl[1] + R[7] * pl[2];
  u[2] = R[2] * pl[0] + R[5] * pl[1] + R[8] * pl[2];
  u[3] = pose[0] * pl[0] + pose[1] * pl[1] + pose[2] * pl[2] + pl[3];
}

// Plane3d::transform_from(oTw) = normalise(oTw^T pi)  (src/isam_plane3d.h:186-188)
PPS_HD void plane_transform_from(const double pl[4], const double pose[7], double o[4]) {
  double R[9];
  quat_to_R(pose + 3, R);
  const double C0 = -(R[0] * pose[0] + R[3] * pose[1] + R[6] * pose[2]);
  const double C1 = -(R[1] * pose[0] + R[4] * pose[1] + R[7] * pose[2]);
  const double C2 = -(R[2] * pose[0] + R[5] * pose[1] + R[8] * pose[2]);
  o[0] = R[0] * pl[0] + R[1] * pl[1] + R[2] * pl[2];
  o[1] = R[3] * pl[0] + R[4] * pl[1] + R[5] * pl[2];
  o[2] = R[6] * pl[0] + R[7] * pl[1] + R[8] * pl[2];
  o[3] = C0 * pl[0] + C1 * pl[1] + C2 * pl[2] + pl[3];
  normalize4(o);
}

// e = Log(q * conj(qm)) through Eigen::AngleAxisd with the wrap of src/isam_plane3d.h:286-294.
// Closed form valid for both Eigen generations: e = v/|v| * 2 atan2(|v|,|w|) * sign(w).
PPS_HD void log_diff(const double q[4], const double qm[4], double e[3], double dq[4]) {
  const double c[4] = {-qm[0], -qm[1], -qm[2], qm[3]};
  quat_mul(q, c, dq);
  const double nn = dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2];
  if (nn != 0.0) {
    const double ri = rsqrt_acc(nn);              // n = |v| = nn / sqrt(nn), angle / n = angle * ri: one reciprocal root, no division
    const double angle = 2.0 * atan2(nn * ri, fabs(dq[3]));
    const double s = dq[3] < 0 ? -(angle * ri) : angle * ri;
    e[0] = dq[0] * s; e[1] = dq[1] * s; e[2] = dq[2] * s;
  } else {
    e[0] = e[1] = e[2] = 0.0;
  }
}

// ---- residuals (unwhitened) -------------------------------------------------------------

// Pose3d_Plane3d_Factor::basic_error  (src/isam_plane3d.h:271-304)
PPS_HD void res_plane_obs(const double pose[7], const double plane[4], const double meas[4], double e[3]) {
  double R[9], u[4], dq[4];
  quat_to_R(pose + 3, R);
  plane_transform_to_raw(plane, pose, R, u);
  normalize4_r(u);
  log_diff(u, meas, e, dq);
}

// isam::get_wall_plane_equation (src/isam_plane3d.cpp:20-55) for one segment + the normalisation of
// Pose3d_Plane3d_Factor2::basic_error (src/isam_plane3d.h:392-394): the wall plane, in the sensor frame, whose
// ground edge is seen along the two rays `ray` (K^-1 (u,v,1) of the edge's end points, precompute_edge_ray
// :361-373), for the camera pose `pose`.  fp64 like the reference's twin of the fp32 pop-up.
PPS_HD void repop_wall_plane(const double pose[7], const double ray[6], double out[4]) {
  double R[9];
  quat_to_R(pose + 3, R);
  // ground_plane_sensor = wTo^T (0,0,-1,0)
  double gs[4];
  gs[0] = R[0] * 0.0 + R[3] * 0.0 + R[6] * -1.0 + 0.0 * 0.0;
  gs[1] = R[1] * 0.0 + R[4] * 0.0 + R[7] * -1.0 + 0.0 * 0.0;
  gs[2] = R[2] * 0.0 + R[5] * 0.0 + R[8] * -1.0 + 0.0 * 0.0;
  gs[3] = pose[0] * 0.0 + pose[1] * 0.0 + pose[2] * -1.0 + 1.0 * 0.0;
  double P[2][3];
  for (int j = 0; j < 2; j++) {
    const double* r = ray + 3 * j;
    const double frac = -gs[3] / (gs[0] * r[0] + gs[1] * r[1] + gs[2] * r[2]);   // ray_plane_interact :13-17
    P[j][0] = frac * r[0]; P[j][1] = frac * r[1]; P[j][2] = frac * r[2];
  }
  const double t1[3] = {P[1][0] - P[0][0], P[1][1] - P[0][1], P[1][2] - P[0][2]};
  const double n[3] = {t1[1] * gs[2] - t1[2] * gs[1], t1[2] * gs[0] - t1[0] * gs[2], t1[0] * gs[1] - t1[1] * gs[0]};
  out[0] = n[0]; out[1] = n[1]; out[2] = n[2];
  out[3] = -(n[0] * P[0][0] + n[1] * P[0][1] + n[2] * P[0][2]);
  normalize4(out);
}

// Pose3d_Plane3d_Factor2::basic_error (src/isam_plane3d.h:381-420): the measurement is re-popped at every evaluation
PPS_HD void res_plane_obs2(const double pose[7], const double plane[4], const double ray[6], double e[3]) {
  double ms[4];
  repop_wall_plane(pose, ray, ms);
  res_plane_obs(pose, plane, ms, e);
}

// Plane3d_Factor::basic_error  (src/isam_plane3d.h:449-473)
PPS_HD void res_plane_prior(const double plane[4], const double meas[4], double e[3]) {
  double dq[4];
  log_diff(plane, meas, e, dq);
}

// Pose3d_Factor::basic_error  (isam/slam3d.h:82-88)
PPS_HD void res_pose_prior(const double pose[7], const double meas6[6], double e[6]) {
  double ypr[3];
  quat_to_euler(pose + 3, ypr);
  e[0] = pose[0] - meas6[0]; e[1] = pose[1] - meas6[1]; e[2] = pose[2] - meas6[2];
  e[3] = standard_rad(ypr[0] - meas6[3]);
  e[4] = standard_rad(ypr[1] - meas6[4]);
  e[5] = standard_rad(ypr[2] - meas6[5]);
}

// p2.ominus(p1) = Pose3d(p1.oTw() * p2.wTo())  (isam/Pose3d.h:233-235): rotation matrix and translation
PPS_HD void ominus_Rt(const double p2[7], const double p1[7], double R12[9], double t12[3], double R1[9]) {
  double R2[9];
  quat_to_R(p1 + 3, R1);
  quat_to_R(p2 + 3, R2);
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double C = -(R1[0 * 3 + i] * p1[0] + R1[1 * 3 + i] * p1[1] + R1[2 * 3 + i] * p1[2]);
#pragma unroll
    for (int j = 0; j < 3; j++)
      R12[i * 3 + j] = R1[0 * 3 + i] * R2[0 * 3 + j] + R1[1 * 3 + i] * R2[1 * 3 + j] + R1[2 * 3 + i] * R2[2 * 3 + j];
    t12[i] = R1[0 * 3 + i] * p2[0] + R1[1 * 3 + i] * p2[1] + R1[2 * 3 + i] * p2[2] + C;
  }
}

// the rotational half of a relative-pose residual: R12 -> quaternion -> Euler angles (as Pose3d(Matrix4d) does) -> wrapped differences
PPS_HD void euler_residual(const double R12[9], const double meas6[6], double e3[3]) {
  double q[4], ypr[3];
  R_to_quat(R12, q);
  quat_to_euler(q, ypr);
  e3[0] = standard_rad(ypr[0] - meas6[3]);
  e3[1] = standard_rad(ypr[1] - meas6[4]);
  e3[2] = standard_rad(ypr[2] - meas6[5]);
}

// Pose3d_Pose3d_Factor::basic_error  (isam/slam3d.h:174-191): matrix -> quaternion -> Euler, as Pose3d(Matrix4d)
PPS_HD void res_odometry(const double p1[7], const double p2[7], const double meas6[6], double e[6]) {
  double R12[9], t12[3], R1[9];
  ominus_Rt(p2, p1, R12, t12, R1);
  e[0] = t12[0] - meas6[0]; e[1] = t12[1] - meas6[1]; e[2] = t12[2] - meas6[2];
  euler_residual(R12, meas6, e + 3);
}

// The translation of p2.ominus(p1) alone, for a p1 / p2 whose ROTATION is the one R1 was built from: the same expressions, in
// the same order, as ominus_Rt's (C = -R1^T t1, then R1^T t2 + C).  What a translation column of the numerical Jacobian needs.
PPS_HD void ominus_t(const double R1[9], const double t1[3], const double t2[3], double t12[3]) {
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double C = -(R1[0 * 3 + i] * t1[0] + R1[1 * 3 + i] * t1[1] + R1[2 * 3 + i] * t1[2]);
    t12[i] = R1[0 * 3 + i] * t2[0] + R1[1 * 3 + i] * t2[1] + R1[2 * 3 + i] * t2[2] + C;
  }
}

// res_plane_obs from the un-normalised transformed plane u = wTo^T pi (src/isam_plane3d.h:180-182, 271-304)
PPS_HD void res_plane_obs_u(const double u_raw[4], const double meas[4], double e[3]) {
  double u[4] = {u_raw[0], u_raw[1], u_raw[2], u_raw[3]}, dq[4];
  normalize4_r(u);
  log_diff(u, meas, e, dq);
}

// r = U e for a packed upper-triangular U (Factor::error, isam/Factor.h:67-77)
template <int M>
PPS_HD void whiten(const double* ut, const double e[M], double r[M]) {
  int k = 0;
#pragma unroll
  for (int i = 0; i < M; i++) {
    double s = 0;
#pragma unroll
    for (int j = i; j < M; j++) s += ut[k++] * e[j];
    r[i] = s;
  }
}

// J <- U J for a packed upper-triangular U; J is M x N row-major, in place (rows processed top-down)
template <int M, int N>
PPS_HD void whiten_rows(const double* ut, double* J) {
  int k = 0;
#pragma unroll
  for (int i = 0; i < M; i++) {
#pragma unroll
    for (int c = 0; c < N; c++) {
      double s = 0;
#pragma unroll
      for (int j = i; j < M; j++) s += ut[k + (j - i)] * J[j * N + c];
      J[i * N + c] = s;
    }
    k += M - i;
  }
}

// ---- analytic derivatives ---------------------------------------------------------------

// d Log(dq) / d dq, dq = (v,w) unit; D is 3x4 (columns x,y,z,w)
PPS_HD void dlog_dq(const double dq[4], double D[12]) {
  const double x = dq[0], y = dq[1], z = dq[2], w = dq[3];
  const double s2 = x * x + y * y + z * z, s = sqrt(s2);
  const double nn = s2 + w * w;
  if (s < 1e-12) {
#pragma unroll
    for (int i = 0; i < 12; i++) D[i] = 0.0;
    D[0] = D[5] = D[10] = 2.0 / w;
    return;
  }
  const double sg = (w < 0) ? -1.0 : 1.0;
  const double phi = 2.0 * atan2(s, fabs(w)) * sg;
  const double a = phi / s;
  const double dphids = 2.0 * w / nn;
  const double dphidw = -2.0 * s / nn;
  const double v[3] = {x, y, z};
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const double vv = v[i] * v[j] / s2;
      D[i * 4 + j] = a * ((i == j ? 1.0 : 0.0) - vv) + dphids * vv;
    }
    D[i * 4 + 3] = dphidw * v[i] / s;
  }
}

// A(3x4) = DL(3x4) * Q(4x4) where Q = d(q * conj(qm))/dq
PPS_HD void dlog_times_Q(const double DL[12], const double qm[4], double A[12]) {
  const double cx = -qm[0], cy = -qm[1], cz = -qm[2], cw = qm[3];
  const double Q[16] = {cw, cz, -cy, cx, -cz, cw, cx, cy, cy, -cx, cw, cz, -cx, -cy, -cz, cw};
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      double s = 0;
#pragma unroll
      for (int l = 0; l < 4; l++) s += DL[i * 4 + l] * Q[l * 4 + j];
      A[i * 4 + j] = s;
    }
}

// d(yaw,pitch,roll)/d(omega) for a right (body-frame) perturbation R Exp(omega), ZYX Euler angles
PPS_HD void dypr_domega(double pitch, double roll, double E[9]) {
  const double sr = sin(roll), cr = cos(roll), cp = cos(pitch), tp = tan(pitch);
  E[0] = 0.0; E[1] = sr / cp; E[2] = cr / cp;
  E[3] = 0.0; E[4] = cr;      E[5] = -sr;
  E[6] = 1.0; E[7] = sr * tp; E[8] = cr * tp;
}

// Pose-plane edge: residual e (3), Jp = de/d(pose tangent) 3x6, Jl = de/d(plane tangent) 3x3 (unwhitened)
PPS_HD void jac_plane_obs(const double pose[7], const double plane[4], const double meas[4], double e[3],
                          double Jp[18], double Jl[9]) {
  double R[9], u[4], dq[4];
  quat_to_R(pose + 3, R);
  plane_transform_to_raw(plane, pose, R, u);
  const double un = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2] + u[3] * u[3]);
  const double p[4] = {u[0] / un, u[1] / un, u[2] / un, u[3] / un};
  log_diff(p, meas, e, dq);
  double DL[12], A[12], B[12];
  dlog_dq(dq, DL);
  dlog_times_Q(DL, meas, A);
  // B = A * (I - p p^T)/|u|   (3x4)
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double ap = A[i * 4 + 0] * p[0] + A[i * 4 + 1] * p[1] + A[i * 4 + 2] * p[2] + A[i * 4 + 3] * p[3];
#pragma unroll
    for (int j = 0; j < 4; j++) B[i * 4 + j] = (A[i * 4 + j] - ap * p[j]) / un;
  }
  // pose tangent: translation moves u3 by n.dt ; rotation R<-R Exp(dth): m=R^T n -> m + [m]x dth
  const double m0 = u[0], m1 = u[1], m2 = u[2];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double b0 = B[i * 4 + 0], b1 = B[i * 4 + 1], b2 = B[i * 4 + 2], b3 = B[i * 4 + 3];
    Jp[i * 6 + 0] = b3 * plane[0];
    Jp[i * 6 + 1] = b3 * plane[1];
    Jp[i * 6 + 2] = b3 * plane[2];
    Jp[i * 6 + 3] = b1 * m2 - b2 * m1;
    Jp[i * 6 + 4] = b2 * m0 - b0 * m2;
    Jp[i * 6 + 5] = b0 * m1 - b1 * m0;
  }
  // plane tangent: q' = Exp(d) * q  ->  d(pi)/dd = 0.5 [w I - [v]x ; -v^T], then u = M pi, M = [R^T 0; t^T 1]
  const double a = plane[0], b = plane[1], c = plane[2], d = plane[3];
  const double Dp[12] = {0.5 * d, 0.5 * c, -0.5 * b, -0.5 * c, 0.5 * d, 0.5 * a,
                         0.5 * b, -0.5 * a, 0.5 * d, -0.5 * a, -0.5 * b, -0.5 * c};
  double MD[12];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const double n0 = Dp[0 * 3 + j], n1 = Dp[1 * 3 + j], n2 = Dp[2 * 3 + j], n3 = Dp[3 * 3 + j];
    MD[0 * 3 + j] = R[0] * n0 + R[3] * n1 + R[6] * n2;
    MD[1 * 3 + j] = R[1] * n0 + R[4] * n1 + R[7] * n2;
    MD[2 * 3 + j] = R[2] * n0 + R[5] * n1 + R[8] * n2;
    MD[3 * 3 + j] = pose[0] * n0 + pose[1] * n1 + pose[2] * n2 + n3;
  }
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      Jl[i * 3 + j] = B[i * 4 + 0] * MD[0 * 3 + j] + B[i * 4 + 1] * MD[1 * 3 + j] + B[i * 4 + 2] * MD[2 * 3 + j] +
                      B[i * 4 + 3] * MD[3 * 3 + j];
}

PPS_HD void jac_plane_prior(const double plane[4], const double meas[4], double e[3], double Jl[9]) {
  double dq[4], DL[12], A[12];
  log_diff(plane, meas, e, dq);
  dlog_dq(dq, DL);
  dlog_times_Q(DL, meas, A);
  const double a = plane[0], b = plane[1], c = plane[2], d = plane[3];
  const double Dp[12] = {0.5 * d, 0.5 * c, -0.5 * b, -0.5 * c, 0.5 * d, 0.5 * a,
                         0.5 * b, -0.5 * a, 0.5 * d, -0.5 * a, -0.5 * b, -0.5 * c};
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      Jl[i * 3 + j] = A[i * 4 + 0] * Dp[0 * 3 + j] + A[i * 4 + 1] * Dp[1 * 3 + j] + A[i * 4 + 2] * Dp[2 * 3 + j] +
                      A[i * 4 + 3] * Dp[3 * 3 + j];
}

PPS_HD void jac_pose_prior(const double pose[7], const double meas6[6], double e[6], double J[36]) {
  double ypr[3], E[9];
  quat_to_euler(pose + 3, ypr);
  e[0] = pose[0] - meas6[0]; e[1] = pose[1] - meas6[1]; e[2] = pose[2] - meas6[2];
  e[3] = standard_rad(ypr[0] - meas6[3]);
  e[4] = standard_rad(ypr[1] - meas6[4]);
  e[5] = standard_rad(ypr[2] - meas6[5]);
  dypr_domega(ypr[1], ypr[2], E);
#pragma unroll
  for (int i = 0; i < 36; i++) J[i] = 0.0;
  J[0] = J[7] = J[14] = 1.0;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) J[(3 + i) * 6 + 3 + j] = E[i * 3 + j];
}

// Odometry edge: J1 = de/d(pose1 tangent), J2 = de/d(pose2 tangent), both 6x6 (unwhitened)
PPS_HD void jac_odometry(const double p1[7], const double p2[7], const double meas6[6], double e[6], double J1[36],
                         double J2[36]) {
  double R12[9], t12[3], R1[9], q[4], ypr[3], E[9];
  ominus_Rt(p2, p1, R12, t12, R1);
  R_to_quat(R12, q);
  quat_to_euler(q, ypr);
  e[0] = t12[0] - meas6[0]; e[1] = t12[1] - meas6[1]; e[2] = t12[2] - meas6[2];
  e[3] = standard_rad(ypr[0] - meas6[3]);
  e[4] = standard_rad(ypr[1] - meas6[4]);
  e[5] = standard_rad(ypr[2] - meas6[5]);
  dypr_domega(ypr[1], ypr[2], E);
#pragma unroll
  for (int i = 0; i < 36; i++) { J1[i] = 0.0; J2[i] = 0.0; }
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      J1[i * 6 + j] = -R1[j * 3 + i];
      J2[i * 6 + j] = R1[j * 3 + i];
    }
  J1[0 * 6 + 4] = -t12[2]; J1[0 * 6 + 5] = t12[1];
  J1[1 * 6 + 3] = t12[2];  J1[1 * 6 + 5] = -t12[0];
  J1[2 * 6 + 3] = -t12[1]; J1[2 * 6 + 4] = t12[0];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const double er = E[i * 3 + 0] * R12[j * 3 + 0] + E[i * 3 + 1] * R12[j * 3 + 1] + E[i * 3 + 2] * R12[j * 3 + 2];
      J1[(3 + i) * 6 + 3 + j] = -er;
      J2[(3 + i) * 6 + 3 + j] = E[i * 3 + j];
    }
}

}  // namespace pps
