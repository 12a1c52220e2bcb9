// pps_isam.hpp -- header-only C++ facade: the subset of the reference's iSAM / plane-factor surface
// that pop_planar_slam's mapper uses, forwarding to the C-ABI of libpps.so (include/pps.h).
//
// Mirrors (paths relative to /root/reference/pop_planar_slam):
//   isam::Pose3d                 Thirdparty/isam/include/isam/Pose3d.h:70-274, Rot3d.h:46-278
//   isam::Plane3d                src/isam_plane3d.h:27-193
//   isam::Noise / Covariance / SqrtInformation / Information   Thirdparty/isam/include/isam/Noise.h:36-62
//   isam::Pose3d_Node, Plane3d_Node                            slam3d.h:39, src/isam_plane3d.h:197-210, Node.h:99-154
//   isam::Pose3d_Factor, Pose3d_Pose3d_Factor                  slam3d.h:58-193
//   isam::Pose3d_Plane3d_Factor, Plane3d_Factor                src/isam_plane3d.h:221-308,428-474
//   isam::Properties, isam::Slam                               Properties.h:37-110, Slam.h:66-277
// Same raw-pointer, non-owning semantics as the reference ("the node itself is not deallocated",
// Slam.h:122-134).  The image has no Eigen, so small fixed-size std::array types stand in for
// Eigen::Vector/Matrix; a maintainer maps them with Eigen::Map.
//
// Everything numeric about the solve happens on the GPU behind the C-ABI; the host math kept here is
// what the mapper itself evaluates between solves (oplus/ominus for the odometry guess,
// transform_from to initialise a new landmark, Covariance -> sqrt information).
#pragma once

#include <array>
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "pps.h"

namespace isam {

typedef std::array<double, 3> Vector3d;
typedef std::array<double, 4> Vector4d;
typedef std::array<double, 6> Vector6d;
typedef std::array<double, 9> Matrix3d;    // row-major
typedef std::array<double, 16> Matrix4d;   // row-major

namespace detail {
inline double standardRad(double t) {   // util.h:101-108
  const double PI = 3.14159265358979323846, TWOPI = 2 * PI;
  if (t >= 0.) t = std::fmod(t + PI, TWOPI) - PI;
  else t = std::fmod(t - PI, -TWOPI) + PI;
  return t;
}
inline Matrix3d quat_to_R(const Vector4d& q) {   // Eigen toRotationMatrix; q = (x,y,z,w)
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  return {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)};
}
inline Vector4d R_to_quat(const Matrix3d& R) {   // Eigen::Quaterniond(Matrix3d)
  Vector4d q{};
  double t = R[0] + R[4] + R[8];
  if (t > 0) {
    t = std::sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
    q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 3 + i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
    q[i] = 0.5 * t; t = 0.5 / t;
    q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t;
    q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
  }
  return q;
}
inline void check(int rc, pps_graph* g, const char* what) {
  if (rc != PPS_OK) throw std::runtime_error(std::string(what) + ": " + (g ? pps_last_error(g) : "pps error"));
}
}  // namespace detail

// ---- Pose3d -----------------------------------------------------------------------------------
class Pose3d {
  Vector3d _t{{0, 0, 0}};
  Vector4d _q{{0, 0, 0, 1}};   // (x,y,z,w)
public:
  static const int dim = 6;
  Pose3d() {}
  Pose3d(double x, double y, double z, double yaw, double pitch, double roll) { set(x, y, z, yaw, pitch, roll); }
  explicit Pose3d(const Matrix4d& wTo) {   // Pose3d(Matrix 4x4), Pose3d.h:90-104
    const double s = wTo[15];
    _t = {wTo[3] / s, wTo[7] / s, wTo[11] / s};
    _q = detail::R_to_quat({wTo[0] / s, wTo[1] / s, wTo[2] / s, wTo[4] / s, wTo[5] / s, wTo[6] / s, wTo[8] / s, wTo[9] / s, wTo[10] / s});
  }
  static Pose3d from_tq(const double tq[7]) { Pose3d p; p._t = {tq[0], tq[1], tq[2]}; p._q = {tq[3], tq[4], tq[5], tq[6]}; return p; }
  void to_tq(double tq[7]) const { tq[0] = _t[0]; tq[1] = _t[1]; tq[2] = _t[2]; tq[3] = _q[0]; tq[4] = _q[1]; tq[5] = _q[2]; tq[6] = _q[3]; }
  double x() const { return _t[0]; }
  double y() const { return _t[1]; }
  double z() const { return _t[2]; }
  const Vector4d& quaternion_xyzw() const { return _q; }
  void ypr(double& yaw, double& pitch, double& roll) const {   // Rot3d::quat_to_euler, Rot3d.h:114-124
    const double q0 = _q[3], q1 = _q[0], q2 = _q[1], q3 = _q[2];
    roll = std::atan2(2.0 * (q0 * q1 + q2 * q3), q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3);
    pitch = std::asin(2.0 * (q0 * q2 - q3 * q1));
    yaw = std::atan2(2.0 * (q0 * q3 + q1 * q2), q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3);
  }
  double yaw() const { double a, b, c; ypr(a, b, c); return a; }
  double pitch() const { double a, b, c; ypr(a, b, c); return b; }
  double roll() const { double a, b, c; ypr(a, b, c); return c; }
  Vector6d vector() const { double Y, P, R; ypr(Y, P, R); return {_t[0], _t[1], _t[2], Y, P, R}; }   // Pose3d.h:138-145
  void set(double x, double y, double z, double yaw, double pitch, double roll) {   // Rot3d::euler_to_quat, Rot3d.h:100-112
    _t = {x, y, z};
    const double sy = std::sin(yaw * 0.5), cy = std::cos(yaw * 0.5), sp = std::sin(pitch * 0.5), cp = std::cos(pitch * 0.5),
                 sr = std::sin(roll * 0.5), cr = std::cos(roll * 0.5);
    _q = {sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy};
  }
  void set(const Vector6d& v) { set(v[0], v[1], v[2], detail::standardRad(v[3]), detail::standardRad(v[4]), detail::standardRad(v[5])); }
  Matrix3d wRo() const { return detail::quat_to_R(_q); }
  Matrix4d wTo() const {   // Pose3d.h:188-194
    const Matrix3d R = wRo();
    return {R[0], R[1], R[2], _t[0], R[3], R[4], R[5], _t[1], R[6], R[7], R[8], _t[2], 0, 0, 0, 1};
  }
  Matrix4d oTw() const {   // Pose3d.h:203-213
    const Matrix3d R = wRo();
    const double C0 = -(R[0] * _t[0] + R[3] * _t[1] + R[6] * _t[2]), C1 = -(R[1] * _t[0] + R[4] * _t[1] + R[7] * _t[2]),
                 C2 = -(R[2] * _t[0] + R[5] * _t[1] + R[8] * _t[2]);
    return {R[0], R[3], R[6], C0, R[1], R[4], R[7], C1, R[2], R[5], R[8], C2, 0, 0, 0, 1};
  }
  static Matrix4d mul(const Matrix4d& A, const Matrix4d& B) {
    Matrix4d C{};
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double s = 0; for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 4 + j]; C[i * 4 + j] = s; }
    return C;
  }
  Pose3d oplus(const Pose3d& d) const { return Pose3d(mul(wTo(), d.wTo())); }      // Pose3d.h:222-224
  Pose3d ominus(const Pose3d& b) const { return Pose3d(mul(b.oTw(), wTo())); }    // Pose3d.h:233-235
};

// ---- Plane3d ----------------------------------------------------------------------------------
class Plane3d {
  Vector4d _abcd{{1, 0, 0, 0}};
  void _normalize() { const double n = std::sqrt(_abcd[0] * _abcd[0] + _abcd[1] * _abcd[1] + _abcd[2] * _abcd[2] + _abcd[3] * _abcd[3]); for (double& v : _abcd) v /= n; }
public:
  static const int dim = 3;
  Plane3d() {}
  explicit Plane3d(const Vector4d& vec) : _abcd(vec) { _normalize(); }   // src/isam_plane3d.h:59-66
  Vector4d vector() const { return _abcd; }
  void set(const Vector4d& v) { _abcd = v; _normalize(); }
  Vector3d normal() const { const double n = std::sqrt(_abcd[0] * _abcd[0] + _abcd[1] * _abcd[1] + _abcd[2] * _abcd[2]); return {_abcd[0] / n, _abcd[1] / n, _abcd[2] / n}; }
  double d() const { return -_abcd[3] / std::sqrt(_abcd[0] * _abcd[0] + _abcd[1] * _abcd[1] + _abcd[2] * _abcd[2]); }
  double distance() const { return std::fabs(d()); }
  Vector3d point0() const { const Vector3d n = normal(); const double dd = d(); return {dd * n[0], dd * n[1], dd * n[2]}; }   // :161-163
  double distance(const Vector3d& pt) const {                                                                                  // :165-167
    const Vector3d n = normal(), p0 = point0();
    return std::fabs(n[0] * (pt[0] - p0[0]) + n[1] * (pt[1] - p0[1]) + n[2] * (pt[2] - p0[2]));
  }
  Vector3d project_to_plane(const Vector3d& pt) const {   // :169-174, used by Mapper_mono::reproj_to_newplane (Mapping.cpp:609-632)
    const Vector3d n = normal();
    const double k = n[0] * pt[0] + n[1] * pt[1] + n[2] * pt[2] - d();
    return {pt[0] - n[0] * k, pt[1] - n[1] * k, pt[2] - n[2] * k};
  }
  static Vector4d tmul(const Matrix4d& T, const Vector4d& v) {   // T^T v
    Vector4d o{};
    for (int k = 0; k < 4; k++) o[k] = T[0 * 4 + k] * v[0] + T[1 * 4 + k] * v[1] + T[2 * 4 + k] * v[2] + T[3 * 4 + k] * v[3];
    return o;
  }
  Plane3d transform_to(const Matrix4d& wTo_pose) const { return Plane3d(tmul(wTo_pose, _abcd)); }      // :180-182
  Plane3d transform_from(const Matrix4d& oTw_pose) const { return Plane3d(tmul(oTw_pose, _abcd)); }    // :186-188
};

// ---- noise models (Noise.h:36-62); sqrtinf kept as a packed upper triangle ---------------------
class Noise {
public:
  std::vector<double> _ut;   // packed upper-triangular sqrt information, row-major
  int _n = 0;
  const std::vector<double>& sqrtinf_ut() const { return _ut; }
};
class SqrtInformation : public Noise {
public:
  SqrtInformation(const double* sqrtinf_rowmajor, int n) { _n = n; for (int r = 0; r < n; r++) for (int c = r; c < n; c++) _ut.push_back(sqrtinf_rowmajor[r * n + c]); }
};
// Covariance(cov): sqrtinf = chol(cov^-1) upper (Noise.h:57-62).  The app only builds diagonal covariances
// (Mapping.cpp:64-67,510-512) -> diag(1/sigma); full SPD matrices go through a dense inverse + Cholesky.
class Covariance : public Noise {
public:
  static Covariance diagonal(const double* variances, int n) {
    Covariance c; c._n = n;
    for (int r = 0; r < n; r++) for (int cc = r; cc < n; cc++) c._ut.push_back(r == cc ? 1.0 / std::sqrt(variances[r]) : 0.0);
    return c;
  }
  Covariance() {}
  Covariance(const double* cov_rowmajor, int n) {
    _n = n;
    // information = cov^-1 by Gauss-Jordan, then upper Cholesky factor U with U^T U = information
    std::vector<double> A(cov_rowmajor, cov_rowmajor + n * n), I(n * n, 0.0);
    for (int i = 0; i < n; i++) I[i * n + i] = 1.0;
    for (int c = 0; c < n; c++) {
      int piv = c;
      for (int r = c + 1; r < n; r++) if (std::fabs(A[r * n + c]) > std::fabs(A[piv * n + c])) piv = r;
      for (int k = 0; k < n; k++) { std::swap(A[c * n + k], A[piv * n + k]); std::swap(I[c * n + k], I[piv * n + k]); }
      const double d = A[c * n + c];
      for (int k = 0; k < n; k++) { A[c * n + k] /= d; I[c * n + k] /= d; }
      for (int r = 0; r < n; r++) if (r != c) { const double f = A[r * n + c]; for (int k = 0; k < n; k++) { A[r * n + k] -= f * A[c * n + k]; I[r * n + k] -= f * I[c * n + k]; } }
    }
    std::vector<double> L(n * n, 0.0);   // lower Cholesky of the information matrix; U = L^T
    for (int i = 0; i < n; i++)
      for (int j = 0; j <= i; j++) {
        double s = I[i * n + j];
        for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k];
        L[i * n + j] = (i == j) ? std::sqrt(s) : s / L[j * n + j];
      }
    for (int r = 0; r < n; r++) for (int c = r; c < n; c++) _ut.push_back(L[c * n + r]);
  }
};

class Slam;

// ---- nodes ------------------------------------------------------------------------------------
class Node {
protected:
  friend class Slam;
  Slam* _slam = nullptr;
  int _id = -1;   // backend id once the value is known
  bool _init = false;
public:
  virtual ~Node() {}
  virtual int dim() const = 0;
  bool initialized() const { return _init; }
  int backend_id() const { return _id; }
protected:
  virtual void push() = 0;   // create / overwrite the backend node from the host value
};

class Pose3d_Node : public Node {
  Pose3d _v;
public:
  int dim() const { return 6; }
  void init(const Pose3d& p);          // NodeT::init, Node.h:121-124
  Pose3d value() const;                // NodeT::value(ESTIMATE), Node.h:130
protected:
  void push();
};

class Plane3d_Node : public Node {
  Plane3d _v;
public:
  int dim() const { return 3; }
  void init(const Plane3d& p);
  Plane3d value() const;
protected:
  void push();
};

// ---- factors ----------------------------------------------------------------------------------
class Factor {
protected:
  friend class Slam;
  Slam* _slam = nullptr;
  int _id = -1;
  std::vector<double> _ut;
public:
  virtual ~Factor() {}
  int backend_id() const { return _id; }
  virtual void initialize() = 0;       // Factor::initialize, called from Slam::add_factor (Slam.cpp:96-99)
protected:
  virtual void push(pps_graph* g) = 0;
};

class Pose3d_Factor : public Factor {           // slam3d.h:58-89
  Pose3d_Node* _pose; Pose3d _measure;
public:
  Pose3d_Factor(Pose3d_Node* pose, const Pose3d& prior, const Noise& noise) : _pose(pose), _measure(prior) { _ut = noise.sqrtinf_ut(); }
  void initialize() { if (!_pose->initialized()) _pose->init(_measure); }
protected:
  void push(pps_graph* g) {
    const Vector6d m = _measure.vector();
    detail::check(pps_add_pose_prior(g, _pose->backend_id(), m.data(), _ut.data(), &_id), g, "pps_add_pose_prior");
  }
};

class Pose3d_Pose3d_Factor : public Factor {    // slam3d.h:91-193
  Pose3d_Node *_pose1, *_pose2; Pose3d _measure;
public:
  Pose3d_Pose3d_Factor(Pose3d_Node* pose1, Pose3d_Node* pose2, const Pose3d& measure, const Noise& noise)
      : _pose1(pose1), _pose2(pose2), _measure(measure) { _ut = noise.sqrtinf_ut(); }
  void initialize() {
    if (!_pose1->initialized() && !_pose2->initialized()) throw std::runtime_error("slam3d: Pose3d_Pose3d_Factor requires pose1 or pose2 to be initialized");
    if (!_pose1->initialized()) { Pose3d z; _pose1->init(_pose2->value().oplus(z.ominus(_measure))); }
    else if (!_pose2->initialized()) _pose2->init(_pose1->value().oplus(_measure));
  }
protected:
  void push(pps_graph* g) {
    const Vector6d m = _measure.vector();
    detail::check(pps_add_odometry(g, _pose1->backend_id(), _pose2->backend_id(), m.data(), _ut.data(), &_id), g, "pps_add_odometry");
  }
};

class Pose3d_Plane3d_Factor : public Factor {   // src/isam_plane3d.h:221-308, relative = false
  Pose3d_Node* _pose; Plane3d_Node* _plane; Plane3d _measure;
public:
  Pose3d_Plane3d_Factor(Pose3d_Node* pose, Plane3d_Node* plane, const Plane3d& measure, const Noise& noise, bool relative = false)
      : _pose(pose), _plane(plane), _measure(measure) {
    if (relative) throw std::runtime_error("Pose3d_Plane3d_Factor: relative parameterisation is not used by the mapper (Mapping.cpp:21)");
    _ut = noise.sqrtinf_ut();
  }
  void initialize() {
    if (!_pose->initialized()) throw std::runtime_error("Plane3d: Pose3d_Plane3d_Factor requires pose to be initialized");
    if (!_plane->initialized()) _plane->init(_measure.transform_from(_pose->value().oTw()));
  }
  const Plane3d& measurement() const { return _measure; }
  void set_measurement(const Plane3d& m);       // FactorT::set_measurement, Factor.h:206
protected:
  void push(pps_graph* g) {
    const Vector4d m = _measure.vector();
    detail::check(pps_add_plane_obs(g, _pose->backend_id(), _plane->backend_id(), m.data(), _ut.data(), &_id), g, "pps_add_plane_obs");
  }
};

// src/isam_plane3d.h:314-424: a plane observation whose measurement is re-popped from the two ground-edge rays at every
// evaluation (get_wall_plane_equation, src/isam_plane3d.cpp:13-55) -- the variant Mapping.cpp:515-521 keeps next to the plain
// factor.  As in the reference precompute_edge_ray() must be called before the factor is added; not for the ground plane.
class Pose3d_Plane3d_Factor2 : public Factor {
  Pose3d_Node* _pose; Plane3d_Node* _plane; Plane3d _measure;
  double _ray[6] = {0, 0, 0, 0, 0, 0}; bool _have_ray = false;
public:
  Pose3d_Plane3d_Factor2(Pose3d_Node* pose, Plane3d_Node* plane, const Plane3d& measure, const Noise& noise, bool relative = false)
      : _pose(pose), _plane(plane), _measure(measure) {
    if (relative) throw std::runtime_error("Pose3d_Plane3d_Factor2: relative parameterisation is not used by the mapper (Mapping.cpp:21)");
    _ut = noise.sqrtinf_ut();
  }
  void initialize() {
    if (!_pose->initialized()) throw std::runtime_error("Plane3d: Pose3d_Plane3d_Factor requires pose to be initialized");
    if (!_plane->initialized()) _plane->init(_measure.transform_from(_pose->value().oTw()));
  }
  // invK: row-major 3x3 (fp32, as the reference's Eigen::Matrix3f), ground_seg2d_line: (u0, v0, u1, v1) of the ONE ground edge
  void precompute_edge_ray(const float invK[9], const float ground_seg2d_line[4]) {
    if (pps_edge_ray(invK, ground_seg2d_line, _ray) != PPS_OK) throw std::runtime_error("Pose3d_Plane3d_Factor2: precompute_edge_ray");
    _have_ray = true;
  }
  const Plane3d& measurement() const { return _measure; }
  const double* edge_ray() const { return _ray; }
protected:
  void push(pps_graph* g) {
    if (!_have_ray) throw std::runtime_error("Pose3d_Plane3d_Factor2: precompute_edge_ray() before add_factor() (isam_plane3d.h:359)");
    const Vector4d m = _measure.vector();
    detail::check(pps_add_plane_obs2(g, _pose->backend_id(), _plane->backend_id(), m.data(), _ray, _ut.data(), &_id), g, "pps_add_plane_obs2");
  }
};

class Plane3d_Factor : public Factor {          // src/isam_plane3d.h:428-474
  Plane3d_Node* _plane; Plane3d _measure;
public:
  Plane3d_Factor(Plane3d_Node* plane, const Plane3d& prior, const Noise& noise) : _plane(plane), _measure(prior) { _ut = noise.sqrtinf_ut(); }
  void initialize() { if (!_plane->initialized()) _plane->init(_measure); }
protected:
  void push(pps_graph* g) {
    const Vector4d m = _measure.vector();
    detail::check(pps_add_plane_prior(g, _plane->backend_id(), m.data(), _ut.data(), &_id), g, "pps_add_plane_prior");
  }
};

// ---- Properties / Slam ------------------------------------------------------------------------
enum Method { GAUSS_NEWTON, LEVENBERG_MARQUARDT, DOG_LEG };

class Properties {   // Properties.h:37-110 (defaults of the reference, not of the app)
public:
  bool verbose = false, quiet = false, force_numerical_jacobian = false;
  Method method = GAUSS_NEWTON;
  double epsilon1 = 1e-2, epsilon2 = 1e-2, epsilon3 = 1e-2, epsilon_abs = 1e-3, epsilon_rel = 1e-5;
  int max_iterations = 500;
  double lm_lambda0 = 1e-6, lm_lambda_factor = 10.;
  int mod_update = 1, mod_batch = 100, mod_solve = 1;
  // backend selection (the analogue of Cholesky::Create, Cholesky.cpp:393-399)
  int jacobian_mode = PPS_JAC_NUMERIC;
  int device = 0;
};

class Slam {
  pps_graph* _g = nullptr;
  Properties _prop;
  int _num_nodes = 0, _num_factors = 0;
  void sync_props() {
    pps_props p; pps_default_props(&p);
    p.epsilon2 = _prop.epsilon2; p.epsilon_abs = _prop.epsilon_abs; p.epsilon_rel = _prop.epsilon_rel;
    p.max_iterations = _prop.max_iterations; p.lm_lambda0 = _prop.lm_lambda0; p.lm_lambda_factor = _prop.lm_lambda_factor;
    p.jacobian_mode = _prop.jacobian_mode; p.device = _prop.device; p.verbose = (_prop.verbose && !_prop.quiet) ? 1 : 0;
    detail::check(pps_set_props(_g, &p), _g, "pps_set_props");
  }
public:
  Slam() { detail::check(pps_graph_create(nullptr, &_g), nullptr, "pps_graph_create"); sync_props(); }
  ~Slam() { if (_g) pps_graph_destroy(_g); }
  Slam(const Slam&) = delete;
  Slam& operator=(const Slam&) = delete;
  pps_graph* handle() { return _g; }
  const Properties& properties() const { return _prop; }
  void set_properties(const Properties& p) {
    if (p.method != LEVENBERG_MARQUARDT) throw std::runtime_error("pps backend implements the mapper's configuration: method = LEVENBERG_MARQUARDT (Mapping.cpp:33)");
    if (p.mod_batch != 1) throw std::runtime_error("pps backend implements mod_batch = 1 (Mapping.cpp:34): every update() is a batch step");
    _prop = p;
    sync_props();
  }
  void add_node(Node* node) {          // Slam::add_node, Slam.cpp:91-94
    node->_slam = this;
    if (node->_init && node->_id < 0) node->push();
    _num_nodes++;
  }
  void add_factor(Factor* factor) {    // Slam::add_factor, Slam.cpp:96-105
    factor->_slam = this;
    factor->initialize();
    factor->push(_g);
    _num_factors++;
  }
  void remove_node(Node* node) { detail::check(pps_remove_node(_g, node->_id), _g, "pps_remove_node"); node->_id = -1; node->_slam = nullptr; _num_nodes--; }
  void remove_factor(Factor* factor) { detail::check(pps_remove_factor(_g, factor->_id), _g, "pps_remove_factor"); factor->_id = -1; _num_factors--; }
  void update() { detail::check(pps_update(_g), _g, "pps_update"); }                         // Slam.cpp:157-196
  int batch_optimization() { int it = 0; detail::check(pps_batch_optimize(_g, &it), _g, "pps_batch_optimize"); return it; }   // :198-210
  double chi2() { double c = 0; detail::check(pps_chi2(_g, &c), _g, "pps_chi2"); return c; }   // :266-268
  int num_nodes() const { int n = 0; pps_num_nodes(_g, &n); return n; }
  int num_factors() const { int n = 0; pps_num_factors(_g, &n); return n; }
  // Slam::save (Slam.cpp:84-89): factors then nodes, the reference's text format (6 significant digits)
  void save(const std::string fname) const { detail::check(pps_graph_save(_g, fname.c_str(), 0), _g, "pps_graph_save"); }
};

// ---- out-of-line members that need Slam ---------------------------------------------------------
inline void Pose3d_Node::init(const Pose3d& p) { _v = p; _init = true; if (_slam) push(); }
inline void Pose3d_Node::push() {
  double tq[7]; _v.to_tq(tq);
  pps_graph* g = _slam->handle();
  if (_id < 0) detail::check(pps_add_pose(g, tq, &_id), g, "pps_add_pose");
  else detail::check(pps_set_pose(g, _id, tq), g, "pps_set_pose");
}
inline Pose3d Pose3d_Node::value() const {
  if (_slam && _id >= 0) { double tq[7]; detail::check(pps_get_pose(_slam->handle(), _id, tq), _slam->handle(), "pps_get_pose"); return Pose3d::from_tq(tq); }
  return _v;
}
inline void Plane3d_Node::init(const Plane3d& p) { _v = p; _init = true; if (_slam) push(); }
inline void Plane3d_Node::push() {
  const Vector4d v = _v.vector();
  pps_graph* g = _slam->handle();
  if (_id < 0) detail::check(pps_add_plane(g, v.data(), &_id), g, "pps_add_plane");
  else detail::check(pps_set_plane(g, _id, v.data()), g, "pps_set_plane");
}
inline Plane3d Plane3d_Node::value() const {
  if (_slam && _id >= 0) { Vector4d v; detail::check(pps_get_plane(_slam->handle(), _id, v.data()), _slam->handle(), "pps_get_plane"); return Plane3d(v); }
  return _v;
}
inline void Pose3d_Plane3d_Factor::set_measurement(const Plane3d& m) {
  _measure = m;
  if (_slam && _id >= 0) { const Vector4d v = m.vector(); detail::check(pps_set_measurement(_slam->handle(), _id, v.data()), _slam->handle(), "pps_set_measurement"); }
}

}  // namespace isam
