/*
 * pps_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain C99 / fp64 / single-thread restatement of the reference algorithm for
 * the plane-SLAM graph solve.  See pps_oracle.h for scope and the
 * "PARITY UNPINNED" note.  Every function cites the reference file:line it
 * follows (paths relative to /root/reference/pop_planar_slam unless stated;
 * "isam/" = Thirdparty/isam/include/isam, "isamlib/" = Thirdparty/isam/isamlib).
 *
 * The sparse direct solver stands in for SuiteSparse CHOLMOD (un-vendored,
 * unpinned libsuitesparse-dev: install_dependenices.sh:11).  It reproduces
 * CHOLMOD's *result* -- delta = (J'J + lambda*diag(J'J))^-1 J'b through a
 * fill-reduced simplicial LL' (isamlib/Cholesky.cpp:68-147) -- not its pivot
 * order.  Algorithms restated from their published descriptions: minimum
 * degree ordering with dense-row removal (Amestoy/Davis/Duff AMD, dense
 * threshold 10*sqrt(n)), elimination tree + up-looking Cholesky (Davis,
 * "Direct Methods for Sparse Linear Systems", ch. 4).
 */
#include "pps_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define PI_ 3.14159265358979323846
#define TWOPI_ (2.0 * PI_)

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------------- */
/* geometry: quaternions stored (x,y,z,w)                                     */
/* ------------------------------------------------------------------------- */

/* isam/util.h:101-108 */
static double standardRad(double t) {
  if (t >= 0.) t = fmod(t + PI_, TWOPI_) - PI_;
  else         t = fmod(t - PI_, -TWOPI_) + PI_;
  return t;
}

/* Eigen quaternion product (Hamilton) a*b */
static void quat_mul(const double a[4], const double b[4], double o[4]) {
  double ax = a[0], ay = a[1], az = a[2], aw = a[3];
  double bx = b[0], by = b[1], bz = b[2], bw = b[3];
  o[3] = aw * bw - ax * bx - ay * by - az * bz;
  o[0] = aw * bx + ax * bw + ay * bz - az * by;
  o[1] = aw * by + ay * bw + az * bx - ax * bz;
  o[2] = aw * bz + az * bw + ax * by - ay * bx;
}

/* Eigen::Matrix3d(quat) = toRotationMatrix (isam/Rot3d.h:96-98); R row-major */
static void quat_to_R(const double q[4], double R[9]) {
  double x = q[0], y = q[1], z = q[2], w = q[3];
  double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  double twx = tx * w, twy = ty * w, twz = tz * w;
  double txx = tx * x, txy = ty * x, txz = tz * x;
  double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

/* Eigen::Quaterniond(Matrix3d) (isam/Rot3d.h:92-94) */
static void R_to_quat(const double R[9], double q[4]) {
  double t = R[0] + R[4] + R[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[7] - R[5]) * t;
    q[1] = (R[2] - R[6]) * t;
    q[2] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 3 + i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t;
    q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
  }
}

/* isam/Rot3d.h:100-112 */
static void euler_to_quat(double yaw, double pitch, double roll, double q[4]) {
  double sy = sin(yaw * 0.5), cy = cos(yaw * 0.5);
  double sp = sin(pitch * 0.5), cp = cos(pitch * 0.5);
  double sr = sin(roll * 0.5), cr = cos(roll * 0.5);
  q[3] = cr * cp * cy + sr * sp * sy;
  q[0] = sr * cp * cy - cr * sp * sy;
  q[1] = cr * sp * cy + sr * cp * sy;
  q[2] = cr * cp * sy - sr * sp * cy;
}

/* isam/Rot3d.h:114-124 */
static void quat_to_euler(const double q[4], double* yaw, double* pitch, double* roll) {
  const double q0 = q[3], q1 = q[0], q2 = q[1], q3 = q[2];
  *roll = atan2(2.0 * (q0 * q1 + q2 * q3), q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3);
  *pitch = asin(2.0 * (q0 * q2 - q3 * q1));
  *yaw = atan2(2.0 * (q0 * q3 + q1 * q2), q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3);
}

/* isam/Rot3d.h:126-136 */
static void rot_delta3_to_quat(const double d[3], double q[4]) {
  double theta = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  double S;
  if (theta < 0.0001) S = 0.5 + theta * theta / 48.;
  else S = sin(0.5 * theta) / theta;
  double C = cos(0.5 * theta);
  q[3] = C; q[0] = S * d[0]; q[1] = S * d[1]; q[2] = S * d[2];
}

/* boost::math::sinc_pi (boost 1.5x, sinc.hpp): Taylor below eps^(1/4) */
static double sinc_pi(double x) {
  const double eps = 2.220446049250313e-16;
  const double t2 = sqrt(eps), tn = sqrt(t2);
  if (fabs(x) >= tn) return sin(x) / x;
  double r = 1.0;
  if (fabs(x) >= eps) {
    double x2 = x * x;
    r -= x2 / 6.0;
    if (fabs(x) >= t2) r += (x2 * x2) / 120.0;
  }
  return r;
}

/* src/isam_plane3d.h:78-93 */
static void plane_delta3_to_quat(const double d[3], double q[4]) {
  double theta = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  double S = 0.5 * sinc_pi(0.5 * theta);
  double C = cos(0.5 * theta);
  q[3] = C; q[0] = S * d[0]; q[1] = S * d[1]; q[2] = S * d[2];
}

static void normalize4(double v[4]) {
  double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
  v[0] /= n; v[1] /= n; v[2] /= n; v[3] /= n;
}

/* Pose3d = Point3d + Rot3d with the mutable ypr cache of isam/Rot3d.h:153-173 */
typedef struct {
  double t[3];
  double q[4];
  int ypr_cached;
  double ypr[3]; /* yaw, pitch, roll */
} pose_t;

typedef struct { double p[4]; } plane_t;

static void pose_ensure_ypr(pose_t* p) { /* isam/Rot3d.h:167-172 */
  if (!p->ypr_cached) {
    quat_to_euler(p->q, &p->ypr[0], &p->ypr[1], &p->ypr[2]);
    p->ypr_cached = 1;
  }
}

/* isam/Pose3d.h:138-145 */
static void pose_vector(pose_t* p, double v[6]) {
  pose_ensure_ypr(p);
  v[0] = p->t[0]; v[1] = p->t[1]; v[2] = p->t[2];
  v[3] = p->ypr[0]; v[4] = p->ypr[1]; v[5] = p->ypr[2];
}

/* isam/Pose3d.h:152-155 + Rot3d::set (Rot3d.h:218-225) */
static void pose_set_vector(pose_t* p, const double v[6]) {
  p->t[0] = v[0]; p->t[1] = v[1]; p->t[2] = v[2];
  p->ypr[0] = standardRad(v[3]); p->ypr[1] = standardRad(v[4]); p->ypr[2] = standardRad(v[5]);
  p->ypr_cached = 1;
  euler_to_quat(p->ypr[0], p->ypr[1], p->ypr[2], p->q);
}

/* isam/Pose3d.h:131-136, Point3d.h:63-69, Rot3d.h:229-233 */
static void pose_exmap(const pose_t* p, const double d[6], pose_t* out) {
  double dq[4], q[4];
  rot_delta3_to_quat(d + 3, dq);
  quat_mul(p->q, dq, q);
  out->t[0] = p->t[0] + d[0]; out->t[1] = p->t[1] + d[1]; out->t[2] = p->t[2] + d[2];
  memcpy(out->q, q, sizeof q);
  out->ypr_cached = 0;
  out->ypr[0] = out->ypr[1] = out->ypr[2] = 0;
}

/* Pose3d(Matrix4d) (isam/Pose3d.h:90-104): quaternion from the rotation block */
static void pose_from_Rt(const double R[9], const double t[3], pose_t* out) {
  out->t[0] = t[0]; out->t[1] = t[1]; out->t[2] = t[2];
  R_to_quat(R, out->q);
  out->ypr_cached = 0;
  out->ypr[0] = out->ypr[1] = out->ypr[2] = 0;
}

/* a.ominus(b) = Pose3d(b.oTw() * a.wTo())  (isam/Pose3d.h:233-235, 203-213) */
static void pose_ominus(const pose_t* a, const pose_t* b, pose_t* out) {
  double Ra[9], Rb[9], R[9], C[3], t[3];
  quat_to_R(a->q, Ra);
  quat_to_R(b->q, Rb);
  /* oRw_b = Rb^T ; C = -oRw_b * tb */
  for (int i = 0; i < 3; i++)
    C[i] = -(Rb[0 * 3 + i] * b->t[0] + Rb[1 * 3 + i] * b->t[1] + Rb[2 * 3 + i] * b->t[2]);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++)
      R[i * 3 + j] = Rb[0 * 3 + i] * Ra[0 * 3 + j] + Rb[1 * 3 + i] * Ra[1 * 3 + j] + Rb[2 * 3 + i] * Ra[2 * 3 + j];
    t[i] = Rb[0 * 3 + i] * a->t[0] + Rb[1 * 3 + i] * a->t[1] + Rb[2 * 3 + i] * a->t[2] + C[i];
  }
  pose_from_Rt(R, t, out);
}

/* a.oplus(d) = Pose3d(a.wTo() * d.wTo())  (isam/Pose3d.h:222-224) */
static void pose_oplus(const pose_t* a, const pose_t* d, pose_t* out) {
  double Ra[9], Rd[9], R[9], t[3];
  quat_to_R(a->q, Ra);
  quat_to_R(d->q, Rd);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++)
      R[i * 3 + j] = Ra[i * 3 + 0] * Rd[0 * 3 + j] + Ra[i * 3 + 1] * Rd[1 * 3 + j] + Ra[i * 3 + 2] * Rd[2 * 3 + j];
    t[i] = Ra[i * 3 + 0] * d->t[0] + Ra[i * 3 + 1] * d->t[1] + Ra[i * 3 + 2] * d->t[2] + a->t[i];
  }
  pose_from_Rt(R, t, out);
}

/* Plane3d::transform_to(wTo) = normalise(wTo^T * pi)  (src/isam_plane3d.h:180-182) */
static void plane_transform_to(const plane_t* pl, const pose_t* pose, plane_t* out) {
  double R[9];
  quat_to_R(pose->q, R);
  const double* v = pl->p;
  for (int k = 0; k < 3; k++) out->p[k] = R[0 * 3 + k] * v[0] + R[1 * 3 + k] * v[1] + R[2 * 3 + k] * v[2];
  out->p[3] = pose->t[0] * v[0] + pose->t[1] * v[1] + pose->t[2] * v[2] + v[3];
  normalize4(out->p);
}

/* Plane3d::transform_from(oTw) = normalise(oTw^T * pi)  (src/isam_plane3d.h:186-188) */
static void plane_transform_from(const plane_t* pl, const pose_t* pose, plane_t* out) {
  double R[9], C[3];
  quat_to_R(pose->q, R);
  for (int i = 0; i < 3; i++)
    C[i] = -(R[0 * 3 + i] * pose->t[0] + R[1 * 3 + i] * pose->t[1] + R[2 * 3 + i] * pose->t[2]);
  const double* v = pl->p;
  /* oTw^T = [R 0; C^T 1] */
  for (int k = 0; k < 3; k++) out->p[k] = R[k * 3 + 0] * v[0] + R[k * 3 + 1] * v[1] + R[k * 3 + 2] * v[2];
  out->p[3] = C[0] * v[0] + C[1] * v[1] + C[2] * v[2] + v[3];
  normalize4(out->p);
}

/* Plane3d::exmap_3dof (src/isam_plane3d.h:101-127), plane_type == -1 */
static void plane_exmap(const plane_t* pl, const double d[3], plane_t* out) {
  double dq[4], q[4];
  plane_delta3_to_quat(d, dq);
  quat_mul(dq, pl->p, q);
  memcpy(out->p, q, sizeof q);
  normalize4(out->p);
}

/* Log(q * conj(qm)) via Eigen::AngleAxisd (src/isam_plane3d.h:286-294).
 * Eigen>=3.3 form: angle = 2 atan2(|v|, |w|), axis = v / (+-|v|), zero if |v|==0. */
static void quat_logmap_diff(const double q[4], const double qm[4], double e[3]) {
  double c[4] = {-qm[0], -qm[1], -qm[2], qm[3]};
  double dq[4];
  quat_mul(q, c, dq);
  double n = sqrt(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2]);
  if (n != 0.0) {
    double angle = 2.0 * atan2(n, fabs(dq[3]));
    if (dq[3] < 0) n = -n;
    if (angle > PI_) angle -= TWOPI_;
    if (angle < -PI_) angle += TWOPI_;
    e[0] = dq[0] / n * angle; e[1] = dq[1] / n * angle; e[2] = dq[2] / n * angle;
  } else {
    e[0] = e[1] = e[2] = 0.0; /* axis (1,0,0) * angle 0 */
  }
}

/* ------------------------------------------------------------------------- */
/* graph                                                                      */
/* ------------------------------------------------------------------------- */
enum { NODE_POSE = 0, NODE_PLANE = 1 };
enum { F_POSE_PRIOR = 0, F_ODOMETRY = 1, F_PLANE_OBS = 2, F_PLANE_PRIOR = 3 };

typedef struct {
  int type, dim, deleted, start;
  pose_t pose, pose0;     /* estimate / linearisation point (isam/Node.h:104-105) */
  plane_t plane, plane0;
} node_t;

typedef struct {
  int type, dim, nn, deleted;
  int n[2];
  double meas[6];       /* pose: x,y,z,yaw,pitch,roll ; plane: a,b,c,d (normalised) */
  double sqrtinf[36];   /* dim x dim row-major upper triangular */
  int repop;            /* Pose3d_Plane3d_Factor2: measurement re-popped from `ray` at every evaluation */
  double ray[6];        /* ground_edge_ray, column-major 3x2 (src/isam_plane3d.h:318,361-373) */
} factor_t;

typedef struct { double lambda, chi2; int accepted; } trace_t;

struct ora_graph {
  ora_props prop;
  node_t* nodes; int nnodes, cap_nodes;
  factor_t* factors; int nfactors, cap_factors;
  int dim_nodes, dim_measure;
  /* linear system workspace */
  int J_rows, J_cols;
  int* Jp; int* Ji; double* Jx; double* Jrhs;     /* CSR of J (== CSC of J^T, Cholesky.cpp:171-194) */
  long Jcap; int Jrowcap;
  /* cached ordering */
  int* perm; int perm_n; int perm_valid; int topo_version, perm_version;
  /* trace + timers */
  trace_t* trace; int ntrace, cap_trace;
  double chi2_init;
  double tim[4];
  long nnzL;
};

void ora_default_props(ora_props* p) {
  /* Properties.h:86-109 x Mapping.cpp:33-39 */
  p->epsilon2 = 1e-2 * 0.1;
  p->epsilon_abs = 1e-3 * 0.1;
  p->epsilon_rel = 1e-5 * 0.1;
  p->max_iterations = 500;
  p->lm_lambda0 = 1e-6;
  p->lm_lambda_factor = 10.;
  p->analytic = 0;
  p->cache_ordering = 0;
  p->threads = 1;
}

ora_graph* ora_create(const ora_props* p) {
  ora_graph* g = (ora_graph*)calloc(1, sizeof(ora_graph));
  if (p) g->prop = *p; else ora_default_props(&g->prop);
  return g;
}

void ora_destroy(ora_graph* g) {
  if (!g) return;
  free(g->nodes); free(g->factors);
  free(g->Jp); free(g->Ji); free(g->Jx); free(g->Jrhs);
  free(g->perm); free(g->trace);
  free(g);
}

static node_t* new_node(ora_graph* g) {
  if (g->nnodes == g->cap_nodes) {
    g->cap_nodes = g->cap_nodes ? 2 * g->cap_nodes : 256;
    g->nodes = (node_t*)realloc(g->nodes, sizeof(node_t) * (size_t)g->cap_nodes);
  }
  node_t* n = &g->nodes[g->nnodes++];
  memset(n, 0, sizeof *n);
  g->topo_version++;
  return n;
}

static void pose_from_tq(const double tq[7], pose_t* p) {
  p->t[0] = tq[0]; p->t[1] = tq[1]; p->t[2] = tq[2];
  p->q[0] = tq[3]; p->q[1] = tq[4]; p->q[2] = tq[5]; p->q[3] = tq[6];
  p->ypr_cached = 0; p->ypr[0] = p->ypr[1] = p->ypr[2] = 0;
}
static void pose_to_tq(const pose_t* p, double tq[7]) {
  tq[0] = p->t[0]; tq[1] = p->t[1]; tq[2] = p->t[2];
  tq[3] = p->q[0]; tq[4] = p->q[1]; tq[5] = p->q[2]; tq[6] = p->q[3];
}

int ora_add_pose(ora_graph* g, const double tq[7]) {
  node_t* n = new_node(g);
  n->type = NODE_POSE; n->dim = 6;
  pose_from_tq(tq, &n->pose);
  n->pose0 = n->pose;
  g->dim_nodes += 6;
  return g->nnodes - 1;
}

int ora_add_plane(ora_graph* g, const double abcd[4]) {
  node_t* n = new_node(g);
  n->type = NODE_PLANE; n->dim = 3;
  memcpy(n->plane.p, abcd, 4 * sizeof(double));
  normalize4(n->plane.p); /* Plane3d(Vector4d), src/isam_plane3d.h:59-66 */
  n->plane0 = n->plane;
  g->dim_nodes += 3;
  return g->nnodes - 1;
}

static factor_t* new_factor(ora_graph* g, int type, int dim, const double* ut) {
  if (g->nfactors == g->cap_factors) {
    g->cap_factors = g->cap_factors ? 2 * g->cap_factors : 1024;
    g->factors = (factor_t*)realloc(g->factors, sizeof(factor_t) * (size_t)g->cap_factors);
  }
  factor_t* f = &g->factors[g->nfactors++];
  memset(f, 0, sizeof *f);
  f->type = type; f->dim = dim;
  int k = 0;
  for (int r = 0; r < dim; r++)
    for (int c = r; c < dim; c++) f->sqrtinf[r * dim + c] = ut[k++];
  g->dim_measure += dim;
  g->topo_version++;
  return f;
}

int ora_add_pose_prior(ora_graph* g, int pose, const double meas6[6], const double ut[21]) {
  factor_t* f = new_factor(g, F_POSE_PRIOR, 6, ut);
  f->nn = 1; f->n[0] = pose; f->n[1] = -1;
  memcpy(f->meas, meas6, 6 * sizeof(double));
  return g->nfactors - 1;
}
int ora_add_odometry(ora_graph* g, int p1, int p2, const double meas6[6], const double ut[21]) {
  factor_t* f = new_factor(g, F_ODOMETRY, 6, ut);
  f->nn = 2; f->n[0] = p1; f->n[1] = p2;
  memcpy(f->meas, meas6, 6 * sizeof(double));
  return g->nfactors - 1;
}
int ora_add_plane_obs(ora_graph* g, int pose, int plane, const double meas4[4], const double ut[6]) {
  factor_t* f = new_factor(g, F_PLANE_OBS, 3, ut);
  f->nn = 2; f->n[0] = pose; f->n[1] = plane;
  memcpy(f->meas, meas4, 4 * sizeof(double));
  normalize4(f->meas);
  return g->nfactors - 1;
}
/* Pose3d_Plane3d_Factor2 (src/isam_plane3d.h:314-424) */
int ora_add_plane_obs2(ora_graph* g, int pose, int plane, const double meas4[4], const double ray6[6], const double ut[6]) {
  int id = ora_add_plane_obs(g, pose, plane, meas4, ut);
  g->factors[id].repop = 1;
  memcpy(g->factors[id].ray, ray6, 6 * sizeof(double));
  return id;
}
/* precompute_edge_ray (src/isam_plane3d.h:361-373): fp32 invK * (u,v,1), cast to double */
void ora_edge_ray(const float invK[9], const float seg2d[4], double ray6[6]) {
  for (int e = 0; e < 2; e++) {
    const float u = seg2d[2 * e], v = seg2d[2 * e + 1];
    for (int i = 0; i < 3; i++) ray6[3 * e + i] = (double)(invK[i * 3 + 0] * u + invK[i * 3 + 1] * v + invK[i * 3 + 2] * 1.f);
  }
}
/* isam::get_wall_plane_equation for one segment (src/isam_plane3d.cpp:20-55) + normalisation (isam_plane3d.h:392-393) */
static void repop_wall_plane(const pose_t* pose, const double ray[6], double out[4]) {
  double R[9], gs[4], P[2][3];
  quat_to_R(pose->q, R);
  /* ground_plane_sensor = transToWorld^T * (0,0,-1,0) */
  for (int k = 0; k < 3; k++) gs[k] = R[0 * 3 + k] * 0.0 + R[1 * 3 + k] * 0.0 + R[2 * 3 + k] * -1.0 + 0.0 * 0.0;
  gs[3] = pose->t[0] * 0.0 + pose->t[1] * 0.0 + pose->t[2] * -1.0 + 1.0 * 0.0;
  for (int j = 0; j < 2; j++) {
    const double* r = ray + 3 * j;
    const double frac = -gs[3] / (gs[0] * r[0] + gs[1] * r[1] + gs[2] * r[2]);   /* ray_plane_interact :13-17 */
    for (int k = 0; k < 3; k++) P[j][k] = frac * r[k];
  }
  const double t1[3] = {P[1][0] - P[0][0], P[1][1] - P[0][1], P[1][2] - P[0][2]};
  const double n[3] = {t1[1] * gs[2] - t1[2] * gs[1], t1[2] * gs[0] - t1[0] * gs[2], t1[0] * gs[1] - t1[1] * gs[0]};
  out[0] = n[0]; out[1] = n[1]; out[2] = n[2];
  out[3] = -(n[0] * P[0][0] + n[1] * P[0][1] + n[2] * P[0][2]);
  normalize4(out);
}
void ora_repop_wall_plane(const double tq[7], const double ray6[6], double out4[4]) {
  pose_t p; pose_from_tq(tq, &p); repop_wall_plane(&p, ray6, out4);
}
int ora_add_plane_prior(ora_graph* g, int plane, const double meas4[4], const double ut[6]) {
  factor_t* f = new_factor(g, F_PLANE_PRIOR, 3, ut);
  f->nn = 1; f->n[0] = plane; f->n[1] = -1;
  memcpy(f->meas, meas4, 4 * sizeof(double));
  normalize4(f->meas);
  return g->nfactors - 1;
}
void ora_set_measurement(ora_graph* g, int fid, const double meas4[4]) {
  factor_t* f = &g->factors[fid];
  memcpy(f->meas, meas4, 4 * sizeof(double));
  normalize4(f->meas);
}
/* Slam::remove_factor / remove_node, isamlib/Slam.cpp:107-126 */
void ora_remove_factor(ora_graph* g, int fid) {
  factor_t* f = &g->factors[fid];
  if (f->deleted) return;
  f->deleted = 1;
  g->dim_measure -= f->dim;
  g->topo_version++;
}
void ora_remove_node(ora_graph* g, int nid) {
  node_t* n = &g->nodes[nid];
  if (n->deleted) return;
  for (int i = 0; i < g->nfactors; i++) {
    factor_t* f = &g->factors[i];
    if (!f->deleted && (f->n[0] == nid || (f->nn > 1 && f->n[1] == nid))) ora_remove_factor(g, i);
  }
  n->deleted = 1;
  g->dim_nodes -= n->dim;
  g->topo_version++;
}

int ora_num_nodes(const ora_graph* g) { return g->nnodes; }
int ora_num_factors(const ora_graph* g) { return g->nfactors; }
int ora_node_dim(const ora_graph* g, int nid) { return g->nodes[nid].dim; }
void ora_get_pose(const ora_graph* g, int nid, double tq[7]) { pose_to_tq(&g->nodes[nid].pose, tq); }
void ora_get_plane(const ora_graph* g, int nid, double abcd[4]) { memcpy(abcd, g->nodes[nid].plane.p, 32); }
void ora_set_pose(ora_graph* g, int nid, const double tq[7]) {
  pose_from_tq(tq, &g->nodes[nid].pose);
  g->nodes[nid].pose0 = g->nodes[nid].pose;
}
void ora_set_plane(ora_graph* g, int nid, const double abcd[4]) {
  memcpy(g->nodes[nid].plane.p, abcd, 32);
  normalize4(g->nodes[nid].plane.p);
  g->nodes[nid].plane0 = g->nodes[nid].plane;
}

/* ------------------------------------------------------------------------- */
/* residuals                                                                  */
/* ------------------------------------------------------------------------- */
#define SEL_LIN 0
#define SEL_EST 1

static pose_t* node_pose(node_t* n, int sel) { return sel == SEL_EST ? &n->pose : &n->pose0; }
static plane_t* node_plane(node_t* n, int sel) { return sel == SEL_EST ? &n->plane : &n->plane0; }

/* basic_error of the four factor types */
static void basic_error(ora_graph* g, const factor_t* f, int sel, double* e) {
  switch (f->type) {
    case F_PLANE_OBS: { /* src/isam_plane3d.h:271-304 */
      pose_t pose = *node_pose(&g->nodes[f->n[0]], sel); /* copy, like Node::value(s) */
      const plane_t* gp = node_plane(&g->nodes[f->n[1]], sel);
      plane_t local;
      plane_transform_to(gp, &pose, &local);
      if (f->repop) {   /* Factor2: src/isam_plane3d.h:390-394 */
        double ms[4];
        repop_wall_plane(&pose, f->ray, ms);
        quat_logmap_diff(local.p, ms, e);
      } else
        quat_logmap_diff(local.p, f->meas, e);
      break;
    }
    case F_PLANE_PRIOR: { /* src/isam_plane3d.h:449-473 */
      const plane_t* pl = node_plane(&g->nodes[f->n[0]], sel);
      quat_logmap_diff(pl->p, f->meas, e);
      break;
    }
    case F_POSE_PRIOR: { /* isam/slam3d.h:82-88: vector() on the node's own object (fills its ypr cache) */
      double v[6];
      pose_vector(node_pose(&g->nodes[f->n[0]], sel), v);
      for (int i = 0; i < 6; i++) e[i] = v[i] - f->meas[i];
      e[3] = standardRad(e[3]); e[4] = standardRad(e[4]); e[5] = standardRad(e[5]);
      break;
    }
    case F_ODOMETRY: { /* isam/slam3d.h:174-191 */
      const pose_t* p1 = node_pose(&g->nodes[f->n[0]], sel);
      const pose_t* p2 = node_pose(&g->nodes[f->n[1]], sel);
      pose_t pred;
      double v[6];
      pose_ominus(p2, p1, &pred);
      pose_vector(&pred, v);
      for (int i = 0; i < 6; i++) e[i] = v[i] - f->meas[i];
      e[3] = standardRad(e[3]); e[4] = standardRad(e[4]); e[5] = standardRad(e[5]);
      break;
    }
  }
}

/* Factor::error = sqrtinf * basic_error  (isam/Factor.h:67-77; no robust cost: Slam.cpp:73) */
static void factor_error(ora_graph* g, const factor_t* f, int sel, double* r) {
  double e[6];
  basic_error(g, f, sel, e);
  int m = f->dim;
  for (int i = 0; i < m; i++) {
    double s = 0;
    for (int j = 0; j < m; j++) s += f->sqrtinf[i * m + j] * e[j];
    r[i] = s;
  }
}

/* numericalDiff (isamlib/numericalDiff.cpp:41-87): symmetric differences, eps = 1e-4,
 * perturb through self_exmap on the linearisation point, restore through
 * update0(vector0()) -- which for poses round-trips the quaternion through the
 * (cached) Euler angles (isam/Pose3d.h:152-155).  H is dim x ncols row-major. */
static void numerical_jacobian(ora_graph* g, const factor_t* f, double* H, int ncols) {
  const double epsilon = 0.0001;
  int m = f->dim, col = 0;
  for (int k = 0; k < f->nn; k++) {
    node_t* node = &g->nodes[f->n[k]];
    int dn = node->dim;
    for (int j = 0; j < dn; j++, col++) {
      double delta[6] = {0, 0, 0, 0, 0, 0};
      double yp[6], ym[6];
      if (node->type == NODE_POSE) {
        double original[6];
        pose_t tmp;
        pose_vector(&node->pose0, original);           /* node->vector0() */
        delta[j] = epsilon;
        pose_exmap(&node->pose0, delta, &tmp); node->pose0 = tmp;   /* self_exmap */
        factor_error(g, f, SEL_LIN, yp);
        pose_set_vector(&node->pose0, original);       /* update0(original) */
        delta[j] = -epsilon;
        pose_exmap(&node->pose0, delta, &tmp); node->pose0 = tmp;
        factor_error(g, f, SEL_LIN, ym);
        pose_set_vector(&node->pose0, original);
      } else {
        double original[4];
        plane_t tmp;
        memcpy(original, node->plane0.p, 32);           /* vector0() */
        delta[j] = epsilon;
        plane_exmap(&node->plane0, delta, &tmp); node->plane0 = tmp;
        factor_error(g, f, SEL_LIN, yp);
        memcpy(node->plane0.p, original, 32); normalize4(node->plane0.p);  /* set(): src/isam_plane3d.h:142-145 */
        delta[j] = -epsilon;
        plane_exmap(&node->plane0, delta, &tmp); node->plane0 = tmp;
        factor_error(g, f, SEL_LIN, ym);
        memcpy(node->plane0.p, original, 32); normalize4(node->plane0.p);
      }
      for (int r = 0; r < m; r++) H[r * ncols + col] = (yp[r] - ym[r]) / (epsilon + epsilon);
    }
  }
}

/* ---- analytic Jacobians (the "optimised CPU" variant; not in the reference,
 * which leaves Factor::jacobian() numerical: src/isam_plane3d.h:306-307) ---- */

/* d e / d dq for e = Log(dq), dq=(v,w) unit: 3x4, columns (x,y,z,w) */
static void dlog_dq(const double dq[4], double D[12]) {
  double x = dq[0], y = dq[1], z = dq[2], w = dq[3];
  double s2 = x * x + y * y + z * z, s = sqrt(s2);
  double nn = s2 + w * w;
  double a, b; /* e = a*v ; a = phi/s */
  if (s < 1e-12) {
    a = 2.0 / w; b = 0.0;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) D[i * 4 + j] = (i == j) ? a : 0.0;
    (void)b;
    return;
  }
  double sg = (w < 0) ? -1.0 : 1.0;
  double phi = 2.0 * atan2(s, fabs(w)) * sg;
  a = phi / s;
  double dphids = 2.0 * w / nn;      /* d phi / d s */
  double dphidw = -2.0 * s / nn;     /* d phi / d w */
  double v[3] = {x, y, z};
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) {
      double vv = v[i] * v[j] / s2;
      D[i * 4 + j] = a * ((i == j ? 1.0 : 0.0) - vv) + dphids * vv;
    }
    D[i * 4 + 3] = dphidw * v[i] / s;
  }
}

/* d(q (x) conj(qm)) / dq : 4x4 (rows/cols x,y,z,w) */
static void dqmulconj_dq(const double qm[4], double M[16]) {
  double cx = -qm[0], cy = -qm[1], cz = -qm[2], cw = qm[3];
  /* o = a*c with a=q: o.x = aw cx + ax cw + ay cz - az cy ... */
  double m[16] = {
      cw,  cz, -cy, cx,
     -cz,  cw,  cx, cy,
      cy, -cx,  cw, cz,
     -cx, -cy, -cz, cw};
  memcpy(M, m, sizeof m);
}

/* normalisation Jacobian d(u/|u|)/du = (I - pp^T)/|u| */
static void dnormalize4(const double u[4], double N[16]) {
  double n = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2] + u[3] * u[3]);
  double p[4] = {u[0] / n, u[1] / n, u[2] / n, u[3] / n};
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) N[i * 4 + j] = ((i == j ? 1.0 : 0.0) - p[i] * p[j]) / n;
}

static void matmul(const double* A, const double* B, double* C, int m, int k, int n) {
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) {
      double s = 0;
      for (int l = 0; l < k; l++) s += A[i * k + l] * B[l * n + j];
      C[i * n + j] = s;
    }
}

/* d plane(4) / d delta(3) for q' = Exp(delta) (x) q, at delta=0, followed by normalisation (identity on the
 * tangent because q is unit): rows x,y,z,w */
static void dplane_ddelta(const double p[4], double D[12]) {
  double a = p[0], b = p[1], c = p[2], d = p[3];
  /* dv = 0.5 (w delta + delta x v), dw = -0.5 v.delta */
  double m[12] = {
      0.5 * d,  0.5 * c, -0.5 * b,
     -0.5 * c,  0.5 * d,  0.5 * a,
      0.5 * b, -0.5 * a,  0.5 * d,
     -0.5 * a, -0.5 * b, -0.5 * c};
  memcpy(D, m, sizeof m);
}

/* Euler-rate matrix: d(yaw,pitch,roll)/d(omega) for a right (body) perturbation R Exp(omega) */
static void dypr_domega(double pitch, double roll, double E[9]) {
  double sr = sin(roll), cr = cos(roll), cp = cos(pitch), tp = tan(pitch);
  double m[9] = {
      0.0, sr / cp, cr / cp,   /* yaw   */
      0.0, cr,      -sr,       /* pitch */
      1.0, sr * tp, cr * tp};  /* roll  */
  memcpy(E, m, sizeof m);
}

/* unwhitened analytic Jacobian He (dim x ncols) */
static void analytic_basic_jacobian(ora_graph* g, const factor_t* f, double* He, int ncols) {
  memset(He, 0, sizeof(double) * (size_t)(f->dim * ncols));
  switch (f->type) {
    case F_PLANE_OBS: {
      const pose_t* pose = &g->nodes[f->n[0]].pose0;
      const plane_t* pl = &g->nodes[f->n[1]].plane0;
      double R[9]; quat_to_R(pose->q, R);
      const double* v = pl->p;
      double u[4];
      for (int k = 0; k < 3; k++) u[k] = R[0 * 3 + k] * v[0] + R[1 * 3 + k] * v[1] + R[2 * 3 + k] * v[2];
      u[3] = pose->t[0] * v[0] + pose->t[1] * v[1] + pose->t[2] * v[2] + v[3];
      double N[16]; dnormalize4(u, N);
      double pl_l[4] = {u[0], u[1], u[2], u[3]}; normalize4(pl_l);
      double c[4] = {-f->meas[0], -f->meas[1], -f->meas[2], f->meas[3]}, dq[4];
      quat_mul(pl_l, c, dq);
      double DL[12]; dlog_dq(dq, DL);
      double Q[16]; dqmulconj_dq(f->meas, Q);
      double A[12], B[12];
      matmul(DL, Q, A, 3, 4, 4);   /* 3x4 */
      matmul(A, N, B, 3, 4, 4);    /* d e / d u : 3x4 */
      /* du/d(pose delta): 4x6. translation: u3 += n.dt ; rotation: m=R^T n -> m + [m]x dtheta */
      double Up[24]; memset(Up, 0, sizeof Up);
      Up[3 * 6 + 0] = v[0]; Up[3 * 6 + 1] = v[1]; Up[3 * 6 + 2] = v[2];
      double m0 = u[0], m1 = u[1], m2 = u[2];
      /* [m]x = [0 -m2 m1; m2 0 -m0; -m1 m0 0] */
      Up[0 * 6 + 3] = 0;   Up[0 * 6 + 4] = -m2; Up[0 * 6 + 5] = m1;
      Up[1 * 6 + 3] = m2;  Up[1 * 6 + 4] = 0;   Up[1 * 6 + 5] = -m0;
      Up[2 * 6 + 3] = -m1; Up[2 * 6 + 4] = m0;  Up[2 * 6 + 5] = 0;
      double Jp[18]; matmul(B, Up, Jp, 3, 4, 6);
      /* du/d(plane delta) = M * dpi/ddelta, M = [R^T 0; t^T 1] */
      double Dp[12]; dplane_ddelta(pl->p, Dp);
      double M[16] = {R[0], R[3], R[6], 0, R[1], R[4], R[7], 0, R[2], R[5], R[8], 0,
                      pose->t[0], pose->t[1], pose->t[2], 1};
      double MD[12]; matmul(M, Dp, MD, 4, 4, 3);
      double Jl[9]; matmul(B, MD, Jl, 3, 4, 3);
      for (int r = 0; r < 3; r++) {
        for (int cidx = 0; cidx < 6; cidx++) He[r * ncols + cidx] = Jp[r * 6 + cidx];
        for (int cidx = 0; cidx < 3; cidx++) He[r * ncols + 6 + cidx] = Jl[r * 3 + cidx];
      }
      break;
    }
    case F_PLANE_PRIOR: {
      const plane_t* pl = &g->nodes[f->n[0]].plane0;
      double c[4] = {-f->meas[0], -f->meas[1], -f->meas[2], f->meas[3]}, dq[4];
      quat_mul(pl->p, c, dq);
      double DL[12]; dlog_dq(dq, DL);
      double Q[16]; dqmulconj_dq(f->meas, Q);
      double A[12]; matmul(DL, Q, A, 3, 4, 4);
      double Dp[12]; dplane_ddelta(pl->p, Dp);
      double Jl[9]; matmul(A, Dp, Jl, 3, 4, 3);
      for (int r = 0; r < 3; r++) for (int cidx = 0; cidx < 3; cidx++) He[r * ncols + cidx] = Jl[r * 3 + cidx];
      break;
    }
    case F_POSE_PRIOR: {
      pose_t p = g->nodes[f->n[0]].pose0;
      double v[6]; pose_vector(&p, v);
      double E[9]; dypr_domega(v[4], v[5], E);
      for (int i = 0; i < 3; i++) He[i * ncols + i] = 1.0;
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) He[(3 + i) * ncols + 3 + j] = E[i * 3 + j];
      break;
    }
    case F_ODOMETRY: {
      const pose_t* p1 = &g->nodes[f->n[0]].pose0;
      const pose_t* p2 = &g->nodes[f->n[1]].pose0;
      pose_t pred; pose_ominus(p2, p1, &pred);
      double v[6]; pose_vector(&pred, v);
      double E[9]; dypr_domega(v[4], v[5], E);
      double R1[9], R12[9];
      quat_to_R(p1->q, R1);
      quat_to_R(pred.q, R12);
      double* t12 = pred.t;
      /* translation rows */
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
          He[i * ncols + j] = -R1[j * 3 + i];       /* d t12 / d t1 = -R1^T */
          He[i * ncols + 6 + j] = R1[j * 3 + i];    /* d t12 / d t2 =  R1^T */
        }
      /* d t12 / d theta1 = [t12]x */
      He[0 * ncols + 3] = 0;        He[0 * ncols + 4] = -t12[2]; He[0 * ncols + 5] = t12[1];
      He[1 * ncols + 3] = t12[2];   He[1 * ncols + 4] = 0;       He[1 * ncols + 5] = -t12[0];
      He[2 * ncols + 3] = -t12[1];  He[2 * ncols + 4] = t12[0];  He[2 * ncols + 5] = 0;
      /* rotation rows: omega = dtheta2 - R12^T dtheta1 */
      double ER[9];
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
          double s = 0;
          for (int l = 0; l < 3; l++) s += E[i * 3 + l] * R12[j * 3 + l]; /* E * R12^T */
          ER[i * 3 + j] = s;
        }
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
          He[(3 + i) * ncols + 3 + j] = -ER[i * 3 + j];
          He[(3 + i) * ncols + 9 + j] = E[i * 3 + j];
        }
      break;
    }
  }
}

static int factor_cols(const ora_graph* g, const factor_t* f) {
  int c = 0;
  for (int k = 0; k < f->nn; k++) c += g->nodes[f->n[k]].dim;
  return c;
}

/* Factor::jacobian (isam/Factor.h:126-139): H then r = error(LINPOINT) */
static void factor_jacobian(ora_graph* g, const factor_t* f, int analytic, double* H, double* r) {
  int ncols = factor_cols(g, f), m = f->dim;
  if (!analytic || f->repop) {   /* Factor2 has no closed form here: numerical like the reference */
    numerical_jacobian(g, f, H, ncols);
  } else {
    double He[6 * 12];
    analytic_basic_jacobian(g, f, He, ncols);
    for (int i = 0; i < m; i++)
      for (int j = 0; j < ncols; j++) {
        double s = 0;
        for (int l = 0; l < m; l++) s += f->sqrtinf[i * m + l] * He[l * ncols + j];
        H[i * ncols + j] = s;
      }
  }
  factor_error(g, f, SEL_LIN, r);
}

int ora_factor_dim(const ora_graph* g, int fid) { return g->factors[fid].dim; }
int ora_factor_cols(const ora_graph* g, int fid) { return factor_cols(g, &g->factors[fid]); }
void ora_factor_error(ora_graph* g, int fid, int sel, double* r_out) { factor_error(g, &g->factors[fid], sel, r_out); }
void ora_factor_jacobian(ora_graph* g, int fid, int analytic, double* H_out, double* r_out) {
  factor_jacobian(g, &g->factors[fid], analytic, H_out, r_out);
}

/* ------------------------------------------------------------------------- */
/* Slam-level operations                                                      */
/* ------------------------------------------------------------------------- */

/* Slam::update_starts (isamlib/Slam.cpp:59-67) */
static void update_starts(ora_graph* g) {
  int start = 0;
  for (int i = 0; i < g->nnodes; i++) {
    node_t* n = &g->nodes[i];
    if (n->deleted) { n->start = -1; continue; }
    n->start = start;
    start += n->dim;
  }
}

/* Slam::jacobian_partial(-1) (isamlib/Slam.cpp:395-432): row-sparse J, rhs = -r (isam/Jacobian.h:98).
 * Zero entries are kept (Slam.cpp:425). */
static void build_jacobian(ora_graph* g) {
  update_starts(g);
  int rows = g->dim_measure;
  long nnz_cap = 0;
  for (int i = 0; i < g->nfactors; i++)
    if (!g->factors[i].deleted) nnz_cap += (long)g->factors[i].dim * factor_cols(g, &g->factors[i]);
  if (nnz_cap > g->Jcap) {
    g->Jcap = nnz_cap;
    g->Ji = (int*)realloc(g->Ji, sizeof(int) * (size_t)nnz_cap);
    g->Jx = (double*)realloc(g->Jx, sizeof(double) * (size_t)nnz_cap);
  }
  if (rows + 1 > g->Jrowcap) {
    g->Jrowcap = rows + 1;
    g->Jp = (int*)realloc(g->Jp, sizeof(int) * (size_t)(rows + 1));
    g->Jrhs = (double*)realloc(g->Jrhs, sizeof(double) * (size_t)(rows + 1));
  }
  /* row / entry offset of every factor first, so that the factors can be linearised independently */
  int* frow = (int*)malloc(sizeof(int) * (size_t)(g->nfactors + 1));
  long* fnz = (long*)malloc(sizeof(long) * (size_t)(g->nfactors + 1));
  int row = 0, any_numeric = !g->prop.analytic;
  long nz = 0;
  g->Jp[0] = 0;
  for (int i = 0; i < g->nfactors; i++) {
    factor_t* f = &g->factors[i];
    frow[i] = row; fnz[i] = nz;
    if (f->deleted) continue;
    if (f->repop) any_numeric = 1;
    row += f->dim; nz += (long)f->dim * factor_cols(g, f);
  }
  /* numericalDiff perturbs the nodes in place (like the reference): only the closed-form sweep may run in parallel */
  const int nthreads = (g->prop.threads > 1 && !any_numeric) ? g->prop.threads : 1;
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static) if (nthreads > 1)
#endif
  for (int i = 0; i < g->nfactors; i++) {
    factor_t* f = &g->factors[i];
    if (f->deleted) continue;
    double H[6 * 12], r[6];
    int ncols = factor_cols(g, f);
    int row = frow[i];
    long nz = fnz[i];
    factor_jacobian(g, f, g->prop.analytic, H, r);
    /* terms are appended in node order of the factor; SparseVector keeps indices sorted */
    int order[2] = {0, 1};
    if (f->nn == 2 && g->nodes[f->n[1]].start < g->nodes[f->n[0]].start) { order[0] = 1; order[1] = 0; }
    for (int rr = 0; rr < f->dim; rr++) {
      for (int kk = 0; kk < f->nn; kk++) {
        int k = order[kk];
        int coff = 0;
        for (int k2 = 0; k2 < k; k2++) coff += g->nodes[f->n[k2]].dim;
        node_t* nd = &g->nodes[f->n[k]];
        for (int c = 0; c < nd->dim; c++) {
          g->Ji[nz] = nd->start + c;
          g->Jx[nz] = H[rr * ncols + coff + c];
          nz++;
        }
      }
      g->Jrhs[row + rr] = -r[rr];
      g->Jp[row + rr + 1] = (int)nz;
    }
  }
  free(frow); free(fnz);
  g->J_rows = rows;
  g->J_cols = g->dim_nodes;
}

/* Slam::weighted_errors + squaredNorm (isamlib/Slam.cpp:254-268) */
static double chi2_sel(ora_graph* g, int sel) {
  double s = 0;
  for (int i = 0; i < g->nfactors; i++) {
    factor_t* f = &g->factors[i];
    if (f->deleted) continue;
    double r[6];
    factor_error(g, f, sel, r);
    for (int k = 0; k < f->dim; k++) s += r[k] * r[k];
  }
  return s;
}
double ora_chi2(ora_graph* g) { return chi2_sel(g, SEL_EST); }

/* Slam::self_exmap / apply_exmap (isamlib/Slam.cpp:216-234, isam/Node.h:141-146) */
static void self_exmap(ora_graph* g, const double* x) {
  for (int i = 0; i < g->nnodes; i++) {
    node_t* n = &g->nodes[i];
    if (n->deleted) continue;
    if (n->type == NODE_POSE) { pose_t t; pose_exmap(&n->pose0, x + n->start, &t); n->pose0 = t; }
    else { plane_t t; plane_exmap(&n->plane0, x + n->start, &t); n->plane0 = t; }
  }
}
static void apply_exmap(ora_graph* g, const double* x) {
  for (int i = 0; i < g->nnodes; i++) {
    node_t* n = &g->nodes[i];
    if (n->deleted) continue;
    if (n->type == NODE_POSE) pose_exmap(&n->pose0, x + n->start, &n->pose);
    else plane_exmap(&n->plane0, x + n->start, &n->plane);
  }
}
static void estimate_to_linpoint(ora_graph* g) {
  for (int i = 0; i < g->nnodes; i++) { g->nodes[i].pose0 = g->nodes[i].pose; g->nodes[i].plane0 = g->nodes[i].plane; }
}
static void linpoint_to_estimate(ora_graph* g) {
  for (int i = 0; i < g->nnodes; i++) { g->nodes[i].pose = g->nodes[i].pose0; g->nodes[i].plane = g->nodes[i].plane0; }
}

/* ------------------------------------------------------------------------- */
/* sparse direct solve (stand-in for CHOLMOD, isamlib/Cholesky.cpp:68-147)    */
/* ------------------------------------------------------------------------- */

typedef struct { int* a; int n, cap; } ivec;
static void ivec_push(ivec* v, int x) {
  if (v->n == v->cap) { v->cap = v->cap ? 2 * v->cap : 8; v->a = (int*)realloc(v->a, sizeof(int) * (size_t)v->cap); }
  v->a[v->n++] = x;
}
static int cmp_int(const void* a, const void* b) { int x = *(const int*)a, y = *(const int*)b; return (x > y) - (x < y); }

typedef struct { int deg, v; } heap_e;
typedef struct { heap_e* a; int n, cap; } heap_t;
static void heap_push(heap_t* h, int deg, int v) {
  if (h->n == h->cap) { h->cap = h->cap ? 2 * h->cap : 1024; h->a = (heap_e*)realloc(h->a, sizeof(heap_e) * (size_t)h->cap); }
  int i = h->n++;
  h->a[i].deg = deg; h->a[i].v = v;
  while (i > 0) {
    int p = (i - 1) / 2;
    if (h->a[p].deg < h->a[i].deg || (h->a[p].deg == h->a[i].deg && h->a[p].v < h->a[i].v)) break;
    heap_e t = h->a[p]; h->a[p] = h->a[i]; h->a[i] = t; i = p;
  }
}
static heap_e heap_pop(heap_t* h) {
  heap_e top = h->a[0];
  h->a[0] = h->a[--h->n];
  int i = 0;
  for (;;) {
    int l = 2 * i + 1, r = l + 1, m = i;
    if (l < h->n && (h->a[l].deg < h->a[m].deg || (h->a[l].deg == h->a[m].deg && h->a[l].v < h->a[m].v))) m = l;
    if (r < h->n && (h->a[r].deg < h->a[m].deg || (h->a[r].deg == h->a[m].deg && h->a[r].v < h->a[m].v))) m = r;
    if (m == i) break;
    heap_e t = h->a[m]; h->a[m] = h->a[i]; h->a[i] = t; i = m;
  }
  return top;
}

/* Block-level minimum-degree ordering with AMD's dense-node rule.
 * Output: perm over scalar columns (perm[k] = original column of pivot k). */
static void compute_ordering(ora_graph* g) {
  int nn = g->nnodes;
  ivec* adj = (ivec*)calloc((size_t)nn, sizeof(ivec));
  int nlive = 0;
  for (int i = 0; i < nn; i++) if (!g->nodes[i].deleted) nlive++;
  for (int i = 0; i < g->nfactors; i++) {
    factor_t* f = &g->factors[i];
    if (f->deleted || f->nn < 2) continue;
    ivec_push(&adj[f->n[0]], f->n[1]);
    ivec_push(&adj[f->n[1]], f->n[0]);
  }
  for (int i = 0; i < nn; i++) { /* sort + unique */
    ivec* v = &adj[i];
    if (v->n == 0) continue;
    qsort(v->a, (size_t)v->n, sizeof(int), cmp_int);
    int m = 0;
    for (int k = 0; k < v->n; k++) if (k == 0 || v->a[k] != v->a[k - 1]) v->a[m++] = v->a[k];
    v->n = m;
  }
  double dense_thr = 10.0 * sqrt((double)nlive);
  if (dense_thr < 16) dense_thr = 16;
  char* dense = (char*)calloc((size_t)nn, 1);
  char* elim = (char*)calloc((size_t)nn, 1);
  for (int i = 0; i < nn; i++) if (!g->nodes[i].deleted && adj[i].n > dense_thr) dense[i] = 1;
  /* strip dense nodes from the lists */
  for (int i = 0; i < nn; i++) {
    ivec* v = &adj[i];
    int m = 0;
    for (int k = 0; k < v->n; k++) if (!dense[v->a[k]]) v->a[m++] = v->a[k];
    v->n = m;
  }
  int* deg = (int*)calloc((size_t)nn, sizeof(int));
  heap_t heap = {0, 0, 0};
  for (int i = 0; i < nn; i++) {
    if (g->nodes[i].deleted || dense[i]) continue;
    int d = 0;
    for (int k = 0; k < adj[i].n; k++) d += g->nodes[adj[i].a[k]].dim;
    deg[i] = d;
    heap_push(&heap, d, i);
  }
  int* order = (int*)malloc(sizeof(int) * (size_t)nn);
  int no = 0;
  int* tmp = NULL; int tmpcap = 0;
  while (heap.n > 0) {
    heap_e e = heap_pop(&heap);
    int v = e.v;
    if (elim[v] || e.deg != deg[v]) continue; /* stale */
    elim[v] = 1;
    order[no++] = v;
    ivec* N = &adj[v];
    for (int a = 0; a < N->n; a++) {
      int u = N->a[a];
      ivec* U = &adj[u];
      int need = U->n + N->n;
      if (need > tmpcap) { tmpcap = 2 * need; tmp = (int*)realloc(tmp, sizeof(int) * (size_t)tmpcap); }
      int i = 0, j = 0, m = 0;
      while (i < U->n || j < N->n) { /* merge, dropping u and v */
        int x;
        if (j >= N->n || (i < U->n && U->a[i] <= N->a[j])) { x = U->a[i]; if (j < N->n && N->a[j] == x) j++; i++; }
        else { x = N->a[j]; j++; }
        if (x != u && x != v) tmp[m++] = x;
      }
      if (m > U->cap) { U->cap = m * 2; U->a = (int*)realloc(U->a, sizeof(int) * (size_t)U->cap); }
      memcpy(U->a, tmp, sizeof(int) * (size_t)m);
      U->n = m;
      int d = 0;
      for (int k = 0; k < m; k++) d += g->nodes[U->a[k]].dim;
      deg[u] = d;
      heap_push(&heap, d, u);
    }
    N->n = 0;
  }
  for (int i = 0; i < nn; i++) if (!g->nodes[i].deleted && dense[i]) order[no++] = i;
  /* expand to scalar permutation */
  if (g->perm_n < g->dim_nodes) { g->perm = (int*)realloc(g->perm, sizeof(int) * (size_t)g->dim_nodes); }
  g->perm_n = g->dim_nodes;
  int k = 0;
  for (int i = 0; i < no; i++) {
    node_t* n = &g->nodes[order[i]];
    for (int c = 0; c < n->dim; c++) g->perm[k++] = n->start + c;
  }
  for (int i = 0; i < nn; i++) free(adj[i].a);
  free(adj); free(dense); free(elim); free(deg); free(heap.a); free(order); free(tmp);
}

/* delta = (J'J + lambda diag(J'J))^-1 J' rhs  (isamlib/Cholesky.cpp:68-147,
 * Optimizer::compute_gauss_newton_step isamlib/Optimizer.cpp:49-67).  Returns 0 on success. */
static int solve_normal_equations(ora_graph* g, double lambda, double* delta) {
  int n = g->J_cols, m = g->J_rows;
  const int* Jp = g->Jp; const int* Ji = g->Ji; const double* Jx = g->Jx;
  double t_ord0 = now_s();
  if (!(g->prop.cache_ordering && g->perm_valid && g->perm_version == g->topo_version && g->perm_n == n)) {
    compute_ordering(g);
    g->perm_valid = 1; g->perm_version = g->topo_version;
  }
  g->tim[3] += now_s() - t_ord0;
  const int* perm = g->perm;
  int* pinv = (int*)malloc(sizeof(int) * (size_t)n);
  for (int k = 0; k < n; k++) pinv[perm[k]] = k;

  /* A = CSC of J (transpose of the row form), cholmod_transpose Cholesky.cpp:86 */
  long nnz = Jp[m];
  int* Ap = (int*)calloc((size_t)n + 1, sizeof(int));
  int* Ai = (int*)malloc(sizeof(int) * (size_t)nnz);
  double* Ax = (double*)malloc(sizeof(double) * (size_t)nnz);
  for (long p = 0; p < nnz; p++) Ap[Ji[p] + 1]++;
  for (int j = 0; j < n; j++) Ap[j + 1] += Ap[j];
  int* w = (int*)malloc(sizeof(int) * (size_t)n);
  memcpy(w, Ap, sizeof(int) * (size_t)n);
  for (int r = 0; r < m; r++)
    for (int p = Jp[r]; p < Jp[r + 1]; p++) { int q = w[Ji[p]]++; Ai[q] = r; Ax[q] = Jx[p]; }

  /* C = P (A'A with damped diagonal) P', upper triangle, CSC; Gustavson product (cholmod_ssmult, Cholesky.cpp:87-89) */
  int* Cp = (int*)calloc((size_t)n + 1, sizeof(int));
  ivec Ci = {0, 0, 0};
  double* Cx = NULL; long Cxcap = 0, cnz = 0;
  double* acc = (double*)calloc((size_t)n, sizeof(double));
  int* mark = (int*)malloc(sizeof(int) * (size_t)n);
  int* list = (int*)malloc(sizeof(int) * (size_t)n);
  for (int j = 0; j < n; j++) mark[j] = -1;
  double* Atb = (double*)calloc((size_t)n, sizeof(double));
  for (int kcol = 0; kcol < n; kcol++) {
    int j = perm[kcol];
    int nl = 0;
    double atb = 0;
    for (int p = Ap[j]; p < Ap[j + 1]; p++) {
      int r = Ai[p]; double ajr = Ax[p];
      atb += ajr * g->Jrhs[r];        /* cholmod_sdmult, Cholesky.cpp:120 */
      for (int q = Jp[r]; q < Jp[r + 1]; q++) {
        int i = pinv[Ji[q]];
        if (i > kcol) continue;        /* upper part only (stype=1) */
        if (mark[i] != kcol) { mark[i] = kcol; list[nl++] = i; acc[i] = 0; }
        acc[i] += Jx[q] * ajr;
      }
    }
    Atb[kcol] = atb;                   /* already permuted: CHOLMOD_P solve, Cholesky.cpp:122 */
    qsort(list, (size_t)nl, sizeof(int), cmp_int);
    if (cnz + nl > Cxcap) { Cxcap = 2 * (cnz + nl) + 1024; Cx = (double*)realloc(Cx, sizeof(double) * (size_t)Cxcap); }
    for (int t = 0; t < nl; t++) {
      int i = list[t];
      double val = acc[i];
      if (i == kcol && lambda > 0) val *= (1.0 + lambda);   /* Cholesky.cpp:94-97 */
      ivec_push(&Ci, i);
      Cx[cnz++] = val;
    }
    Cp[kcol + 1] = (int)cnz;
  }

  /* elimination tree */
  int* parent = (int*)malloc(sizeof(int) * (size_t)n);
  int* anc = (int*)malloc(sizeof(int) * (size_t)n);
  for (int k = 0; k < n; k++) {
    parent[k] = -1; anc[k] = -1;
    for (int p = Cp[k]; p < Cp[k + 1]; p++) {
      int i = Ci.a[p];
      while (i != -1 && i < k) { int inext = anc[i]; anc[i] = k; if (inext == -1) parent[i] = k; i = inext; }
    }
  }
  /* symbolic: column counts through row reaches */
  int* s = (int*)malloc(sizeof(int) * (size_t)n);
  int* flag = (int*)malloc(sizeof(int) * (size_t)n);
  int* cnt = (int*)calloc((size_t)n, sizeof(int));
  for (int k = 0; k < n; k++) flag[k] = -1;
  for (int k = 0; k < n; k++) {
    flag[k] = k; cnt[k]++;
    for (int p = Cp[k]; p < Cp[k + 1]; p++) {
      int i = Ci.a[p];
      for (; i < k && flag[i] != k; i = parent[i]) { flag[i] = k; cnt[i]++; }
    }
  }
  long* Lp = (long*)malloc(sizeof(long) * ((size_t)n + 1));
  Lp[0] = 0;
  for (int k = 0; k < n; k++) Lp[k + 1] = Lp[k] + cnt[k];
  long lnz = Lp[n];
  g->nnzL = lnz;
  int* Li = (int*)malloc(sizeof(int) * (size_t)lnz);
  double* Lx = (double*)malloc(sizeof(double) * (size_t)lnz);
  long* c = (long*)malloc(sizeof(long) * (size_t)n);
  double* x = (double*)calloc((size_t)n, sizeof(double));
  for (int k = 0; k < n; k++) { c[k] = Lp[k]; flag[k] = -1; }
  int ok = 0;
  /* up-looking numeric Cholesky (cholmod_factorize + change_factor to simplicial LL', Cholesky.cpp:100,111) */
  for (int k = 0; k < n; k++) {
    int top = n;
    flag[k] = k;
    for (int p = Cp[k]; p < Cp[k + 1]; p++) {
      int i = Ci.a[p];
      if (i > k) continue;
      x[i] = Cx[p];
      int len = 0;
      for (; flag[i] != k; i = parent[i]) { s[len++] = i; flag[i] = k; }
      while (len > 0) s[--top] = s[--len];
    }
    double d = x[k]; x[k] = 0;
    for (; top < n; top++) {
      int i = s[top];
      double lki = x[i] / Lx[Lp[i]];
      x[i] = 0;
      for (long p = Lp[i] + 1; p < c[i]; p++) x[Li[p]] -= Lx[p] * lki;
      d -= lki * lki;
      long p = c[i]++;
      Li[p] = k; Lx[p] = lki;
    }
    if (!(d > 0)) { ok = -1; break; }
    long p = c[k]++;
    Li[p] = k; Lx[p] = sqrt(d);
  }
  if (ok == 0) {
    /* L y = P A'b ; L' x = y (Cholesky.cpp:124-128) */
    double* y = Atb;
    for (int j = 0; j < n; j++) {
      y[j] /= Lx[Lp[j]];
      for (long p = Lp[j] + 1; p < Lp[j + 1]; p++) y[Li[p]] -= Lx[p] * y[j];
    }
    for (int j = n - 1; j >= 0; j--) {
      for (long p = Lp[j] + 1; p < Lp[j + 1]; p++) y[j] -= Lx[p] * y[Li[p]];
      y[j] /= Lx[Lp[j]];
    }
    /* permute_vector: delta(order[i]) = delta_ordered(i) (Optimizer.cpp:42-47) */
    for (int k = 0; k < n; k++) delta[perm[k]] = y[k];
  }
  free(pinv); free(Ap); free(Ai); free(Ax); free(w); free(Cp); free(Ci.a); free(Cx); free(acc); free(mark);
  free(list); free(Atb); free(parent); free(anc); free(s); free(flag); free(cnt); free(Lp); free(Li); free(Lx);
  free(c); free(x);
  return ok;
}

static double vec_norm(const double* v, int n) {
  double s = 0;
  for (int i = 0; i < n; i++) s += v[i] * v[i];
  return sqrt(s);
}

static void trace_push(ora_graph* g, double lambda, double chi2, int accepted) {
  if (g->ntrace == g->cap_trace) {
    g->cap_trace = g->cap_trace ? 2 * g->cap_trace : 64;
    g->trace = (trace_t*)realloc(g->trace, sizeof(trace_t) * (size_t)g->cap_trace);
  }
  g->trace[g->ntrace].lambda = lambda; g->trace[g->ntrace].chi2 = chi2; g->trace[g->ntrace].accepted = accepted;
  g->ntrace++;
}

/* Optimizer::relinearize (isamlib/Optimizer.cpp:114-185) for method != DOG_LEG,
 * reached from Slam::update with mod_batch = 1 (Slam.cpp:157-175, Mapping.cpp:34). */
void ora_update(ora_graph* g) {
  memset(g->tim, 0, sizeof g->tim);
  double t0 = now_s();
  estimate_to_linpoint(g);
  build_jacobian(g);
  double t1 = now_s();
  g->tim[0] += t1 - t0;
  double* h = (double*)calloc((size_t)g->dim_nodes, sizeof(double));
  if (solve_normal_equations(g, 0.0, h) != 0) fprintf(stderr, "ora_update: matrix not positive definite\n");
  double t2 = now_s();
  g->tim[1] += t2 - t1;
  apply_exmap(g, h);
  g->tim[2] += now_s() - t2;
  free(h);
}

/* Optimizer::levenberg_marquardt (isamlib/Optimizer.cpp:371-467) */
int ora_batch_optimize(ora_graph* g) {
  const ora_props* prop = &g->prop;
  memset(g->tim, 0, sizeof g->tim);
  g->ntrace = 0;
  int num_iter = 0;
  double lambda = prop->lm_lambda0;
  int n = g->dim_nodes;
  double* delta = (double*)calloc((size_t)n, sizeof(double));
  double t0 = now_s();
  estimate_to_linpoint(g);
  build_jacobian(g);
  double t1 = now_s(); g->tim[0] += t1 - t0;
  double error = chi2_sel(g, SEL_LIN);
  g->chi2_init = error;
  double t2 = now_s(); g->tim[2] += t2 - t1;
  if (solve_normal_equations(g, lambda, delta) != 0) fprintf(stderr, "ora: not positive definite\n");
  g->tim[1] += now_s() - t2;
  while ((prop->max_iterations <= 0 || num_iter < prop->max_iterations) && vec_norm(delta, n) > prop->epsilon2 &&
         error > prop->epsilon_abs) {
    num_iter++;
    double ta = now_s();
    linpoint_to_estimate(g);           /* remember the last accepted linearisation point */
    self_exmap(g, delta);
    double error_new = chi2_sel(g, SEL_LIN);
    double error_diff = error - error_new;
    double tb = now_s(); g->tim[2] += tb - ta;
    trace_push(g, lambda, error_new, error_diff > 0.);
    if (error_diff > 0.) {
      if (error_diff < prop->epsilon_rel * error) break;
      lambda /= prop->lm_lambda_factor;
      error = error_new;
      build_jacobian(g);
      double tc = now_s(); g->tim[0] += tc - tb; tb = tc;
    } else {
      lambda *= prop->lm_lambda_factor;
      estimate_to_linpoint(g);         /* restore previous estimate */
    }
    if (solve_normal_equations(g, lambda, delta) != 0) fprintf(stderr, "ora: not positive definite\n");
    g->tim[1] += now_s() - tb;
  }
  linpoint_to_estimate(g);
  free(delta);
  return num_iter;
}

int ora_trace_len(const ora_graph* g) { return g->ntrace; }
void ora_trace_get(const ora_graph* g, int i, double* lambda, double* chi2_new, int* accepted) {
  *lambda = g->trace[i].lambda; *chi2_new = g->trace[i].chi2; *accepted = g->trace[i].accepted;
}
double ora_initial_chi2(const ora_graph* g) { return g->chi2_init; }
void ora_timers(const ora_graph* g, double t[4]) { memcpy(t, g->tim, sizeof g->tim); }
long ora_last_nnzL(const ora_graph* g) { return g->nnzL; }

/* ------------------------------------------------------------------------- */
/* free-standing helpers                                                      */
/* ------------------------------------------------------------------------- */
void ora_plane_transform_to(const double abcd[4], const double tq[7], double out[4]) {
  plane_t p, o; pose_t ps; memcpy(p.p, abcd, 32); pose_from_tq(tq, &ps);
  plane_transform_to(&p, &ps, &o); memcpy(out, o.p, 32);
}
void ora_plane_transform_from(const double abcd[4], const double tq[7], double out[4]) {
  plane_t p, o; pose_t ps; memcpy(p.p, abcd, 32); pose_from_tq(tq, &ps);
  plane_transform_from(&p, &ps, &o); memcpy(out, o.p, 32);
}
void ora_plane_exmap(const double abcd[4], const double d[3], double out[4]) {
  plane_t p, o; memcpy(p.p, abcd, 32); plane_exmap(&p, d, &o); memcpy(out, o.p, 32);
}
void ora_pose_exmap(const double tq[7], const double d[6], double out[7]) {
  pose_t p, o; pose_from_tq(tq, &p); pose_exmap(&p, d, &o); pose_to_tq(&o, out);
}
void ora_pose_vector(const double tq[7], double v6[6]) { pose_t p; pose_from_tq(tq, &p); pose_vector(&p, v6); }
void ora_pose_from_vector(const double v6[6], double tq[7]) { pose_t p; pose_set_vector(&p, v6); pose_to_tq(&p, tq); }
void ora_pose_oplus(const double a[7], const double d[7], double out[7]) {
  pose_t pa, pd, o; pose_from_tq(a, &pa); pose_from_tq(d, &pd); pose_oplus(&pa, &pd, &o); pose_to_tq(&o, out);
}
void ora_pose_ominus(const double a[7], const double b[7], double out[7]) {
  pose_t pa, pb, o; pose_from_tq(a, &pa); pose_from_tq(b, &pb); pose_ominus(&pa, &pb, &o); pose_to_tq(&o, out);
}

/* ------------------------------------------------------------------------- */
/* plane data association, src/Mapping.cpp:256-397                             */
/* ------------------------------------------------------------------------- */

/* Plane3d::normal / d / point0 / distance(point)  (src/isam_plane3d.h:148-171) */
static double plane_norm3(const double* p) { return sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]); }
static void plane_normal(const double* p, double n[3]) {
  const double l = plane_norm3(p);
  for (int k = 0; k < 3; k++) n[k] = p[k] / l;
}
static double plane_d(const double* p) { return -p[3] / plane_norm3(p); }
static void plane_point0(const double* p, double x[3]) {
  double n[3]; plane_normal(p, n);
  const double d = plane_d(p);
  for (int k = 0; k < 3; k++) x[k] = d * n[k];
}
static double plane_distance_to(const double* p, const double pt[3]) {
  double n[3], x0[3]; plane_normal(p, n); plane_point0(p, x0);
  return fabs(n[0] * (pt[0] - x0[0]) + n[1] * (pt[1] - x0[1]) + n[2] * (pt[2] - x0[2]));
}
static float norm2f(float x, float y) { return sqrtf(x * x + y * y); }

/* Mapping.cpp:112-126: parameter t of the projection of the query on the segment, clamped to [0,1];
 * a degenerate segment returns the distance to its first point instead (kept as is). */
float ora_point_proj_to_lineseg(const float b[2], const float e[2], const float q[2]) {
  const float length = norm2f(e[0] - b[0], e[1] - b[1]);
  if (length < 0.001) return norm2f(q[0] - b[0], q[1] - b[1]);
  const float t = ((q[0] - b[0]) * (e[0] - b[0]) + (q[1] - b[1]) * (e[1] - b[1])) / length / length;
  if (t > 1.0) return 1.0f;
  if (t < 0.0) return 0.0f;
  return t;
}

/* Plane3d::project_to_plane (src/isam_plane3d.h:173-178) as used by reproj_to_newplane (Mapping.cpp:617-618) */
void ora_project_to_plane(const double abcd[4], const float pt[3], float out[3]) {
  double n[3]; plane_normal(abcd, n);
  const double d = plane_d(abcd);
  const double x = pt[0], y = pt[1], z = pt[2];
  const double s = (n[0] * x + n[1] * y + n[2] * z) - d;
  out[0] = (float)(x - n[0] * s); out[1] = (float)(y - n[1] * s); out[2] = (float)(z - n[2] * s);
}

void ora_find_closest_plane(const double est_pose[7], const double plane_local[4], int fpi, int frame_seq_id,
                            const float seg2d[4], const float seg3d[4], const ora_landmark* lm, int n_lm,
                            const ora_assoc_params* prm, int* best, double* best_err) {
  int numMatches = 0, bestMatch = -1;
  double bestReprojError = -1;
  pose_t pose; pose_from_tq(est_pose, &pose);
  plane_t cur_local, cur_world;
  memcpy(cur_local.p, plane_local, 32);
  plane_transform_from(&cur_local, &pose, &cur_world);                       /* :264 */
  double n_cur[3]; plane_normal(cur_world.p, n_cur);
  for (int c = 0; c < n_lm; c++) {
    const ora_landmark* L = &lm[c];
    if (L->deleted) continue;                                                 /* :275 */
    if (fpi == 0 && L->frame_plane_indice == 0) { numMatches++; bestMatch = c; break; }   /* :277-281 */
    if ((fpi == 0 && L->frame_plane_indice >= 1) || (fpi >= 1 && L->frame_plane_indice == 0)) continue;  /* :282-284 */
    plane_t old_world, old_local;
    memcpy(old_world.p, L->plane, 32);
    plane_transform_to(&old_world, &pose, &old_local);                        /* :291 */
    if (frame_seq_id - L->frame_seq_id > prm->assoc_near_frames) continue;    /* :295 */
    double n_old[3]; plane_normal(old_world.p, n_old);
    const double angle = acos(n_cur[0] * n_old[0] + n_cur[1] * n_old[1] + n_cur[2] * n_old[2]) * 180.0 / PI_;  /* :298 */
    if (angle > prm->edge_asso_angle) continue;                               /* :304 */
    double thre2d = prm->edge_asso_2ddist, thre_cov = prm->edge_asso_proj;
    if (angle < 25.0) {                                                       /* :309-315 */
      thre_cov = prm->edge_asso_proj / 3;
      thre2d = prm->edge_asso_2ddist * 1.5;
      if (angle <= 10.0) thre_cov = prm->edge_asso_proj / 2;
    }
    double x0[3]; plane_point0(old_local.p, x0);
    const double plane_dist = plane_distance_to(cur_local.p, x0);             /* :318 */
    if (plane_dist > prm->edge_asso_planedist) continue;                      /* :324 */
    if (plane_dist < 1.5) thre_cov = thre_cov / 2;                            /* :326 */
    float d2 = 0, d2c = 0;                                                     /* :330-346, fp32 */
    for (int i = 0; i < 2; i++) {
      const float a = norm2f(seg2d[2 * i] - L->seg2d[0], seg2d[2 * i + 1] - L->seg2d[1]);
      const float b = norm2f(seg2d[2 * i] - L->seg2d[2], seg2d[2 * i + 1] - L->seg2d[3]);
      d2 += (a < b ? a : b);
    }
    d2 = d2 / 2;
    for (int i = 0; i < 2; i++) {
      const float a = norm2f(seg2d[0] - L->seg2d[2 * i], seg2d[1] - L->seg2d[2 * i + 1]);
      const float b = norm2f(seg2d[2] - L->seg2d[2 * i], seg2d[3] - L->seg2d[2 * i + 1]);
      d2c += (a < b ? a : b);
    }
    d2c = d2c / 2;
    if (d2 > thre2d || d2c > thre2d) continue;                                /* :350 */
    const float o_bg = ora_point_proj_to_lineseg(seg3d, seg3d + 2, L->seg3d);           /* :355-360 */
    const float o_ed = ora_point_proj_to_lineseg(seg3d, seg3d + 2, L->seg3d + 2);
    const float cov_on = fabsf(o_bg - o_ed);
    const float n_bg = ora_point_proj_to_lineseg(L->seg3d, L->seg3d + 2, seg3d);
    const float n_ed = ora_point_proj_to_lineseg(L->seg3d, L->seg3d + 2, seg3d + 2);
    const float cov_no = fabsf(n_bg - n_ed);
    if (cov_on < thre_cov || cov_no < thre_cov) continue;                     /* :364 */
    numMatches++;
    double total = angle / prm->edge_asso_angle * 3 + (1 - cov_on) + (1 - cov_no);      /* :368 */
    total += (d2 > d2c ? d2 : d2c) / thre2d + plane_dist / 4;                             /* :369 */
    if (numMatches == 1 || total < bestReprojError) { bestMatch = c; bestReprojError = total; }   /* :374-377 */
  }
  *best = bestMatch;
  *best_err = bestReprojError;
}

/* ------------------------------------------------------------------------- */
/* pop-up (fp32), /root/reference/pop_up_wall                                  */
/* ------------------------------------------------------------------------- */

/* popup_plane::update_plane_equation_from_seg (libs/popup_plane.cpp:654-705),
 * ray_plane_interact (libs/matrix_utils.cpp:189-193) */
void ora_popup_planes_ex(const float* seg2d, int n, const float invK[9], const float T[16], float* out, float* seg3d_world) {
  if (n <= 0) return;
  /* ground_plane_sensor = T^T * (0,0,-1,0) */
  float gs[4];
  for (int k = 0; k < 4; k++) gs[k] = T[0 * 4 + k] * 0.f + T[1 * 4 + k] * 0.f + T[2 * 4 + k] * -1.f + T[3 * 4 + k] * 0.f;
  out[0] = gs[0]; out[1] = gs[1]; out[2] = gs[2]; out[3] = gs[3];
  for (int sgi = 0; sgi < n; sgi++) {
    float Pw[2][3];
    for (int e = 0; e < 2; e++) {
      float u = seg2d[sgi * 4 + 2 * e], v = seg2d[sgi * 4 + 2 * e + 1];
      float ray[3];
      for (int i = 0; i < 3; i++) ray[i] = invK[i * 3 + 0] * u + invK[i * 3 + 1] * v + invK[i * 3 + 2] * 1.f;
      float frac = -gs[3] / (gs[0] * ray[0] + gs[1] * ray[1] + gs[2] * ray[2]);
      float Ps[4] = {frac * ray[0], frac * ray[1], frac * ray[2], 1.f};
      float Ph[4];
      for (int i = 0; i < 4; i++) Ph[i] = T[i * 4 + 0] * Ps[0] + T[i * 4 + 1] * Ps[1] + T[i * 4 + 2] * Ps[2] + T[i * 4 + 3] * Ps[3];
      for (int i = 0; i < 3; i++) Pw[e][i] = Ph[i] / Ph[3];   /* homo_to_real_coord */
    }
    Pw[0][2] = 0.f; Pw[1][2] = 0.f;   /* "make it exact zero", popup_plane.cpp:575-578 */
    if (seg3d_world)                  /* ground_seg3d_lines_world row (x0,y0,0,x1,y1,0) */
      for (int e = 0; e < 2; e++)
        for (int i = 0; i < 3; i++) seg3d_world[sgi * 6 + 3 * e + i] = Pw[e][i];
    float t1[3] = {Pw[1][0] - Pw[0][0], Pw[1][1] - Pw[0][1], Pw[1][2] - Pw[0][2]};
    float t2[3] = {0.f, 0.f, -1.f};
    float nw[3] = {t1[1] * t2[2] - t1[2] * t2[1], t1[2] * t2[0] - t1[0] * t2[2], t1[0] * t2[1] - t1[1] * t2[0]};
    float dist = -(nw[0] * Pw[0][0] + nw[1] * Pw[0][1] + nw[2] * Pw[0][2]);
    float pw[4] = {nw[0], nw[1], nw[2], dist};
    for (int k = 0; k < 4; k++)
      out[(sgi + 1) * 4 + k] = T[0 * 4 + k] * pw[0] + T[1 * 4 + k] * pw[1] + T[2 * 4 + k] * pw[2] + T[3 * 4 + k] * pw[3];
  }
}
void ora_popup_planes(const float* seg2d, int n, const float invK[9], const float T[16], float* out) {
  ora_popup_planes_ex(seg2d, n, invK, T, out, NULL);
}

/* generate_cloud per-pixel math + matrixToCloud filters (libs/popup_plane.cpp:826-831, 948-960) */
void ora_popup_cloud(const int* plane_id, int width, int height, const float invK[9], const float T[16],
                     const float* planes, int nplanes, float depth_thre, float ceiling_thre,
                     float* xyz, unsigned char* valid) {
  for (int y = 0; y < height; y++)
    for (int x = 0; x < width; x++) {
      int idx = y * width + x;
      int pid = plane_id[idx];
      valid[idx] = 0; xyz[3 * idx] = xyz[3 * idx + 1] = xyz[3 * idx + 2] = 0.f;
      if (pid < 0 || pid >= nplanes) continue;
      const float* pl = planes + 4 * pid;
      float ray[3];
      for (int i = 0; i < 3; i++) ray[i] = invK[i * 3 + 0] * (float)x + invK[i * 3 + 1] * (float)y + invK[i * 3 + 2] * 1.f;
      float frac = -pl[3] / (pl[0] * ray[0] + pl[1] * ray[1] + pl[2] * ray[2]);
      float Ps[3] = {frac * ray[0], frac * ray[1], frac * ray[2]};
      float Pw[3];
      for (int i = 0; i < 3; i++) Pw[i] = T[i * 4 + 0] * Ps[0] + T[i * 4 + 1] * Ps[1] + T[i * 4 + 2] * Ps[2];
      for (int i = 0; i < 3; i++) Pw[i] += T[i * 4 + 3];
      if (Ps[2] < 0) continue;
      if (Ps[2] > depth_thre) continue;
      if (Pw[2] < -0.2f) continue;
      xyz[3 * idx] = Pw[0]; xyz[3 * idx + 1] = Pw[1];
      xyz[3 * idx + 2] = Pw[2] < ceiling_thre ? Pw[2] : ceiling_thre;
      valid[idx] = 1;
    }
}

/* get_depth_map_good (libs/popup_plane.cpp:866-921), full resolution */
void ora_popup_depth(const int* plane_id, int width, int height, const float invK[9], const float T[16],
                     const float* planes, int nplanes, const float ceil_s[4], float ceiling_thre, float* depth) {
  for (int y = 0; y < height; y++)
    for (int x = 0; x < width; x++) {
      int idx = y * width + x;
      int pid = plane_id[idx];
      depth[idx] = 0.f;
      if (pid < 0 || pid >= nplanes) continue;
      const float* pl = planes + 4 * pid;
      float ray[3];
      for (int i = 0; i < 3; i++) ray[i] = invK[i * 3 + 0] * (float)x + invK[i * 3 + 1] * (float)y + invK[i * 3 + 2] * 1.f;
      float frac = -pl[3] / (pl[0] * ray[0] + pl[1] * ray[1] + pl[2] * ray[2]);
      float Ps[3] = {frac * ray[0], frac * ray[1], frac * ray[2]};
      float hgt = T[2 * 4 + 0] * Ps[0] + T[2 * 4 + 1] * Ps[1] + T[2 * 4 + 2] * Ps[2];
      hgt += T[2 * 4 + 3];
      if (hgt < ceiling_thre) {
        if (Ps[2] < 0) continue;
        depth[idx] = Ps[2];
      } else {
        float fc = -ceil_s[3] / (ceil_s[0] * ray[0] + ceil_s[1] * ray[1] + ceil_s[2] * ray[2]);
        float z = fc * ray[2];
        if (z < 0) continue;
        depth[idx] = z;
      }
    }
}


/* popup_plane.cpp:913-917.  cv::resize(src, dst, Size(), 0.5, 0.5) with the default INTER_LINEAR takes the INTER_AREA
 * path for an exact factor of 2: dst = (s00 + s01 + s10 + s11) * 0.25.  convertTo(-1, 4) multiplies by 4.
 * cv::resize(..., 2, 2): fx = (float)((dx + 0.5) * 0.5 - 0.5), sx = floor(fx), fx -= sx, clamped at both borders;
 * a horizontal pass S[sx] * (1 - fx) + S[sx + 1] * fx on two source rows, then the vertical blend. */
void ora_depth_fill_half(const float* sparse, int w, int h, float* out) {
  const int hw = w / 2, hh = h / 2;
  float* half = (float*)malloc(sizeof(float) * (size_t)hw * hh);
  for (int y = 0; y < hh; y++)
    for (int x = 0; x < hw; x++) {
      const float* s = sparse + (size_t)(2 * y) * w + 2 * x;
      const float sum = s[0] + s[1] + s[w] + s[w + 1];
      half[(size_t)y * hw + x] = sum * 0.25f * 4.0f;
    }
  for (int Y = 0; Y < h; Y++) {
    float fy = (float)((Y + 0.5) * 0.5 - 0.5);
    int sy = (int)floorf(fy);
    fy -= sy;
    if (sy < 0) { fy = 0; sy = 0; }
    if (sy >= hh - 1) { fy = 0; sy = hh - 1; }
    const int sy1 = sy + 1 < hh ? sy + 1 : hh - 1;
    for (int X = 0; X < w; X++) {
      float fx = (float)((X + 0.5) * 0.5 - 0.5);
      int sx = (int)floorf(fx);
      fx -= sx;
      if (sx < 0) { fx = 0; sx = 0; }
      if (sx >= hw - 1) { fx = 0; sx = hw - 1; }
      const int sx1 = sx + 1 < hw ? sx + 1 : hw - 1;
      const float r0 = half[(size_t)sy * hw + sx] * (1.f - fx) + half[(size_t)sy * hw + sx1] * fx;
      const float r1 = half[(size_t)sy1 * hw + sx] * (1.f - fx) + half[(size_t)sy1 * hw + sx1] * fx;
      out[(size_t)Y * w + X] = r0 * (1.f - fy) + r1 * fy;
    }
  }
  free(half);
}


/* matrix_utils.cpp:290-304 */
static float point_dist_lineseg_f(const float b[2], const float e[2], const float q[2]) {
  const float length = sqrtf((e[0] - b[0]) * (e[0] - b[0]) + (e[1] - b[1]) * (e[1] - b[1]));
  if (length < 0.001) return sqrtf((q[0] - b[0]) * (q[0] - b[0]) + (q[1] - b[1]) * (q[1] - b[1]));
  const float t = ((q[0] - b[0]) * (e[0] - b[0]) + (q[1] - b[1]) * (e[1] - b[1])) / length / length;
  if (t < 0.0) return sqrtf((q[0] - b[0]) * (q[0] - b[0]) + (q[1] - b[1]) * (q[1] - b[1]));
  else if (t > 1.0) return sqrtf((q[0] - e[0]) * (q[0] - e[0]) + (q[1] - e[1]) * (q[1] - e[1]));
  const float p[2] = {b[0] + t * (e[0] - b[0]), b[1] + t * (e[1] - b[1])};
  return sqrtf((q[0] - p[0]) * (q[0] - p[0]) + (q[1] - p[1]) * (q[1] - p[1]));
}

/* direction_hit_boundary (libs/matrix_utils.cpp:229-270): where the ray pt + lambda * direc leaves the image.  The
 * literals 0.0 / 1.0 are doubles there, so lambda is a double quotient rounded to float.  (-1, -1) = no hit. */
static void direction_hit_boundary(const float pt[2], const float direc[2], int img_width, int img_height, float hit[2]) {
  float lambd;
  if (direc[1] < 0) {                                   /* top edge */
    lambd = (float)((0.0 - pt[1]) / direc[1]);
    if (lambd >= 0) {
      const float hx = pt[0] + lambd * direc[0], hy = pt[1] + lambd * direc[1];
      if ((0 <= (int)hx) && ((int)hx <= img_width - 1)) { hit[0] = hx; hit[1] = hy; return; }
    }
  }
  if (direc[1] > 0) {                                   /* bottom edge (nobottom = false) */
    lambd = (float)((img_height - 1.0 - pt[1]) / direc[1]);
    if (lambd >= 0) {
      const float hx = pt[0] + lambd * direc[0], hy = pt[1] + lambd * direc[1];
      if ((0 <= (int)hx) && ((int)hx <= img_width - 1)) { hit[0] = hx; hit[1] = hy; return; }
    }
  }
  if (direc[0] > 0) {                                   /* right edge */
    lambd = (float)((img_width - 1.0 - pt[0]) / direc[0]);
    if (lambd >= 0) {
      const float hx = pt[0] + lambd * direc[0], hy = pt[1] + lambd * direc[1];
      if ((0 <= (int)hy) && ((int)hy <= img_height - 1)) { hit[0] = hx; hit[1] = hy; return; }
    }
  }
  if (direc[0] < 0) {                                   /* left edge */
    lambd = (float)((0.0 - pt[0]) / direc[0]);
    if (lambd >= 0) {
      const float hx = pt[0] + lambd * direc[0], hy = pt[1] + lambd * direc[1];
      if ((0 <= (int)hy) && ((int)hy <= img_height - 1)) { hit[0] = hx; hit[1] = hy; return; }
    }
  }
  hit[0] = -1.f; hit[1] = -1.f;
}

/* popup_plane::find_2d_3d_closed_polygon_simplemode (libs/popup_plane.cpp:409-500) with walllength_threshold <= 0 (the class
 * default, popup_plane.h:81; the cut of :502-546 needs cv::intersectConvexConvex and is not restated): one closed 2-D polygon
 * per wall -- the ground segment, the image-boundary hits of the world-vertical lines through its two end points, and the
 * image corners between them.  Plane 0 (ground) has no polygon in this mode (:489).
 * The reference inverts transToWolrd with Eigen's general 4x4 inverse; here the rigid-body inverse [R' | -R' t] is used (same
 * matrix up to fp32 rounding; Eigen's operation order cannot be restated without Eigen).
 * verts: (x, y) pairs, at most 8 per wall; off[n + 2]: vertex offsets of planes 0 .. n.  Returns the vertex count. */
int ora_popup_polygons_simple(const float* seg2d, int n, const float K[9], const float invK[9], const float T[16], int width, int height,
                              float* verts, int* off) {
  off[0] = 0; off[1] = 0;                                /* ground: void */
  if (n <= 0) return 0;
  float* planes = (float*)malloc(sizeof(float) * 4 * (size_t)(n + 1));
  float* seg3d = (float*)malloc(sizeof(float) * 6 * (size_t)n);
  ora_popup_planes_ex(seg2d, n, invK, T, planes, seg3d);   /* ground_seg3d_lines_world (:569-578) */
  float iT[12];                                          /* rows 0..2 of invT */
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) iT[i * 4 + j] = T[j * 4 + i];
    iT[i * 4 + 3] = -(T[0 * 4 + i] * T[3] + T[1 * 4 + i] * T[7] + T[2 * 4 + i] * T[11]);
  }
  int nv = 0;
  for (int sg = 0; sg < n; sg++) {
    float hitb[2][2];
    for (int e = 0; e < 2; e++) {
      float img[2][2];
      for (int c = 0; c < 2; c++) {                      /* the ground point and the point 2 m above it (:419-427) */
        const float Pw[3] = {seg3d[sg * 6 + 3 * e], seg3d[sg * 6 + 3 * e + 1], seg3d[sg * 6 + 3 * e + 2] + (c ? 2.f : 0.f)};
        float Ps[3], h[3];
        for (int i = 0; i < 3; i++) Ps[i] = iT[i * 4 + 0] * Pw[0] + iT[i * 4 + 1] * Pw[1] + iT[i * 4 + 2] * Pw[2] + iT[i * 4 + 3] * 1.f;
        for (int i = 0; i < 3; i++) h[i] = K[i * 3 + 0] * Ps[0] + K[i * 3 + 1] * Ps[1] + K[i * 3 + 2] * Ps[2];
        img[c][0] = h[0] / h[2]; img[c][1] = h[1] / h[2];
      }
      float dir[2] = {img[1][0] - img[0][0], img[1][1] - img[0][1]};
      if (dir[1] > 0) { dir[0] = -dir[0]; dir[1] = -dir[1]; }                  /* :428-429 */
      direction_hit_boundary(seg2d + sg * 4 + 2 * e, dir, width, height, hitb[e]);   /* :436-441 */
    }
    const float* p0 = seg2d + sg * 4;
    const float* p1 = seg2d + sg * 4 + 2;
    const float* bh = hitb[0];
    const float* eh = hitb[1];
    float* v = verts + 2 * (size_t)nv;
    int k = 0;
#define PUSH(x, y) do { v[2 * k] = (x); v[2 * k + 1] = (y); k++; } while (0)
    PUSH(p0[0], p0[1]); PUSH(p1[0], p1[1]);
    if ((eh[0] != p1[0]) || (eh[1] != p1[1])) PUSH(eh[0], eh[1]);                                  /* :462 */
    if (0 < bh[0] && bh[0] < width - 1 && eh[0] == width - 1) PUSH((float)(width - 1), 0.f);     /* :464-466 */
    if (bh[0] == 0 && eh[0] == width - 1) { PUSH((float)(width - 1), 0.f); PUSH(0.f, 0.f); }      /* :467-470 */
    if (bh[0] == 0 && 0 < eh[0] && eh[0] < width - 1) PUSH(0.f, 0.f);                             /* :471-473 */
    if ((bh[0] != p0[0]) || (bh[1] != p0[1])) PUSH(bh[0], bh[1]);                                  /* :474 */
    PUSH(p0[0], p0[1]);                                                                            /* final close (:477) */
#undef PUSH
    if ((bh[0] == -1) || (eh[0] == -1)) k = 0;                                                    /* :480-481 */
    nv += k;
    off[sg + 2] = nv;
  }
  free(planes); free(seg3d);
  return nv;
}

/* popup_plane.cpp:616-640 */
void ora_popup_plane_info(const float* seg2d, int n, const float invK[9], const float T[16], float plane_cam_dist_thre,
                          const int* actual, int n_actual, float* dist_to_cam, int* good) {
  float gs[4];
  for (int k = 0; k < 4; k++) gs[k] = T[0 * 4 + k] * 0.f + T[1 * 4 + k] * 0.f + T[2 * 4 + k] * -1.f + T[3 * 4 + k] * 0.f;
  dist_to_cam[0] = T[2 * 4 + 3];
  good[0] = 1;
  const float cam[2] = {T[0 * 4 + 3], T[1 * 4 + 3]};
  for (int sgi = 0; sgi < n; sgi++) {
    float Pw[2][2], zs[2];
    for (int e = 0; e < 2; e++) {
      const float u = seg2d[sgi * 4 + 2 * e], v = seg2d[sgi * 4 + 2 * e + 1];
      float ray[3];
      for (int i = 0; i < 3; i++) ray[i] = invK[i * 3 + 0] * u + invK[i * 3 + 1] * v + invK[i * 3 + 2] * 1.f;
      const float frac = -gs[3] / (gs[0] * ray[0] + gs[1] * ray[1] + gs[2] * ray[2]);
      const float Ps[4] = {frac * ray[0], frac * ray[1], frac * ray[2], 1.f};
      zs[e] = Ps[2];                        /* ground_seg3d_lines_sensor(seg, 2 / 5) */
      float Ph[4];
      for (int i = 0; i < 4; i++) Ph[i] = T[i * 4 + 0] * Ps[0] + T[i * 4 + 1] * Ps[1] + T[i * 4 + 2] * Ps[2] + T[i * 4 + 3] * Ps[3];
      Pw[e][0] = Ph[0] / Ph[3]; Pw[e][1] = Ph[1] / Ph[3];
    }
    dist_to_cam[sgi + 1] = point_dist_lineseg_f(Pw[0], Pw[1], cam);
    int ok = zs[0] > 0 && zs[1] > 0 && dist_to_cam[sgi + 1] < plane_cam_dist_thre;
    if (ok && n_actual > 0) {
      ok = 0;
      for (int k = 0; k < n_actual; k++) if (actual[k] == sgi + 1) ok = 1;
    }
    good[sgi + 1] = ok;
  }
}
