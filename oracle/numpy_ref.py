"""Independent numpy/scipy evaluation of the reference arithmetic -> golden fixtures.

TEST INFRASTRUCTURE.  Written separately from oracle/pps_oracle.c (different language,
different building blocks: scipy.spatial.transform.Rotation for every rotation /
Euler / log-map operation, dense linear algebra for the solve) so that the two
restatements pin each other.  The reference itself ships no golden vectors and
cannot be built here -- these fixtures are SELF-GENERATED:

    python oracle/numpy_ref.py            # rewrites tests/golden/*.json

Reference semantics followed (paths relative to /root/reference/pop_planar_slam):
  residuals        src/isam_plane3d.h:271-304,449-473 ; Thirdparty/isam/include/isam/slam3d.h:82-88,174-191
  retractions      src/isam_plane3d.h:101-127 ; isam/Pose3d.h:131-136
  Jacobians        Thirdparty/isam/isamlib/numericalDiff.cpp:41-87 (central differences, eps = 1e-4)
  normal equations Thirdparty/isam/isamlib/Cholesky.cpp:86-128 ((J'J with diag*(1+lambda)) delta = J'b, b = -r)
  LM loop          Thirdparty/isam/isamlib/Optimizer.cpp:371-467 with the app's Properties (Mapping.cpp:32-43)
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
from scipy.spatial.transform import Rotation as Rot

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EPS = 1e-4
F_POSE_PRIOR, F_ODOMETRY, F_PLANE_OBS, F_PLANE_PRIOR = 0, 1, 2, 3


def wrap(a):
    return (a + np.pi) % (2 * np.pi) - np.pi


def unpack_ut(ut, m):
    U = np.zeros((m, m))
    k = 0
    for r in range(m):
        for c in range(r, m):
            U[r, c] = ut[k]
            k += 1
    return U


# ---- state ------------------------------------------------------------------
def pose_exmap(p, d):
    q = (Rot.from_quat(p[3:]) * Rot.from_rotvec(d[3:])).as_quat()
    return np.concatenate([p[:3] + d[:3], q])


def plane_exmap(pl, d):
    q = (Rot.from_rotvec(d) * Rot.from_quat(pl)).as_quat()
    return q / np.linalg.norm(q)


def pose_vec(p):
    ypr = Rot.from_quat(p[3:]).as_euler("ZYX")
    return np.concatenate([p[:3], ypr])


# ---- residuals --------------------------------------------------------------
def quat_log_diff(q, qm):
    return (Rot.from_quat(q) * Rot.from_quat(qm).inv()).as_rotvec()


def res_plane_obs(pose, plane, meas):
    R = Rot.from_quat(pose[3:]).as_matrix()
    u = np.concatenate([R.T @ plane[:3], [plane[:3] @ pose[:3] + plane[3]]])
    return quat_log_diff(u / np.linalg.norm(u), meas)


def res_plane_prior(plane, meas):
    return quat_log_diff(plane, meas)


def res_pose_prior(pose, meas6):
    e = pose_vec(pose) - meas6
    e[3:] = wrap(e[3:])
    return e


def res_odometry(p1, p2, meas6):
    R1 = Rot.from_quat(p1[3:]).as_matrix()
    R2 = Rot.from_quat(p2[3:]).as_matrix()
    R12 = R1.T @ R2
    t12 = R1.T @ (p2[:3] - p1[:3])
    ypr = Rot.from_matrix(R12).as_euler("ZYX")
    e = np.concatenate([t12, ypr]) - meas6
    e[3:] = wrap(e[3:])
    return e


class Graph:
    def __init__(self, spec):
        self.node_type = spec.node_type.copy()
        self.x = [spec.node_init[i, :7].copy() if spec.node_type[i] == 0 else
                  spec.node_init[i, :4] / np.linalg.norm(spec.node_init[i, :4]) for i in range(len(spec.node_type))]
        self.dim = [6 if t == 0 else 3 for t in spec.node_type]
        self.start = np.concatenate([[0], np.cumsum(self.dim)])
        self.f_type = spec.f_type
        self.f_nodes = spec.f_nodes
        self.f_meas = []
        self.f_U = []
        for k, t in enumerate(spec.f_type):
            if t in (F_POSE_PRIOR, F_ODOMETRY):
                self.f_meas.append(spec.f_meas[k, :6].copy())
                self.f_U.append(unpack_ut(spec.f_sqrtinf[k, :21], 6))
            else:
                m = spec.f_meas[k, :4]
                self.f_meas.append(m / np.linalg.norm(m))
                self.f_U.append(unpack_ut(spec.f_sqrtinf[k, :6], 3))

    def n(self):
        return int(self.start[-1])

    def basic_error(self, k, x):
        t = self.f_type[k]
        a, b = self.f_nodes[k]
        if t == F_PLANE_OBS:
            return res_plane_obs(x[a], x[b], self.f_meas[k])
        if t == F_PLANE_PRIOR:
            return res_plane_prior(x[a], self.f_meas[k])
        if t == F_POSE_PRIOR:
            return res_pose_prior(x[a], self.f_meas[k])
        return res_odometry(x[a], x[b], self.f_meas[k])

    def error(self, k, x):
        return self.f_U[k] @ self.basic_error(k, x)

    def retract_node(self, i, xi, d):
        return pose_exmap(xi, d) if self.node_type[i] == 0 else plane_exmap(xi, d)

    def factor_jacobian(self, k, x):
        """(H, r): H = whitened central-difference Jacobian [cols of node a | cols of node b]"""
        nodes = [n for n in self.f_nodes[k] if n >= 0]
        cols = []
        for n in nodes:
            for j in range(self.dim[n]):
                d = np.zeros(self.dim[n])
                xs = list(x)
                d[j] = EPS
                xs[n] = self.retract_node(n, x[n], d)
                yp = self.error(k, xs)
                d[j] = -EPS
                xs[n] = self.retract_node(n, x[n], d)
                ym = self.error(k, xs)
                cols.append((yp - ym) / (EPS + EPS))
        return np.array(cols).T, self.error(k, x)

    def jacobian(self, x):
        rows, rhs = [], []
        for k in range(len(self.f_type)):
            H, r = self.factor_jacobian(k, x)
            R = np.zeros((H.shape[0], self.n()))
            c = 0
            for n in self.f_nodes[k]:
                if n < 0:
                    continue
                R[:, self.start[n]:self.start[n] + self.dim[n]] = H[:, c:c + self.dim[n]]
                c += self.dim[n]
            rows.append(R)
            rhs.append(-r)
        return np.vstack(rows), np.concatenate(rhs)

    def chi2(self, x):
        return float(sum(np.sum(self.error(k, x) ** 2) for k in range(len(self.f_type))))

    def retract(self, x, delta):
        return [self.retract_node(i, x[i], delta[self.start[i]:self.start[i + 1]]) for i in range(len(x))]

    @staticmethod
    def solve(J, b, lam):
        H = J.T @ J
        H[np.diag_indices_from(H)] *= (1.0 + lam)
        return np.linalg.solve(H, J.T @ b)

    def gauss_newton_step(self):
        J, b = self.jacobian(self.x)
        self.x = self.retract(self.x, self.solve(J, b, 0.0))

    def levenberg_marquardt(self, eps2=1e-3, eps_abs=1e-4, eps_rel=1e-6, max_it=500, lam0=1e-6, fac=10.0):
        lam = lam0
        x0 = self.x
        J, b = self.jacobian(x0)
        error = self.chi2(x0)
        chi0 = error
        delta = self.solve(J, b, lam)
        trace = []
        it = 0
        while it < max_it and np.linalg.norm(delta) > eps2 and error > eps_abs:
            it += 1
            saved = x0
            x0 = self.retract(x0, delta)
            enew = self.chi2(x0)
            diff = error - enew
            trace.append((lam, enew, bool(diff > 0)))
            if diff > 0:
                if diff < eps_rel * error:
                    break
                lam /= fac
                error = enew
                J, b = self.jacobian(x0)
            else:
                lam *= fac
                x0 = saved
            delta = self.solve(J, b, lam)
        self.x = x0
        return it, chi0, trace


def fixture_for(spec):
    g = Graph(spec)
    fx = {"name": spec.name, "n_poses": spec.n_poses, "n_planes": spec.n_planes,
          "spec": {"node_type": spec.node_type.tolist(), "node_init": spec.node_init.tolist(),
                   "f_type": spec.f_type.tolist(), "f_nodes": spec.f_nodes.tolist(), "f_meas": spec.f_meas.tolist(),
                   "f_sqrtinf": spec.f_sqrtinf.tolist(),
                   "factor_after_node": spec.meta["factor_after_node"].tolist()}}
    fx["chi2_initial"] = g.chi2(g.x)
    fac = []
    for k in range(len(spec.f_type)):
        H, r = g.factor_jacobian(k, g.x)
        fac.append({"r": r.tolist(), "H": H.tolist()})
    fx["factors"] = fac
    J, b = g.jacobian(g.x)
    fx["gn_delta"] = g.solve(J, b, 0.0).tolist()
    fx["lm_delta_lambda_1e-3"] = g.solve(J, b, 1e-3).tolist()
    g2 = Graph(spec)
    g2.gauss_newton_step()
    fx["chi2_after_gn"] = g2.chi2(g2.x)
    it, chi0, trace = g.levenberg_marquardt()
    fx["lm_iterations"] = it
    fx["lm_trace"] = [[l, c, a] for l, c, a in trace]
    fx["chi2_final"] = g.chi2(g.x)
    fx["final_state"] = [xi.tolist() for xi in g.x]
    return fx


def special_cases():
    """Hand-picked residual inputs: double-cover (w < 0), near-identity dq, the ground plane."""
    rng = np.random.default_rng(11)
    cases = []
    for name in ("random", "w_negative", "near_identity", "ground"):
        for _ in range(4):
            q = rng.normal(size=4); q /= np.linalg.norm(q)
            t = rng.normal(size=3)
            pose = np.concatenate([t, q])
            pl = rng.normal(size=4); pl /= np.linalg.norm(pl)
            if name == "ground":
                pl = np.array([0.0, 0.0, -1.0, 0.0])
            R = Rot.from_quat(q).as_matrix()
            u = np.concatenate([R.T @ pl[:3], [pl[:3] @ t + pl[3]]]); u /= np.linalg.norm(u)
            if name == "near_identity":
                meas = (Rot.from_rotvec(rng.normal(size=3) * 1e-9) * Rot.from_quat(u)).as_quat()
            elif name == "w_negative":
                meas = -(Rot.from_rotvec(rng.normal(size=3) * 0.3) * Rot.from_quat(u)).as_quat()
            else:
                meas = rng.normal(size=4); meas /= np.linalg.norm(meas)
            p2 = np.concatenate([rng.normal(size=3), (lambda v: v / np.linalg.norm(v))(rng.normal(size=4))])
            m6 = np.concatenate([rng.normal(size=3), rng.uniform(-3, 3, 3) * np.array([1, 0.4, 1])])
            cases.append({"kind": name, "pose": pose.tolist(), "plane": pl.tolist(), "meas": meas.tolist(),
                          "pose2": p2.tolist(), "meas6": m6.tolist(),
                          "e_plane_obs": res_plane_obs(pose, pl, meas).tolist(),
                          "e_plane_prior": res_plane_prior(pl, meas).tolist(),
                          "e_pose_prior": res_pose_prior(pose, m6).tolist(),
                          "e_odometry": res_odometry(pose, p2, m6).tolist(),
                          "pose_vector": pose_vec(pose).tolist(),
                          "pose_exmap": pose_exmap(pose, m6 * 0.1).tolist(),
                          "plane_exmap": plane_exmap(pl, m6[:3] * 0.1).tolist()})
    return cases


def main():
    from pop_up_slam_amd import synth
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    specs = [synth.small_world(5, 3, seed=1), synth.small_world(20, 6, seed=2, obs_per_pose=5),
             synth.small_world(50, 10, seed=3, obs_per_pose=5),
             synth.small_world(30, 8, seed=4, obs_per_pose=5, odo_scale=25.0, meas_sigma=0.03, name="hard_30p_8l"),
             # badly initialised: long LM trajectory with many rejected trials
             synth.small_world(40, 6, seed=9, obs_per_pose=3, odo_scale=100.0, meas_sigma=0.05, name="hard_40p_6l")]
    for spec in specs:
        fx = fixture_for(spec)
        with open(os.path.join(out, f"{spec.name}.json"), "w") as f:
            json.dump(fx, f)
        print(spec.name, "chi2", fx["chi2_initial"], "->", fx["chi2_final"], "iters", fx["lm_iterations"])
    with open(os.path.join(out, "residual_cases.json"), "w") as f:
        json.dump(special_cases(), f)


# ---- edge-case graphs (round 5): what the synthetic worlds never produce ---------------------------------------------
def _ut_pack(U):
    m = U.shape[0]
    return np.array([U[r, c] for r in range(m) for c in range(r, m)])


def dense_sqrtinf_spec():
    """A small world whose factors carry NON-DIAGONAL upper-triangular square-root information matrices (Noise.h:36-62 accepts any
    sqrtinf; Covariance() produces one from a full covariance): the diagonal of the application's weights times (I + strictly upper
    random part)."""
    from pop_up_slam_amd import synth
    spec = synth.small_world(12, 4, seed=21, obs_per_pose=3, odo_scale=25.0, meas_sigma=0.03, name="dense_sqrtinf_12p_4l")
    rng = np.random.default_rng(77)
    sq = spec.f_sqrtinf.copy()
    for k, t in enumerate(spec.f_type):
        m = 6 if t in (F_POSE_PRIOR, F_ODOMETRY) else 3
        U = unpack_ut(sq[k, :m * (m + 1) // 2], m)
        S = np.triu(rng.uniform(-0.6, 0.6, (m, m)), 1)
        U = np.diag(np.diag(U)) @ (np.eye(m) + S)
        sq[k, :] = 0.0
        sq[k, :m * (m + 1) // 2] = _ut_pack(U)
    spec.f_sqrtinf = sq
    return spec


def pi_wrap_spec():
    """Poses whose yaw -- absolute (pose prior) and relative (odometry) -- sits within 1e-6 of +-pi with the measurement on the OTHER
    side of the wrap (slam3d.h:82-88,174-191 + standardRad, util.h:101-108), and plane observations whose measured 4-vector is the
    negated (double-cover) representative, so that dq.w < 0 inside a solve (isam_plane3d.h:286-294)."""
    from pop_up_slam_amd import synth
    from scipy.optimize import brentq
    rng = np.random.default_rng(123)
    n = 8

    def quat(y, pt, r):
        return Rot.from_euler("ZYX", [y, pt, r]).as_quat()

    def rel_yaw(q1, q2):
        return (Rot.from_quat(q1).inv() * Rot.from_quat(q2)).as_euler("ZYX")[0]

    # pose 0: absolute yaw pi - 3e-7.  Edges 0->1 and 4->5 are U-turns whose relative yaw is +pi - 5e-7 / -pi + 4e-7.
    pr = rng.normal(0.0, 0.02, (n, 2))
    yaws = [np.pi - 3e-7]
    targets = {0: np.pi - 5e-7, 4: -np.pi + 4e-7}
    q = [quat(yaws[0], *pr[0])]
    for k in range(n - 1):
        if k in targets:
            tgt = targets[k]
            f = lambda dy: wrap(rel_yaw(q[k], quat(yaws[k] + dy, *pr[k + 1])) - tgt)      # noqa: E731
            dy = brentq(f, tgt - 0.2, tgt + 0.2, xtol=1e-15, rtol=1e-15)
        else:
            dy = rng.normal(0.4, 0.2)
        yaws.append(yaws[k] + dy)
        q.append(quat(yaws[-1], *pr[k + 1]))
    t = np.cumsum(rng.normal(0.0, 0.3, (n, 3)), axis=0)
    truth_pose = [np.concatenate([t[k], q[k]]) for k in range(n)]
    planes = [np.array([0.0, 0.0, -1.0, 0.0])]
    for _ in range(2):
        v = rng.normal(size=3); v /= np.linalg.norm(v)
        pl = np.append(v, rng.uniform(1.0, 4.0)); planes.append(pl / np.linalg.norm(pl))
    nt, ni, ft, fn, fm, fs, after = [], [], [], [], [], [], []

    def node(tp, init):
        nt.append(tp); v = np.zeros(7); v[:len(init)] = init; ni.append(v); return len(nt) - 1

    def factor(tp, a, b, meas, sq):
        ft.append(tp); fn.append((a, b)); m = np.zeros(6); m[:len(meas)] = meas; fm.append(m)
        s = np.zeros(21); s[:len(sq)] = sq; fs.append(s); after.append(len(nt) - 1)

    pose_sq = _ut_pack(np.diag(1.0 / np.array([0.01, 0.01, 0.01, 3e-3, 3e-3, 3e-3])))    # physical weights: chi2 ~ dof
    exact = {0, 1, 4, 5}                      # estimates that start exactly on the wrap
    pnode, lnode = [], {}
    for k in range(n):
        est = truth_pose[k] if k in exact else pose_exmap(truth_pose[k], rng.normal(0.0, 0.05, 6))
        pn = node(0, est); pnode.append(pn)
        if k == 0:
            m6 = pose_vec(truth_pose[0]); m6[3] = -np.pi + 4e-7            # the other side of the wrap: e_yaw = -7e-7
            factor(F_POSE_PRIOR, pn, -1, m6, pose_sq)
        else:
            R1 = Rot.from_quat(truth_pose[k - 1][3:]); R2 = Rot.from_quat(truth_pose[k][3:])
            m6 = np.concatenate([R1.inv().apply(truth_pose[k][:3] - truth_pose[k - 1][:3]), (R1.inv() * R2).as_euler("ZYX")])
            m6 += rng.normal(0.0, 1.0, 6) * np.array([0.01, 0.01, 0.01, 3e-3, 3e-3, 3e-3])
            if k - 1 == 0:
                m6[3] = -np.pi + 3e-7          # true relative yaw +pi - 5e-7
            if k - 1 == 4:
                m6[3] = np.pi - 2e-7           # true relative yaw -pi + 4e-7
            factor(F_ODOMETRY, pnode[k - 1], pn, m6, pose_sq)
        seen = [0, 1 + (k % 2)] + ([2 - (k % 2)] if k % 3 == 0 else [])
        meas = {}
        for j in seen:
            meas[j] = plane_exmap(res_plane_local(truth_pose[k], planes[j]), rng.normal(0.0, 0.03, 3))
            if (k + j) % 2:
                meas[j] = -meas[j]             # double cover: the same plane, dq.w < 0
        for j in seen:
            if j not in lnode:
                lnode[j] = node(1, plane_exmap(planes[j], rng.normal(0.0, 0.05, 3)))
                if j == 0:
                    factor(F_PLANE_PRIOR, lnode[j], -1, planes[0], _ut_pack(np.eye(3) * 20.0))
        for j in seen:
            factor(F_PLANE_OBS, pn, lnode[j], meas[j], _ut_pack(np.eye(3) / 0.03))
    return synth.GraphSpec(name="pi_wrap_8p_3l", node_type=np.array(nt, dtype=np.int32), node_init=np.array(ni),
                           f_type=np.array(ft, dtype=np.int32), f_nodes=np.array(fn, dtype=np.int32), f_meas=np.array(fm),
                           f_sqrtinf=np.array(fs), meta={"factor_after_node": np.array(after, dtype=np.int64)})


def res_plane_local(pose, plane):
    R = Rot.from_quat(pose[3:]).as_matrix()
    u = np.concatenate([R.T @ plane[:3], [plane[:3] @ pose[:3] + plane[3]]])
    return u / np.linalg.norm(u)


def edge_main():
    """python oracle/numpy_ref.py edge  -- writes the round-5 edge-case fixtures only (the files main() writes stay untouched)"""
    out = os.path.join(ROOT, "tests", "golden")
    for spec in (dense_sqrtinf_spec(), pi_wrap_spec()):
        fx = fixture_for(spec)
        with open(os.path.join(out, f"{spec.name}.json"), "w") as f:
            json.dump(fx, f)
        print(spec.name, "chi2", fx["chi2_initial"], "->", fx["chi2_final"], "iters", fx["lm_iterations"],
              "rejected", sum(1 for _, _, a in fx["lm_trace"] if not a))


if __name__ == "__main__":
    edge_main() if sys.argv[1:] == ["edge"] else main()
