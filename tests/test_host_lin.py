"""K1's thread form evaluates once what a central-difference step provably leaves bit-identical (csrc/pps_lin.h: R and R^T n for the
translation and plane columns of a plane observation; the rotation matrix -> quaternion -> Euler chain for the twelve translation
steps of an odometry edge).  tests/cpp/lin_host.cpp holds the plain form -- 2 n + 1 complete evaluations through exmap,
numericalDiff.cpp:41-87 statement by statement -- and both are compiled for the host without multiply-add contraction: the two
records must agree to the last bit, on random states and on the axis-aligned ones a corridor graph is full of."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_dp = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def lin():
    src = os.path.join(ROOT, "tests", "cpp", "lin_host.cpp")
    out = os.path.join(ROOT, "tests", "cpp", "liblin_host.so")
    deps = [src] + [os.path.join(ROOT, "pop_up_slam_amd", "csrc", h) for h in ("pps_lin.h", "pps_geom.h")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-ffp-contract=off",
                               "-I" + os.path.join(ROOT, "pop_up_slam_amd", "csrc"), src, "-o", out])
    lib = C.CDLL(out)
    lib.lin_host_compare.restype = C.c_double
    lib.lin_host_compare.argtypes = [C.c_int] + [_dp] * 6

    def run(kind, a, b, ms, w):
        n = (30, 78, 42)[kind]
        arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (a, b, ms, w)]
        s = np.zeros(n); p = np.zeros(n)
        worst = lib.lin_host_compare(kind, *[x.ctypes.data_as(_dp) for x in arrs], s.ctypes.data_as(_dp), p.ctypes.data_as(_dp))
        return worst, s, p
    return run


def _pose(rng, axis_aligned=False):
    if axis_aligned:
        yaw = rng.integers(0, 4) * np.pi / 2
        q = np.array([0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2)])
    else:
        q = rng.standard_normal(4); q /= np.linalg.norm(q)
    return np.concatenate([rng.standard_normal(3) * 5, q])


def _plane(rng, axis_aligned=False):
    if axis_aligned:
        n = np.zeros(4); n[rng.integers(0, 3)] = rng.choice([-1.0, 1.0]); n[3] = rng.standard_normal() * 3
        return n / np.linalg.norm(n)
    n = rng.standard_normal(4)
    return n / np.linalg.norm(n)


def _ut(rng, m):
    U = np.triu(rng.standard_normal((m, m))) + 3 * np.eye(m)
    return np.concatenate([U[i, i:] for i in range(m)])


@pytest.mark.parametrize("axis_aligned", [False, True])
def test_structured_central_differences_equal_the_plain_ones_bit_for_bit(lin, axis_aligned):
    rng = np.random.default_rng(11 + axis_aligned)
    seen = 0.0
    for _ in range(400):
        w3, w6 = _ut(rng, 3), _ut(rng, 6)
        worst, s, _ = lin(0, _pose(rng, axis_aligned), _plane(rng, axis_aligned), _plane(rng, axis_aligned), w3)
        assert worst == 0.0
        seen = max(seen, np.abs(s).max())
        worst, s, _ = lin(1, _pose(rng, axis_aligned), _pose(rng, axis_aligned), rng.standard_normal(6), w6)
        assert worst == 0.0
        # the exact zeros of the reference: angle rows against translation columns, translation rows against the second pose's rotation
        J1, J2 = s[:36].reshape(6, 6), s[36:72].reshape(6, 6)
        assert np.all(J1[3:, :3] == 0.0) and np.all(J2[3:, :3] == 0.0)
        worst, s, _ = lin(2, _pose(rng, axis_aligned), np.zeros(4), rng.standard_normal(6), w6)
        assert worst == 0.0
    assert seen > 0.1          # (the records are not all zeros)
