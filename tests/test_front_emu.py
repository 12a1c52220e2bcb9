"""csrc/pps_front_reg.h -- the register-tile elimination of one front (8-column panels: two chained 4 x 4 pivot blocks per LDS round
trip, two MFMAs per trailing tile) -- executed on the CPU by the wave emulation of tests/cpp/wave_emu.h and compared with numpy's
Cholesky.  The SAME source the GPU kernels include, compiled by g++ with 64 coroutines standing in for the lanes, so that an index
slip in the panel logic shows up here and not on the GPU box; the hardware run of the same cases is tests/test_gpu_fronts.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_dp = C.POINTER(C.c_double)


def _emu_lib(fused):
    """the emulation library; fused: built with -mfma -ffp-contract=fast, so that a - b * c is ONE operation on the host as it is in the GPU's
    `#pragma clang fp contract(fast)` regions (g++ ignores the pragma) -- what a bit-for-bit comparison of two forms needs, since the emulated
    MFMA accumulates with fused multiply-adds like the hardware"""
    src = os.path.join(ROOT, "tests", "cpp", "front_emu.cpp")
    out = os.path.join(ROOT, "tests", "cpp", "libfront_emu_fma.so" if fused else "libfront_emu.so")
    deps = [src, os.path.join(ROOT, "tests", "cpp", "wave_emu.h")] + [os.path.join(ROOT, "pop_up_slam_amd", "csrc", h) for h in ("pps_front_reg.h", "pps_regtile.h")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        fp = ["-mfma", "-ffp-contract=fast"] if fused else ["-ffp-contract=off"]
        subprocess.check_call(["g++", "-O2" if fused else "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-psabi", "-Wno-unknown-pragmas"] + fp +
                              ["-I" + os.path.join(ROOT, "pop_up_slam_amd", "csrc"), src, "-o", out])
    return C.CDLL(out)


def _runner(lib):
    lib.emu_front_factor.argtypes = [C.c_int] * 4 + [_dp, _dp, _dp, _dp, C.POINTER(C.c_longlong)]
    lib.emu_front_factor_w4.argtypes = lib.emu_front_factor.argtypes

    def run(tri, p, b, tiles=0, strip=False, w=8):
        fa = p + b + 1
        a = np.ascontiguousarray(tri, dtype=np.float64)
        assert a.size == fa * (fa + 1) // 2
        L = np.zeros((fa, p)); U = np.zeros((b + 1) * (b + 2) // 2); bad = C.c_double(); cnt = (C.c_longlong * 3)()
        rc = {8: lib.emu_front_factor, 4: lib.emu_front_factor_w4}[w](tiles, int(strip), p, b, a.ctypes.data_as(_dp), L.ctypes.data_as(_dp), U.ctypes.data_as(_dp), C.byref(bad), cnt)
        if rc != 0:
            raise ValueError(rc)
        return L, U, bad.value, list(cnt)
    return run


@pytest.fixture(scope="module")
def emu():
    return _runner(_emu_lib(False))


@pytest.fixture(scope="module")
def emu_fused():
    with open("/proc/cpuinfo") as f:
        if " fma " not in f.read():
            pytest.skip("host CPU without FMA: the fused build of the emulation cannot run")
    return _runner(_emu_lib(True))


def _front(p, b, seed):
    rng = np.random.default_rng(seed)
    f = p + b
    M = rng.standard_normal((f, f + 20))
    H = M @ M.T + f * np.eye(f)
    rhs = rng.standard_normal(f)
    full = np.zeros((f + 1, f + 1)); full[:f, :f] = H; full[f, :f] = rhs; full[f, f] = 7.0
    tri = np.concatenate([full[i, :i + 1] for i in range(f + 1)])
    return H, rhs, tri


def _check(run, p, b, tiles, strip, seed=0, w=8):
    H, rhs, tri = _front(p, b, seed)
    f = p + b
    L, U, bad, cnt = run(tri, p, b, tiles, strip, w)
    assert bad == 0.0
    LA = np.linalg.cholesky(H[:p, :p])
    LB = np.linalg.solve(LA, H[p:, :p].T).T
    y = np.linalg.solve(LA, rhs[:p])
    scale = np.abs(LA).max()
    assert np.abs(np.tril(L[:p]) - LA).max() <= 1e-12 * scale
    if b:
        assert np.abs(L[p:f] - LB).max() <= 1e-12 * scale
    assert np.abs(L[f] - y).max() <= 1e-12 * max(1.0, np.abs(y).max())
    S = H[p:, p:] - LB @ LB.T
    r = rhs[p:] - LB @ y
    Ut = np.zeros((b + 1, b + 1)); Ut[np.tril_indices(b + 1)] = U
    if b:
        assert np.abs(np.tril(Ut[:b, :b]) - np.tril(S)).max() <= 1e-11 * np.abs(S).max()
        assert np.abs(Ut[b, :b] - r).max() <= 1e-11 * max(1.0, np.abs(r).max())
    return cnt


def test_every_pivot_count_up_to_64_rows(emu):
    """p = 1 .. 63 (every panel width 1 .. 8, panels that end a tile column or stop short of it) with boundary blocks of 0, 1, a
    random size and the largest one; the tile count the kernels pick and every larger one"""
    rng = np.random.default_rng(5)
    for p in range(1, 64):
        for b in sorted({0, 1, int(rng.integers(0, 64 - p)), 63 - p}):
            if p + b + 1 > 64:
                continue
            _check(emu, p, b, 0, False, seed=100 * p + b)
    for p, b in [(4, 8), (6, 8), (8, 8), (9, 20), (15, 16), (15, 33), (18, 30), (27, 20)]:
        for tiles in (2, 3, 4):
            if p + b <= 16 * tiles:
                _check(emu, p, b, tiles, False, seed=tiles)


def test_fronts_of_65_to_80_rows_both_ways(emu):
    """fifth tile row in registers / four tile rows + LDS strip"""
    for p, b in [(4, 61), (6, 70), (8, 70), (13, 60), (21, 50), (27, 52), (24, 55), (48, 31), (60, 19), (64, 15)]:
        _check(emu, p, b, 5, False, seed=p)
        _check(emu, p, b, 4, True, seed=p)


def test_four_column_panels(emu):
    """the register-only band kernels run the same code with one pivot block per panel step (kPanelWBand = 4, pps_k3.hip: a lone wave per
    SIMD is bound by the pivot chain, and two 4-column steps are shorter than one 8-column step there)"""
    for p, b in [(1, 2), (4, 8), (6, 8), (13, 0), (15, 33), (18, 30), (27, 36), (33, 30), (48, 15), (63, 0)]:
        _check(emu, p, b, 0, False, seed=p, w=4)
    for p, b in [(6, 70), (21, 50), (48, 31), (64, 15)]:
        _check(emu, p, b, 5, False, seed=p, w=4)
        _check(emu, p, b, 4, True, seed=p, w=4)


def test_four_and_eight_column_panels_agree_bit_for_bit(emu_fused):
    """per entry the two panel widths perform the same operations in the same order (the emulated MFMA accumulates with fused multiply-adds in k
    order like the hardware): a band kernel (W = 4) and a level kernel (W = 8) leave the same bits"""
    for p, b in [(1, 2), (6, 8), (13, 0), (15, 33), (18, 30), (27, 36), (33, 30), (48, 15), (63, 0)]:
        _, _, tri = _front(p, b, 100 * p + b)
        L4, U4, bad4, _ = emu_fused(tri, p, b, 0, False, 4)
        L8, U8, bad8, _ = emu_fused(tri, p, b, 0, False, 8)
        ok = np.tril(np.ones((p + b + 1, p), dtype=bool))          # (above the diagonal of L_A: unspecified in both)
        assert np.array_equal(L4[ok], L8[ok]) and np.array_equal(U4, U8) and bad4 == bad8, (p, b)


def test_cross_lane_operations_per_front(emu):
    """what the 8-column panel buys, counted by the emulation: a separator front of a C2 tree (p = 15, b = 33) is two panel steps --
    7 wave barriers and 24 MFMAs -- where the 4-column form took four steps (13 barriers)"""
    cnt = _check(emu, 15, 33, 0, False)
    assert cnt[2] == 7 and cnt[1] == 24


def test_not_positive_definite_in_either_pivot_block(emu):
    _, _, tri = _front(14, 10, 1)
    t = tri.copy(); t[2 * 3 // 2 + 2] = -1.0                     # H[2][2]: first block of the first panel
    assert emu(t, 14, 10)[2] == 1.0
    t = tri.copy(); t[13 * 14 // 2 + 13] = -1e6                  # H[13][13]: second block of the second panel
    assert emu(t, 14, 10)[2] == 1.0
    assert emu(tri, 14, 10)[2] == 0.0
