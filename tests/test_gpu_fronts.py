"""The register-tile elimination of one front (pps_debug_front_factor) against numpy's Cholesky: every tile count the band kernels
use, fronts of one panel and of many (8-column panels since round 4: every panel width 1 .. 8, panels that end a tile column or stop
short of it), the LDS strip and the fifth tile row for fronts of 65 .. 80 rows.  tests/test_front_emu.py runs the same source on the
CPU (wave emulation) for every pivot count.  The case that motivated the
test: with fifteen accumulator tiles hipcc 7.2 miscompiled the panel loop once it peeled the first panel (rows 8 and up of every
later panel wrong, a whole frame-loop stage not positive definite); no graph-level test isolates a single front."""
import numpy as np
import pytest

import pop_up_slam_amd as P

pytestmark = pytest.mark.gpu


def _front(p, b, seed):
    rng = np.random.default_rng(seed)
    f = p + b
    M = rng.standard_normal((f, f + 20))
    H = M @ M.T + f * np.eye(f)
    rhs = rng.standard_normal(f)
    full = np.zeros((f + 1, f + 1)); full[:f, :f] = H; full[f, :f] = rhs; full[f, f] = 7.0
    tri = np.concatenate([full[i, :i + 1] for i in range(f + 1)])
    return H, rhs, tri


def _check(p, b, tiles, strip, seed=0):
    H, rhs, tri = _front(p, b, seed)
    f = p + b
    L, U, bad = P.debug_front_factor(tri, p, b, tiles=tiles, strip=strip)
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


@pytest.mark.parametrize("p,b", [(4, 8), (6, 8), (15, 16), (15, 33), (18, 30), (27, 36), (6, 57), (33, 30), (48, 15), (63, 0),
                                 (1, 2), (5, 3), (8, 8), (9, 20), (12, 0), (13, 0), (16, 10), (17, 30), (24, 24), (40, 23), (56, 7)])
def test_fronts_up_to_64_rows(built, p, b):
    """the solver's own choice of tile rows, then every larger tile count that holds the front"""
    fa = p + b + 1
    _check(p, b, 0, False)
    for tiles in (2, 3, 4):
        if p + b <= 16 * tiles:
            _check(p, b, tiles, False, seed=tiles)
    assert fa <= 64


@pytest.mark.parametrize("p,b", [(4, 61), (6, 70), (8, 70), (13, 60), (21, 50), (27, 52), (24, 55), (48, 31), (60, 19), (64, 15)])
def test_fronts_of_65_to_80_rows_both_ways(built, p, b):
    """fifteen register tiles (what the band kernels run) and four tile rows + LDS strip: same factor, same update matrix"""
    _check(p, b, 5, False)
    _check(p, b, 4, True)
    H, rhs, tri = _front(p, b, 3)
    L5, U5, _ = P.debug_front_factor(tri, p, b, tiles=5)
    L4, U4, _ = P.debug_front_factor(tri, p, b, tiles=4, strip=True)
    f = p + b
    np.testing.assert_allclose(np.tril(L5[:p]), np.tril(L4[:p]), rtol=0, atol=1e-12 * np.abs(L4).max())
    np.testing.assert_allclose(L5[p:], L4[p:], rtol=0, atol=1e-12 * np.abs(L4).max())
    np.testing.assert_allclose(U5, U4, rtol=0, atol=1e-11 * np.abs(U4).max())


def test_not_positive_definite_is_reported(built):
    H, rhs, tri = _front(6, 10, 1)
    tri = tri.copy(); tri[2 * 3 // 2 + 2] = -1.0            # H[2][2] < 0
    _, _, bad = P.debug_front_factor(tri, 6, 10)
    assert bad == 1.0


def test_arguments_no_path_takes(built):
    _, _, tri = _front(6, 10, 1)
    for kw in (dict(tiles=5), dict(tiles=4, strip=True), dict(tiles=1)):
        with pytest.raises(P.PpsError):
            P.debug_front_factor(tri, 6, 10, **kw)
    _, _, tri = _front(20, 50, 1)
    for kw in (dict(tiles=3), dict(tiles=4)):
        with pytest.raises(P.PpsError):
            P.debug_front_factor(tri, 20, 50, **kw)
