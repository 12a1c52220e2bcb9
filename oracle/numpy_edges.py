"""Independent evaluation of the ground-edge selection (test infrastructure, NOT product code).

A third formulation next to oracle/pps_edges_oracle.c and the product's host stage, written against the same
published behaviour (OpenCV dilate / erode / nearest resize, skimage.measure.find_contours, intervaltree) and the
reference call sites pop_up_wall/libs/select_edge.cpp:66-409 and pop_up_python/.../pop_up_fun.py:85-204, but with
different machinery: scipy.ndimage for the morphology, vectorised numpy for the marching-squares cells, Python
dictionaries and deques for the contour linking, a plain list of (begin, end, payload) tuples for the interval set.
Running it (python -m oracle.numpy_edges) writes tests/golden/edges_cases.json, which tests/test_oracle_edges.py
checks the C oracle against.  float32 arithmetic is kept where the reference computes in float (Eigen MatrixXf);
numpy's arctan2 / sqrt may differ from libm in the last bit, which only matters within 1e-5 of a threshold.
"""
import json
import os
import sys
from collections import deque

import numpy as np
import scipy.ndimage as ndi

F = np.float32
DEFAULTS = dict(downsample_contour=0, dilation_distance=11, erosion_distance=11, pre_vertical_thre=15.0, pre_minium_len=15.0,
                pre_contour_close_thre=50.0, interval_overlap_thre=20.0, post_short_thre=30.0, post_bind_dist_thre=10.0,
                post_merge_dist_thre=20.0, post_merge_angle_thre=10.0, post_extend_thre=15.0, pre_boundary_thre=5.0,
                pre_merge_angle_thre=10.0, pre_merge_dist_thre=10.0, pre_proj_angle_thre=20.0, pre_proj_cover_thre=0.6,
                pre_proj_cover_large_thre=0.8, pre_proj_dist_thre=100.0)


def preprocess(label, prm):
    lab = np.asarray(label, np.uint8)
    if prm["downsample_contour"]:
        h, w = lab.shape
        hh, hw = int(np.rint(h * 0.5)), int(np.rint(w * 0.5))
        lab = lab[np.minimum(2 * np.arange(hh), h - 1)][:, np.minimum(2 * np.arange(hw), w - 1)]
        kd = ke = 8
    else:
        kd, ke = prm["dilation_distance"], prm["erosion_distance"]
    grown = ndi.maximum_filter(lab, size=(kd, kd), mode="constant", cval=0)
    closed = ndi.minimum_filter(grown, size=(ke, ke), mode="constant", cval=255)
    return (255 - closed).astype(np.uint8)


# marching squares: from -> to edge per case (T top, B bottom, L left, R right), 'low' vertex connectivity
_ARCS = {1: ["TL"], 2: ["RT"], 3: ["RL"], 4: ["LB"], 5: ["TB"], 6: ["RT", "LB"], 7: ["RB"], 8: ["BR"], 9: ["TL", "BR"],
         10: ["BT"], 11: ["BL"], 12: ["LR"], 13: ["TR"], 14: ["LT"]}


def cell_arcs(pre):
    hi = np.asarray(pre) > 0
    ul, ur, ll, lr = hi[:-1, :-1], hi[:-1, 1:], hi[1:, :-1], hi[1:, 1:]
    code = ul * 1 + ur * 2 + ll * 4 + lr * 8
    arcs = []
    for r, c in zip(*np.nonzero((code > 0) & (code < 15))):
        r, c = int(r), int(c)
        # level 0: the crossing of an edge lies on its zero end
        pt = {"T": (r, c + 1 if ul[r, c] else c), "B": (r + 1, c + 1 if ll[r, c] else c),
              "L": (r + 1 if ul[r, c] else r, c), "R": (r + 1 if ur[r, c] else r, c + 1)}
        arcs += [(pt[a[0]], pt[a[1]]) for a in _ARCS[int(code[r, c])]]
    return arcs


def link(arcs):
    """contours in order of creation: a new arc extends the contour that ends at its start and / or the one that starts
    at its end; when it joins two contours the older one survives"""
    born = 0
    alive, first_of, last_of = {}, {}, {}
    for a, b in arcs:
        if a == b:
            continue
        nxt = first_of.get(b)     # contour starting where the arc ends
        prv = last_of.get(a)      # contour ending where the arc starts
        if nxt is not None and prv is not None:
            if nxt is prv:
                prv[1].append(b)
                del first_of[b], last_of[a]
            elif nxt[0] > prv[0]:
                prv[1].extend(nxt[1])
                del first_of[b], last_of[nxt[1][-1]], alive[nxt[0]], last_of[a]
                last_of[prv[1][-1]] = prv
            else:
                nxt[1].extendleft(reversed(prv[1]))
                del first_of[prv[1][0]], last_of[a], alive[prv[0]], first_of[b]
                first_of[nxt[1][0]] = nxt
        elif nxt is None and prv is None:
            born += 1
            rec = (born, deque([a, b]))
            alive[born] = rec
            first_of[a] = rec
            last_of[b] = rec
        elif nxt is not None:
            nxt[1].appendleft(a)
            del first_of[b]
            first_of[a] = nxt
        else:
            prv[1].append(b)
            del last_of[a]
            last_of[b] = prv
    return [np.array(alive[k][1], dtype=float) for k in sorted(alive)]


def ground_contour(pre, downsample):
    cs = link(cell_arcs(pre))
    if not cs:
        return np.zeros((0, 2), F), 0, 0
    gap = [np.linalg.norm(c[0] - c[-1]) for c in cs]
    c = cs[int(np.argmax(gap))]
    xy = c[:, ::-1][0:-1:20].astype(F) * F(2 if downsample else 1)
    return xy, len(cs), len(c)


# ---- float32 geometry -------------------------------------------------------------------------------------------------
def _len(v):
    return np.sqrt(v[0] * v[0] + v[1] * v[1])


def _angle(l):
    a = F(float(np.arctan2(l[3] - l[1], l[2] - l[0])) / 3.14159265 * 180)
    return a - F(180) if a > 90 else (a + F(180) if a < -90 else a)


def _gap(a, b):
    d = abs(a - b)
    return min(d, F(180) - d)


def _foot(bg, ed, q):
    ln = _len(ed - bg)
    if ln < 0.001:
        return _len(q - bg), None
    t = ((q - bg)[0] * (ed - bg)[0] + (q - bg)[1] * (ed - bg)[1]) / ln / ln
    return _len(q - (bg + t * (ed - bg))), t


def _far_from_contour(l, cxy):
    worst = F(-np.inf)
    for k in range(10):
        s = l[:2] + F(k / 10.0) * (l[2:] - l[:2])
        d = cxy - s
        worst = max(worst, np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).min() if len(cxy) else F(np.inf))
    return worst


def _to_border(pt, dr, w, h):
    """first image border the ray pt + lam * dr reaches, tried in the reference's order: top, bottom, right, left"""
    def attempt(lam, along_x):
        if lam >= 0:
            hit = pt + lam * dr
            v, lim = (hit[0], w - 1) if along_x else (hit[1], h - 1)
            if 0 <= int(v) <= lim:
                return hit
        return None
    tries = []
    if dr[1] < 0: tries.append((F((0.0 - float(pt[1])) / float(dr[1])), True))
    if dr[1] > 0: tries.append((F((h - 1.0 - float(pt[1])) / float(dr[1])), True))
    if dr[0] > 0: tries.append((F((w - 1.0 - float(pt[0])) / float(dr[0])), False))
    if dr[0] < 0: tries.append((F((0.0 - float(pt[0])) / float(dr[0])), False))
    for lam, along_x in tries:
        hit = attempt(lam, along_x)
        if hit is not None:
            return hit
    return np.array([-1, -1], F)


# ---- interval cover -----------------------------------------------------------------------------------------------------
def cover_optimize(lines, overlap_thre):
    lines = [tuple(float(x) for x in l) for l in np.asarray(lines, F).reshape(-1, 4)]
    if not lines:
        return np.zeros((0, 4), F)
    length = [float(_len(np.array([l[2] - l[0], l[3] - l[1]], F))) for l in lines]
    todo = list(range(len(lines)))
    cover = set()

    def admit(cands):
        best = max(cands, key=lambda k: (length[todo[k]], -k))     # longest, first on ties
        cover.add((lines[todo[best]][0], lines[todo[best]][2], lines[todo[best]]))
        todo.pop(best)

    admit(list(range(len(todo))))
    while todo:
        ok = []
        for k, idx in enumerate(todo):
            q0, q1 = lines[idx][0], lines[idx][2]
            shared = sum(min(q1 - b, e - q0, abs(q1 - q0), e - b) for b, e, _ in cover if q0 < q1 and b < q1 and e > q0)
            if shared < overlap_thre:
                ok.append(k)
        if not ok:
            break
        admit(ok)
    before = len(cover)
    cuts = sorted({x for b, e, _ in cover for x in (b, e)})
    if len(cuts) > 2:
        cover = {(lo, hi, d) for lo, hi in zip(cuts[:-1], cuts[1:]) for b, e, d in cover if b <= lo < e}
    split = len(cover) != before
    if split:
        srt = sorted(cover)
        drop = set()
        for i in range(len(srt)):
            for j in range(i + 1, len(srt)):
                if srt[i][:2] == srt[j][:2]:
                    drop.add(srt[i] if (srt[i][2][2] - srt[i][2][0]) < (srt[j][2][2] - srt[j][2][0]) else srt[j])
        cover -= drop
        for _ in range(100):
            srt = sorted(cover)
            pair = next(((p, q) for i, p in enumerate(srt) for q in srt[i + 1:] if (p[0] == q[1] or p[1] == q[0]) and p[2] == q[2]), None)
            if pair is None:
                break
            p, q = pair
            cover -= {p, q}
            cover.add((min(p[0], q[0]), max(p[1], q[1]), q[2]))
    out = []
    for b, e, raw in sorted(cover):
        if split and (b != raw[0] or e != raw[2]):
            f1, f2 = (b - raw[0]) / (raw[2] - raw[0]), (e - raw[0]) / (raw[2] - raw[0])
            out.append([b, int(f1 * (raw[3] - raw[1]) + raw[1]), e, int(f2 * (raw[3] - raw[1]) + raw[1])])
        else:
            out.append(list(raw))
    return np.array(out, F).reshape(-1, 4)


# ---- the selection ----------------------------------------------------------------------------------------------------
def select(label, lsd, **kw):
    prm = dict(DEFAULTS); prm.update(kw)
    lab = np.asarray(label, np.uint8)
    H, W = lab.shape
    pre = preprocess(lab, prm)
    cxy, _, _ = ground_contour(pre, prm["downsample_contour"])
    empty = (np.zeros((0, 4), F), np.zeros((0, 4), F), np.zeros(0, F))
    if len(cxy) == 0:
        return empty
    keep = []
    m = prm["pre_boundary_thre"]
    for l in np.asarray(lsd, F).reshape(-1, 4):
        if _len(l[2:] - l[:2]) < prm["pre_minium_len"]:
            continue
        if (l[0] < m and l[2] < m) or (l[0] > W - m and l[2] > W - m) or (l[1] < m and l[3] < m) or (l[1] > H - m and l[3] > H - m):
            continue
        if not abs(abs(_angle(l)) - F(90)) > prm["pre_vertical_thre"]:
            continue
        if not _far_from_contour(l, cxy) < prm["pre_contour_close_thre"]:
            continue
        keep.append(l[[2, 3, 0, 1]].copy() if l[0] > l[2] else l.copy())
    changed = True
    rounds = 0
    while changed and rounds < 100:          # chain near-collinear pieces end to start
        rounds += 1; changed = False
        ang = [_angle(l) for l in keep]
        for a in range(len(keep)):
            for b in range(a + 1, len(keep)):
                if _gap(ang[a], ang[b]) < prm["pre_merge_angle_thre"]:
                    if _len(keep[a][2:] - keep[b][:2]) < prm["pre_merge_dist_thre"]:
                        keep[a][2:] = keep[b][2:]
                    elif _len(keep[b][2:] - keep[a][:2]) < prm["pre_merge_dist_thre"]:
                        keep[a][:2] = keep[b][:2]
                    else:
                        continue
                    keep.pop(b); changed = True
                    break
            if changed:
                break
    changed = True
    rounds = 0
    while changed and rounds < 100:          # mutually covering near-parallel lines: one goes
        rounds += 1; changed = False
        ang = [_angle(l) for l in keep]
        for a in range(len(keep)):
            for b in range(a + 1, len(keep)):
                if not _gap(ang[a], ang[b]) < prm["pre_proj_angle_thre"]:
                    continue
                A, B = keep[a], keep[b]
                (d1, t1), (d2, t2) = _foot(B[:2], B[2:], A[:2]), _foot(B[:2], B[2:], A[2:])
                (d3, t3), (d4, t4) = _foot(A[:2], A[2:], B[:2]), _foot(A[:2], A[2:], B[2:])
                if not all(d < prm["pre_proj_dist_thre"] for d in (d1, d2, d3, d4)):
                    continue
                clip = lambda t: F(-1) if t is None else min(max(t, F(0)), F(1))
                cab, cba = abs(clip(t1) - clip(t2)), abs(clip(t3) - clip(t4))
                if not (cab > prm["pre_proj_cover_thre"] or cba > prm["pre_proj_cover_thre"]):
                    continue
                if min(cab, cba) < prm["pre_proj_cover_large_thre"]:
                    victim = b if cab > cba else a
                else:
                    victim = a if _far_from_contour(A, cxy) > _far_from_contour(B, cxy) else b
                keep.pop(victim); changed = True
                break
            if changed:
                break
    if not keep:
        return empty
    segs = [s for s in cover_optimize(np.array(keep, F), prm["interval_overlap_thre"]) if _len(s[2:] - s[:2]) > prm["post_short_thre"]]
    segs = [s.copy() for s in segs]
    if not segs:
        return empty
    for _ in range(100):
        moved = False
        for s in range(len(segs) - 1):
            A, B = segs[s], segs[s + 1]
            if (A[2] != B[0] or A[3] != B[1]) and _len(A[2:] - B[:2]) < prm["post_bind_dist_thre"]:
                mid = ((A[2:] + B[:2]) / F(2)).astype(np.int32).astype(F)
                A[2:] = mid; B[:2] = mid; moved = True
        if not moved:
            break
    for _ in range(100):
        merged = False
        for s in range(len(segs) - 1):
            A, B = segs[s], segs[s + 1]
            if _gap(_angle(A), _angle(B)) < prm["post_merge_angle_thre"]:
                lim = prm["post_merge_dist_thre"]
                line_dist = lambda bg, ed, q: _foot(bg, ed, q)[0]
                if (line_dist(B[:2], B[2:], A[:2]) < lim and line_dist(B[:2], B[2:], A[2:]) < lim) or \
                   (line_dist(A[:2], A[2:], B[:2]) < lim and line_dist(A[:2], A[2:], B[2:]) < lim):
                    A[2:] = B[2:]; segs.pop(s + 1); merged = True
                    break
        if not merged:
            break
    first, last = segs[0], segs[-1]
    start0, end0 = first[:2].copy(), last[2:].copy()
    s_hit = _to_border(first[2:].copy(), first[:2] - first[2:], W, H)
    e_hit = _to_border(last[:2].copy(), last[2:] - last[:2], W, H)
    first[:2] = s_hit.astype(np.int32).astype(F)
    last[2:] = e_hit.astype(np.int32).astype(F)
    d_first, d_last = _far_from_contour(first, cxy), _far_from_contour(last, cxy)
    if d_first > prm["post_extend_thre"]: first[:2] = start0
    if d_last > prm["post_extend_thre"]: last[2:] = end0
    closed, where = [], []
    for s, sg in enumerate(segs):
        if s and (segs[s - 1][2] != sg[0] or segs[s - 1][3] != sg[1]):
            closed.append([segs[s - 1][2], segs[s - 1][3], sg[0], sg[1]])
        where.append(len(closed))
        closed.append(list(sg))
    return np.array(segs, F).reshape(-1, 4), np.array(closed, F).reshape(-1, 4), np.array(where, F)


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests"))
    import edge_helpers as E
    configs = [dict(), dict(pre_contour_close_thre=400.0, post_short_thre=20.0, post_bind_dist_thre=30.0, post_merge_dist_thre=50.0,
                            post_merge_angle_thre=20.0, post_extend_thre=0.0),
               dict(downsample_contour=1, post_short_thre=20.0, post_bind_dist_thre=15.0, post_merge_dist_thre=10.0, post_extend_thre=50.0),
               dict(interval_overlap_thre=60.0, pre_merge_dist_thre=3.0, pre_proj_cover_thre=2.0)]
    cases = []
    for seed in range(16):
        kw = configs[seed % 4]
        lab, lines = E.random_scene(500 + seed, n_knots=3 + seed % 4, holes=8)
        prm = dict(DEFAULTS); prm.update(kw)
        pre = preprocess(lab, prm)
        cxy, nc, npnt = ground_contour(pre, prm["downsample_contour"])
        o, c, w = select(lab, lines, **kw)
        cases.append(dict(seed=500 + seed, n_knots=3 + seed % 4, holes=8, params=kw, n_contours=nc, n_points=npnt,
                          contour=cxy.ravel().tolist(), open=o.ravel().tolist(), closed=c.ravel().tolist(), open_in_closed=w.tolist()))
    with open(os.path.join(root, "tests", "golden", "edges_cases.json"), "w") as f:
        json.dump(cases, f)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
