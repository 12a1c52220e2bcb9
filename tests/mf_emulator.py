"""numpy emulation of the device-side multifrontal solve, driven ONLY by the flat arrays that
pps_analysis_dump exports.  Used by the CPU tests to validate the host-side symbolic analysis
(ordering, fronts, scatter maps, H-block contribution lists) without a GPU."""
import numpy as np


def solve_with_analysis(A, Jbuf, lam):
    """A: parsed dump (pop_up_slam_amd.parse_analysis_dump); Jbuf: flat J buffer in the device layout.
    Returns delta in elimination order."""
    nseg = A["n_segs"]
    H = np.zeros(A["H_size"])
    ctr = A["contrib"].reshape(-1, 4)
    for s in range(nseg):
        blk = A["seg_blk"][s]
        rows, cols, size = A["blk_rows"][blk], A["blk_cols"][blk], A["blk_size"][blk]
        acc = np.zeros(size)
        for c in range(A["seg_c0"][s], A["seg_c0"][s] + A["seg_cnt"][s]):
            jv, ju, roff, m = ctr[c]
            Jv = Jbuf[jv:jv + m * rows].reshape(m, rows)
            Ju = Jbuf[ju:ju + m * cols].reshape(m, cols)
            acc[:rows * cols] += (Jv.T @ Ju).ravel()
            if size > rows * cols:
                acc[rows * cols:] -= Jv.T @ Jbuf[roff:roff + m]
        H[A["seg_hoff"][s]:A["seg_hoff"][s] + size] = acc
    F_ = A["n_fronts"]
    Ls, Us = [None] * F_, [None] * F_
    for s in range(F_):
        p, b = A["f_p"][s], A["f_b"][s]
        f = p + b
        fa = f + 1
        Fm = np.zeros((fa, fa))
        for a in range(A["f_asm_off"][s], A["f_asm_off"][s + 1]):
            blk, lrow, lcol = A["asm_blk"][a], A["asm_lrow"][a], A["asm_lcol"][a]
            rows, cols, size, ns = A["blk_rows"][blk], A["blk_cols"][blk], A["blk_size"][blk], A["blk_nseg"][blk]
            v = np.zeros(size)
            for q in range(ns):
                v += H[A["blk_hoff"][blk] + q * size: A["blk_hoff"][blk] + (q + 1) * size]
            B = v[:rows * cols].reshape(rows, cols).copy()
            diag = size > rows * cols
            if diag:
                B[np.diag_indices(rows)] *= (1.0 + lam)
                B = np.tril(B)
                Fm[f, lcol:lcol + rows] += v[rows * cols:]
            assert lrow >= lcol
            Fm[lrow:lrow + rows, lcol:lcol + cols] += B
        for ci in range(A["f_child_off"][s], A["f_child_off"][s + 1]):
            c = A["child"][ci]
            cm = A["cmap"][A["f_cmap_off"][c]:A["f_cmap_off"][c + 1]]
            assert len(cm) == A["f_b"][c] + 1
            assert np.all(np.diff(cm) > 0), "child map must be increasing"
            U = Us[c]
            for i in range(len(cm)):
                for j in range(i + 1):
                    Fm[cm[i], cm[j]] += U[i, j]
        # symmetrise the lower triangle and eliminate the p pivots
        Fs = np.tril(Fm) + np.tril(Fm, -1).T
        Aa = Fs[:p, :p]
        La = np.linalg.cholesky(Aa)
        Bb = Fs[p:, :p]
        Lb = np.linalg.solve(La, Bb.T).T
        Us[s] = Fs[p:, p:] - Lb @ Lb.T
        Ls[s] = np.vstack([La, Lb])
    delta = np.zeros(A["n_scalars"])
    for s in range(F_ - 1, -1, -1):
        p, b = A["f_p"][s], A["f_b"][s]
        L = Ls[s]
        y = L[p + b, :]
        bi = A["bidx"][A["f_bidx_off"][s]:A["f_bidx_off"][s + 1]]
        t = y - L[p:p + b, :].T @ delta[bi]
        x = np.linalg.solve(L[:p, :].T, t)
        delta[A["f_poff"][s]:A["f_poff"][s] + p] = x
    return delta
