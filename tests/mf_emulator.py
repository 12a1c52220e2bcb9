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
            m = m & 255                      # (bit 8: the contribution also names a product record in ju -- this emulation multiplies the Jacobians)
            Jv = Jbuf[jv:jv + m * rows].reshape(m, rows)
            Ju = Jv if size > rows * cols else Jbuf[ju:ju + m * cols].reshape(m, cols)
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
        delta[A["pidx"][A["f_poff"][s]:A["f_poff"][s] + p]] = x
    return delta


def _tri(i):
    return i * (i + 1) // 2


def solve_with_band_schedule(A, Jbuf, lam, Pbuf=None):
    """Emulates the wave-per-front band kernels from the arrays THEY read: packed segment / front / child
    records, the front-ordered H (Hf) with its flat gather targets, the packed extend-add targets and the
    stage -> group -> local level -> front schedule.  Returns delta in elimination order."""
    srec = A["srec"].reshape(-1, 8)
    ctr = A["contrib"].reshape(-1, 4)
    H = np.zeros(A["H_size"]); Hf = np.zeros(len(A["el_src"]))
    for rows, cols, size, c0, cnt, hoff, doff, nsegb in srec:
        acc = np.zeros(size)
        for c in range(c0, c0 + cnt):
            jv, ju, roff, m = ctr[c]
            if m >= 256 and Pbuf is not None:      # a plane observation's product record (what K2 sums since round 4)
                assert size > rows * cols
                acc += Pbuf[ju:ju + size]
                continue
            m = m & 255
            Jv = Jbuf[jv:jv + m * rows].reshape(m, rows)
            Ju = Jv if size > rows * cols else Jbuf[ju:ju + m * cols].reshape(m, cols)
            acc[:rows * cols] += (Jv.T @ Ju).ravel()
            if size > rows * cols:
                acc[rows * cols:] -= Jv.T @ Jbuf[roff:roff + m]
        H[hoff:hoff + size] = acc
        if nsegb == 1:
            dst = A["blk_dst"][doff:doff + size]
            Hf[dst[dst >= 0]] = acc[dst >= 0]
    for blk in np.where(A["blk_nseg"] > 1)[0]:                      # k_hreduce
        size, ns, ho = A["blk_size"][blk], A["blk_nseg"][blk], A["blk_hoff"][blk]
        v = H[ho:ho + ns * size].reshape(ns, size).sum(axis=0)
        dst = A["blk_dst"][A["blk_doff"][blk]:A["blk_doff"][blk] + size]
        Hf[dst[dst >= 0]] = v[dst >= 0]
    frec = A["frec"].reshape(-1, 16); crec = A["crec"].reshape(-1, 8)
    Lst, Ust, done = {}, {}, set()
    order = []
    for st in range(A["n_stages"]):
        for g in range(A["stage_grp_off"][st], A["stage_grp_off"][st + 1]):
            for l in range(A["grp_lvl_off"][g], A["grp_lvl_off"][g + 1]):
                for i in range(A["glvl_front_off"][l], A["glvl_front_off"][l + 1]):
                    order.append(i)
                    r = frec[i]
                    s, p, b, e0, e1, cr0, nch = r[0], r[1], r[2], r[3], r[4], r[5], r[6]
                    assert s == A["glvl_fronts"][i] and p == A["f_p"][s] and b == A["f_b"][s]
                    f = p + b; fa = f + 1
                    F = np.zeros(_tri(fa))
                    tg = A["el_tgt"][e0:e1]
                    v = Hf[e0:e1] * np.where(tg & (1 << 30), 1.0 + lam, 1.0)
                    np.add.at(F, tg & 0x3fffffff, v)
                    for cj in range(nch):
                        n, ulo, uhi, elo, ehi, c = crec[cr0 + cj][:6]
                        assert c in done, "child not finished before its parent (band schedule order)"
                        assert n == _tri(A["f_b"][c] + 1) and ((uhi << 32) | ulo) == A["f_Uoff"][c]
                        eo = (ehi << 32) | elo
                        np.add.at(F, A["ea_tgt"][eo:eo + n], Ust[c])
                    M = np.zeros((fa, fa))
                    M[np.tril_indices(fa)] = F
                    M = M + np.tril(M, -1).T
                    La = np.linalg.cholesky(M[:p, :p])
                    Lb = np.linalg.solve(La, M[p:, :p].T).T
                    U = M[p:, p:] - Lb @ Lb.T
                    Ust[s] = U[np.tril_indices(b + 1)]
                    Lst[s] = np.vstack([La, Lb])
                    done.add(s)
    assert len(done) == A["n_fronts"]
    delta = np.zeros(A["n_scalars"])
    # back-substitution with the hand-down of the solve kernel: a front whose parent is in the same group takes its
    # boundary values from the parent's local solution vector [x_p | x_b] through cmap (record fields 14, 15)
    glo = A["glvl_front_off"]
    xloc = {}
    for i in reversed(order):
        r = frec[i]
        s, p, b, poff, boff = r[0], r[1], r[2], r[7], r[8]
        L = Lst[s]
        xb_ref = delta[A["bidx"][boff:boff + b]]
        if r[14] >= 0:
            lvl = int(np.searchsorted(glo, i, side="right")) - 1          # own local level -> group -> its first position
            grp = int(np.searchsorted(A["grp_lvl_off"], lvl, side="right")) - 1
            par_pos = glo[A["grp_lvl_off"][grp]] + r[14]
            assert par_pos > i and frec[par_pos][0] == A["f_parent"][s]
            cm = A["cmap"][r[15]:r[15] + b]
            xb = xloc[par_pos][cm]
            assert np.array_equal(xb, xb_ref), (i, s)
        else:
            xb = xb_ref
        t = L[p + b, :] - L[p:p + b, :].T @ xb
        xp = np.linalg.solve(L[:p, :].T, t)
        delta[A["pidx"][poff:poff + p]] = xp
        xloc[i] = np.concatenate([xp, xb])
    return delta
