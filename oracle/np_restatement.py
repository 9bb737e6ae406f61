"""Independent numpy/scipy restatement of the reference's EKF-SLAM / FastSLAM-2 hot path.

TEST INFRASTRUCTURE ONLY.  Purpose: a second, independently written statement of the same reference
behaviour (matrix-level numpy + LAPACK instead of the C oracle's scalar loops), used by
tools/gen_golden.py to produce the fixtures under tests/golden/ that pin the C oracle.  It follows the
reference's Eigen expressions line by line (citations relative to /root/reference) -- it does not call
or read the reference at run time and contains none of its source.

PARITY UNPINNED: neither restatement can be checked against the reference itself, which cannot be built
here (Eigen 3.4.0 / Boost 1.84.0 absent) and ships no golden vectors.
"""
from __future__ import annotations

import numpy as np

PI = 3.14159265358979323846264338327950288  # MSVC std::_Pi_val
Q_LOWER_CHOL_GAIN = 1
Q_PREDICT_NM4 = 2
REF_EXACT = 3
TEXTBOOK = 0


class NpSlam:
    def __init__(self, dtype=np.float32, quirks=REF_EXACT):
        self.t = np.dtype(dtype).type
        self.dtype = np.dtype(dtype)
        self.quirks = quirks

    # ------------------------------------------------------------ slam.h:816-829
    def pi2pi(self, angle):
        t = self.t
        angle = t(np.fmod(t(angle), t(2.0 * PI)))
        if float(angle) > PI:
            angle = t(float(angle) - 2.0 * PI)
        if float(angle) < -PI:
            angle = t(float(angle) + 2.0 * PI)
        return angle

    # ------------------------------------------------------------ slam.h:776-779
    def make_symmetric(self, A):
        return ((A + A.T) * self.t(0.5)).astype(self.dtype)

    # ------------------------------------------------------------ slam.h:413-436
    def cholesky_decomposition(self, M):
        M = np.asarray(M, dtype=self.dtype)
        try:
            # LLT reads the lower triangle only
            Ml = np.tril(M) + np.tril(M, -1).T
            L = np.linalg.cholesky(Ml.astype(self.dtype)).astype(self.dtype)
            if np.any(np.diag(L) <= 0) or not np.all(np.isfinite(L)):
                raise np.linalg.LinAlgError
        except np.linalg.LinAlgError:
            Ml = np.tril(M) + np.tril(M, -1).T
            ev, V = np.linalg.eigh(Ml.astype(self.dtype))  # ascending, like SelfAdjointEigenSolver
            with np.errstate(invalid="ignore"):
                L = (V * np.sqrt(ev)[None, :]).astype(self.dtype)
        if not np.all(np.isfinite(L)):
            L = np.zeros_like(L)
        return L

    def inverse(self, A):
        A = np.asarray(A, dtype=self.dtype)
        with np.errstate(all="ignore"):
            try:
                return np.linalg.inv(A).astype(self.dtype)
            except np.linalg.LinAlgError:
                return np.full_like(A, np.inf)

    def gain_factor(self, S):
        L = self.cholesky_decomposition(S)
        G = self.inverse(L)  # slam.h:251
        if not (self.quirks & Q_LOWER_CHOL_GAIN):
            G = np.ascontiguousarray(G.T)
        if not np.all(np.isfinite(G)):
            G = np.zeros_like(G)  # slam.h:252-255
        return G

    # ------------------------------------------------------------ slam.h:235-266
    def cholesky_update(self, X, P, V, R, H):
        if V.shape[0] == 0:
            return X, P
        PHT = P @ H.T
        S = H @ PHT + R
        S = self.make_symmetric(S)
        G = self.gain_factor(S)
        W1 = PHT @ G
        W = W1 @ G.T
        X = X + W @ V
        P = P - W1 @ W1.T
        return X.astype(self.dtype), P.astype(self.dtype)

    # ------------------------------------------------------------ slam.h:700-725
    def joseph_update(self, X, P, V, R, H):
        t = self.t
        PHT = P @ H.T
        S = H @ PHT + R
        SI = self.inverse(S)
        SI = self.make_symmetric(SI)
        W = PHT @ SI
        X = X + W @ V
        n = P.shape[0]
        Cm = np.eye(n, dtype=self.dtype) - W @ H
        P = Cm @ P @ Cm.T + W @ R @ W.T
        P = P + np.eye(n, dtype=self.dtype) * t(np.finfo(np.float32).tiny)
        return X.astype(self.dtype), P.astype(self.dtype)

    # ------------------------------------------------------------ EKF.cpp:354-404
    def observe_model(self, X, idf):
        t = self.t
        n = X.shape[0]
        H = np.zeros((2, n), dtype=self.dtype)
        Z = np.zeros(2, dtype=self.dtype)
        fpos = 3 + idf * 2 - 1
        if n > 3:
            dx = t(X[fpos - 1] - X[0])
            dy = t(X[fpos] - X[1])
            d2 = t(dx * dx + dy * dy)
            d = t(np.sqrt(d2))
            xd, yd, xd2, yd2 = t(dx / d), t(dy / d), t(dx / d2), t(dy / d2)
            Z[0] = d
            Z[1] = t(np.arctan2(dy, dx)) - X[2]
            H[:, 0:3] = np.array([[-xd, -yd, 0], [yd2, -xd2, -1]], dtype=self.dtype)
            H[:, fpos - 1:fpos + 1] = np.array([[xd, yd], [-yd2, xd2]], dtype=self.dtype)
        return Z, H

    def _motion_jacobians(self, phi, v, swa, wb, dt):
        t = self.t
        phi, v, swa, wb, dt = t(phi), t(v), t(swa), t(wb), t(dt)
        s, c = t(np.sin(t(swa + phi))), t(np.cos(t(swa + phi)))
        Gv = np.array([[1, 0, -v * dt * s], [0, 1, v * dt * c], [0, 0, 1]], dtype=self.dtype)
        Gu = np.array([[dt * c, -v * dt * s], [dt * s, v * dt * c],
                       [dt * t(np.sin(swa)) / wb, v * dt * t(np.cos(swa)) / wb]], dtype=self.dtype)
        return Gv, Gu

    # ------------------------------------------------------------ EKF.cpp:406-455
    def predict(self, X, P, v, swa, Q, wb, dt):
        t = self.t
        X, P = X.copy(), P.copy()
        n = X.shape[0]
        phi = X[2]
        v, swa, wb, dt = t(v), t(swa), t(wb), t(dt)
        Gv, Gu = self._motion_jacobians(phi, v, swa, wb, dt)
        P[0:3, 0:3] = Gv @ P[0:3, 0:3] @ Gv.T + Gu @ Q @ Gu.T
        if n > 3:
            w = (n - 4) if (self.quirks & Q_PREDICT_NM4) else (n - 3)
            if w > 0:
                P[0:3, 3:3 + w] = Gv @ P[0:3, 3:3 + w]
                P[3:3 + w, 0:3] = P[0:3, 3:3 + w].T
        X[0] = X[0] + v * dt * t(np.cos(t(swa + phi)))
        X[1] = X[1] + v * dt * t(np.sin(t(swa + phi)))
        X[2] = self.pi2pi(X[2] + v * dt * t(np.sin(swa)) / wb)
        return X, P

    # ------------------------------------------------------------ EKF.cpp:93-129
    def batch_update(self, X, P, Z, R, idf):
        m = Z.shape[1]
        n = X.shape[0]
        if m == 0:
            return X.copy(), P.copy()
        H = np.zeros((2 * m, n), dtype=self.dtype)
        V = np.zeros(2 * m, dtype=self.dtype)
        RR = np.zeros((2 * m, 2 * m), dtype=self.dtype)
        for i in range(m):
            Zp, HT = self.observe_model(X, int(idf[i]))
            H[2 * i:2 * i + 2, :] = HT
            V[2 * i] = Z[0, i] - Zp[0]
            V[2 * i + 1] = self.pi2pi(Z[1, i] - Zp[1])
            RR[2 * i:2 * i + 2, 2 * i:2 * i + 2] = R
        return self.cholesky_update(X, P, V, RR, H)

    # ------------------------------------------------------------ EKF.cpp:457-479
    def single_update(self, X, P, Z, R, idf):
        X, P = X.copy(), P.copy()
        for i in range(Z.shape[1]):
            Zp, H = self.observe_model(X, int(idf[i]))
            V = np.array([Z[0, i] - Zp[0], self.pi2pi(Z[1, i] - Zp[1])], dtype=self.dtype)
            X, P = self.cholesky_update(X, P, V, R, H)
        return X, P

    def update(self, X, P, Z, R, idf, batch):
        return self.batch_update(X, P, Z, R, idf) if batch else self.single_update(X, P, Z, R, idf)

    # ------------------------------------------------------------ EKF.cpp:9-91
    def augment(self, X, P, Z, R):
        t = self.t
        X, P = X.copy(), P.copy()
        for i in range(Z.shape[1]):
            ln = X.shape[0]
            r, b = t(Z[0, i]), t(Z[1, i])
            s, c = t(np.sin(t(X[2] + b))), t(np.cos(t(X[2] + b)))
            X = np.concatenate([X, np.array([X[0] + r * c, X[1] + r * s], dtype=self.dtype)])
            Gv = np.array([[1, 0, -r * s], [0, 1, r * c]], dtype=self.dtype)
            Gz = np.array([[c, -r * s], [s, r * c]], dtype=self.dtype)
            AP = P
            P = np.zeros((ln + 2, ln + 2), dtype=self.dtype)
            P[:ln, :ln] = AP
            P[ln:, ln:] = Gv @ AP[0:3, 0:3] @ Gv.T + Gz @ R @ Gz.T
            P[ln:, 0:3] = Gv @ P[0:3, 0:3]
            P[0:3, ln:] = P[ln:, 0:3].T
            if ln > 3:
                P[ln:, 3:ln] = Gv @ P[0:3, 3:ln]
                P[3:ln, ln:] = P[ln:, 3:ln].T
        return X, P

    # ------------------------------------------------------------ EKF.cpp:328-352
    def observe_heading(self, X, P, phi, use=True):
        if not use:
            return X.copy(), P.copy()
        t = self.t
        sigma = t((float(np.float32(0.01)) * PI) / 180.0)
        n = X.shape[0]
        H = np.zeros((1, n), dtype=self.dtype)
        H[0, 2] = 1
        V = np.array([self.pi2pi(t(phi) - X[2])], dtype=self.dtype)
        R = np.array([[sigma * sigma]], dtype=self.dtype)
        return self.joseph_update(X, P, V, R, H)

    # ------------------------------------------------------------ simulator side
    def vehicle_model(self, Xv, v, swa, wb, dt):
        t = self.t
        v, swa, wb, dt = t(v), t(swa), t(wb), t(dt)
        x, y, phi = Xv
        return np.array([x + v * dt * t(np.cos(t(swa + phi))), y + v * dt * t(np.sin(t(swa + phi))),
                         self.pi2pi(phi + v * dt * t(np.sin(swa)) / wb)], dtype=self.dtype)

    def compute_swa(self, Xv, WP, iwp, minD, swa, rateSWA, maxSWA, dt):
        """slam.h:279-332 including signum<int>() truncating its float argument to int."""
        t = self.t
        minD, swa, rateSWA, maxSWA, dt = t(minD), t(swa), t(rateSWA), t(maxSWA), t(dt)
        cwp = WP[:, iwp - 1]
        d2 = t((cwp[0] - Xv[0]) * (cwp[0] - Xv[0]) + (cwp[1] - Xv[1]) * (cwp[1] - Xv[1]))
        if d2 < minD * minD:
            iwp += 1
            if iwp > WP.shape[1]:
                return 0, swa
            cwp = WP[:, iwp - 1]
        dG = self.pi2pi(t(np.arctan2(t(cwp[1] - Xv[1]), t(cwp[0] - Xv[0]))) - Xv[2] - swa)
        maxDelta = t(rateSWA * dt)
        if abs(dG) > maxDelta:
            dG = t(maxDelta * t(np.sign(int(dG))))
        swa = t(swa + dG)
        if abs(swa) > maxSWA:
            swa = t(t(np.sign(int(swa))) * maxSWA)
        return iwp, swa

    def get_observations(self, Xv, LM, rmax):
        t = self.t
        dx = (LM[0, :] - Xv[0]).astype(self.dtype).astype(np.float64)
        dy = (LM[1, :] - Xv[1]).astype(self.dtype).astype(np.float64)
        phi = float(Xv[2])
        rm = float(t(rmax))
        vis = (np.abs(dx) < rm) & (np.abs(dy) < rm) & ((dx * np.cos(phi) + dy * np.sin(phi)) > 0.0) & (
            (dx * dx + dy * dy) < rm * rm)
        idx = np.nonzero(vis)[0]
        fx = (LM[0, idx] - Xv[0]).astype(self.dtype)
        fy = (LM[1, idx] - Xv[1]).astype(self.dtype)
        Z = np.zeros((2, idx.shape[0]), dtype=self.dtype)
        Z[0, :] = np.sqrt(fx * fx + fy * fy)
        Z[1, :] = np.arctan2(fy, fx).astype(self.dtype) - Xv[2]
        return Z, (idx + 1).astype(np.int32)

    @staticmethod
    def data_associate_table(Z, tags, table, nf):
        """EKF.cpp:146-233 (table modified in place)."""
        zf, zn, idf, idn = [], [], [], []
        for i, tag in enumerate(tags):
            if table[tag - 1] == 0:
                zn.append(Z[:, i])
                idn.append(tag)
            else:
                zf.append(Z[:, i])
                idf.append(table[tag - 1])
        for i, tag in enumerate(idn):
            table[tag - 1] = nf + i + 1
        ZF = np.stack(zf, axis=1) if zf else np.zeros((2, 0), dtype=Z.dtype)
        ZN = np.stack(zn, axis=1) if zn else np.zeros((2, 0), dtype=Z.dtype)
        return ZF, ZN, np.array(idf, dtype=np.int32)

    # ------------------------------------------------------------ particle filter (per particle)
    def pf_predict(self, Xv, Pv, v, swa, Q, wb, dt):
        t = self.t
        v, swa, wb, dt = t(v), t(swa), t(wb), t(dt)
        phi = Xv[2]
        Gv, Gu = self._motion_jacobians(phi, v, swa, wb, dt)
        P = (Gv @ Pv @ Gv.T + Gu @ Q @ Gu.T).astype(self.dtype)
        X = np.array([Xv[0] + v * dt * t(np.cos(t(swa + phi))), Xv[1] + v * dt * t(np.sin(t(swa + phi))),
                      self.pi2pi(Xv[2] + v * dt * t(np.sin(swa)) / wb)], dtype=self.dtype)
        return X, P

    def pf_compute_jacobians(self, Xv, XF, PF, idf, R):
        """PF.cpp:70-135. XF 2 x Nf; PF list/array of 2x2. Returns ZP (2 x len), HV, HF, SF lists."""
        t = self.t
        ZP = np.zeros((2, len(idf)), dtype=self.dtype)
        HV, HF, SF = [], [], []
        for i, f in enumerate(idf):
            xf = XF[:, f - 1]
            pf = PF[f - 1]
            dx, dy = t(xf[0] - Xv[0]), t(xf[1] - Xv[1])
            d2 = t(dx * dx + dy * dy)
            d = t(np.sqrt(d2))
            ZP[0, i] = d
            ZP[1, i] = self.pi2pi(t(np.arctan2(dy, dx)) - Xv[2])
            hv = np.array([[-dx / d, -dy / d, 0], [dy / d2, -dx / d2, -1]], dtype=self.dtype)
            hf = np.array([[dx / d, dy / d], [-dy / d2, dx / d2]], dtype=self.dtype)
            HV.append(hv)
            HF.append(hf)
            SF.append((hf @ pf @ hf.T + R).astype(self.dtype))
        return ZP, HV, HF, SF

    def pf_gauss_evaluate(self, V, S, log_flag=False):
        t = self.t
        D = V.shape[0]
        SC = self.cholesky_decomposition(S).T
        with np.errstate(all="ignore"):
            nin = self.inverse(SC) @ V
            E = t(-0.5) * t(np.sum((nin * nin).astype(self.dtype)))
            if not log_flag:
                Cn = float(np.power(2.0 * PI, float(np.float32(D) / np.float32(2.0)))) * float(np.prod(np.diag(SC)))
                return t(float(t(np.exp(E))) / Cn)
            Cn = 0.5 * D * np.log(2.0 * PI) + float(np.sum(np.diag(SC)))
            return t(float(E) - Cn)

    def pf_likelihood(self, Xv, XF, PF, Z, idf, R):
        w = self.t(1)
        for i in range(len(idf)):
            ZP, _, _, SF = self.pf_compute_jacobians(Xv, XF, PF, [int(idf[i])], R)
            V = np.array([Z[0, i] - ZP[0, 0], self.pi2pi(Z[1, i] - ZP[1, 0])], dtype=self.dtype)
            w = self.t(w * self.pf_gauss_evaluate(V, SF[0]))
        return w

    def pf_sample_proposal(self, w, Xv, Pv, XF, PF, Z, idf, R, normals):
        """PF.cpp:502-544 with the 3 standard-normal draws as input. Returns (w, Xv, Pv)."""
        X, P = Xv.copy(), Pv.copy()
        X0, P0 = X.copy(), P.copy()
        pX = Xv.copy()
        for i in range(len(idf)):
            ZP, HV, _, SF = self.pf_compute_jacobians(pX, XF, PF, [int(idf[i])], R)
            SFI = self.inverse(SF[0])
            VI = np.array([Z[0, i] - ZP[0, 0], self.pi2pi(Z[1, i] - ZP[1, 0])], dtype=self.dtype)
            PT = HV[0].T @ SFI @ HV[0] + self.inverse(P)
            P = self.inverse(PT.astype(self.dtype))
            X = (X + P @ HV[0].T @ SFI @ VI).astype(self.dtype)
            pX = X.copy()
        L = self.cholesky_decomposition(P)
        XS = (L @ np.asarray(normals, dtype=self.dtype) + X).astype(self.dtype)
        like = self.pf_likelihood(XS, XF, PF, Z, idf, R)
        d1 = np.array([X0[0] - XS[0], X0[1] - XS[1], self.pi2pi(X0[2] - XS[2])], dtype=self.dtype)
        d2 = np.array([X[0] - XS[0], X[1] - XS[1], self.pi2pi(X[2] - XS[2])], dtype=self.dtype)
        prior = self.pf_gauss_evaluate(d1, P0)
        prop = self.pf_gauss_evaluate(d2, P)
        with np.errstate(all="ignore"):
            wn = self.t(self.t(self.t(w * like) * prior) / prop)
        return wn, XS, np.zeros((3, 3), dtype=self.dtype)

    def pf_feature_update(self, Xv, XF, PF, Z, idf, R):
        """PF.cpp:222-277. Returns new (XF, PF)."""
        XF = XF.copy()
        PF = [p.copy() for p in PF]
        ZP, _, HF, _ = self.pf_compute_jacobians(Xv, XF, PF, [int(f) for f in idf], R)
        for i, f in enumerate(idf):
            V = np.array([Z[0, i] - ZP[0, i], self.pi2pi(Z[1, i] - ZP[1, i])], dtype=self.dtype)
            x, p = self.cholesky_update(XF[:, f - 1].copy(), PF[f - 1], V, R, HF[i])
            XF[:, f - 1] = x
            PF[f - 1] = p
        return XF, PF

    def pf_add_features(self, Xv, Z, R):
        """PF.cpp:9-60. Returns the appended XF columns (2 x q) and PF blocks."""
        t = self.t
        xf, pf = [], []
        for i in range(Z.shape[1]):
            r, b = t(Z[0, i]), t(Z[1, i])
            s, c = t(np.sin(t(Xv[2] + b))), t(np.cos(t(Xv[2] + b)))
            xf.append([Xv[0] + r * c, Xv[1] + r * s])
            Gz = np.array([[c, -r * s], [s, r * c]], dtype=self.dtype)
            pf.append((Gz @ R @ Gz.T).astype(self.dtype))
        return np.array(xf, dtype=self.dtype).T.reshape(2, -1), pf

    def pf_stratified_resample(self, w, select, ref_exact=False):
        """PF.cpp:546-577 -> (keep, neff). ref_exact reproduces the shipped loop."""
        t = self.t
        w = (np.asarray(w, dtype=self.dtype) / t(np.sum(np.asarray(w, dtype=self.dtype)))).astype(self.dtype)
        neff = t(1) / t(np.sum((w * w).astype(self.dtype)))
        n = w.shape[0]
        cum = np.cumsum(w, dtype=self.dtype)
        keep = np.zeros(n, dtype=np.int32)
        ctr = 1
        for i in range(n):
            while ctr <= n and (select[i] if ref_exact else select[ctr - 1]) < cum[i]:
                keep[ctr - 1] = i
                ctr += 1
        return keep, neff
