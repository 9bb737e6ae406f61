"""Deterministic synthetic EKF-SLAM workloads -- input generation only, no filter math.

Everything comes from a counter-based generator, splitmix64(seed, index) -> uniform double in [0,1) ->
Box-Muller normal, so that the CPU oracle, the tests and bench.py all see identical bytes.

Map:        landmark i uniform in [-5000, 5000]^2 m (the demo map's extent, test/main.cpp:24-54), seed 1.
State:      pose (0,0,0); landmark estimates = truth + N(0, 1 m^2), seed 2.
Covariance: P0 = D + U U^T, U n x 8 iid N(0, 0.25), D = I, with the pose block / pose cross terms scaled
            so that P0[0:3,0:3] is 1e-2 of that (SPD, kappa <~ 1e3), seed 3.
Per step t: v = 83.33, swa = 0.05 sin(0.01 t), dt = 0.01, wb = 73 (slam.h:65-69); QE = 2 Q, RE = 8 R as in
            test/main.cpp:93-129; m distinct observed landmarks from seed (4,t); Z = exact range/bearing of the
            TRUE landmark from the TRUE pose + N(0, R), seed (5,t).
"""
from __future__ import annotations

import numpy as np

PI = 3.14159265358979323846264338327950288


def splitmix64(seed: int, idx) -> np.ndarray:
    """Vectorised splitmix64 finaliser of (seed * golden + idx + golden)."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.asarray(idx, dtype=np.uint64)
             + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, idx) -> np.ndarray:
    """Uniform doubles in [0,1) from the top 53 bits."""
    return (splitmix64(seed, idx) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(seed: int, idx) -> np.ndarray:
    """Standard normals by Box-Muller on two uniform streams of the same seed."""
    idx = np.asarray(idx, dtype=np.uint64)
    u1 = uniform01(seed, idx * np.uint64(2))
    u2 = uniform01(seed, idx * np.uint64(2) + np.uint64(1))
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


def noise_matrices(dtype=np.float32):
    """Q, R (slam.h:72-81, test/main.cpp:93-103) and the inflated QE = 2Q, RE = 8R (main.cpp:125-129)."""
    f = np.float32
    sv, ss = f(0.3), f(float(f(1.0)) * PI / 180.0)
    sr, sb = f(0.1), f(float(f(1.0)) * PI / 180.0)
    Q = np.array([[sv * sv, 0], [0, ss * ss]], dtype=dtype, order="F")
    R = np.array([[sr * sr, 0], [0, sb * sb]], dtype=dtype, order="F")
    return Q, R, (2 * Q).astype(dtype, order="F"), (8 * R).astype(dtype, order="F")


class Workload:
    """A synthetic map + initial filter state + a stream of (controls, observations) per step."""

    def __init__(self, n_landmarks: int, m_obs: int, dtype=np.float32, seed: int = 0, build_p: bool = True,
                 corr: float = 0.5):
        self.N = int(n_landmarks)
        self.m = int(m_obs)
        self.n = 3 + 2 * self.N
        self.dtype = np.dtype(dtype)
        self.seed = int(seed)
        N, n = self.N, self.n
        s = 1000 * self.seed
        idx = np.arange(N, dtype=np.uint64)
        self.LM = np.empty((2, N), dtype=np.float64)
        self.LM[0] = -5000.0 + 10000.0 * uniform01(s + 1, 2 * idx)
        self.LM[1] = -5000.0 + 10000.0 * uniform01(s + 1, 2 * idx + np.uint64(1))
        X0 = np.zeros(n, dtype=np.float64)
        X0[3::2] = self.LM[0] + normal(s + 2, 2 * idx)
        X0[4::2] = self.LM[1] + normal(s + 2, 2 * idx + np.uint64(1))
        self.X0 = X0.astype(self.dtype)
        # corr: standard deviation of U's entries (0.5 = SURVEY 8d's strongly correlated P0; <= 0.1 gives the weakly
        # correlated P0 on which the reference's own lower-Cholesky gain stays healthy, see tests/test_timed_path_gpu.py)
        self.corr = float(corr)
        self.U = (self.corr * normal(s + 3, np.arange(n * 8, dtype=np.uint64))).reshape(n, 8).astype(self.dtype)
        self.U[0:3, :] *= self.dtype.type(0.1)  # pose block 1e-2, pose<->map cross terms 1e-1
        self.P0 = self.make_p0() if build_p else None
        self.Q, self.R, self.QE, self.RE = noise_matrices(self.dtype)
        self.v, self.wb, self.dt = 83.33, 73.0, 0.01
        self._true_pose = np.zeros(3, dtype=np.float64)
        self._t = 0

    def make_p0(self) -> np.ndarray:
        n = self.n
        P = np.asfortranarray((self.U @ self.U.T).astype(self.dtype))
        d = np.ones(n, dtype=self.dtype)
        d[0:3] = self.dtype.type(0.01)
        P[np.arange(n), np.arange(n)] += d
        return P

    # ---- per-step inputs -------------------------------------------------------------------
    def controls(self, t: int):
        return self.v, 0.05 * np.sin(0.01 * t)

    def true_pose_after(self, t: int) -> np.ndarray:
        """Noise-free vehicle model (slam.h:952-966) integrated in float64 up to and including step t."""
        while self._t <= t:
            v, swa = self.controls(self._t)
            x, y, phi = self._true_pose
            x += v * self.dt * np.cos(swa + phi)
            y += v * self.dt * np.sin(swa + phi)
            phi += v * self.dt * np.sin(swa) / self.wb
            phi = (phi + np.pi) % (2 * np.pi) - np.pi
            self._true_pose = np.array([x, y, phi])
            self._t += 1
        assert self._t == t + 1, "steps must be requested in order"
        return self._true_pose

    def observations(self, t: int):
        """(Z 2 x m Fortran, idf m int32, 1-based) for step t; the true pose must have been advanced to t."""
        N, m = self.N, self.m
        s = 1000 * self.seed
        # m distinct landmarks: rank the first draws of a per-step stream
        keys = splitmix64(s + 4 + 7919 * (t + 1), np.arange(N, dtype=np.uint64))
        pick = np.sort(np.argpartition(keys, m - 1)[:m]) if m < N else np.arange(N)
        pose = self.true_pose_after(t)
        dx = self.LM[0, pick] - pose[0]
        dy = self.LM[1, pick] - pose[1]
        rng = np.sqrt(dx * dx + dy * dy)
        brg = np.arctan2(dy, dx) - pose[2]
        nz = normal(s + 5 + 104729 * (t + 1), np.arange(2 * m, dtype=np.uint64))
        Z = np.empty((2, m), dtype=np.float64)
        Z[0] = rng + nz[0::2] * np.sqrt(float(self.R[0, 0]))
        Z[1] = brg + nz[1::2] * np.sqrt(float(self.R[1, 1]))
        return np.asfortranarray(Z.astype(self.dtype)), (pick + 1).astype(np.int32)
