"""Shared helpers of the parity tests: scenario builders and the tolerance rules of SURVEY.md 8d."""
import numpy as np

from pyoracle import Oracle

# per-call tolerances, GPU vs CPU oracle at the same dtype and quirks (SURVEY.md 8d "Parity tolerance")
X_RTOL = {np.dtype(np.float32): 1e-5, np.dtype(np.float64): 1e-12}
P_RTOL = {np.dtype(np.float32): 1e-4, np.dtype(np.float64): 1e-10}


def make_scenario(N, dtype, seed=0, corr=0.5, pose_scale=1e-2):
    """A well-conditioned SPD covariance P0 = D + U U^T and a state with N landmarks in [-500,500]^2."""
    rng = np.random.default_rng(seed)
    n = 3 + 2 * N
    U = rng.normal(size=(n, 4)) * corr
    P = np.eye(n) + U @ U.T
    s = np.ones(n)
    s[:3] = np.sqrt(pose_scale)
    P = P * s[:, None] * s[None, :]
    X = np.concatenate([[1.0, -2.0, 0.3], rng.uniform(-500, 500, size=2 * N)])
    return np.array(X, dtype=dtype), np.array(P, dtype=dtype, order="F")


def make_obs(X, idf, dtype, seed=1, sr=0.3, sb=0.02):
    """Noisy range/bearing observations of the listed (1-based) features from the estimated pose."""
    rng = np.random.default_rng(seed)
    X = np.asarray(X, dtype=np.float64)
    Z = np.zeros((2, len(idf)), dtype=np.float64)
    for i, f in enumerate(idf):
        fx = 3 + 2 * (f - 1)
        dx, dy = X[fx] - X[0], X[fx + 1] - X[1]
        Z[0, i] = np.hypot(dx, dy) + rng.normal() * sr
        Z[1, i] = np.arctan2(dy, dx) - X[2] + rng.normal() * sb
    return np.asfortranarray(Z.astype(dtype))


def assert_close(name, got, ref, rtol, hi=None, fair=4.0):
    """|got-ref|_max <= rtol*max(1,|ref|_max), or -- the fairness rule -- the error of `got` against the
    high-precision result `hi` is no worse than `fair` x the error of `ref` (CPU, same dtype) against it."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert np.all(np.isfinite(got)) == np.all(np.isfinite(ref)), f"{name}: finiteness differs"
    if not np.all(np.isfinite(ref)):
        return
    scale = max(1.0, float(np.abs(ref).max()) if ref.size else 1.0)
    err = float(np.abs(got - ref).max()) if ref.size else 0.0
    if err <= rtol * scale:
        return
    if hi is not None:
        hi = np.asarray(hi, dtype=np.float64)
        e_got = float(np.abs(got - hi).max())
        e_ref = float(np.abs(ref - hi).max())
        if e_got <= fair * e_ref + rtol * scale:
            return
        raise AssertionError(f"{name}: err {err:.3e} > {rtol:.1e}*{scale:.3e}; vs f64: got {e_got:.3e}, cpu {e_ref:.3e}")
    raise AssertionError(f"{name}: max abs err {err:.3e} > {rtol:.1e} * {scale:.3e}")


class OracleState:
    """X / P pair driven through the CPU oracle with spare capacity for augmentation."""

    def __init__(self, X, P, dtype, quirks, extra=0):
        self.o = Oracle(dtype, quirks)
        n = X.shape[0]
        cap = n + 2 * extra
        self.X = np.zeros(cap, dtype=dtype)
        self.P = np.zeros((cap, cap), dtype=dtype, order="F")
        self.X[:n] = X
        self.P[:n, :n] = P
        self.n = n

    def predict(self, v, swa, Q, wb, dt):
        self.o.predict(self.X, self.P, self.n, v, swa, Q, wb, dt)

    def update(self, Z, R, idf, batch, fast=False):
        return self.o.update(self.X, self.P, self.n, Z, R, idf, batch, fast=fast)

    def augment(self, Z, R):
        self.n = self.o.augment(self.X, self.P, self.n, Z, R)

    def observe_heading(self, phi, use=True):
        self.o.observe_heading(self.X, self.P, self.n, phi, use)

    def x(self):
        return self.X[: self.n].copy()

    def p(self):
        return np.array(self.P[: self.n, : self.n], order="F")


class EngineBackend:
    """Adapter giving the HIP engine (conan_slam_amd.EKF) the surface sim_driver.run_demo() drives."""

    def __init__(self, dtype=np.float32, quirks=3, max_landmarks=64):
        from conan_slam_amd import EKF

        self.ekf = EKF(max_landmarks, dtype=dtype, quirks=quirks)

    @property
    def n(self):
        return self.ekf.n

    def predict(self, v, swa, Q, wb, dt):
        self.ekf.predict(v, swa, Q, wb, dt)

    def observe_heading(self, phi, use):
        self.ekf.observe_heading(phi, use)

    def update(self, Z, R, idf, batch):
        self.ekf.update(Z, R, idf, batch)

    def augment(self, Z, R):
        self.ekf.augment(Z, R)

    def get_x(self):
        return self.ekf.get_x()

    def get_p(self):
        return self.ekf.get_p()
