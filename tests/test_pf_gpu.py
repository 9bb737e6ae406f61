"""GPU parity tests of the FastSLAM-2 per-particle path and of the resample step: the HIP particle store,
through the C ABI (include/cslam.h), against the CPU oracle particle by particle on the same seeded inputs."""
import numpy as np
import pytest

from helpers import assert_close
from pyoracle import Oracle, REF_EXACT, TEXTBOOK

pytestmark = pytest.mark.gpu

DTYPES = [np.float32, np.float64]
TOL = {np.dtype(np.float32): 2e-5, np.dtype(np.float64): 1e-12}


def _random_particles(np_, nf, dtype, seed=0):
    """Well-conditioned particles around a common pose with nf mapped features each."""
    rng = np.random.default_rng(seed)
    parts = []
    for _ in range(np_):
        Xv = np.array([rng.normal(0, 2.0), rng.normal(0, 2.0), rng.normal(0.2, 0.05)], dtype=dtype)
        A = rng.normal(size=(3, 3)) * np.array([0.3, 0.3, 0.02])[:, None]
        Pv = np.asfortranarray((A @ A.T + np.diag([0.05, 0.05, 1e-4])).astype(dtype))
        XF = np.asfortranarray(rng.uniform(-300, 300, size=(2, nf)).astype(dtype))
        PF = np.zeros((4, nf), dtype=dtype, order="F")
        for f in range(nf):
            B = rng.normal(size=(2, 2)) * 0.5
            PF[:, f] = (B @ B.T + 0.2 * np.eye(2)).reshape(-1, order="F")
        w = dtype(rng.uniform(0.5, 1.5) / np_)
        parts.append([w, Xv, Pv, XF, PF])
    return parts


def _shard_from(parts, nfcap, dtype, quirks=REF_EXACT):
    from conan_slam_amd.pf import ParticleShard

    sh = ParticleShard(len(parts), nfcap, dtype=dtype, quirks=quirks)
    for i, (w, Xv, Pv, XF, PF) in enumerate(parts):
        sh.set_particle(i, w, Xv, Pv, XF, PF)
    return sh


def _obs_for(parts, idf, dtype, seed=3):
    """Observations of the listed features as seen from the mean particle pose (+ noise)."""
    rng = np.random.default_rng(seed)
    X = np.mean([p[1] for p in parts], axis=0).astype(np.float64)
    XF = parts[0][3].astype(np.float64)
    Z = np.zeros((2, len(idf)))
    for i, f in enumerate(idf):
        dx, dy = XF[0, f - 1] - X[0], XF[1, f - 1] - X[1]
        Z[0, i] = np.hypot(dx, dy) + rng.normal() * 0.2
        Z[1, i] = np.arctan2(dy, dx) - X[2] + rng.normal() * 0.01
    return np.asfortranarray(Z.astype(dtype))


def _compare(sh, parts, dtype, tag, wtol=None):
    tol = TOL[np.dtype(dtype)]
    for i, (w, Xv, Pv, XF, PF) in enumerate(parts):
        gw, gX, gP, gXF, gPF = sh.get_particle(i)
        assert_close(f"{tag} Xv[{i}]", gX, Xv, tol)
        assert_close(f"{tag} Pv[{i}]", gP, Pv, tol)
        assert_close(f"{tag} XF[{i}]", gXF, XF, tol)
        assert_close(f"{tag} PF[{i}]", gPF, PF, tol)
        rel = abs(float(gw) - float(w)) / max(abs(float(w)), 1e-300)
        assert rel <= (wtol if wtol is not None else 50 * tol), (tag, i, float(gw), float(w))


@pytest.mark.parametrize("dtype", DTYPES)
def test_initial_state_and_roundtrip(gpu_required, dtype):
    from conan_slam_amd.pf import ParticleShard

    sh = ParticleShard(10, 4, dtype=dtype, n_global=40)
    w, Xv, Pv, XF, PF = sh.get_particle(3)
    assert w == dtype(1.0 / 40) and not Xv.any() and not Pv.any() and XF.shape == (2, 0)  # PF.cpp:319-341
    parts = _random_particles(10, 3, dtype, seed=1)
    sh2 = _shard_from(parts, 5, dtype)
    _compare(sh2, parts, dtype, "roundtrip", wtol=0.0)
    sh.close()
    sh2.close()


@pytest.mark.parametrize("dtype", DTYPES)
def test_predict_and_heading(gpu_required, dtype):
    o = Oracle(dtype)
    parts = _random_particles(70, 2, dtype, seed=2)
    sh = _shard_from(parts, 2, dtype)
    Q = np.diag([0.18, 6e-4]).astype(dtype)
    for step in range(3):
        sh.predict(83.33, 0.04 * step, Q, 73.0, 0.01)
        for p in parts:
            o.pf_predict(p[1], p[2], 83.33, 0.04 * step, Q, 73.0, 0.01)
    _compare(sh, parts, dtype, "predict", wtol=0.0)
    sh.observe_heading(0.25, True)
    for p in parts:
        o.pf_observe_heading(p[1], p[2], 0.25, True)
    tol = 2e-3 if dtype == np.float32 else 1e-9  # 1 - W[2] cancellation, see test_ekf_gpu.test_observe_heading
    for i, p in enumerate(parts):
        _, gX, gP, _, _ = sh.get_particle(i)
        assert_close("heading Xv", gX, p[1], TOL[np.dtype(dtype)])
        assert_close("heading Pv", gP, p[2], tol)
    sh.close()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("quirks", [REF_EXACT, TEXTBOOK])
@pytest.mark.parametrize("m", [1, 5])
def test_feature_update(gpu_required, dtype, quirks, m):
    o = Oracle(dtype, quirks)
    nf = 8
    parts = _random_particles(65, nf, dtype, seed=4)
    sh = _shard_from(parts, nf, dtype, quirks)
    idf = np.array([2, 7, 1, 5, 8][:m], dtype=np.int32)
    Z = _obs_for(parts, idf, dtype)
    R = np.diag([0.08, 0.0024]).astype(dtype)
    sh.feature_update(Z, idf, R)
    for p in parts:
        o.pf_feature_update(p[1], p[3], p[4], Z, idf, R)
    _compare(sh, parts, dtype, f"feature_update m={m}", wtol=0.0)
    sh.close()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("m", [1, 4])
def test_sample_proposal(gpu_required, dtype, m):
    o = Oracle(dtype)
    o64 = Oracle(np.float64)
    nf = 6
    npart = 130
    parts = _random_particles(npart, nf, dtype, seed=5)
    hi = [[np.float64(p[0])] + [np.array(a, dtype=np.float64, order="F") for a in p[1:]] for p in parts]
    sh = _shard_from(parts, nf, dtype)
    idf = np.array([3, 1, 6, 4][:m], dtype=np.int32)
    Z = _obs_for(parts, idf, dtype)
    R = np.diag([0.08, 0.0024]).astype(dtype)
    normals = np.random.default_rng(9).normal(size=(3, npart)).astype(dtype)
    sh.sample_proposal(Z, idf, R, normals)
    for i, (p, h) in enumerate(zip(parts, hi)):
        w = np.array([p[0]], dtype=dtype)
        o.pf_sample_proposal(w, p[1], p[2], p[3], p[4], Z, idf, R, normals[:, i].copy())
        p[0] = w[0]
        wh = np.array([h[0]], dtype=np.float64)
        o64.pf_sample_proposal(wh, h[1], h[2], h[3], h[4], Z.astype(np.float64), idf, R.astype(np.float64),
                               normals[:, i].astype(np.float64))
        h[0] = wh[0]
    # pose / map against the same-precision oracle; the weight is a product of Gaussian densities (exp of a
    # quadratic form in an inverse): the fairness rule of SURVEY 8d against the f64 oracle replaces a fixed tolerance
    _compare(sh, parts, dtype, f"sample_proposal m={m}", wtol=np.inf)
    wg = sh.get_weights().astype(np.float64)
    wc = np.array([p[0] for p in parts], dtype=np.float64)
    wh = np.array([h[0] for h in hi], dtype=np.float64)
    ok = wh > 1e-30  # (densities far out in the tails underflow to zero in f32, on the device as in the oracle)
    assert ok.sum() >= 1 and np.all(np.abs(wg[~ok]) <= 1e-30) and np.all(np.isfinite(wg))
    e_gpu, e_cpu = np.abs(wg[ok] - wh[ok]) / wh[ok], np.abs(wc[ok] - wh[ok]) / wh[ok]
    if dtype == np.float64:
        assert np.abs(wg - wc).max() <= 1e-9 * np.abs(wc).max()
    else:
        assert e_gpu.max() <= 4.0 * e_cpu.max() + 1e-6, (e_gpu.max(), e_cpu.max())
        assert np.median(e_gpu) <= 4.0 * np.median(e_cpu) + 1e-7, (np.median(e_gpu), np.median(e_cpu))
    sh.close()


@pytest.mark.parametrize("dtype", DTYPES)
def test_add_features(gpu_required, dtype):
    from conan_slam_amd import CslamError, _capi

    o = Oracle(dtype)
    parts = _random_particles(33, 2, dtype, seed=6)
    sh = _shard_from(parts, 5, dtype)
    Zn = np.asfortranarray(np.array([[120.0, 300.0, 45.0], [0.3, -1.1, 2.0]], dtype=dtype))
    R = np.diag([0.08, 0.0024]).astype(dtype)
    sh.add_features(Zn, R)
    assert sh.n_features == 5
    for p in parts:
        XF = np.zeros((2, 5), dtype=dtype, order="F")
        PF = np.zeros((4, 5), dtype=dtype, order="F")
        XF[:, :2], PF[:, :2] = p[3], p[4]
        nf = o.pf_add_features(p[1], XF, PF, 2, Zn, R)
        assert nf == 5
        p[3], p[4] = XF, PF
    _compare(sh, parts, dtype, "add_features", wtol=0.0)
    with pytest.raises(CslamError) as ei:
        sh.add_features(Zn[:, :1], R)
    assert ei.value.code == _capi.ERR_CAPACITY
    sh.close()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("force", [True, False])
def test_resample_single_shard(gpu_required, dtype, force):
    """PF.cpp:473-500 on one shard against the oracle's corrected normalise/resample."""
    from conan_slam_amd.pf import SingleComm, resample_particles, stratified_random

    o = Oracle(dtype)
    npart, nf = 64, 3
    parts = _random_particles(npart, nf, dtype, seed=7)
    rng = np.random.default_rng(8)
    w = rng.uniform(0.0, 1.0, npart) ** (6 if force else 0.05)  # skewed weights => small Neff
    for p, wi in zip(parts, w):
        p[0] = dtype(wi)
    sh = _shard_from(parts, nf, dtype)
    u = rng.uniform(size=npart)
    select = stratified_random(npart, u, dtype)
    assert np.allclose(select, o.pf_stratified_random(npart, u.astype(dtype), ref_exact=False), rtol=1e-6)
    neff, did = resample_particles(sh, SingleComm(), int(0.75 * npart), True, select=select)
    wref = np.array([p[0] for p in parts], dtype=dtype)
    neff_ref, did_ref, keep = o.pf_normalize_resample(wref, int(0.75 * npart), True, select)
    assert did == did_ref == force
    assert abs(neff - float(neff_ref)) <= 1e-3 * float(neff_ref)
    new = [[wref[i], *[a.copy() for a in parts[keep[i]][1:]]] for i in range(npart)] if did else \
        [[wref[i], *parts[i][1:]] for i in range(npart)]
    _compare(sh, new, dtype, "resample", wtol=1e-5)
    sh.close()


def test_pack_unpack_roundtrip(gpu_required):
    import torch

    dtype = np.float32
    parts = _random_particles(20, 4, dtype, seed=11)
    sh = _shard_from(parts, 6, dtype)
    src = np.array([5, 5, 0, 19, 7], dtype=np.int32)
    buf = sh.pack(src)
    assert buf.shape == (5, 13 + 6 * 4)
    host = buf.cpu().numpy()
    assert host[0, 0] == parts[5][0] and np.array_equal(host[3, 1:4], parts[19][1])
    dst = np.array([1, 2, 3, 4, 6], dtype=np.int32)
    sh.unpack(dst, buf)
    torch.cuda.synchronize()
    for d, s in zip(dst, src):
        gw, gX, gP, gXF, gPF = sh.get_particle(int(d))
        assert gw == parts[s][0] and np.array_equal(gX, parts[s][1]) and np.array_equal(gPF, parts[s][4])
    sh.close()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("npart", [512, 20000])
def test_device_resample_equals_the_host_planned_one(gpu_required, dtype, npart):
    """cslam_pf_resample_local (plan on the device) against weight_sums + scale + host keep[] + gather_local: the
    same particles in the same slots, bit for bit, and the same Neff.  20 000 particles: more than one LDS stage of
    the sequential running sum (kPfPlanMax = 8192 per stage)."""
    from conan_slam_amd.pf import SingleComm, resample_particles, stratified_random

    nf = 5 if npart <= 1024 else 1
    parts = _random_particles(npart, nf, dtype, seed=31)
    rng = np.random.default_rng(32)
    w = rng.uniform(0.0, 1.0, npart) ** 5
    for p, wi in zip(parts, w):
        p[0] = dtype(wi)
    a = _shard_from(parts, nf, dtype)
    b = _shard_from(parts, nf, dtype)
    b.host_resample = True
    select = stratified_random(npart, rng.uniform(size=npart), dtype)
    ra = resample_particles(a, SingleComm(), int(0.75 * npart), True, select=select)
    rb = resample_particles(b, SingleComm(), int(0.75 * npart), True, select=select)
    assert ra[1] and rb[1] and abs(ra[0] - rb[0]) <= 1e-9 * abs(rb[0])
    for i in range(0, npart, 37 if npart <= 1024 else 997):
        pa, pb = a.get_particle(i), b.get_particle(i)
        for x, y in zip(pa, pb):
            assert np.array_equal(np.asarray(x), np.asarray(y)), i
    assert np.array_equal(a.get_weights(), b.get_weights())
    a.close()
    b.close()


@pytest.mark.parametrize("dtype", DTYPES)
def test_sharded_resample_behind_the_c_abi_world_of_one(gpu_required, dtype):
    """cslam_pf_resample_sharded (RCCL all-reduce + all-gather, keep[] and the exchange plan on the device, grouped
    send/recv, include/cslam.h) with a communicator of ONE rank -- all a one-GPU box can host -- against
    cslam_pf_resample_local: same particles in the same slots bit for bit, same Neff, same decision."""
    import ctypes as C

    from conan_slam_amd import _capi
    from conan_slam_amd.pf import SingleComm, resample_particles, stratified_random

    L = _capi.lib()
    ident = (C.c_ubyte * 128)()
    _capi.check(L.cslam_comm_unique_id(ident))
    comm = C.c_void_p(None)
    _capi.check(L.cslam_comm_create(ident, C.c_int(0), C.c_int(1), C.c_int(-1), C.byref(comm)))

    class OneRank:  # the shape resample_sharded expects of a communicator object
        _h, world, rank = comm, 1, 0

    npart, nf = 384, 7
    parts = _random_particles(npart, nf, dtype, seed=51)
    rng = np.random.default_rng(52)
    for force in (True, False):
        w = rng.uniform(0.0, 1.0, npart) ** (5 if force else 0.05)
        for p, wi in zip(parts, w):
            p[0] = dtype(wi)
        a = _shard_from(parts, nf, dtype)
        b = _shard_from(parts, nf, dtype)
        select = stratified_random(npart, rng.uniform(size=npart), dtype)
        ra = a.resample_sharded(OneRank, select, int(0.75 * npart), True)
        rb = resample_particles(b, SingleComm(), int(0.75 * npart), True, select=select)
        assert ra[1] == rb[1] == force
        assert abs(ra[0] - rb[0]) <= 1e-9 * abs(rb[0])
        assert np.array_equal(a.get_weights(), b.get_weights())
        for i in range(0, npart, 29):
            for x, y in zip(a.get_particle(i), b.get_particle(i)):
                assert np.array_equal(np.asarray(x), np.asarray(y)), (force, i)
        a.close()
        b.close()
    _capi.check(L.cslam_comm_destroy(comm))


def _run_ranks(fns, timeout=180.0):
    """One host thread per rank (the ranks meet inside the call, as processes do over RCCL); re-raises a rank's error."""
    import threading

    out, err = [None] * len(fns), [None] * len(fns)

    def body(r):
        try:
            out[r] = fns[r]()
        except BaseException as e:  # noqa: BLE001 -- carried to the test thread
            err[r] = e

    th = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(len(fns))]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout)
    assert not any(t.is_alive() for t in th), "a rank hung inside the sharded resample"
    for e in err:
        if e is not None:
            raise e
    return out


@pytest.mark.parametrize("world,npart,nf,skew", [(2, 64, 3, "mild"), (4, 96, 5, "one_rank"), (8, 128, 2, "one_particle"),
                                                  (8, 512, 1000, "mild"), (2, 96, 4, "none")])
def test_sharded_resample_loopback_world_gt_1(gpu_required, world, npart, nf, skew):
    """cslam_pf_resample_sharded with MORE than one rank (PF.cpp:473-500, 546-577 over block-partitioned particles):
    `world` shards on one device behind the in-process loopback communicator (cslam_comm_create_loopback), one host
    thread per rank.  The ranks > 0 paths -- pf_exchange_plan_kernel for rank > 0, the grouped send / recv, the receive
    ordering by source rank -- must leave the concatenated shards bit-equal to cslam_pf_resample_local on the whole set
    and to the oracle's keep[]; the device-side plan must equal conan_slam_amd/pf.py::plan_exchange for every rank.
    512 x 1000 is BASELINE configs[3]'s particle set over 8 ranks."""
    from conan_slam_amd.pf import LoopbackComm, SingleComm, plan_exchange, resample_particles, stratified_random

    dtype = np.float32
    L = npart // world
    parts = _random_particles(npart, nf, dtype, seed=1000 + world)
    rng = np.random.default_rng(7 * world + npart)
    if skew == "mild":
        w = rng.uniform(0.0, 1.0, npart) ** 5
    elif skew == "one_rank":              # almost all the weight on rank 2's particles: everybody receives from it
        w = np.full(npart, 1e-6)
        w[2 * L:3 * L] = rng.uniform(0.5, 1.0, L)
    elif skew == "one_particle":          # one survivor: every slot of every rank is a copy of it
        w = np.full(npart, 1e-9)
        w[npart - 3] = 1.0
    else:                                 # uniform weights: Neff = N, no resample (weights only normalised)
        w = np.full(npart, 0.37)
    for p, wi in zip(parts, w):
        p[0] = dtype(wi)
    whole = _shard_from(parts, nf, dtype)
    shards = [_shard_from(parts[r * L:(r + 1) * L], nf, dtype) for r in range(world)]
    comms = LoopbackComm.create(world)
    select = stratified_random(npart, rng.uniform(size=npart), dtype)
    nmin = int(0.75 * npart)
    res = _run_ranks([(lambda r=r: shards[r].resample_sharded(comms[r], select, nmin, True)) for r in range(world)])
    ref = resample_particles(whole, SingleComm(), nmin, True, select=select)
    o = Oracle(dtype)
    wref = np.array([p[0] for p in parts], dtype=dtype)
    neff_o, did_o, keep = o.pf_normalize_resample(wref, nmin, True, select)
    expect = skew != "none"
    assert ref[1] == did_o == expect
    for r in range(world):
        assert res[r][1] == expect and abs(res[r][0] - ref[0]) <= 1e-9 * abs(ref[0]), (r, res[r], ref)
    # particles: shard r slot i == whole slot r*L + i == original particle keep[r*L + i], bit for bit
    step = 1 if npart * nf <= 20000 else 3
    for g in range(0, npart, step):
        got = shards[g // L].get_particle(g % L)
        ws = whole.get_particle(g)
        src = parts[keep[g]] if expect else parts[g]
        for a, b in zip(got[1:], ws[1:]):
            assert np.array_equal(np.asarray(a), np.asarray(b)), ("vs whole set", g)
        for a, b in zip(got[1:], src[1:]):
            assert np.array_equal(np.asarray(a), np.asarray(b)), ("vs oracle keep", g, int(keep[g]))
    wall = np.concatenate([sh.get_weights() for sh in shards])
    if expect:
        assert np.array_equal(wall, whole.get_weights()) and np.all(wall == dtype(1.0 / npart))
    else:
        # no resample: w / ws only; the all-reduce adds the per-rank partial sums, the whole set adds in one sweep
        assert np.allclose(wall, whole.get_weights(), rtol=2e-7, atol=0)
    # the exchange plan of every rank against the host planner
    for r in range(world):
        send_c, recv_c, send_idx = shards[r].debug_last_exchange(world)
        if expect:
            src_l, sc, _, rc_ = plan_exchange(np.asarray(keep), r, world, L)
            assert send_c == sc and recv_c == rc_, (r, send_c, sc, recv_c, rc_)
            assert np.array_equal(send_idx, src_l), r
        else:
            assert not any(send_c) and not any(recv_c)
    for c in comms:
        c.close()
    for sh in shards + [whole]:
        sh.close()


def test_loopback_communicator_breaks_instead_of_hanging(gpu_required):
    """A rank that never arrives must not hang its peers: with only ONE of two ranks calling, the loopback barrier
    times out... that would take its full 60 s, so this test checks the cheap half -- bad arguments are refused before
    any collective starts (the rule the sharded resample follows: everything fallible comes before the first one)."""
    import ctypes as C

    from conan_slam_amd import _capi
    from conan_slam_amd.pf import LoopbackComm

    L = _capi.lib()
    arr = (C.c_void_p * 17)()
    assert L.cslam_comm_create_loopback(C.c_int(17), C.c_int(-1), arr) == _capi.ERR_BAD_ARG
    assert L.cslam_comm_create_loopback(C.c_int(0), C.c_int(-1), arr) == _capi.ERR_BAD_ARG
    comms = LoopbackComm.create(2)
    parts = _random_particles(8, 2, np.float32, seed=5)
    sh = _shard_from(parts, 2, np.float32)
    assert L.cslam_pf_resample_sharded(sh._h, comms[0]._h, None, C.c_double(6.0), C.c_int(1), None, None) == _capi.ERR_BAD_ARG
    for c in comms:
        c.close()
    sh.close()


@pytest.mark.parametrize("dtype", DTYPES)
def test_fused_observation_step_equals_the_separate_calls(gpu_required, dtype):
    """cslam_pf_observation_step (one staged copy, nothing returned) against predict + sampleProposal + featureUpdate +
    resampleParticles issued one by one: the same kernels on the same inputs, so the stores must agree bit for bit."""
    from conan_slam_amd.pf import SingleComm, resample_particles, stratified_random

    npart, nf, m = 200, 9, 4
    parts = _random_particles(npart, nf, dtype, seed=61)
    a = _shard_from(parts, nf, dtype)
    b = _shard_from(parts, nf, dtype)
    rng = np.random.default_rng(62)
    Q = np.diag([0.18, 6e-4]).astype(dtype)
    R = np.diag([0.08, 0.0024]).astype(dtype)
    n_res = 0
    for step in range(4):
        idf = np.sort(rng.permutation(nf)[:m] + 1).astype(np.int32)
        Z = _obs_for(parts, idf, dtype, seed=step)
        nrm = rng.normal(size=(3, npart)).astype(dtype)
        sel = stratified_random(npart, rng.uniform(size=npart), dtype)
        nmin = npart + 1 if step % 2 == 0 else int(0.5 * npart)
        a.observation_step(83.33, 0.02 * step, Q, 73.0, 0.01, Z, idf, R, nrm, sel, nmin, True)
        b.predict(83.33, 0.02 * step, Q, 73.0, 0.01)
        b.sample_proposal(Z, idf, R, nrm)
        b.feature_update(Z, idf, R)
        _, did = resample_particles(b, SingleComm(), nmin, True, select=sel)
        n_res += int(did)
    calls, resamples, _ = a.resample_stats()
    assert calls == 4 and resamples == n_res and n_res >= 2
    assert np.array_equal(a.get_weights(), b.get_weights())
    for i in range(0, npart, 13):
        for x, y in zip(a.get_particle(i), b.get_particle(i)):
            assert np.array_equal(np.asarray(x), np.asarray(y)), i
    a.close()
    b.close()
