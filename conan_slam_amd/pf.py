"""Host-side mirror of the reference's FastSLAM-2 back-end (PF.cpp) over the C ABI, sharded one shard per GPU.

`ParticleShard` wraps one `cslam_pf_t` handle: the particles this process owns, structure-of-arrays in HBM.
Its methods carry the reference's names and argument meaning applied to EVERY owned particle:
predict / observe_heading / sample_proposal / feature_update / add_features (PF.cpp:419-471, 382-417,
502-544, 222-277, 9-60).

`resample_particles` is the one step that couples particles (PF.cpp:473-500).  With the particle set
block-partitioned over `world` ranks it runs as SURVEY.md 8e describes:
  1. all-reduce(SUM) of the two local scalars [sum w, sum w^2]  -> global weight sum, Neff
  2. every rank scales its weights by 1/ws                        (PF.cpp:482-487)
  3. only if Neff < Nmin and resampling is on (PF.cpp:490):
       all-gather of the normalised weights -> every rank derives the identical keep[] from the shared
       select[] (stratified resample, PF.cpp:546-577 with the indexing defect of SURVEY 2.1 #8 removed),
       particle records whose source rank differs from the destination rank travel in ONE all-to-all-v;
       weights become 1/N.
The collectives go through a small `Comm` interface: `TorchComm` is torch.distributed (backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests), `SingleComm` is the world-size-1 case with no
communication at all.  The planning logic (who sends which particle where) is pure host code and is what the
world_size-2 gloo tests exercise with a numpy stand-in for the shard.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np

from . import _capi
from ._capi import F32, F64, Q_REF_EXACT, check


def _vp(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------------------------------------
# stratified resampling on the host (N is at most a few thousand scalars)
# ------------------------------------------------------------------------------------------------
def stratified_random(n: int, uniforms: np.ndarray, dtype=np.float32) -> np.ndarray:
    """PF.cpp:579-596 with uniform[0,1) strata offsets (Bailey's stratified_random; the reference's Gaussian
    offsets are defect #10 of SURVEY 2.1): select[i] = k/2 + i*k + (u[i]*k - k/2), k = 1/n."""
    t = np.dtype(dtype).type
    k = t(1) / t(n)
    # di_0 = k/2, di_i = di_(i-1) + k: a running sum in the particle dtype (np.cumsum accumulates sequentially in the
    # requested dtype, i.e. with the same roundings as the reference's loop); the rest is elementwise in that dtype
    steps = np.full(n, k, dtype=dtype)
    steps[0] = k / t(2)
    di = np.cumsum(steps, dtype=dtype)
    u = np.asarray(uniforms)[:n].astype(dtype)
    return (di + (u * k - k / t(2))).astype(dtype)


def stratified_keep(w_norm: np.ndarray, select: np.ndarray) -> np.ndarray:
    """PF.cpp:559-574 (intended form): 0-based index of the particle each slot keeps = the first i whose running weight
    sum exceeds the slot's stratum position (none: 0, as the reference's zero-initialised Keep).  The running sum is
    sequential in the particle dtype (np.cumsum accumulates in index order in the requested dtype: the reference's
    sequence of roundings); the search is what pf_resample_plan_kernel / pf_keep_kernel do on the device."""
    dtype = w_norm.dtype
    n = w_norm.shape[0]
    cum = np.cumsum(w_norm, dtype=dtype)
    keep = np.searchsorted(cum, np.asarray(select, dtype=dtype), side="right")
    keep[keep >= n] = 0
    return keep.astype(np.int32)


def plan_exchange(keep: np.ndarray, rank: int, world: int, n_local: int):
    """Who sends what where.  Global slot g lives on rank g // n_local at local index g % n_local.
    Returns (send_src_local, send_counts, recv_dst_local, recv_counts):
      send_src_local: local source indices, grouped by destination rank (rank order, ascending slot)
      recv_dst_local: local destination slots, grouped by source rank in the same canonical order."""
    n = keep.shape[0]
    assert n == world * n_local
    dst_rank = np.arange(n) // n_local
    src_rank = keep // n_local
    send_src, send_counts, recv_dst, recv_counts = [], [], [], []
    for d in range(world):
        sel = np.nonzero((src_rank == rank) & (dst_rank == d))[0]
        send_src.extend((keep[sel] % n_local).tolist())
        send_counts.append(int(sel.shape[0]))
    for s in range(world):
        sel = np.nonzero((dst_rank == rank) & (src_rank == s))[0]
        recv_dst.extend((sel % n_local).tolist())
        recv_counts.append(int(sel.shape[0]))
    return (np.array(send_src, dtype=np.int32), send_counts, np.array(recv_dst, dtype=np.int32), recv_counts)


# ------------------------------------------------------------------------------------------------
# communicators
# ------------------------------------------------------------------------------------------------
class SingleComm:
    rank, world = 0, 1

    def all_reduce_sum(self, vals: Sequence[float]) -> List[float]:
        return list(vals)


class TorchComm:
    """torch.distributed process group: backend 'nccl' is RCCL over xGMI on ROCm, 'gloo' on CPU."""

    def __init__(self, group=None, device=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = device if device is not None else torch.device("cpu")

    def all_reduce_sum(self, vals):
        t = self.torch.tensor(list(vals), dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().tolist()

    def all_gather(self, local):
        local = local.contiguous().reshape(-1)
        out = self.torch.empty(self.world * local.numel(), dtype=local.dtype, device=local.device)
        self.dist.all_gather_into_tensor(out, local, group=self.group)
        return out

    def all_to_all_v(self, send, send_counts, recv_counts, rec_len):
        out = self.torch.empty((sum(recv_counts), rec_len), dtype=send.dtype, device=send.device)
        self.dist.all_to_all_single(out, send, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts),
                                    group=self.group)
        return out


class RcclComm:
    """The engine's own RCCL communicator (cslam_comm_create, include/cslam.h): the sharded resample then runs entirely
    behind the C ABI (cslam_pf_resample_sharded) -- all-reduce, all-gather, the keep[] plan on the device and one grouped
    send/recv -- with no tensor library in the data path.  The 128-byte unique id is created on rank 0 and handed to
    the other ranks through torch.distributed (any backend; gloo will do), which is used for nothing else."""

    def __init__(self, device: int, group=None):
        import torch
        import torch.distributed as dist

        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        L = _capi.lib()
        buf = (C.c_ubyte * 128)()
        if self.rank == 0:
            check(L.cslam_comm_unique_id(buf))
        t = torch.tensor(list(buf), dtype=torch.uint8)
        backend = dist.get_backend(group)
        if backend == "nccl":
            t = t.cuda(device)
        dist.broadcast(t, src=0, group=group)
        raw = bytes(t.cpu().tolist())
        self._h = C.c_void_p(None)
        check(L.cslam_comm_create(C.c_char_p(raw), C.c_int(self.rank), C.c_int(self.world), C.c_int(device),
                                  C.byref(self._h)))
        self._L = L

    def all_reduce_sum(self, vals):  # (kept for callers that only want the sums)
        raise NotImplementedError("RcclComm is used through ParticleShard.resample_sharded")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.cslam_comm_destroy(self._h)
            self._h = C.c_void_p(None)


class LoopbackComm:
    """One rank of an in-process loopback communicator (cslam_comm_create_loopback): `world` ranks on ONE device, each
    driven from its own host thread.  Lets the multi-rank paths of cslam_pf_resample_sharded run on a one-GPU box."""

    def __init__(self, handle, rank: int, world: int, owner):
        self._h, self.rank, self.world, self._owner = handle, rank, world, owner

    @staticmethod
    def create(world: int, device: int = -1):
        L = _capi.lib()
        arr = (C.c_void_p * world)()
        check(L.cslam_comm_create_loopback(C.c_int(world), C.c_int(device), arr))
        owner = {"L": L, "open": world}
        return [LoopbackComm(C.c_void_p(arr[r]), r, world, owner) for r in range(world)]

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._owner["L"].cslam_comm_destroy(self._h)
            self._h = C.c_void_p(None)


# ------------------------------------------------------------------------------------------------
# the shard on the GPU
# ------------------------------------------------------------------------------------------------
class ParticleShard:
    def __init__(self, n_particles: int, max_features: int, dtype=np.float32, device: int = -1,
                 quirks: int = Q_REF_EXACT, n_global: int | None = None):
        self.dtype = np.dtype(dtype)
        self._L = _capi.lib()
        self._h = C.c_void_p(None)
        check(self._L.cslam_pf_create(C.c_int(n_particles), C.c_int(max_features),
                                      C.c_int(F32 if self.dtype == np.float32 else F64), C.c_int(device),
                                      C.c_int(quirks), C.byref(self._h)))
        self.n_local = n_particles
        self.n_global = n_global or n_particles
        # PF.cpp:327: w = 1/numParticles of the whole filter
        check(self._L.cslam_pf_set_uniform_weight(self._h, C.c_double(1.0 / self.n_global)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.cslam_pf_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers
    def _m22(self, M):
        return np.asfortranarray(np.asarray(M, dtype=self.dtype).reshape(2, 2))

    def _z(self, Z):
        Z = np.asarray(Z, dtype=self.dtype)
        m = 0 if Z.size == 0 else Z.reshape(2, -1, order="F").shape[1]
        return (np.ascontiguousarray(Z.reshape(-1, order="F")) if m else np.zeros(2, self.dtype)), m

    @property
    def n_features(self) -> int:
        a, b = C.c_int(0), C.c_int(0)
        check(self._L.cslam_pf_get_counts(self._h, C.byref(a), C.byref(b)))
        return b.value

    def synchronize(self):
        check(self._L.cslam_pf_synchronize(self._h))

    # ---- the per-particle path (every owned particle)
    def predict(self, v, swa, Q, wb, dt):
        """PF::predict -- PF.cpp:419-471."""
        Q = self._m22(Q)
        check(self._L.cslam_pf_predict(self._h, C.c_double(float(v)), C.c_double(float(swa)), _vp(Q),
                                       C.c_double(float(wb)), C.c_double(float(dt))))

    def observe_heading(self, phi, use_heading=False):
        """PF::observeHeading -- PF.cpp:382-417."""
        check(self._L.cslam_pf_observe_heading(self._h, C.c_double(float(phi)), C.c_int(1 if use_heading else 0)))

    def sample_proposal(self, Z, idf, R, normals):
        """PF::sampleProposal -- PF.cpp:502-544. normals: 3 x n_local standard-normal draws (input)."""
        Zc, m = self._z(Z)
        idf = np.ascontiguousarray(idf, dtype=np.int32)
        nrm = np.ascontiguousarray(np.asarray(normals, dtype=self.dtype).reshape(3, self.n_local))
        check(self._L.cslam_pf_sample_proposal(self._h, _vp(Zc), C.c_int(m), _vp(idf) if m else None,
                                               _vp(self._m22(R)), _vp(nrm)))

    def feature_update(self, Z, idf, R):
        """PF::featureUpdate -- PF.cpp:222-277."""
        Zc, m = self._z(Z)
        idf = np.ascontiguousarray(idf, dtype=np.int32)
        check(self._L.cslam_pf_feature_update(self._h, _vp(Zc), C.c_int(m), _vp(idf) if m else None,
                                              _vp(self._m22(R))))

    def add_features(self, Z, R):
        """PF::addOneNewFeature -- PF.cpp:9-60."""
        Zc, q = self._z(Z)
        check(self._L.cslam_pf_add_features(self._h, _vp(Zc), C.c_int(q), _vp(self._m22(R))))

    # ---- pieces of the resample step
    def weight_sums(self) -> Tuple[float, float]:
        s = (C.c_double * 2)()
        check(self._L.cslam_pf_weight_sums(self._h, s))
        return s[0], s[1]

    def scale_weights(self, scale: float):
        check(self._L.cslam_pf_scale_weights(self._h, C.c_double(float(scale))))

    def set_uniform_weight(self, w0: float):
        check(self._L.cslam_pf_set_uniform_weight(self._h, C.c_double(float(w0))))

    def get_weights(self) -> np.ndarray:
        w = np.zeros(self.n_local, dtype=self.dtype)
        check(self._L.cslam_pf_get_weights(self._h, _vp(w)))
        return w

    def set_weights(self, w):
        w = np.ascontiguousarray(w, dtype=self.dtype)
        assert w.shape[0] == self.n_local
        check(self._L.cslam_pf_set_weights(self._h, _vp(w)))

    def weights_device_ptr(self) -> int:
        p = C.c_void_p(None)
        check(self._L.cslam_pf_weights_device_ptr(self._h, C.byref(p)))
        return p.value

    @property
    def record_len(self) -> int:
        b = C.c_longlong(0)
        check(self._L.cslam_pf_record_bytes(self._h, C.byref(b)))
        return b.value // self.dtype.itemsize

    def gather_local(self, keep, w_new: float):
        keep = np.ascontiguousarray(keep, dtype=np.int32)
        assert keep.shape[0] == self.n_local
        check(self._L.cslam_pf_gather_local(self._h, _vp(keep), C.c_double(float(w_new))))

    def resample_local(self, select, n_effective: float, resample_status: bool, want_result: bool = True):
        """PF::resampleParticles (PF.cpp:473-500) for a shard that holds every particle, entirely on the device.
        -> (neff, resampled) or None when want_result is False (nothing returns to the host)."""
        select = np.ascontiguousarray(select, dtype=self.dtype)
        assert select.shape[0] == self.n_local
        if not want_result:
            check(self._L.cslam_pf_resample_local(self._h, _vp(select), C.c_double(float(n_effective)),
                                                  C.c_int(1 if resample_status else 0), None, None))
            return None
        neff, did = C.c_double(0.0), C.c_int(0)
        check(self._L.cslam_pf_resample_local(self._h, _vp(select), C.c_double(float(n_effective)),
                                              C.c_int(1 if resample_status else 0), C.byref(neff), C.byref(did)))
        return float(neff.value), bool(did.value)

    def resample_sharded(self, comm: "RcclComm", select, n_effective: float, resample_status: bool):
        """PF::resampleParticles over the particle set sharded across comm.world ranks, behind the C ABI
        (cslam_pf_resample_sharded).  select: the world * n_local strata positions, identical on every rank."""
        select = np.ascontiguousarray(select, dtype=self.dtype)
        assert select.shape[0] == self.n_local * comm.world
        neff, did = C.c_double(0.0), C.c_int(0)
        check(self._L.cslam_pf_resample_sharded(self._h, comm._h, _vp(select), C.c_double(float(n_effective)),
                                                C.c_int(1 if resample_status else 0), C.byref(neff), C.byref(did)))
        return float(neff.value), bool(did.value)

    def debug_last_exchange(self, world: int):
        """(send counts per destination, receive counts per source, local send indices) of the last sharded resample."""
        counts = (C.c_int * (2 * world))()
        n_send = C.c_int(0)
        check(self._L.cslam_pf_debug_last_exchange(self._h, counts, None, C.c_int(0), C.byref(n_send)))
        idx = np.zeros(max(n_send.value, 1), dtype=np.int32)
        if n_send.value:
            check(self._L.cslam_pf_debug_last_exchange(self._h, None, idx.ctypes.data_as(C.c_void_p), C.c_int(idx.shape[0]),
                                                       C.byref(n_send)))
        c = list(counts)
        return c[:world], c[world:], idx[: n_send.value]

    def observation_step(self, v, swa, Q, wb, dt, Z, idf, R, normals, select, n_effective: float, resample_status: bool):
        """predict + sampleProposal + featureUpdate + resampleParticles for a shard that holds every particle, in one
        C call with one staged copy; nothing returns to the host (see resample_stats)."""
        Zc, m = self._z(Z)
        idf = np.ascontiguousarray(idf, dtype=np.int32)
        nrm = np.ascontiguousarray(np.asarray(normals, dtype=self.dtype).reshape(3, self.n_local))
        sel = np.ascontiguousarray(select, dtype=self.dtype)
        assert sel.shape[0] == self.n_local
        check(self._L.cslam_pf_observation_step(self._h, C.c_double(float(v)), C.c_double(float(swa)), _vp(self._m22(Q)),
                                                C.c_double(float(wb)), C.c_double(float(dt)), _vp(Zc), C.c_int(m),
                                                _vp(idf) if m else None, _vp(self._m22(R)), _vp(nrm), _vp(sel),
                                                C.c_double(float(n_effective)), C.c_int(1 if resample_status else 0)))

    def resample_stats(self):
        """(resample calls, resamples performed, last Neff) from the device-side counters."""
        a, b, c = C.c_double(0.0), C.c_double(0.0), C.c_double(0.0)
        check(self._L.cslam_pf_resample_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return int(a.value), int(b.value), float(c.value)

    def pack_into(self, src_idx: np.ndarray, dptr: int):
        src_idx = np.ascontiguousarray(src_idx, dtype=np.int32)
        check(self._L.cslam_pf_pack(self._h, _vp(src_idx), C.c_int(src_idx.shape[0]), C.c_void_p(dptr)))

    def unpack_from(self, dst_idx: np.ndarray, dptr: int):
        dst_idx = np.ascontiguousarray(dst_idx, dtype=np.int32)
        check(self._L.cslam_pf_unpack(self._h, _vp(dst_idx), C.c_int(dst_idx.shape[0]), C.c_void_p(dptr)))

    # torch-tensor flavoured pack/unpack used by resample_particles (device buffers come from torch: plumbing)
    def pack(self, src_idx):
        import torch

        tdt = torch.float32 if self.dtype == np.float32 else torch.float64
        buf = torch.empty((len(src_idx), self.record_len), dtype=tdt, device="cuda")
        if len(src_idx):
            self.pack_into(src_idx, buf.data_ptr())
        return buf

    def unpack(self, dst_idx, buf):
        import torch

        if len(dst_idx):
            torch.cuda.synchronize()
            self.unpack_from(dst_idx, buf.contiguous().data_ptr())

    def weights_tensor(self):
        import torch

        return torch.from_numpy(self.get_weights()).cuda()

    # ---- host access to single particles (tests, reporting)
    def get_particle(self, i: int):
        nf = self.n_features
        w = np.zeros(1, self.dtype)
        Xv = np.zeros(3, self.dtype)
        Pv = np.zeros((3, 3), self.dtype, order="F")
        XF = np.zeros((2, nf), self.dtype, order="F")
        PF = np.zeros((4, nf), self.dtype, order="F")
        check(self._L.cslam_pf_get_particle(self._h, C.c_int(i), _vp(w), _vp(Xv), _vp(Pv), _vp(XF) if nf else None,
                                            _vp(PF) if nf else None))
        return w[0], Xv, Pv, XF, PF

    def set_particle(self, i: int, w, Xv, Pv, XF, PF):
        XF = np.asfortranarray(np.asarray(XF, dtype=self.dtype).reshape(2, -1, order="F"))
        nf = XF.shape[1]
        PF = np.asfortranarray(np.asarray(PF, dtype=self.dtype).reshape(4, -1, order="F"))
        wv = np.array([w], dtype=self.dtype)
        Xv = np.ascontiguousarray(Xv, dtype=self.dtype)
        Pv = np.asfortranarray(np.asarray(Pv, dtype=self.dtype).reshape(3, 3))
        check(self._L.cslam_pf_set_particle(self._h, C.c_int(i), _vp(wv), _vp(Xv), _vp(Pv), _vp(XF) if nf else None,
                                            _vp(PF) if nf else None, C.c_int(nf)))


# ------------------------------------------------------------------------------------------------
# PF::resampleParticles over a sharded particle set
# ------------------------------------------------------------------------------------------------
def resample_particles(shard, comm, n_effective: int, resample_status: bool, select: np.ndarray | None = None,
                       uniforms: np.ndarray | None = None):
    """PF::resampleParticles(particles, numEffective, resampleStatus) -- PF.cpp:473-500 -- for the particle set
    block-partitioned over comm.world shards.  `shard` is a ParticleShard (or anything with its resample
    surface: weight_sums, scale_weights, weights_tensor, pack, unpack, gather_local, set_uniform_weight,
    n_local, dtype).  `select` are the N strata positions (PF.cpp:557); when None they are built from
    `uniforms` (N uniform[0,1) draws that every rank must pass identically).  Returns (neff, resampled)."""
    n_local = shard.n_local
    n = n_local * comm.world
    if comm.world == 1 and hasattr(shard, "resample_local") and not getattr(shard, "host_resample", False):
        # one shard holds everything: sums, normalisation, Neff, decision, keep[] and the moves stay on the device
        if select is None:
            assert uniforms is not None, "pass select[] or the uniform draws it is built from"
            select = stratified_random(n, uniforms, shard.dtype)
        return shard.resample_local(np.asarray(select, dtype=shard.dtype), n_effective, resample_status)
    if isinstance(comm, RcclComm):
        # sharded over GPUs: the three collectives, the plan and the moves all run behind the C ABI on the device
        if select is None:
            assert uniforms is not None, "pass select[] or the uniform draws it is built from"
            select = stratified_random(n, uniforms, shard.dtype)
        return shard.resample_sharded(comm, select, n_effective, resample_status)
    s1, s2 = shard.weight_sums()
    ws, ws2 = comm.all_reduce_sum([s1, s2])          # collective 1: two scalars
    shard.scale_weights(1.0 / ws)                     # PF.cpp:482-487
    neff = (ws * ws) / ws2 if ws2 > 0 else 0.0        # 1 / sum (w/ws)^2, PF.cpp:549-554
    if not (neff < n_effective and resample_status):  # PF.cpp:490
        return neff, False
    if comm.world == 1:
        w_all = shard.get_weights()
    else:
        w_all = comm.all_gather(shard.weights_tensor()).cpu().numpy()  # collective 2: N weights
    w_all = np.ascontiguousarray(w_all, dtype=shard.dtype)
    if select is None:
        assert uniforms is not None, "pass select[] or the uniform draws it is built from"
        select = stratified_random(n, uniforms, shard.dtype)
    keep = stratified_keep(w_all, np.asarray(select, dtype=shard.dtype))
    if comm.world == 1:
        shard.gather_local(keep, 1.0 / n)
        return neff, True
    send_src, send_counts, recv_dst, recv_counts = plan_exchange(keep, comm.rank, comm.world, n_local)
    send = shard.pack(send_src)
    rec_len = send.shape[1]
    recv = comm.all_to_all_v(send, send_counts, recv_counts, rec_len)  # collective 3: particle records
    shard.unpack(recv_dst, recv)
    shard.set_uniform_weight(1.0 / n)                 # PF.cpp:495
    return neff, True
