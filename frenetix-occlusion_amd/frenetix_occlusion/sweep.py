"""Batched trajectory x agent criticality sweep on the GPU (host side).

Mirrors what M calls of the reference's ``Metric.evaluate_metrics`` compute (metrics/metric.py:35-100), but for the
whole candidate batch in one launch of the HIP kernel behind ``fo_sweep_run`` (include/fo_hip.h).
PyTorch only owns the device buffers and the stream; all arithmetic happens in libfo_hip.so.
"""
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _native as N

# entries of harm_params.json that the reference reads (config/harm_params.json:26-29,40-45,98-101)
DEFAULT_HARM_COEFF = dict(lr4s_const=-4.457, lr4s_speed=0.177, lr4s_side=0.244, lr4s_rear=-0.431,
                          lr1s_const=-4.591, lr1s_speed=0.185, ped_const=3.164, ped_speed=0.288)
DEFAULT_METRICS = ("hr", "ttc", "ttce", "dce", "wttc", "cp")  # config/config.yaml:6-12


def list_views(raw, A, Tm1, M):
    """The five per-timestep lists (order of ``_native.LST``: cp, ego_harm, obst_harm, ego_risk, obst_risk) as strided
    ``[A, T-1, M]`` views of the raw list buffer ``fo_sweep_run`` fills (layout: include/fo_hip.h -- cp dense, the
    harms and the risks as interleaved pairs).  Works for torch tensors and numpy arrays alike."""
    n = A * Tm1 * M
    cp = raw[:n].reshape(A, Tm1, M)
    harm = raw[n:3 * n].reshape(A, Tm1, M, 2)
    risk = raw[3 * n:5 * n].reshape(A, Tm1, M, 2)
    return [cp, harm[..., 0], harm[..., 1], risk[..., 0], risk[..., 1]]


@dataclass
class SweepResult:
    cost: torch.Tensor                    # [M, 16] float64
    safe: torch.Tensor                    # [M] uint8
    pair_f: Optional[torch.Tensor] = None  # [12, A, M] float64
    pair_i: Optional[torch.Tensor] = None  # [4, A, M] int32
    lists_raw: Optional[torch.Tensor] = None  # [5 A (T-1) M] float64 (or float32, lists="f32"), layout of include/fo_hip.h
    lists_shape: tuple = (0, 0, 0)         # (A, T-1, M)
    # trajectory batch split over the ranks of a process group (distributed.CostGather): every output above holds this
    # rank's rows [rows[0], rows[1]) of the batch; cost_all is the all-gathered cost matrix [M_total, 16] of every rank
    cost_all: Optional[torch.Tensor] = None
    rows: Optional[tuple] = None

    def list_views(self):
        """five strided [A, T-1, M] views (no copy), order of ``_native.LST``"""
        return None if self.lists_raw is None else list_views(self.lists_raw, *self.lists_shape)

    @property
    def lists(self):
        """dense [5, A, T-1, M] copy (tests and small batches; large batches should use ``list_views``)"""
        return None if self.lists_raw is None else torch.stack(self.list_views())


def _vehicle_tuple(vp):
    if isinstance(vp, (tuple, list, np.ndarray)):
        return tuple(float(q) for q in vp)
    return tuple(float(getattr(vp, k)) for k in ("length", "width", "wb_rear_axle", "mass", "a_max"))


class MetricSweep:
    def __init__(self, vehicle_params, dt, metrics=DEFAULT_METRICS, thresholds=None, harm_coeff=None, device=0, ctx=None):
        if not torch.cuda.is_available():
            raise RuntimeError("MetricSweep needs a ROCm GPU (no CPU fallback)")
        self.device = torch.device("cuda", int(device) if not isinstance(device, torch.device) else device.index or 0)
        self._dev_index = self.device.index
        self.ctx = ctx or N.Context(self.device.index)   # one fo_ctx per ego per GPU; shared with the scene stage
        self.dt = float(dt)
        self.metrics = tuple(metrics)
        self.A = 0
        self.Ta = 0
        self._agent_tensors = None
        self.configure(vehicle_params, thresholds, harm_coeff)

    def configure(self, vehicle_params, thresholds=None, harm_coeff=None, metrics=None):
        if metrics is not None:
            self.metrics = tuple(metrics)
        self.vehicle = _vehicle_tuple(vehicle_params)
        veh = N.Vehicle(*self.vehicle)
        hc = N.HarmCoeff(**(harm_coeff or DEFAULT_HARM_COEFF))
        thr = thresholds if isinstance(thresholds, N.Thresholds) else N.make_thresholds(thresholds)
        self.ctx.call("fo_sweep_configure", veh, hc, thr, N.metric_mask(self.metrics), self.dt)

    def reserve(self, M, T, A, Ta):
        self.ctx.call("fo_sweep_reserve", int(M), int(T), int(A), int(Ta))

    def _dev(self, t, dtype=torch.float64):
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(np.ascontiguousarray(t))
        return t.to(device=self.device, dtype=dtype).contiguous()

    def _staging(self, n, M, T):
        """(host view, device view) [n][M][T] of the pinned staging buffer and its device twin, free to be written: the
        previous copy out of the pinned buffer has finished, and the kernels of the previous run() that read the device
        buffer are ordered before whatever the current stream does next"""
        need = n * M * T
        if getattr(self, "_pin", None) is None or self._pin.numel() < need:
            self._pin = torch.empty(need, dtype=torch.float64).pin_memory()
            self._stage = torch.empty(need, dtype=torch.float64, device=self.device)
            self._pin_free = None
        if self._pin_free is not None:
            self._pin_free.synchronize()            # the previous copy out of the pinned buffer must have finished
        if getattr(self, "_stage_busy", None) is not None:
            # the device staging buffer is still being read by the kernels of the previous run(); if the caller has
            # switched streams in between, the new copy must wait for them (same stream: a no-op in stream order)
            torch.cuda.current_stream(self.device).wait_event(self._stage_busy)
        return self._pin[:need].view(n, M, T), self._stage[:need].view(n, M, T)

    def _staging_copy(self, host, dev):
        dev.copy_(host, non_blocking=True)
        self._pin_free = torch.cuda.Event()
        self._pin_free.record()

    def _upload_packed(self, arrays):
        """host arrays [M,T] (x, y, theta, v[, a]) -> device tensors through ONE pinned staging buffer and ONE
        host-to-device copy (five pageable copies of a 2 000 x 31 batch cost 0.1 ms more than the sweep itself)"""
        n, (M, T) = len(arrays), arrays[0].shape
        host, dev = self._staging(n, M, T)
        hv = host.numpy()
        for i, arr in enumerate(arrays):
            np.copyto(hv[i], arr, casting="unsafe")
        self._staging_copy(host, dev)
        return [dev[i] for i in range(n)]

    FIELDS = ("x", "y", "theta", "v", "a")

    def upload_trajectory_objects(self, trajectories):
        """list of the planner's trajectory objects (``.cartesian.{x,y,theta,v,a}``: what the reference's per-trajectory call
        receives, interface.py:216-219) -> dict of device tensors [M,T], packed into the pinned staging buffer by the native
        helper (csrc/fo_pyhost.c: 0.45 ms for 2 000 objects against 3.6 ms for the numpy gather) and sent with one
        host-to-device copy.  None when the helper is not built (the caller packs with numpy)."""
        H = N.pyhost()
        M = len(trajectories)
        if H is None or M == 0:
            return None
        T = len(trajectories[0].cartesian.x)
        host, dev = self._staging(5, M, T)
        H.pack_trajectories(trajectories if isinstance(trajectories, (list, tuple)) else list(trajectories), host.numpy(), self.FIELDS)
        self._staging_copy(host, dev)
        return {k: dev[i] for i, k in enumerate(self.FIELDS)}

    def _stream(self):
        return N.current_stream(self._dev_index)

    def set_agents(self, pos, yaw, v, cov, shape, raw_dims, type, len, check=True):
        """pos [A,Ta,2], yaw/v [A,Ta], cov [A,Ta,2,2], shape/raw_dims [A,2], type/len [A] (numpy or torch)."""
        pos, yaw, v, cov = self._dev(pos), self._dev(yaw), self._dev(v), self._dev(cov)
        shape, raw = self._dev(shape), self._dev(raw_dims)
        typ, ln = self._dev(type, torch.int32), self._dev(len, torch.int32)
        A = int(pos.shape[0])
        Ta = int(pos.shape[1]) if A else 0
        if A:
            assert pos.shape == (A, Ta, 2) and yaw.shape == (A, Ta) and v.shape == (A, Ta)
            assert cov.numel() == A * Ta * 4 and shape.shape == (A, 2) and raw.shape == (A, 2)
        self._agent_tensors = (pos, yaw, v, cov, shape, raw, typ, ln)  # keep alive until the stream has consumed them
        p = lambda t: t.data_ptr() if t.numel() else None
        self.ctx.call("fo_sweep_set_agents", A, Ta, p(pos), p(yaw), p(v), p(cov), p(shape), p(raw), p(typ), p(ln),
                      self._stream())
        self.A, self.Ta = A, Ta
        if check:
            self.ctx.call("fo_sweep_check", self._stream())

    def alloc_out(self, M, T, A, mode, lists="f64", cost=None) -> SweepResult:
        """output buffers of one batch shape; ``cost``: a caller's [M,16] float64 block to write the cost rows into (the
        block a rank contributes to the all-gather of a sharded batch, distributed.CostGather.block)"""
        ldt = torch.float64 if lists == "f64" else torch.float32
        out = SweepResult(cost=cost if cost is not None else torch.empty((M, N.NC), dtype=torch.float64, device=self.device),
                          safe=torch.empty((M,), dtype=torch.uint8, device=self.device))
        if mode in ("pair", "full"):
            out.pair_f = torch.empty((N.NPF, A, M), dtype=torch.float64, device=self.device)
            out.pair_i = torch.empty((N.NPI, A, M), dtype=torch.int32, device=self.device)
        if mode == "full":
            out.lists_raw = torch.empty((N.NL * A * max(T - 1, 0) * M,), dtype=ldt, device=self.device)
        out.lists_shape = (A, max(T - 1, 0), M)
        return out

    def _check_out(self, out, M, T, A, mode, ldt=torch.float64):
        """a reused SweepResult must match the batch: the kernels write with the strides of the *current* M/A/T"""
        want = {"cost": ((M, N.NC), torch.float64), "safe": ((M,), torch.uint8),
                "pair_f": ((N.NPF, A, M), torch.float64) if mode in ("pair", "full") else None,
                "pair_i": ((N.NPI, A, M), torch.int32) if mode in ("pair", "full") else None,
                "lists_raw": ((N.NL * A * max(T - 1, 0) * M,), ldt) if mode == "full" else None}
        for name, w in want.items():
            t = getattr(out, name)
            if w is None:
                if t is not None:
                    raise ValueError(f"out.{name} is set but mode '{mode}' does not write it")
                continue
            if t is None or tuple(t.shape) != w[0] or t.dtype != w[1] or t.device != self.device or not t.is_contiguous():
                got = None if t is None else (tuple(t.shape), t.dtype, str(t.device))
                raise ValueError(f"out.{name}: need contiguous {w[0]} {w[1]} on {self.device}, got {got}")

    def run(self, x, y, theta, v, a=None, mode="reduced", out: Optional[SweepResult] = None, lists="f64",
            autotune=0) -> SweepResult:
        """x,y,theta,v[,a]: [M,T].  mode: 'reduced' (cost+safe), 'pair' (+ per-pair scalars), 'full' (+ lists).
        lists: element type of the per-timestep lists of mode 'full' -- 'f64' (the reference's numpy dtype), 'f32'
        (half the bytes; cost / safe / pair outputs are identical, see fo_sweep_set_list_format in include/fo_hip.h) or
        'f32x' (float32 storage of the float64 results).  autotune = n > 0: instead of one run, let the library time its
        agents-per-wave settings on this batch (n launches each) and keep the best for the shape (``last_autotune``)."""
        if lists not in N.LIST_FORMAT:
            raise ValueError(f"unknown list format '{lists}'")
        ldt = torch.float64 if lists == "f64" else torch.float32
        if lists != getattr(self.ctx, "list_format", "f64"):     # per-context state: cached on the Context, not here
            self.ctx.call("fo_sweep_set_list_format", N.LIST_FORMAT[lists])
            self.ctx.list_format = lists
        ins = [x, y, theta, v] + ([a] if a is not None else [])
        if all(isinstance(q, np.ndarray) and q.ndim == 2 and q.shape == ins[0].shape for q in ins) and ins[0].size:
            up = self._upload_packed(ins)
            x, y, theta, v = up[:4]
            a = up[4] if a is not None else None
        else:
            x, y, theta, v = self._dev(x), self._dev(y), self._dev(theta), self._dev(v)
            a = self._dev(a) if a is not None else None
        M, T = int(x.shape[0]), int(x.shape[1])
        A = self.A
        if mode not in ("reduced", "pair", "full"):
            raise ValueError(f"unknown output mode '{mode}'")
        if out is not None:
            self._check_out(out, M, T, A, mode, ldt)
        else:
            out = self.alloc_out(M, T, A, mode, lists)
        p = lambda t: t.data_ptr() if (t is not None and t.numel()) else None
        self._last_inputs = (x, y, theta, v, a)
        if autotune:
            # measure the kernel's agents-per-wave settings on THIS batch and keep the best for the shape (fo_sweep_autotune)
            import ctypes as C
            best, ms4 = C.c_int(), (C.c_double * 4)()
            self.ctx.call("fo_sweep_autotune", M, T, p(x), p(y), p(theta), p(v), p(a), p(out.cost), p(out.safe),
                          p(out.pair_f), p(out.pair_i), p(out.lists_raw), int(autotune), C.byref(best), ms4, self._stream())
            self.last_autotune = {"agents_per_wave": best.value, "ms": {k: ms4[i] for i, k in enumerate((1, 2, 4, 8))}}
        else:
            self.ctx.call("fo_sweep_run", M, T, p(x), p(y), p(theta), p(v), p(a), p(out.cost), p(out.safe),
                          p(out.pair_f), p(out.pair_i), p(out.lists_raw), self._stream())
        if getattr(self, "_stage", None) is not None:
            self._stage_busy = torch.cuda.Event()   # last consumer of the staging buffer (see _upload_packed)
            self._stage_busy.record()
        out.lists_shape = (A, max(T - 1, 0), M)
        return out
