"""ctypes binding of libsot_hip.so (include/sot_hip.h).  PyTorch supplies device memory and
streams only; every computation on the hot path happens inside the HIP library.

There is deliberately NO fallback: if the library is missing or a tensor is not on a HIP device
the call raises.
"""
from __future__ import annotations

import ctypes
import os
import threading

import torch

from . import build as _build

FLAG_SQUARE = 1
FLAG_DONT_NORMALIZE = 2
FLAG_LIMIT_Q = 4
FLAG_REQUIRE_SORT = 8
FLAG_PRENORMALIZED = 16    # weights are used as given (the functional form wasserstein_1d)
FLAG_NO_SPECIALIZE = 32  # diagnostic: generic forward kernel only (include/sot_hip.h)
FLAG_SAME_GRID = 64      # both measures live on one grid: the p = 1 forward runs the merge-free kernel (include/sot_hip.h)
FLAG_NO_AREA = 128       # diagnostic: ignore FLAG_SAME_GRID
FLAG_TIE_FREE_GRADIENT = 256   # opt-in: merge-free p = 1 training form (include/sot_hip.h: SOT_FLAG_TIE_FREE_GRADIENT)

SOT_OK = 0
SOT_ERR_INVALID_P = -1
SOT_ERR_BAD_SHAPE = -2
SOT_ERR_UNSUPPORTED_SIZE = -3
SOT_ERR_NULL_POINTER = -4
SOT_ERR_WORKSPACE = -5
SOT_ERR_LAUNCH = -6
ABI_VERSION = 13                  # include/sot_hip.h: SOT_ABI_VERSION (bumped with every signature change)
COMPLETION_COUNTER_WORDS = 16    # include/sot_hip.h: SOT_COMPLETION_COUNTER_WORDS

_vp = ctypes.c_void_p


class SotProblem(ctypes.Structure):
    _fields_ = [("x", _vp), ("y", _vp), ("xpos", _vp), ("ypos", _vp),
                ("B", ctypes.c_int64), ("n", ctypes.c_int32), ("m", ctypes.c_int32),
                ("x_row_stride", ctypes.c_int64), ("y_row_stride", ctypes.c_int64),
                ("xpos_row_stride", ctypes.c_int64), ("ypos_row_stride", ctypes.c_int64),
                ("p", ctypes.c_float), ("flags", ctypes.c_uint32),
                ("xperm", _vp), ("yperm", _vp), ("perm_is_identity", _vp),
                ("row_perm_out", _vp), ("row_perm_in", _vp)]


EXPORTS = {
    "sot_abi_version": (ctypes.c_int, []),
    "sot_status_string": (ctypes.c_char_p, [ctypes.c_int]),
    "sot_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(SotProblem)]),
    "sot_w1d_loss_and_grad": (ctypes.c_int, [ctypes.POINTER(SotProblem), _vp, ctypes.c_double, _vp, _vp, ctypes.c_float, _vp, _vp, _vp,
                                             ctypes.c_size_t, _vp]),
    "sot_scale_inplace": (ctypes.c_int, [_vp, ctypes.c_int64, _vp, _vp]),
    "sot_profile_next_launch": (ctypes.c_int, [ctypes.c_int]),
    "sot_profile_elapsed_ms": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_float)]),
    "sot_prepare_positions": (ctypes.c_int, [_vp, _vp, ctypes.c_int32, ctypes.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sot_prepare_unit_positions": (ctypes.c_int, [_vp, _vp, ctypes.c_int32, ctypes.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sot_w1d_forward": (ctypes.c_int, [ctypes.POINTER(SotProblem), _vp, _vp, ctypes.c_size_t, _vp]),
    "sot_w1d_reduce_mean": (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_double, ctypes.c_int, ctypes.c_float,
                                           _vp, _vp, _vp]),
    "sot_w1d_backward": (ctypes.c_int, [ctypes.POINTER(SotProblem), _vp, ctypes.c_int64, ctypes.c_float, _vp, _vp, _vp,
                                        ctypes.c_size_t, _vp]),
    "sot_w1d_position_grad": (ctypes.c_int, [ctypes.POINTER(SotProblem), _vp, ctypes.c_int64, ctypes.c_float, _vp, _vp, _vp,
                                             ctypes.c_size_t, _vp]),
    "sot_column_sum": (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64, _vp, _vp]),
    "sot_w1d_loss": (ctypes.c_int, [ctypes.POINTER(SotProblem), _vp, ctypes.c_double, ctypes.c_int, ctypes.c_float, _vp, _vp,
                                    _vp, _vp, ctypes.c_size_t, _vp]),
    "sot_w1d_quantiles": (ctypes.c_int, [ctypes.POINTER(SotProblem), _vp, _vp, _vp, _vp, _vp, _vp,
                                         ctypes.c_size_t, _vp]),
    "sot_w1d_forward_csr": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64,
                                           ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_uint32, _vp, _vp]),
    "sot_segmented_sort": (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64, _vp, _vp, _vp]),
    "sot_stft_frames": (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int]),
    "sot_stft_mag_forward": (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _vp, ctypes.c_int, ctypes.c_int,
                                            _vp, _vp]),
    "sot_oscillator_bank_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int]),
    "sot_oscillator_bank_forward": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_float, _vp, _vp,
                                                   ctypes.c_size_t, _vp]),
    "sot_oscillator_bank_backward": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_float, _vp, _vp,
                                                    _vp, _vp, ctypes.c_size_t, ctypes.c_int, _vp]),
    "sot_synth_envelopes_forward": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                                   ctypes.c_float, _vp, _vp, _vp]),
    "sot_synth_envelopes_backward": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                                    ctypes.c_float, _vp, _vp, _vp, _vp, _vp]),
    "sot_synth_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    "sot_synth_forward": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_float,
                                         _vp, _vp, ctypes.c_size_t, _vp]),
    "sot_synth_backward": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_float,
                                          _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, ctypes.c_int, _vp]),
    "sot_synth_tap_table_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int64]),
    "sot_synth_tap_tables": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int64, _vp, _vp]),
    "sot_mss_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int64, _vp, ctypes.c_int]),
    "sot_mss_loss_and_grad": (ctypes.c_int, [_vp, ctypes.c_int64, _vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _vp, _vp, ctypes.c_int,
                                             ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_float, _vp, _vp, _vp,
                                             ctypes.c_size_t, _vp]),
    "sot_spec_distance_workspace_bytes": (ctypes.c_size_t, []),
    "sot_spec_distance_forward": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int,
                                                 _vp, ctypes.c_int, _vp, ctypes.c_size_t, _vp]),
    "sot_spec_distance_backward": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                                  ctypes.c_int, _vp, ctypes.c_float, _vp, _vp, _vp]),
    "sot_spec_distance_rows_forward": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                                      ctypes.c_int, _vp, ctypes.c_int, _vp]),
    "sot_spec_distance_rows_backward": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                                       ctypes.c_int, _vp, ctypes.c_float, _vp, _vp, _vp]),
    "sot_stft_backward_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    "sot_stft_mag_forward_pair": (ctypes.c_int, [_vp, ctypes.c_int64, _vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _vp, ctypes.c_int,
                                                 ctypes.c_int, _vp, _vp]),
    "sot_stft_mag_backward": (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _vp, ctypes.c_int, ctypes.c_int,
                                             _vp, _vp, _vp, ctypes.c_int, _vp, ctypes.c_size_t, _vp]),
    "sot_stft_mag_forward_spec": (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _vp, ctypes.c_int, ctypes.c_int,
                                                 _vp, _vp, _vp]),
    "sot_stft_mag_forward_pair_spec": (ctypes.c_int, [_vp, ctypes.c_int64, _vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _vp, ctypes.c_int,
                                                      ctypes.c_int, _vp, _vp, _vp]),
    "sot_stft_mag_backward_spec": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _vp, ctypes.c_int, ctypes.c_int,
                                                  _vp, _vp, _vp, ctypes.c_int, _vp, ctypes.c_size_t, _vp]),
}

_lib = None
_lock = threading.Lock()


def library_path() -> str:
    """The in-tree library, or a diagnostic variant named by SOT_LIB_PATH (tools/: A/B builds; never set in production)."""
    return os.environ.get("SOT_LIB_PATH") or _build.LIB


def load(build_if_missing: bool = True):
    """Load libsot_hip.so; raises if it cannot be found/built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = library_path()
        if path == _build.LIB and _build.is_stale():   # missing, or built from other sources than the tree's (build.py: digest)
            if not build_if_missing:
                what = "is missing" if not os.path.exists(path) else "was built from different sources than the ones in this tree"
                raise RuntimeError(f"{path} {what}: run `python __graft_entry__.py` (build()) first")
            _build.build(force=True)
        elif not os.path.exists(path):
            raise RuntimeError(f"{path} is missing")
        lib = ctypes.CDLL(path)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(lib, name)  # AttributeError here = ABI mismatch, reported loudly
            fn.restype = res
            fn.argtypes = args
        if lib.sot_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libsot_hip.so has ABI version {lib.sot_abi_version()}, this binding expects {ABI_VERSION}")
        _lib = lib
    return _lib


class SotError(RuntimeError):
    status = None   # the library's negative status code (include/sot_hip.h: sot_status)


_glue = None
_glue_tried = False


def glue():
    """The C++ host path of the module's default call (csrc/sot_torch_glue.cpp -> _sot_glue.so, built in-tree by build.build_glue):
    one pybind11 call per forward and a C++ autograd node instead of this file's ctypes marshalling + a Python autograd.Function
    (~100 -> ~60 us of host time per forward + backward at the paper's step size).  Returns the bound extension module, or None
    -- with one warning -- when it is missing, stale or disabled (SOT_NO_GLUE=1): the callers then use this file's Python binding,
    which launches the same kernels."""
    global _glue, _glue_tried
    if _glue_tried:
        return _glue
    load()   # before taking the (non-reentrant) lock: load() takes it too
    with _lock:
        if _glue_tried:
            return _glue
        mod = None
        if os.environ.get("SOT_NO_GLUE", "0") != "1":
            try:
                if _build.glue_is_stale():
                    raise RuntimeError(f"{_build.GLUE_LIB} is missing or was built from other sources (run `python __graft_entry__.py`)")
                import importlib.util
                spec = importlib.util.spec_from_file_location("_sot_glue", _build.GLUE_LIB)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                if mod.bind(library_path()) != ABI_VERSION:
                    raise RuntimeError("ABI version mismatch")
            except Exception as exc:  # noqa: BLE001 -- the Python binding is complete; say why the C++ path is off
                import warnings
                warnings.warn(f"sot_amd: the C++ host path is not available ({exc}); using the Python binding (same kernels, more host time)")
                mod = None
        _glue, _glue_tried = mod, True
    return _glue


def problem_flags(p, flags, plan) -> int:
    """The flag word of a call: the caller's flags plus SOT_FLAG_SAME_GRID when the plan says both measures live on one grid --
    asked only by the p = 1 forward without cutoff, the one call with a use for the answer (it may cost one synchronisation per plan)."""
    flags = int(flags)
    if plan is not None and p == 1 and not (flags & (FLAG_LIMIT_Q | FLAG_NO_AREA | FLAG_PRENORMALIZED)) and plan.same_grid():
        flags |= FLAG_SAME_GRID
    return flags


def check(rc: int, p=None):
    if rc == SOT_OK:
        return
    if rc == SOT_ERR_INVALID_P:  # same exception type and text as losses.py:271
        raise AssertionError(f"The OT loss is only valid for p>=1, {p} was given")
    err = SotError(f"libsot_hip: {load().sot_status_string(rc).decode()} (status {rc})")
    err.status = rc
    raise err


def require_hip(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "sot_amd is the MI355X HIP implementation of the SOT loss and has no CPU path: got a "
                f"{t.device} tensor. Move inputs to the GPU (torch device 'cuda' on ROCm).")
        if t.dtype != torch.float32:
            raise TypeError(f"sot_amd computes in float32; got {t.dtype} (convert with .float())")


def stream_ptr(device) -> int:
    """Raw hipStream_t of torch's current stream on `device` (cheap: no Stream object is built)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(idx)


class _on_device:
    """`with torch.cuda.device(dev)` only when dev is not already current (the common case costs nothing)."""
    __slots__ = ("ctx",)

    def __init__(self, device):
        idx = device.index
        self.ctx = None if (idx is None or idx == torch.cuda.current_device()) else torch.cuda.device(idx)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)


def _ptr(t):
    return None if t is None else t.data_ptr()


def rows_view(t: torch.Tensor) -> torch.Tensor:
    """2-D fp32 view with unit inner stride (copies only if the inner dim is strided)."""
    if t.stride(-1) != 1 and t.shape[-1] > 1:
        t = t.contiguous()
    if t.shape[0] > 1 and t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t


def row_permutations(x, y, xpos, ypos, flags):
    """A [B, n + m] uint16 buffer for the per-row sort permutations (sot_problem.row_perm_out / row_perm_in), or None when the call does not
    sort per row (shared positions, no REQUIRE_SORT, rows beyond uint16)."""
    if not (flags & FLAG_REQUIRE_SORT) or xpos.ndim != 2 or ypos.ndim != 2 or max(x.shape[1], y.shape[1]) > 65535:
        return None
    return torch.empty((x.shape[0], x.shape[1] + y.shape[1]), dtype=torch.uint16, device=x.device)


def make_problem(x, y, xpos, ypos, p, flags, plan=None, perm_out=None, perm_in=None) -> SotProblem:
    B, n = x.shape
    m = y.shape[1]
    pr = SotProblem()
    pr.row_perm_out = perm_out.data_ptr() if perm_out is not None else None
    pr.row_perm_in = perm_in.data_ptr() if perm_in is not None else None
    if plan is not None:
        plan.use_on_current_stream(x.device)
    pr.x, pr.y = x.data_ptr(), y.data_ptr()
    pr.B, pr.n, pr.m = B, n, m
    pr.x_row_stride = x.stride(0) if B > 1 else n
    pr.y_row_stride = y.stride(0) if B > 1 else m
    pr.p, pr.flags = float(p), problem_flags(p, flags, plan)
    if plan is not None:
        pr.xpos, pr.ypos = plan.xpos_sorted.data_ptr(), plan.ypos_sorted.data_ptr()
        pr.xperm, pr.yperm = plan.xperm.data_ptr(), plan.yperm.data_ptr()
        pr.perm_is_identity = plan.ident.data_ptr()
        pr.xpos_row_stride = pr.ypos_row_stride = 0
    else:
        pr.xpos, pr.ypos = xpos.data_ptr(), ypos.data_ptr()
        pr.xpos_row_stride = 0 if xpos.ndim == 1 else (xpos.stride(0) if B > 1 else n)
        pr.ypos_row_stride = 0 if ypos.ndim == 1 else (ypos.stride(0) if B > 1 else m)
        pr.xperm = pr.yperm = pr.perm_is_identity = None
    return pr


class PositionPlan:
    """Sorted shared positions + permutation + identity flags (sot_prepare_positions); unit: of xpos / xpos.max() and ypos / ypos.max()
    (sot_prepare_unit_positions: the grid the reference's trainer rebuilds every step, trainer.py:196-197, out of one launch)."""

    def __init__(self, xpos: torch.Tensor, ypos: torch.Tensor, unit: bool = False):
        require_hip(xpos, ypos)
        lib = load()
        n, m = xpos.numel(), ypos.numel()
        dev = xpos.device
        xp, yp = xpos.contiguous(), ypos.contiguous()
        self._same_tensor = n == m and xpos.data_ptr() == ypos.data_ptr()
        g = glue()
        with _on_device(dev):
            self.stream = stream_ptr(dev)
            if g is not None and xp.ndim == 1 and yp.ndim == 1:
                # one allocation + one launch (the reference's trainer makes fresh positions, hence a fresh plan, every step)
                self.xpos_sorted, self.ypos_sorted, self.xperm, self.yperm, self.ident = g.make_plan(xp, yp, bool(unit))
            else:
                self.xpos_sorted = torch.empty(n, dtype=torch.float32, device=dev)
                self.ypos_sorted = torch.empty(m, dtype=torch.float32, device=dev)
                self.xperm = torch.empty(n, dtype=torch.int32, device=dev)
                self.yperm = torch.empty(m, dtype=torch.int32, device=dev)
                self.ident = torch.empty(2, dtype=torch.int32, device=dev)
                prepare = lib.sot_prepare_unit_positions if unit else lib.sot_prepare_positions
                check(prepare(xp.data_ptr(), yp.data_ptr(), n, m, self.xpos_sorted.data_ptr(), self.ypos_sorted.data_ptr(), self.xperm.data_ptr(),
                              self.yperm.data_ptr(), self.ident.data_ptr(), self.stream))
            # the plan is cached and may be consumed from other streams: they wait on this event (see use_plan)
            self.ready = torch.cuda.Event()
            self.ready.record(torch.cuda.current_stream(dev))
        self._same_grid = None   # decided lazily, see same_grid()

    def same_grid(self) -> bool:
        """Do both measures live on ONE grid (every reference call site: y_pos = x_pos.clone(), the fixed_x buffer)?  Then the
        p = 1 forward without cutoff has a merge-free form (SOT_FLAG_SAME_GRID).  Only such a call asks (make_problem), so the
        paper's p = 2 training step never pays for the answer.  Trivial for one tensor passed twice; otherwise ONE device
        comparison per plan, i.e. one host synchronisation the first time a p = 1 call uses the plan -- answered "no" (and not
        remembered) while a stream capture is in progress."""
        if self._same_grid is None:
            if self.xpos_sorted.numel() != self.ypos_sorted.numel():
                self._same_grid = False
            elif self._same_tensor:
                self._same_grid = True
            elif torch.cuda.is_current_stream_capturing():
                return False
            else:
                self._same_grid = bool(torch.equal(self.xpos_sorted, self.ypos_sorted))
        return self._same_grid

    def use_on_current_stream(self, device):
        """Order the current stream after the kernel that produced this plan (no-op on the producing stream)."""
        if stream_ptr(device) != self.stream:
            torch.cuda.current_stream(device).wait_event(self.ready)


_counter_pools = {}   # device index -> ([slots, COMPLETION_COUNTER_WORDS] zero-filled int32 tensor, {stream pointer: slot})
_COUNTER_SLOTS = 64


def completion_counters(device):
    """Device pointer of the completion-counter words (include/sot_hip.h: sot_w1d_loss) for torch's current stream on
    `device`, or None when the in-kernel batch mean cannot be used safely: the kernels leave the words zero, so one
    zero-filled buffer per (device, stream) serves every call that stream orders.  None (-> the separate mean kernel)
    while a stream is being captured before the device's pool exists (no allocation inside a capture) and when more than
    _COUNTER_SLOTS streams have asked."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    pool = _counter_pools.get(idx)
    if pool is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        with _lock:
            pool = _counter_pools.get(idx)
            if pool is None:
                buf = torch.zeros(_COUNTER_SLOTS, COMPLETION_COUNTER_WORDS, dtype=torch.int32, device=torch.device("cuda", idx))
                torch.cuda.current_stream(idx).synchronize()   # once per device: the zero fill is done before any stream uses it
                pool = _counter_pools[idx] = (buf, {})
    buf, slots = pool
    sp = torch._C._cuda_getCurrentRawStream(idx)
    slot = slots.get(sp)
    if slot is None:
        with _lock:
            slot = slots.get(sp)
            if slot is None:
                if len(slots) >= _COUNTER_SLOTS:
                    return None
                slot = slots[sp] = len(slots)
    return buf.data_ptr() + 4 * COMPLETION_COUNTER_WORDS * slot


# Batch mean inside the row kernel's last workgroup (one kernel) instead of the separate mean kernel.  OFF by default: on
# MI355X the hand-off (counter round trips, agent-scope acquire, re-read of the row losses by one workgroup) costs more than
# the kernel boundary it removes -- 8192 x 2048: 55.1 us per call as one kernel, 46.9 us as two (tools/ab_mean.py).
FUSED_MEAN = os.environ.get("SOT_FUSED_MEAN", "0") == "1"


def _want_tail(fused_mean):
    return FUSED_MEAN if fused_mean is None else bool(fused_mean)


def _needs_workspace(pr: SotProblem, flags, plan) -> bool:
    """Shared positions that still have to be planned; or per-row positions nobody has sorted (no row_perm_in / row_perm_out): the
    library's pre-sort kernel leaves the rows' permutations in the workspace (optional there -- without it the row kernel sorts in LDS)."""
    if not (flags & FLAG_REQUIRE_SORT) or plan is not None:
        return False
    return pr.xpos_row_stride == 0 or (not pr.row_perm_in and not pr.row_perm_out)


def workspace(pr: SotProblem, device) -> torch.Tensor:
    nbytes = load().sot_workspace_bytes(ctypes.byref(pr))
    return torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)


def forward_rows(x, y, xpos, ypos, p, flags, plan=None, out=None, perm_out=None, perm_in=None) -> torch.Tensor:
    """row_loss[B] = W_p^p per row (sot_w1d_forward).  perm_out / perm_in: row_permutations() buffers (per-row positions: the sort's
    permutations are left there / taken from there)."""
    lib = load()
    dev = x.device
    B = x.shape[0]
    row_loss = out if out is not None else torch.empty(B, dtype=torch.float32, device=dev)
    pr = make_problem(x, y, xpos, ypos, p, flags, plan, perm_out, perm_in)
    need_ws = _needs_workspace(pr, flags, plan)
    ws = workspace(pr, dev) if need_ws else None
    with _on_device(dev):
        rc = lib.sot_w1d_forward(ctypes.byref(pr), row_loss.data_ptr(), _ptr(ws), ws.numel() if ws is not None else 0,
                                 stream_ptr(dev))
    check(rc, p)
    return row_loss


def loss_fused(x, y, xpos, ypos, p, flags, plan=None, denom=None, hinge=None, want_sum=False, fused_mean=None, row_out=None,
               mean_out=None, sum_out=None):
    """Forward + batch mean behind one FFI call (sot_w1d_loss): the forward kernel and the mean kernel back to back, or --
    fused_mean=True (default: module switch FUSED_MEAN) -- ONE kernel whose last workgroup reduces the row losses (same bits).  Returns (mean 0-d fp32, row_loss [B], sum fp64 or None).
    row_out / mean_out / sum_out: optional preallocated outputs ([B] fp32, 1-element fp32, 1-element fp64)."""
    lib = load()
    dev = x.device
    B = x.shape[0]
    row_loss = row_out if row_out is not None else torch.empty(B, dtype=torch.float32, device=dev)
    mean = mean_out if mean_out is not None else torch.empty((), dtype=torch.float32, device=dev)
    total = sum_out if sum_out is not None else (torch.empty((), dtype=torch.float64, device=dev) if want_sum else None)
    pr = make_problem(x, y, xpos, ypos, p, flags, plan)
    need_ws = _needs_workspace(pr, flags, plan)
    ws = workspace(pr, dev) if need_ws else None
    with _on_device(dev):
        rc = lib.sot_w1d_loss(ctypes.byref(pr), row_loss.data_ptr(), float(B if denom is None else denom),
                              0 if hinge is None else 1, 0.0 if hinge is None else float(hinge), mean.data_ptr(),
                              _ptr(total), completion_counters(dev) if _want_tail(fused_mean) else None,
                              _ptr(ws), ws.numel() if ws is not None else 0, stream_ptr(dev))
    check(rc, p)
    return mean, row_loss, total


def prepared_loss_call(x, y, xpos, ypos, p, flags, plan, row_out, mean_out, sum_out=None, stream=None):
    """sot_w1d_loss on fixed buffers with its arguments marshalled ONCE: returns a zero-argument callable that enqueues the row kernel
    and the fixed-order reduction (row losses -> row_out, mean -> mean_out, fp64 sum -> sum_out) on `stream` (a raw hipStream_t; default:
    torch's current stream at the time of each call).  For loops that issue the same call on the same buffers again and again (bench.py's
    timed loop, a serving loop on preallocated buffers): per call only the C function is entered -- no struct fill, no allocation."""
    lib = load()
    dev = x.device
    require_hip(x, y, row_out, mean_out)
    pr = make_problem(x, y, xpos, ypos, p, flags, plan)
    if (int(flags) & FLAG_REQUIRE_SORT) and plan is None and xpos.ndim == 1:
        raise RuntimeError("prepared_loss_call needs a position plan for shared positions")
    keep = (x, y, xpos, ypos, plan, row_out, mean_out, sum_out, pr)   # the closure owns everything the pointers refer to
    fn = lib.sot_w1d_loss
    ref = ctypes.byref(pr)
    rows_p, mean_p, sum_p, denom = row_out.data_ptr(), mean_out.data_ptr(), _ptr(sum_out), float(x.shape[0])
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    raw = torch._C._cuda_getCurrentRawStream

    def call():
        rc = fn(ref, rows_p, denom, 0, 0.0, mean_p, sum_p, None, None, 0, raw(idx) if stream is None else stream)
        if rc != SOT_OK:
            check(rc, p)
        return keep[6]
    return call


def loss_and_grad(x, y, xpos, ypos, p, flags, plan=None, fused_mean=None):
    """Training form (sot_w1d_loss_and_grad): (mean 0-d fp32, row_loss [B], d mean / d y [B, m]) -- one pass over the rows
    where a compile-time backward kernel exists."""
    lib = load()
    dev = x.device
    B, m = x.shape[0], y.shape[1]
    row_loss = torch.empty(B, dtype=torch.float32, device=dev)
    mean = torch.empty((), dtype=torch.float32, device=dev)
    gy = torch.empty(B, m, dtype=torch.float32, device=dev)
    pr = make_problem(x, y, xpos, ypos, p, flags, plan)
    need_ws = _needs_workspace(pr, flags, plan)
    ws = workspace(pr, dev) if need_ws else None
    with _on_device(dev):
        rc = lib.sot_w1d_loss_and_grad(ctypes.byref(pr), row_loss.data_ptr(), float(B), mean.data_ptr(), None, 1.0 / B, gy.data_ptr(),
                                       completion_counters(dev) if _want_tail(fused_mean) else None,
                                       _ptr(ws), ws.numel() if ws is not None else 0, stream_ptr(dev))
    check(rc, p)
    return mean, row_loss, gy


def profile_next_launch(slot: int):
    """Arm kernel-attached timing (sot_profile_next_launch) for this thread's next launch of a compile-time-length kernel."""
    check(load().sot_profile_next_launch(int(slot)))


def profile_elapsed_ms(slot: int) -> float:
    """Duration of the launch that slot timed (waits for it)."""
    ms = ctypes.c_float()
    check(load().sot_profile_elapsed_ms(int(slot), ctypes.byref(ms)))
    return float(ms.value)


def scale_inplace(data: torch.Tensor, scalar: torch.Tensor) -> torch.Tensor:
    """data *= scalar (0-d / 1-element fp32 device tensor); a no-op kernel when the scalar is exactly 1."""
    require_hip(data, scalar)
    if not data.is_contiguous() or data.dtype != torch.float32 or scalar.dtype != torch.float32 or scalar.numel() != 1:
        raise RuntimeError("scale_inplace expects a contiguous fp32 tensor and a one-element fp32 scalar")
    with _on_device(data.device):
        check(load().sot_scale_inplace(data.data_ptr(), data.numel(), scalar.data_ptr(), stream_ptr(data.device)))
    return data


def reduce_mean(row_loss, denom=None, hinge=None, want_sum=False, sum_out=None):
    """Fixed-order fp64 batch reduction.  sum_out: optional preallocated fp64 tensor (1 element) for the partial sum."""
    lib = load()
    dev = row_loss.device
    B = row_loss.numel()
    mean = torch.empty((), dtype=torch.float32, device=dev)
    total = sum_out if sum_out is not None else (torch.empty((), dtype=torch.float64, device=dev) if want_sum else None)
    want_sum = want_sum or sum_out is not None
    with _on_device(dev):
        check(lib.sot_w1d_reduce_mean(row_loss.data_ptr(), B, float(B if denom is None else denom),
                                      0 if hinge is None else 1, 0.0 if hinge is None else float(hinge),
                                      mean.data_ptr(), _ptr(total), stream_ptr(dev)))
    return (mean, total) if want_sum else mean


def backward_rows(x, y, xpos, ypos, p, flags, grad_row, need_gx=True, need_gy=True, plan=None, grad_scale=1.0, perm_in=None):
    """grad_row: [B] tensor, or a 0-d / 1-element tensor that is broadcast to every row (stride 0).  perm_in: the forward's row_permutations()."""
    lib = load()
    dev = x.device
    B, n = x.shape
    m = y.shape[1]
    gx = torch.empty(B, n, dtype=torch.float32, device=dev) if need_gx else None
    gy = torch.empty(B, m, dtype=torch.float32, device=dev) if need_gy else None
    pr = make_problem(x, y, xpos, ypos, p, flags, plan, perm_in=perm_in)
    need_ws = _needs_workspace(pr, flags, plan)
    ws = workspace(pr, dev) if need_ws else None
    g = grad_row.contiguous()
    stride = 0 if g.numel() == 1 else 1
    if stride == 1 and g.numel() != B:
        raise RuntimeError(f"grad_row has {g.numel()} elements for {B} rows")
    with _on_device(dev):
        rc = lib.sot_w1d_backward(ctypes.byref(pr), g.data_ptr(), stride, float(grad_scale), _ptr(gx), _ptr(gy), _ptr(ws),
                                  ws.numel() if ws is not None else 0, stream_ptr(dev))
    check(rc, p)
    return gx, gy


def position_grads(x, y, xpos, ypos, p, flags, grad_row, need_x=True, need_y=True, plan=None, grad_scale=1.0, perm_in=None):
    """Gradients w.r.t. the support positions (sot_w1d_position_grad): per-row [B, n] / [B, m] for per-row positions; for a shared
    position row the batch sum [n] / [m] (sot_column_sum), which is what autograd's expand backward returns (losses.py:167-170)."""
    lib = load()
    dev = x.device
    B, n = x.shape
    m = y.shape[1]
    gxp = torch.empty(B, n, dtype=torch.float32, device=dev) if need_x else None
    gyp = torch.empty(B, m, dtype=torch.float32, device=dev) if need_y else None
    pr = make_problem(x, y, xpos, ypos, p, flags, plan, perm_in=perm_in)
    need_ws = _needs_workspace(pr, flags, plan)
    ws = workspace(pr, dev) if need_ws else None
    g = grad_row.contiguous()
    stride = 0 if g.numel() == 1 else 1
    if stride == 1 and g.numel() != B:
        raise RuntimeError(f"grad_row has {g.numel()} elements for {B} rows")
    out = []
    with _on_device(dev):
        rc = lib.sot_w1d_position_grad(ctypes.byref(pr), g.data_ptr(), stride, float(grad_scale), _ptr(gxp), _ptr(gyp), _ptr(ws),
                                       ws.numel() if ws is not None else 0, stream_ptr(dev))
        check(rc, p)
        for rows, pos, width in ((gxp, xpos, n), (gyp, ypos, m)):
            if rows is None or pos.ndim == 2:
                out.append(rows)
                continue
            tot = torch.empty(width, dtype=torch.float32, device=dev)
            check(lib.sot_column_sum(rows.data_ptr(), B, width, width, tot.data_ptr(), stream_ptr(dev)), p)
            out.append(tot)
    return out


def quantiles(x, y, xpos, ypos, p, flags, plan=None):
    lib = load()
    dev = x.device
    B, n = x.shape
    m = y.shape[1]
    K = n + m
    uq = torch.empty(B, K, dtype=torch.float32, device=dev)
    vq = torch.empty(B, K, dtype=torch.float32, device=dev)
    Q = torch.empty(B, K, dtype=torch.float32, device=dev)
    U = torch.empty(B, n, dtype=torch.float32, device=dev)
    V = torch.empty(B, m, dtype=torch.float32, device=dev)
    pr = make_problem(x, y, xpos, ypos, p, flags, plan)
    need_ws = _needs_workspace(pr, flags, plan)
    ws = workspace(pr, dev) if need_ws else None
    with _on_device(dev):
        rc = lib.sot_w1d_quantiles(ctypes.byref(pr), uq.data_ptr(), vq.data_ptr(), Q.data_ptr(), U.data_ptr(),
                                   V.data_ptr(), _ptr(ws), ws.numel() if ws is not None else 0, stream_ptr(dev))
    check(rc, p)
    return uq, vq, Q, U, V


def forward_rows_csr(xw, xp, xoff, yw, yp, yoff, max_n, max_m, p, flags):
    """row_loss[B] for ragged supports in CSR form (sot_w1d_forward_csr); offsets are int64 [B+1] on the device."""
    require_hip(xw, xp, yw, yp)
    lib = load()
    dev = xw.device
    for t in (xoff, yoff):
        if not t.is_cuda or t.dtype != torch.int64:
            raise TypeError("CSR offsets must be int64 tensors on the GPU")
    B = xoff.numel() - 1
    if yoff.numel() - 1 != B:
        raise RuntimeError("x and y offsets describe different numbers of rows")
    if xw.numel() != xp.numel() or yw.numel() != yp.numel():
        raise RuntimeError("weights and positions must have the same number of entries")
    xw, xp, yw, yp, xoff, yoff = (t.contiguous() for t in (xw, xp, yw, yp, xoff, yoff))
    row_loss = torch.empty(B, dtype=torch.float32, device=dev)
    with _on_device(dev):
        rc = lib.sot_w1d_forward_csr(xw.data_ptr(), xp.data_ptr(), xoff.data_ptr(), xw.numel(), yw.data_ptr(), yp.data_ptr(),
                                     yoff.data_ptr(), yw.numel(), B, int(max_n), int(max_m), float(p), int(flags),
                                     row_loss.data_ptr(), stream_ptr(dev))
    check(rc, p)
    return row_loss


def segmented_sort(keys: torch.Tensor):
    """(values, int64 indices) of a per-row ascending stable sort (sot_segmented_sort)."""
    require_hip(keys)
    lib = load()
    keys = rows_view(keys)
    B, n = keys.shape
    vals = torch.empty(B, n, dtype=torch.float32, device=keys.device)
    idx = torch.empty(B, n, dtype=torch.int64, device=keys.device)
    with _on_device(keys.device):
        check(lib.sot_segmented_sort(keys.data_ptr(), B, n, keys.stride(0) if B > 1 else n, vals.data_ptr(),
                                     idx.data_ptr(), stream_ptr(keys.device)))
    return vals, idx


def _aligned8(t: torch.Tensor) -> torch.Tensor:
    """the STFT kernels read the window two taps at a time: an odd-offset view is copied to a fresh allocation"""
    return t if t.data_ptr() % 8 == 0 else t.clone()


def stft_mag_forward(audio: torch.Tensor, window: torch.Tensor, n_fft: int, hop: int, want_spec: bool = False):
    """[batch, samples] fp32 on the GPU -> [batch, frames, n_fft/2+1] magnitudes (sot_stft_mag_forward).  want_spec: returns
    (magnitudes, spectrum [batch, frames, n_fft/2+1, 2]) -- the complex spectrum for stft_mag_backward(spec=...) (sot_stft_mag_forward_spec)."""
    require_hip(audio, window)
    lib = load()
    if audio.ndim != 2 or window.numel() != n_fft:
        raise RuntimeError("stft_mag_forward expects audio [batch, samples] and a window of n_fft samples")
    if audio.stride(1) != 1:
        audio = audio.contiguous()
    window = _aligned8(window.contiguous())
    batch, samples = audio.shape
    frames = int(lib.sot_stft_frames(samples, hop))
    mag = torch.empty(batch, frames, n_fft // 2 + 1, dtype=torch.float32, device=audio.device)
    spec = torch.empty(batch, frames, n_fft // 2 + 1, 2, dtype=torch.float32, device=audio.device) if want_spec else None
    with _on_device(audio.device):
        check(lib.sot_stft_mag_forward_spec(audio.data_ptr(), batch, samples, audio.stride(0) if batch > 1 else samples, window.data_ptr(),
                                            int(n_fft), int(hop), mag.data_ptr(), _ptr(spec), stream_ptr(audio.device)))
    return (mag, spec) if want_spec else mag


def stft_mag_forward_pair(audio_a: torch.Tensor, audio_b: torch.Tensor, window: torch.Tensor, n_fft: int, hop: int, want_spec_b: bool = False):
    """Two [batch, samples] signals -> their [batch, frames, n_fft/2+1] magnitudes from ONE launch (sot_stft_mag_forward_pair);
    the two results are the halves of one allocation.  want_spec_b: a third result, the complex spectrum of audio_b
    [batch, frames, n_fft/2+1, 2] for stft_mag_backward(spec=...)."""
    require_hip(audio_a, audio_b, window)
    lib = load()
    if audio_a.ndim != 2 or audio_a.shape != audio_b.shape or window.numel() != n_fft:
        raise RuntimeError("stft_mag_forward_pair expects two audio tensors [batch, samples] of one shape and a window of n_fft samples")
    if audio_a.stride(1) != 1:
        audio_a = audio_a.contiguous()
    if audio_b.stride(1) != 1:
        audio_b = audio_b.contiguous()
    window = _aligned8(window.contiguous())
    batch, samples = audio_a.shape
    frames = int(lib.sot_stft_frames(samples, hop))
    mag = torch.empty(2 * batch, frames, n_fft // 2 + 1, dtype=torch.float32, device=audio_a.device)
    spec = torch.empty(batch, frames, n_fft // 2 + 1, 2, dtype=torch.float32, device=audio_a.device) if want_spec_b else None
    with _on_device(audio_a.device):
        check(lib.sot_stft_mag_forward_pair_spec(audio_a.data_ptr(), audio_a.stride(0) if batch > 1 else samples, audio_b.data_ptr(),
                                                 audio_b.stride(0) if batch > 1 else samples, batch, samples, window.data_ptr(), int(n_fft),
                                                 int(hop), mag.data_ptr(), _ptr(spec), stream_ptr(audio_a.device)))
    return (mag[:batch], mag[batch:], spec) if want_spec_b else (mag[:batch], mag[batch:])


def stft_mag_backward(audio: torch.Tensor, window: torch.Tensor, n_fft: int, hop: int, grad_mag: torch.Tensor,
                      grad_scale: torch.Tensor = None, accumulate_into: torch.Tensor = None, spec: torch.Tensor = None) -> torch.Tensor:
    """dL/d(audio) [batch, samples] from dL/d(mag) [batch, frames, n_fft/2+1] (sot_stft_mag_backward); grad_scale: optional
    one-element fp32 device tensor multiplying grad_mag; accumulate_into: an existing contiguous [batch, samples] gradient
    that the result is added to (and that is returned); spec: the complex spectrum the forward stored (want_spec) -- the kernel
    then skips its own forward transform of every frame (same result bit for bit) and `audio` only supplies the shape."""
    require_hip(audio, window, grad_mag, grad_scale, spec)
    if grad_scale is not None and grad_scale.numel() != 1:
        raise RuntimeError("stft_mag_backward: grad_scale must hold one element")
    lib = load()
    if audio.stride(1) != 1:
        audio = audio.contiguous()
    window = _aligned8(window.contiguous())
    grad_mag = grad_mag.contiguous()
    batch, samples = audio.shape
    if accumulate_into is not None:
        require_hip(accumulate_into)
        if accumulate_into.shape != (batch, samples) or not accumulate_into.is_contiguous():
            raise RuntimeError("stft_mag_backward: accumulate_into must be a contiguous [batch, samples] tensor")
        grad_audio = accumulate_into
    else:
        grad_audio = torch.empty(batch, samples, dtype=torch.float32, device=audio.device)
    ws = torch.empty(max(1, int(lib.sot_stft_backward_workspace_bytes(batch, samples, int(n_fft), int(hop)))), dtype=torch.uint8,
                     device=audio.device)
    with _on_device(audio.device):
        if spec is not None and (not spec.is_contiguous() or spec.shape != (batch, grad_mag.shape[1], n_fft // 2 + 1, 2)):
            raise RuntimeError("stft_mag_backward: spec must be the contiguous [batch, frames, n_fft/2+1, 2] spectrum of the forward pass")
        check(lib.sot_stft_mag_backward_spec(audio.data_ptr(), _ptr(spec), batch, samples, audio.stride(0) if batch > 1 else samples,
                                             window.data_ptr(), int(n_fft), int(hop), grad_mag.data_ptr(), _ptr(grad_scale),
                                             grad_audio.data_ptr(), 0 if accumulate_into is None else 1, ws.data_ptr(), ws.numel(),
                                             stream_ptr(audio.device)))
    return grad_audio


def spec_distance_forward(target: torch.Tensor, value: torch.Tensor, mag_weight: float, logmag_weight: float, eps: float = 1e-5,
                          l2: bool = False, accumulate_into: torch.Tensor = None, per_row: bool = False) -> torch.Tensor:
    """0-d fp32: mag_weight * mean D(t - v) + logmag_weight * mean D(slog t - slog v) (sot_spec_distance_forward);
    accumulate_into: an existing 0-d fp32 device tensor the distance is added to (and that is returned).
    per_row: one value per leading index instead ([rows] fp32; sot_spec_distance_rows_forward)."""
    require_hip(target, value)
    lib = load()
    target, value = target.contiguous(), value.contiguous()
    if target.shape != value.shape or target.numel() == 0:
        raise RuntimeError("spec_distance_forward expects two non-empty tensors of the same shape")
    if per_row:
        rows = target.shape[0]
        out = accumulate_into if accumulate_into is not None else torch.empty(rows, dtype=torch.float32, device=target.device)
        if out.numel() != rows or not out.is_contiguous():
            raise RuntimeError("spec_distance_forward: accumulate_into must hold one contiguous element per row")
        with _on_device(target.device):
            check(lib.sot_spec_distance_rows_forward(target.data_ptr(), value.data_ptr(), rows, target.numel() // rows, float(mag_weight),
                                                     float(logmag_weight), float(eps), int(bool(l2)), out.data_ptr(),
                                                     0 if accumulate_into is None else 1, stream_ptr(target.device)))
        return out
    if accumulate_into is not None:
        require_hip(accumulate_into)
        if accumulate_into.numel() != 1:
            raise RuntimeError("spec_distance_forward: accumulate_into must hold one element")
    out = accumulate_into if accumulate_into is not None else torch.empty((), dtype=torch.float32, device=target.device)
    ws = torch.empty(int(lib.sot_spec_distance_workspace_bytes()), dtype=torch.uint8, device=target.device)
    with _on_device(target.device):
        check(lib.sot_spec_distance_forward(target.data_ptr(), value.data_ptr(), target.numel(), float(mag_weight), float(logmag_weight),
                                            float(eps), int(bool(l2)), out.data_ptr(), 0 if accumulate_into is None else 1,
                                            ws.data_ptr(), ws.numel(), stream_ptr(target.device)))
    return out


def spec_distance_backward(target, value, mag_weight, logmag_weight, upstream, grad_scale=1.0, eps=1e-5, l2=False, need_target=False,
                           need_value=True, per_row=False):
    require_hip(target, value, upstream)
    lib = load()
    target, value = target.contiguous(), value.contiguous()
    gt = torch.empty_like(target) if need_target else None
    gv = torch.empty_like(value) if need_value else None
    if per_row:
        rows = target.shape[0]
        up = upstream.contiguous()
        if up.numel() != rows:
            raise RuntimeError("spec_distance_backward: one upstream gradient per row")
        with _on_device(target.device):
            check(lib.sot_spec_distance_rows_backward(target.data_ptr(), value.data_ptr(), rows, target.numel() // rows, float(mag_weight),
                                                      float(logmag_weight), float(eps), int(bool(l2)), up.data_ptr(), float(grad_scale), _ptr(gt),
                                                      _ptr(gv), stream_ptr(target.device)))
        return gt, gv
    with _on_device(target.device):
        check(lib.sot_spec_distance_backward(target.data_ptr(), value.data_ptr(), target.numel(), float(mag_weight), float(logmag_weight),
                                             float(eps), int(bool(l2)), upstream.contiguous().data_ptr(), float(grad_scale), _ptr(gt), _ptr(gv),
                                             stream_ptr(target.device)))
    return gt, gv


MSS_FUSED_SIZES = (64, 128, 256, 512, 1024, 2048)   # n_fft the two-launch form takes (hop = n_fft / 4); include/sot_hip.h: sot_mss_loss_and_grad


def mss_loss_and_grad(target: torch.Tensor, value: torch.Tensor, fft_sizes, windows, mag_weight: float, logmag_weight: float, eps: float = 1e-5,
                      l2: bool = False, per_clip: bool = False, want_grad: bool = True, post_scale: float = 1.0):
    """MSSLoss (losses.py:365-425) of [batch, samples] float32 audio and d loss / d value in two launches (sot_mss_loss_and_grad):
    -> (loss: 0-d, or [batch] with per_clip; grad: [batch, samples] or None).  windows: one n_fft-tap device tensor per scale.
    post_scale: one more float32 product on the finished loss and gradient (a mix weight, losses.py:360)."""
    require_hip(target, value)
    lib = load()
    if target.ndim != 2 or target.shape != value.shape:
        raise RuntimeError("mss_loss_and_grad expects two [batch, samples] tensors of one shape")
    # rows the library can address: unit inner stride and a row stride of at least one row (an expanded target -- stride 0 -- or an
    # overlapping as_strided view is copied, as the reference's own ops would do); a single row's stride is meaningless to torch
    target, value = rows_view(target), rows_view(value)
    batch, samples = value.shape
    t_stride = target.stride(0) if batch > 1 else samples
    v_stride = value.stride(0) if batch > 1 else samples
    n = len(fft_sizes)
    sizes = (ctypes.c_int * n)(*[int(s) for s in fft_sizes])
    wins = [_aligned8(w) for w in windows]
    wptr = (ctypes.c_void_p * n)(*[w.data_ptr() for w in wins])
    nbytes = int(lib.sot_mss_workspace_bytes(batch, samples, sizes, n))
    if nbytes == 0 and batch > 0:
        raise SotError(f"sot_mss_loss_and_grad does not take fft_sizes={tuple(fft_sizes)} (powers of two in [64, 2048], at most 8)")
    ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=value.device)
    loss = torch.empty(batch if per_clip else (), dtype=torch.float32, device=value.device)
    if batch == 0:   # nothing is launched: the mean over an empty batch is NaN (what the reference's torch.mean returns); per clip: no entries
        loss.fill_(float("nan"))
    grad = torch.empty((batch, samples), dtype=torch.float32, device=value.device) if want_grad else None
    with _on_device(value.device):
        check(lib.sot_mss_loss_and_grad(target.data_ptr(), t_stride, value.data_ptr(), v_stride, batch, samples,
                                        ctypes.cast(sizes, ctypes.c_void_p), ctypes.cast(wptr, ctypes.c_void_p), n, float(mag_weight), float(logmag_weight),
                                        float(eps), int(bool(l2)), int(bool(per_clip)), float(post_scale), loss.data_ptr(), _ptr(grad), ws.data_ptr(), ws.numel(),
                                        stream_ptr(value.device)))
    return loss, grad


def oscillator_bank_forward(freq: torch.Tensor, amp: torch.Tensor, sample_rate: float, return_workspace: bool = False):
    """[batch, samples, sinusoids] envelopes -> [batch, samples] audio (sot_oscillator_bank_forward).  return_workspace: also
    return the scratch buffer (it holds the segment start phases, which oscillator_bank_backward can reuse)."""
    require_hip(freq, amp)
    lib = load()
    if freq.ndim != 3 or freq.shape != amp.shape:
        raise RuntimeError("oscillator_bank_forward expects two [batch, samples, sinusoids] tensors")
    freq, amp = freq.contiguous(), amp.contiguous()
    batch, samples, k = freq.shape
    audio = torch.empty(batch, samples, dtype=torch.float32, device=freq.device)
    ws = _oscillator_workspace(lib, batch, samples, k, freq.device)
    with _on_device(freq.device):
        check(lib.sot_oscillator_bank_forward(freq.data_ptr(), amp.data_ptr(), batch, samples, k, float(sample_rate), audio.data_ptr(),
                                              ws.data_ptr(), ws.numel(), stream_ptr(freq.device)))
    return (audio, ws) if return_workspace else audio


def _oscillator_workspace(lib, batch, samples, k, device):
    need = int(lib.sot_oscillator_bank_workspace_bytes(batch, samples, k))
    if need == 0 and batch > 0:
        raise RuntimeError(f"oscillator bank: {samples} samples x {k} sinusoids is outside what the HIP kernels take")
    return torch.empty(max(need, 8), dtype=torch.uint8, device=device)


def oscillator_bank_backward(freq, amp, sample_rate, grad_audio, need_freq=True, need_amp=True, forward_workspace=None):
    require_hip(freq, amp, grad_audio)
    lib = load()
    freq, amp, grad_audio = freq.contiguous(), amp.contiguous(), grad_audio.contiguous()
    batch, samples, k = freq.shape
    gf = torch.empty_like(freq) if need_freq else None
    ga = torch.empty_like(amp) if need_amp else None
    ws = forward_workspace if forward_workspace is not None else _oscillator_workspace(lib, batch, samples, k, freq.device)
    with _on_device(freq.device):
        check(lib.sot_oscillator_bank_backward(freq.data_ptr(), amp.data_ptr(), batch, samples, k, float(sample_rate), grad_audio.data_ptr(),
                                               _ptr(gf), _ptr(ga), ws.data_ptr(), ws.numel(), 0 if forward_workspace is None else 1,
                                               stream_ptr(freq.device)))
    return gf, ga


def synth_envelopes_forward(amp_frames, freq_frames, window, n_samples: int, sample_rate: float, harmonic: bool):
    """Frame-rate controls [batch, frames, K] (+ [batch, frames, 1] f0 when harmonic) -> sample-rate (amplitude, frequency)
    envelopes [batch, n_samples, K] (sot_synth_envelopes_forward)."""
    require_hip(amp_frames, freq_frames, window)
    lib = load()
    amp_frames, freq_frames, window = amp_frames.contiguous(), freq_frames.contiguous(), window.contiguous()
    batch, frames, k = amp_frames.shape
    if freq_frames.shape != ((batch, frames, 1) if harmonic else (batch, frames, k)) or window.numel() * frames != 2 * n_samples:
        raise RuntimeError("synth_envelopes_forward: control shapes / window length do not fit")
    amp_env = torch.empty(batch, n_samples, k, dtype=torch.float32, device=amp_frames.device)
    freq_env = torch.empty_like(amp_env)
    with _on_device(amp_frames.device):
        check(lib.sot_synth_envelopes_forward(amp_frames.data_ptr(), freq_frames.data_ptr(), window.data_ptr(), batch, frames, k, int(bool(harmonic)),
                                              int(n_samples), float(sample_rate), amp_env.data_ptr(), freq_env.data_ptr(),
                                              stream_ptr(amp_frames.device)))
    return amp_env, freq_env


def synth_envelopes_backward(amp_frames, freq_frames, window, n_samples, sample_rate, harmonic, grad_amp_env, grad_freq_env, need_amp=True,
                             need_freq=True):
    require_hip(*[t for t in (amp_frames, freq_frames, window, grad_amp_env, grad_freq_env) if t is not None])
    lib = load()
    amp_frames, freq_frames, window = amp_frames.contiguous(), freq_frames.contiguous(), window.contiguous()
    batch, frames, k = amp_frames.shape
    ga = torch.empty_like(amp_frames) if need_amp else None
    gf = torch.empty_like(freq_frames) if need_freq else None
    gae = grad_amp_env.contiguous() if grad_amp_env is not None else None
    gfe = grad_freq_env.contiguous() if grad_freq_env is not None else None
    with _on_device(amp_frames.device):
        check(lib.sot_synth_envelopes_backward(amp_frames.data_ptr(), freq_frames.data_ptr(), window.data_ptr(), batch, frames, k, int(bool(harmonic)),
                                               int(n_samples), float(sample_rate), _ptr(gae), _ptr(gfe), _ptr(ga), _ptr(gf),
                                               stream_ptr(amp_frames.device)))
    return ga, gf


def synth_forward(amp_frames, freq_frames, window, n_samples: int, sample_rate: float, harmonic: bool, for_backward: bool = False):
    """Frame-rate controls -> audio [batch, n_samples] (sot_synth_forward).  Returns (audio, workspace); with for_backward the
    workspace is sized for sot_synth_backward, which then reuses the segment start phases in it."""
    require_hip(amp_frames, freq_frames, window)
    lib = load()
    amp_frames, freq_frames, window = amp_frames.contiguous(), freq_frames.contiguous(), window.contiguous()
    batch, frames, k = amp_frames.shape
    if freq_frames.shape != ((batch, frames, 1) if harmonic else (batch, frames, k)) or window.numel() * frames != 2 * n_samples:
        raise RuntimeError("synth_forward: control shapes / window length do not fit")
    dev = amp_frames.device
    audio = torch.empty(batch, n_samples, dtype=torch.float32, device=dev)
    ws = torch.empty(max(8, lib.sot_synth_workspace_bytes(batch, frames, n_samples, k, int(for_backward))), dtype=torch.uint8, device=dev)
    with _on_device(dev):
        check(lib.sot_synth_forward(amp_frames.data_ptr(), freq_frames.data_ptr(), window.data_ptr(), batch, frames, k, int(bool(harmonic)),
                                    int(n_samples), float(sample_rate), audio.data_ptr(), ws.data_ptr(), ws.numel(), stream_ptr(dev)))
    return audio, ws


def synth_tap_tables(window, frames: int, n_samples: int):
    """The weight tables of sot_synth_backward for (window, frames, n_samples), to be kept by the caller (sot_synth_tap_tables)."""
    require_hip(window)
    lib = load()
    window = window.contiguous()
    tables = torch.empty(max(8, lib.sot_synth_tap_table_bytes(int(frames), int(n_samples))), dtype=torch.uint8, device=window.device)
    with _on_device(window.device):
        check(lib.sot_synth_tap_tables(window.data_ptr(), int(frames), int(n_samples), tables.data_ptr(), stream_ptr(window.device)))
    return tables


def synth_backward(amp_frames, freq_frames, window, n_samples, sample_rate, harmonic, grad_audio, need_amp=True, need_freq=True,
                   forward_workspace=None, tap_tables=None):
    require_hip(amp_frames, freq_frames, window, grad_audio)
    lib = load()
    amp_frames, freq_frames, window, grad_audio = amp_frames.contiguous(), freq_frames.contiguous(), window.contiguous(), grad_audio.contiguous()
    batch, frames, k = amp_frames.shape
    dev = amp_frames.device
    need = lib.sot_synth_workspace_bytes(batch, frames, n_samples, k, 1)
    reuse = forward_workspace is not None and forward_workspace.numel() >= need
    ws = forward_workspace if reuse else torch.empty(max(8, need), dtype=torch.uint8, device=dev)
    ga = torch.empty_like(amp_frames) if need_amp else None
    gf = torch.empty_like(freq_frames) if need_freq else None
    with _on_device(dev):
        check(lib.sot_synth_backward(amp_frames.data_ptr(), freq_frames.data_ptr(), window.data_ptr(), batch, frames, k, int(bool(harmonic)),
                                     int(n_samples), float(sample_rate), grad_audio.data_ptr(), _ptr(ga), _ptr(gf), _ptr(tap_tables), ws.data_ptr(),
                                     ws.numel(), int(reuse), stream_ptr(dev)))
    return ga, gf
