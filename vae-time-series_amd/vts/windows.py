"""Training windows cut on the device from a series that stays resident in HBM.

Counterpart of the host-side preparation in the reference's drivers:
``create_sliding_windows`` + ``StandardScaler`` of vae-models/graph/train_ld_2011.py:11-17,47-59 and
the block-maximum targets of vae-models/graph/training_block_max_hurdat.py:81-86,100-106.  The
series is uploaded once; a batch is then a gather by window start (one HBM write of the windows)
instead of a host-to-device copy of B*T*C floats.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib as L

TARGETS = {"next": L.TARGET_NEXT, "block_max": L.TARGET_BLOCK_MAX}


def _dptr(t: Optional[torch.Tensor], dtype, what: str):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise RuntimeError(f"vts windows: {what} must be a contiguous {dtype} tensor on a ROCm device")
    return C.c_void_p(t.data_ptr())


class WindowSource:
    """A raw series ``[Nt, C]`` on the device plus the scalers fitted on its training windows.

    ``window_length`` / ``predict_steps`` as in the reference's ``create_sliding_windows``; window i
    covers series rows i .. i+window_length-1 and its target is column 0 of the following
    ``predict_steps`` rows (``target="next"``) or their maximum (``target="block_max"``).
    """

    def __init__(self, series: torch.Tensor, window_length: int = 12, predict_steps: int = 1, target: str = "next"):
        if series.dim() == 1:
            series = series[:, None]
        if series.dim() != 2:
            raise ValueError("series must be [Nt] or [Nt, C]")
        if target not in TARGETS:
            raise ValueError(f"target must be one of {sorted(TARGETS)}")
        if not series.is_cuda:
            raise RuntimeError("vts windows run on a ROCm device only: move the series to the GPU once")
        self.series = series.to(torch.float32).contiguous()
        self.T, self.P, self.target = int(window_length), int(predict_steps), target
        self.x_stats: Optional[torch.Tensor] = None    # [2, C] float64: means, scales
        self.y_stats: Optional[torch.Tensor] = None    # [2] float64: mean, scale

    def __len__(self) -> int:
        """Number of complete windows, as len(data) - window_length - predict_steps + 1 in the reference."""
        return max(0, self.series.shape[0] - self.T - self.P + 1)

    def _starts(self, starts) -> torch.Tensor:
        if starts is None:
            return torch.arange(len(self), device=self.series.device, dtype=torch.int64)
        return torch.as_tensor(starts, dtype=torch.int64).to(self.series.device).contiguous()

    def set_scalers(self, x_mean=None, x_scale=None, y_mean=None, y_scale=None) -> None:
        """Use statistics fitted elsewhere (e.g. a StandardScaler's ``mean_`` / ``scale_``)."""
        dev = self.series.device
        if x_mean is not None:
            self.x_stats = torch.stack([torch.as_tensor(x_mean, dtype=torch.float64).reshape(-1),
                                        torch.as_tensor(x_scale, dtype=torch.float64).reshape(-1)]).to(dev).contiguous()
            if self.x_stats.shape[1] != self.series.shape[1]:
                raise ValueError("x_mean / x_scale need one entry per channel")
        if y_mean is not None:
            self.y_stats = torch.tensor([float(y_mean), float(y_scale)], dtype=torch.float64, device=dev)

    def fit(self, starts=None, fit_y: bool = True) -> "WindowSource":
        """StandardScaler.fit over the rows of the given windows (and over their next-step targets)."""
        lib = L.load()
        st = self._starts(starts)
        Nt, Cn = self.series.shape
        ws_b = lib.vts_windows_stats_workspace_bytes(Nt)
        ws = L.workspace(ws_b, self.series.device)
        xs = torch.empty(2, Cn, dtype=torch.float64, device=self.series.device)
        want_y = fit_y and self.target == "next"
        ys = torch.empty(2, dtype=torch.float64, device=self.series.device) if want_y else None
        L.check(lib.vts_windows_stats(L.ptr(self.series), Nt, Cn, _dptr(st, torch.int64, "starts"), st.numel(), self.T,
                                      self.P, TARGETS[self.target], _dptr(xs, torch.float64, "x_stats"),
                                      _dptr(ys, torch.float64, "y_stats"), L.wptr(ws), ws_b, L.stream()),
                "vts_windows_stats")
        self.x_stats = xs
        if want_y:
            self.y_stats = ys
        return self

    def batch(self, starts=None, want_x: bool = True, want_y: bool = True) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
        """(X [B, T, C], y [B, P] or [B, 1]) for the windows starting at ``starts`` (all windows if None),
        standardised with whatever scalers are set."""
        lib = L.load()
        st = self._starts(starts)
        B = st.numel()
        Nt, Cn = self.series.shape
        dev = self.series.device
        X = torch.empty(B, self.T, Cn, device=dev) if want_x else None
        y = None
        if want_y:
            y = torch.empty(B, self.P if self.target == "next" else 1, device=dev)
        xm = self.x_stats[0] if self.x_stats is not None else None
        xs = self.x_stats[1] if self.x_stats is not None else None
        L.check(lib.vts_windows_build(L.ptr(self.series), Nt, Cn, _dptr(st, torch.int64, "starts"), B, self.T, self.P,
                                      TARGETS[self.target], _dptr(xm, torch.float64, "x_mean"),
                                      _dptr(xs, torch.float64, "x_scale"), _dptr(self.y_stats, torch.float64, "y_stats"),
                                      L.ptr(X), L.ptr(y), L.stream()),
                "vts_windows_build")
        return X, y


def create_sliding_windows(data: torch.Tensor, window_length: int = 12, predict_steps: int = 1):
    """Device counterpart of the reference's function of the same name (train_ld_2011.py:11-17): every
    window of ``data`` [Nt, C] and its targets; y is flattened as there."""
    X, y = WindowSource(data, window_length, predict_steps).batch()
    return X, y.reshape(-1)
