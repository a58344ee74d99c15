"""CPU oracle (test infrastructure, NOT a product path) for the window builder: a numpy restatement
of the reference's host-side data preparation.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Pinned against the reference: tests/golden/make_goldens_windows.py compiles the reference's own
function definitions (create_sliding_windows of vae-models/graph/train_ld_2011.py:11-17, create_X_data /
create_y_data of vae-models/graph/training_block_max_hurdat.py:81-86,100-106), runs them and
sklearn's StandardScaler on slices of the reference's datasets exactly as train_ld_2011.py:47-59
does, and stores inputs and outputs (tests/golden/windows_*.npz); tests/test_oracle_windows.py replays
them through the functions below, bit for bit.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


def sliding_windows(data: np.ndarray, window_length: int, predict_steps: int,
                    starts: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
    """train_ld_2011.py:11-17: X[i] = data[i : i+T], y[i] = data[i+T : i+T+P, 0]; y is [N, P] here
    (the reference flattens it).  ``starts`` picks windows (default: every complete one, in order)."""
    data = np.asarray(data, dtype=np.float32)
    T, P = window_length, predict_steps
    if starts is None:
        starts = np.arange(len(data) - T - P + 1)
    rows = np.asarray(starts, dtype=np.int64)[:, None] + np.arange(T)[None, :]
    X = data[rows]                                               # [N, T, C]
    yrows = np.asarray(starts, dtype=np.int64)[:, None] + T + np.arange(P)[None, :]
    y = data[yrows, 0]                                           # [N, P]
    return X.astype(np.float32), y.astype(np.float32)


def block_max_targets(data: np.ndarray, window_length: int, predict_steps: int,
                      starts: Optional[np.ndarray] = None) -> np.ndarray:
    """training_block_max_hurdat.py:100-106 (create_y_data): the maximum of the steps after the first
    ``window_length`` of each (window_length + predict_steps)-long block, column 0; [N, 1]."""
    _, y = sliding_windows(data, window_length, predict_steps, starts)
    return np.max(y, axis=1, keepdims=True).astype(np.float32)


def scaler_fit(rows: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """StandardScaler.fit on float rows [n, F] (train_ld_2011.py:54,56): float64 mean and population
    standard deviation, scale 1 for a constant column.  Plain two-pass float64 sums - sklearn's own
    incremental sums differ from this in the last bits (tests compare at 1e-12 relative)."""
    r = np.asarray(rows, dtype=np.float64)
    mean = r.mean(axis=0)
    var = ((r - mean) ** 2).mean(axis=0)
    scale = np.sqrt(var)
    scale[scale < 10 * np.finfo(np.float64).eps] = 1.0
    return mean, scale


def scaler_transform(x: np.ndarray, mean: np.ndarray, scale: np.ndarray) -> np.ndarray:
    """StandardScaler.transform on a float32 array with float64 statistics (train_ld_2011.py:54-59): the
    in-place ``X -= mean_`` and ``X /= scale_`` each compute in float64 and round to float32."""
    x1 = (np.asarray(x, dtype=np.float32).astype(np.float64) - mean).astype(np.float32)
    return (x1.astype(np.float64) / scale).astype(np.float32)
