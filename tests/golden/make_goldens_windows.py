#!/usr/bin/env python3
"""Generate tests/golden/windows_*.npz by running the REFERENCE's data-preparation functions on CPU.

Usage (build container only; the reference tree does not travel to the GPU box):
    python tests/golden/make_goldens_windows.py /root/reference

The reference's drivers run at import time and import packages that are absent here, so only the
function definitions named below are compiled, out of the parsed source of
vae-models/graph/train_ld_2011.py (create_sliding_windows) and
vae-models/graph/training_block_max_hurdat.py (create_X_data, create_y_data).  They are run on
slices of the reference's own datasets, followed by sklearn's StandardScaler exactly as
train_ld_2011.py:47-59 uses it.  Nothing of the reference's text is stored: the fixtures hold arrays
only (series, window starts, windows, targets, scaler statistics).
"""
from __future__ import annotations

import ast
import os
import sys

import numpy as np
import pandas as pd
from sklearn.model_selection import train_test_split
from sklearn.preprocessing import StandardScaler

HERE = os.path.dirname(os.path.abspath(__file__))


def reference_functions(root: str, rel: str, names):
    path = os.path.join(root, rel)
    with open(path, "r") as fh:
        tree = ast.parse(fh.read(), path)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert {n.name for n in body} == set(names), rel
    ns = {"np": np}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


def ld2011_case(root, csv, rows, T, P, tag):
    (create_sliding_windows,) = reference_functions(root, "vae-models/graph/train_ld_2011.py", ["create_sliding_windows"])
    df = pd.read_csv(os.path.join(root, "datasets", csv), delimiter=",").select_dtypes(include=[np.number])
    data = df.values.astype(np.float32)[:rows]                     # load_csv_dataset, train_ld_2011.py:19-22
    X_data, Y_data = create_sliding_windows(data, T, P)             # :39
    idx = np.arange(len(X_data))
    X_train, X_test, y_train, y_test, i_train, i_test = train_test_split(X_data, Y_data.reshape(len(X_data), -1), idx,
                                                                          test_size=0.2, random_state=42)   # :46
    F = X_train.shape[2]
    sx, sy = StandardScaler(), StandardScaler()
    Xs_train = sx.fit_transform(X_train.reshape(-1, F)).reshape(-1, T, F)                                   # :54
    ys_train = sy.fit_transform(y_train.reshape(-1, 1)).reshape(-1, P)                                      # :56
    Xs_test = sx.transform(X_test.reshape(-1, F)).reshape(-1, T, F)                                         # :58
    ys_test = sy.transform(y_test.reshape(-1, 1)).reshape(-1, P)                                            # :60
    np.savez_compressed(os.path.join(HERE, f"windows_{tag}.npz"), series=data, T=T, P=P,
                        X_all=X_data, y_all=Y_data.reshape(len(X_data), -1), train_starts=i_train, test_starts=i_test,
                        x_mean=sx.mean_, x_scale=sx.scale_, y_mean=sy.mean_, y_scale=sy.scale_,
                        X_train=Xs_train.astype(np.float32), y_train=ys_train.astype(np.float32),
                        X_test=Xs_test.astype(np.float32), y_test=ys_test.astype(np.float32))
    print(tag, data.shape, X_data.shape, Xs_train.dtype)


def block_max_case(root, rows, T, P, tag):
    create_X_data, create_y_data = reference_functions(root, "vae-models/graph/training_block_max_hurdat.py",
                                                       ["create_X_data", "create_y_data"])
    s = pd.read_csv(os.path.join(root, "datasets", "sunspots.csv")).select_dtypes(include=[np.number]).values
    series = s[:rows, -1].astype(np.float32)
    # blocks of T + P consecutive values, one starting at every step (the layout the block-maximum drivers feed)
    blocks = np.stack([series[i:i + T + P] for i in range(len(series) - T - P + 1)])
    X = create_X_data(blocks, T)                                     # training_block_max_hurdat.py:81-86
    y = create_y_data(blocks, T)                                     # :100-106
    np.savez_compressed(os.path.join(HERE, f"windows_{tag}.npz"), series=series[:, None], T=T, P=P,
                        X_all=X[:, :, None].astype(np.float32), y_all=y.reshape(-1, 1).astype(np.float32))
    print(tag, series.shape, X.shape, y.shape)


def main():
    if len(sys.argv) < 2 or not os.path.isdir(sys.argv[1]):
        print("usage: make_goldens_windows.py /root/reference  (no reference tree: nothing to do)")
        return
    root = sys.argv[1]
    ld2011_case(root, "LD2011_2014_less.csv", 600, 4, 1, "ld2011_t4")       # the driver's own window_length / predict_steps
    ld2011_case(root, "LD2011_2014.csv", 300, 24, 3, "ld2011wide_t24_p3")   # 45 channels, several target steps
    ld2011_case(root, "sunspots.csv", 500, 12, 1, "sunspots_t12")           # one channel, the function's defaults
    block_max_case(root, 400, 16, 8, "blockmax_t16_p8")                     # 24-step blocks: 16 in, maximum of 8


if __name__ == "__main__":
    main()
