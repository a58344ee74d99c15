"""ctypes binding of libvts_hip.so (the C ABI declared in include/vts.h).

There is no fallback: if the shared library is missing, or a call is made with tensors that
are not on a ROCm device, this module raises.  Tensors are passed as raw device pointers
(``Tensor.data_ptr()``) together with torch's current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvts_hip.so")

ACT_NONE, ACT_RELU, ACT_EXP, ACT_CLAMP_MIN, ACT_RELU_DROPOUT = 0, 1, 2, 3, 4
GAUSSIAN, GPD, MIXTURE, MIXTURE_SPLIT = 0, 1, 2, 3
TARGET_NEXT, TARGET_BLOCK_MAX = 0, 1
FAMILY = {"gaussian": GAUSSIAN, "gpd": GPD, "mixture": MIXTURE}

_i64, _f32, _int, _ptr, _sz, _u64 = C.c_int64, C.c_float, C.c_int, C.c_void_p, C.c_size_t, C.c_uint64

class MlpChain(C.Structure):
    """vts_mlp_chain of include/vts.h."""
    _fields_ = [("n_trunk", C.c_int32), ("n_heads", C.c_int32), ("in_dim", C.c_int32), ("head_dim", C.c_int32),
                ("trunk_dim", C.c_int32 * 3), ("trunk_act", C.c_int32 * 3), ("head_act", C.c_int32 * 8),
                ("keep_scale", C.c_float)]


def ptr_array(tensors):
    """A host array of device pointers (None entries = NULL) for the `const float* const*` arguments."""
    arr = (C.c_void_p * max(len(tensors), 1))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


# name -> (restype, argtypes); mirrors include/vts.h one to one
SIGNATURES = {
    "vts_version": (C.c_char_p, []),
    "vts_last_error": (C.c_char_p, []),
    "vts_lstm_gates_bytes": (_sz, [_i64, _i64, _i64]),
    "vts_lstm_cells_bytes": (_sz, [_i64, _i64, _i64]),
    "vts_lstm_seqgrad_bytes": (_sz, [_i64, _i64, _i64]),
    "vts_lstm_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "vts_lstm_layer_fwd": (_int, [_ptr] * 10 + [_ptr, _sz, _i64, _i64, _i64, _i64, _int, _ptr]),
    "vts_lstm_layer_bwd": (_int, [_ptr] * 14 + [_int, _ptr, _sz, _i64, _i64, _i64, _i64, _int, _ptr]),
    "vts_lstm_recurrent_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "vts_lstm_wgrad_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "vts_lstm_inproj_fwd": (_int, [_ptr] * 5 + [_i64, _i64, _i64, _i64, _int, _ptr]),
    "vts_lstm_recurrent_fwd": (_int, [_ptr] * 6 + [_ptr, _sz, _i64, _i64, _i64, _int, _ptr]),
    "vts_lstm_recurrent_fwd_status": (_int, [_ptr, _sz, _i64, _i64, _i64, _ptr, _ptr]),
    "vts_debug_set_cluster_fault": (None, [_int]),
    "vts_debug_set_cluster_rings": (None, [_int]),
    "vts_debug_set_cluster_bwd": (None, [_int]),
    "vts_debug_set_staged_bwd": (None, [_int]),
    "vts_debug_set_pair_bwd": (None, [_int]),
    "vts_debug_set_pair_fault": (None, [_int]),
    "vts_lstm_recurrent_bwd_status": (_int, [_ptr, _sz, _i64, _i64, _i64, _ptr, _ptr]),
    "vts_lstm_recurrent_bwd": (_int, [_ptr] * 8 + [_int, _ptr, _sz, _i64, _i64, _i64, _int, _ptr]),
    "vts_lstm_dgrad": (_int, [_ptr] * 3 + [_i64, _i64, _i64, _i64, _int, _ptr]),
    "vts_lstm_wgrad": (_int, [_ptr, _ptr, _int, _ptr, _int, _ptr, _sz, _i64, _i64, _i64, _i64, _int, _ptr]),
    "vts_lstm_wgrad_pair_supported": (_int, [_i64, _i64, _i64, _i64, _int]),
    "vts_lstm_wgrad_pair": (_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _int, _ptr, _sz, _i64, _i64, _i64, _i64, _int, _ptr]),
    "vts_lstm_planes_supported": (_int, [_i64, _i64, _i64, _i64, _int]),
    "vts_lstm_planes_bytes": (_sz, [_i64, _i64, _i64, _int]),
    "vts_lstm_planes_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "vts_lstm_planes_from_rows": (_int, [_ptr, _ptr, _i64, _i64, _i64, _i64, _int, _ptr]),
    "vts_lstm_inproj_fwd_planes": (_int, [_ptr, _int, _ptr, _ptr, _ptr, _ptr, _ptr, _sz, _i64, _i64, _i64, _i64, _int,
                                          _ptr]),
    "vts_lstm_recurrent_fwd_planes": (_int, [_ptr] * 7 + [_ptr, _sz, _i64, _i64, _i64, _int, _ptr]),
    "vts_lstm_fused_fwd_supported": (_int, [_i64, _i64, _i64, _i64, _int]),
    "vts_lstm_fused_fwd_preferred": (_int, []),
    "vts_lstm_fused_fwd_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "vts_lstm_fused_fwd_planes": (_int, [_ptr, _int] + [_ptr] * 10 + [_ptr, _sz, _i64, _i64, _i64, _i64, _int, _ptr]),
    "vts_lstm_fused_fwd_status": (_int, [_ptr, _sz, _i64, _i64, _i64, _i64, _ptr, _ptr]),
    "vts_debug_set_fused_fault": (None, [_int]),
    "vts_debug_set_fused_fwd": (None, [_int]),
    "vts_lstm_recurrent_bwd_planes": (_int, [_ptr] * 9 + [_int, _ptr, _sz, _i64, _i64, _i64, _int, _ptr]),
    "vts_lstm_dgrad_planes": (_int, [_ptr, _ptr, _ptr, _ptr, _sz, _i64, _i64, _i64, _i64, _int, _ptr]),
    "vts_lstm_wgrad_planes": (_int, [_ptr, _ptr, _int, _ptr, _ptr, _ptr, _int, _ptr, _sz, _i64, _i64, _i64, _i64,
                                     _int, _ptr]),
    "vts_tile16_convert": (_int, [_ptr, _ptr, _i64, _i64, _i64, _int, _ptr]),
    "vts_linear_fwd": (_int, [_ptr, _i64, _ptr, _ptr, _ptr, _i64, _i64, _i64, _int, _ptr, _f32, _ptr, _int, _ptr]),
    "vts_linear_bwd_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "vts_linear_bwd": (_int, [_ptr, _ptr, _ptr, _int, _f32, _ptr, _i64, _ptr, _ptr, _i64, _int, _ptr, _ptr, _int,
                              _ptr, _sz, _i64, _i64, _i64, _int, _ptr]),
    "vts_mlp_chain_supported": (_int, [_ptr, _i64]),
    "vts_mlp_chain_fwd": (_int, [_ptr, _ptr, _i64, _i64] + [_ptr] * 6 + [_ptr]),
    "vts_mlp_chain_bwd_workspace_bytes": (_sz, [_ptr, _i64]),
    "vts_mlp_chain_bwd": (_int, [_ptr, _ptr, _i64, _i64] + [_ptr] * 6 + [_ptr, _i64, _ptr, _ptr, _ptr, _sz, _ptr]),
    "vts_layernorm_fwd": (_int, [_ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _f32, _ptr]),
    "vts_layernorm_bwd_workspace_bytes": (_sz, [_i64, _i64]),
    "vts_layernorm_bwd": (_int, [_ptr, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _i64, _ptr, _ptr, _int, _ptr, _sz,
                                 _i64, _i64, _ptr]),
    "vts_reparam_fwd": (_int, [_int] + [_ptr] * 9 + [_i64, _i64, _ptr]),
    "vts_reparam_bwd": (_int, [_int] + [_ptr] * 12 + [_i64, _i64, _ptr]),
    "vts_elbo_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "vts_elbo_fwd": (_int, [_int] + [_ptr] * 7 + [_f32, _f32, _ptr, _ptr, _sz, _i64, _i64, _i64, _ptr]),
    "vts_elbo_fwd_static_max_elements": (_i64, []),
    "vts_elbo_fwd_static": (_int, [_int] + [_ptr] * 7 + [_f32, _f32, _ptr, _i64, _i64, _i64, _ptr]),
    "vts_elbo_bwd": (_int, [_int] + [_ptr] * 8 + [_f32] + [_ptr] * 6 + [_i64, _i64, _i64, _ptr]),
    "vts_reparam_kl_blocks": (_i64, [_i64, _i64]),
    "vts_reparam_kl_fwd": (_int, [_int] + [_ptr] * 8 + [_f32, _ptr, _ptr, _i64, _i64, _ptr]),
    "vts_reparam_kl_bwd": (_int, [_int] + [_ptr] * 10 + [_f32] + [_ptr] * 4 + [_i64, _i64, _ptr]),
    "vts_elbo_recon_fwd": (_int, [_int, _ptr, _ptr, _ptr, _i64, _ptr, _f32, _f32, _ptr, _ptr, _sz, _i64, _i64, _i64,
                                  _ptr]),
    "vts_elbo_recon_bwd": (_int, [_int, _ptr, _ptr, _ptr, _ptr, _f32, _ptr, _ptr, _i64, _ptr, _i64, _i64, _ptr]),
    "vts_elbo_sums": (_int, [_int] + [_ptr] * 7 + [_i64, _f32, _ptr, _ptr, _sz, _i64, _i64, _i64, _ptr]),
    "vts_elbo_from_sums": (_int, [_int, _ptr, _ptr, _f32, _f32, _ptr, _i64, _i64, _ptr]),
    "vts_linear_relu_dropout_fwd": (_int, [_ptr, _i64, _ptr, _ptr, _ptr, _i64, _i64, _i64, _f32, _u64, _int, _ptr]),
    "vts_ffn_hidden_bytes": (_sz, [_i64, _i64]),
    "vts_ffn_mask_bytes": (_sz, [_i64, _i64]),
    "vts_ffn_bwd_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "vts_ffn_fwd": (_int, [_ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _f32, _u64, _ptr, _ptr, _ptr, _i64, _i64, _i64,
                           _int, _ptr]),
    "vts_ffn_bwd": (_int, [_ptr, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _f32, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr,
                           _sz, _i64, _i64, _i64, _int, _ptr]),
    "vts_ffn_recompute_supported": (_int, [_ptr, _i64, _i64, _i64, _i64, _int]),
    "vts_ffn_bwd_recompute_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "vts_ffn_bwd_recompute": (_int, [_ptr, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _f32, _ptr, _i64, _ptr, _ptr, _ptr, _ptr,
                                     _ptr, _sz, _i64, _i64, _i64, _int, _ptr]),
    "vts_mha_fwd": (_int, [_ptr, _i64, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _f32, _u64, _i64, _i64, _i64, _i64,
                           _i64, _ptr]),
    "vts_mha_bwd": (_int, [_ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _f32, _u64, _ptr, _i64,
                           _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _ptr]),
    "vts_dropout_add_fwd": (_int, [_ptr, _ptr, _ptr, _f32, _u64, _ptr, _i64, _ptr]),
    "vts_dropout_bwd": (_int, [_ptr, _ptr, _f32, _u64, _ptr, _i64, _ptr]),
    "vts_dropout_mask": (_int, [_u64, _f32, _ptr, _i64, _ptr]),
    "vts_adam_step": (_int, [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _f32, _f32, _f32, _f32, _f32, _ptr]),
    "vts_adam_step_guarded": (_int, [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _f32, _f32, _f32, _f32, _f32, _ptr, _i64, _ptr]),
    "vts_adam_step_counted": (_int, [_ptr, _ptr, _ptr, _ptr, _i64, _ptr, _f32, _f32, _f32, _f32, _f32, _ptr]),
    "vts_gather_chunks": (_int, [_ptr, _ptr, _ptr, _i64, _ptr, _ptr]),
    "vts_scale": (_int, [_ptr, _ptr, _ptr, _i64, _ptr]),
    "vts_reparam3_fwd": (_int, [_ptr] * 6 + [_int] + [_ptr] * 5 + [_i64, _i64, _ptr]),
    "vts_reparam3_bwd": (_int, [_ptr] * 12 + [_i64, _i64, _ptr]),
    "vts_elbo3_workspace_bytes": (_sz, [_i64, _i64]),
    "vts_elbo3_fwd": (_int, [_ptr] * 8 + [_f32, _f32, _ptr, _ptr, _ptr, _sz, _i64, _i64, _ptr]),
    "vts_elbo3_bwd": (_int, [_ptr] * 7 + [_f32, _f32] + [_ptr] * 9 + [_i64, _i64, _ptr]),
    "vts_windows_build": (_int, [_ptr, _i64, _i64, _ptr, _i64, _i64, _i64, _int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "vts_windows_stats_workspace_bytes": (_sz, [_i64]),
    "vts_windows_stats": (_int, [_ptr, _i64, _i64, _ptr, _i64, _i64, _i64, _int, _ptr, _ptr, _ptr, _sz, _ptr]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the HIP library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `make -C vae-time-series_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU or PyTorch fallback")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def version() -> str:
    return load().vts_version().decode()


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {load().vts_last_error().decode()}")


def ptr(t: Optional[torch.Tensor]):
    """Raw device pointer of a tensor (None -> NULL).  Refuses anything not on a GPU."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("vts kernels run on a ROCm device only: got a tensor on " + str(t.device))
    if t.dtype != torch.float32:
        raise RuntimeError(f"vts kernels take float32 tensors, got {t.dtype}")
    return C.c_void_p(t.data_ptr())


def bptr(t: Optional[torch.Tensor]):
    """Raw device pointer of a uint8 mask (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.uint8 or not t.is_contiguous():
        raise RuntimeError("vts byte masks must be contiguous uint8 tensors on a ROCm device")
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def workspace(nbytes: int, device) -> torch.Tensor:
    """Scratch bytes from torch's caching allocator (16-byte aligned by construction)."""
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def wptr(t: torch.Tensor):
    return C.c_void_p(t.data_ptr())


class KernelTimer:
    """Brackets selected C-ABI calls with HIP events on torch's current stream (the stream the
    kernels are enqueued on) and accumulates device time per call name.  Used by bench.py to
    measure the dominant kernels inside the timed region; off by default.  `only`: a set of call names to bracket
    (None = all of them); every bracket puts two markers into the queue, a few microseconds of idle device each."""

    def __init__(self):
        self.records = []   # (name, start_event, end_event)
        self.enabled = False
        self.only = None

    def call(self, name, fn, *args):
        if not self.enabled or (self.only is not None and name not in self.only):
            return fn(*args)
        start = torch.cuda.Event(enable_timing=True)
        end = torch.cuda.Event(enable_timing=True)
        start.record()
        rc = fn(*args)
        end.record()
        self.records.append((name, start, end))
        return rc

    def summary(self):
        """name -> (launches, total_ms); call after torch.cuda.synchronize()."""
        out = {}
        for name, s, e in self.records:
            n, tot = out.get(name, (0, 0.0))
            out[name] = (n + 1, tot + s.elapsed_time(e))
        return out

    def reset(self):
        self.records = []


timer = KernelTimer()
