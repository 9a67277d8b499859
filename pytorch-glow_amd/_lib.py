"""ctypes binding of libglowhip.so (the C ABI declared in include/glowhip.h).

The library is built in-tree by ``pytorch-glow_amd/csrc/Makefile`` (hipcc, gfx950) and loaded from
``pytorch-glow_amd/libglowhip.so``.  There is NO fallback: if the library is missing every compute
entry point raises, so a GPU test can never pass on a silent PyTorch path.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_long, c_size_t, c_void_p

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# GLOWHIP_LIB_PATH: kernel-variant experiments (scripts/) load an alternative build of the same ABI
LIB_PATH = os.environ.get("GLOWHIP_LIB_PATH") or os.path.join(_PKG_DIR, "libglowhip.so")
CSRC_DIR = os.path.join(_PKG_DIR, "csrc")

LAYER_SQUEEZE, LAYER_FLOWSTEP, LAYER_SPLIT2D = 0, 1, 2
PERM_INVCONV, PERM_GATHER = 0, 1
COUPLING_ADDITIVE, COUPLING_AFFINE = 0, 1


class GlowHipError(RuntimeError):
    """Non-zero return code from libglowhip (message = glowhip_last_error())."""


class LayerDesc(ctypes.Structure):
    """Mirror of ``glowhip_layer_desc`` (include/glowhip.h)."""
    _fields_ = [
        ("kind", c_int32), ("C", c_int32), ("H", c_int32), ("W", c_int32),
        ("hidden", c_int32), ("permutation", c_int32), ("coupling", c_int32), ("reserved", c_int32),
        ("an_bias", c_void_p), ("an_logs", c_void_p),
        ("invconv_w", c_void_p),
        ("perm_idx", c_void_p), ("perm_idx_inv", c_void_p),
        ("f0_w", c_void_p), ("f0_an_bias", c_void_p), ("f0_an_logs", c_void_p),
        ("f2_w", c_void_p), ("f2_an_bias", c_void_p), ("f2_an_logs", c_void_p),
        ("f4_w", c_void_p), ("f4_bias", c_void_p), ("f4_logs", c_void_p),
    ]


class LayerGrads(ctypes.Structure):
    """Mirror of ``glowhip_layer_grads``."""
    _fields_ = [(n, c_void_p) for n in ("an_bias", "an_logs", "invconv_w", "f0_w", "f0_an_bias", "f0_an_logs",
                                        "f2_w", "f2_an_bias", "f2_an_logs", "f4_w", "f4_bias", "f4_logs")]


class OptimChunk(ctypes.Structure):
    """Mirror of ``glowhip_optim_chunk``."""
    _fields_ = [("param", c_void_p), ("grad", c_void_p), ("m", c_void_p), ("v", c_void_p), ("n", c_int32), ("pad", c_int32)]


class TimingRecord(ctypes.Structure):
    """Mirror of ``glowhip_timing_record``."""
    _fields_ = [("kind", c_int32), ("layer", c_int32), ("mfma", c_int32), ("ms", c_float)]


K_CHANMIX, K_CONV_F0, K_CONV_F2, K_CONV_F4, K_OTHER = range(5)

_P = c_void_p
# name -> (restype, argtypes); every symbol include/glowhip.h declares (tests/test_abi.py checks the two agree)
SIGNATURES = {
    "glowhip_version": (c_int, []),
    "glowhip_last_error": (c_char_p, []),
    "glowhip_squeeze2d": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "glowhip_actnorm_init": (c_int, [_P, c_long, c_int, c_int, c_int, c_float, _P, _P, _P]),
    "glowhip_actnorm_init_batch_variance": (c_int, [_P, c_long, c_int, c_int, c_int, c_float, _P, _P, _P]),
    "glowhip_actnorm": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "glowhip_invconv_scratch_bytes": (c_size_t, [c_int]),
    "glowhip_invconv_prepare": (c_int, [_P, c_int, _P, _P, _P, _P]),
    "glowhip_invconv": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "glowhip_permute_channels": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "glowhip_conv2d": (c_int, [_P, c_long, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, _P]),
    "glowhip_gaussian_logp": (c_int, [_P, c_long, _P, _P, c_long, c_int, c_int, c_int, _P, _P, _P, _P]),
    "glowhip_plan_create": (_P, [POINTER(LayerDesc), c_int]),
    "glowhip_plan_destroy": (None, [_P]),
    "glowhip_plan_packed_bytes": (c_size_t, [_P]),
    "glowhip_plan_workspace_bytes": (c_size_t, [_P, c_int]),
    "glowhip_plan_pack": (c_int, [_P, _P, c_size_t, _P]),
    "glowhip_plan_pack_for": (c_int, [_P, _P, c_size_t, c_int, _P]),
    "glowhip_plan_pack_sync": (c_int, [_P]),
    "glowhip_plan_forget_packed": (c_int, [_P]),
    "glowhip_plan_encode": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "glowhip_plan_decode": (c_int, [_P, _P, _P, POINTER(c_void_p), c_int, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "glowhip_glow_forward": (c_int, [_P, _P, _P, _P, _P, _P, c_long, c_int, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "glowhip_glow_forward_u8": (c_int, [_P, _P, _P, c_float, _P, _P, _P, c_long, c_int, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "glowhip_plan_set_family": (c_int, [_P, c_int]),
    "glowhip_plan_get_family": (c_int, [_P]),
    "glowhip_plan_status": (c_int, [_P, _P, c_size_t, c_int, _P, c_long, _P, _P]),
    "glowhip_plan_set_dequant_rng": (c_int, [_P, ctypes.c_ulonglong, c_int, POINTER(ctypes.c_ulonglong)]),
    "glowhip_plan_set_dequant_stream": (c_int, [_P, ctypes.c_ulonglong, ctypes.c_ulonglong]),
    "glowhip_dequant_noise": (c_int, [_P, c_long, ctypes.c_ulonglong, ctypes.c_ulonglong, c_int, _P]),
    "glowhip_plan_actnorm_init": (c_int, [_P, _P, c_size_t, _P, _P, c_float, c_int, _P, c_size_t, _P]),
    "glowhip_plan_output_shape": (c_int, [_P, c_int, POINTER(c_int32)]),
    "glowhip_plan_describe": (c_int, [_P, c_char_p, c_size_t]),
    "glowhip_plan_describe_for": (c_int, [_P, c_int, c_char_p, c_size_t]),
    "glowhip_plan_launch_counts": (c_int, [_P, c_char_p, c_size_t, c_int]),
    "glowhip_debug_force_tail_tile": (None, [c_int]),
    "glowhip_plan_tape_bytes": (c_size_t, [_P, c_int]),
    "glowhip_plan_train_workspace_bytes": (c_size_t, [_P, c_int]),
    "glowhip_glow_forward_train": (c_int, [_P, _P, _P, _P, _P, _P, c_long, c_int, _P, _P, _P, c_int, _P, c_size_t, _P,
                                           c_size_t, _P]),
    "glowhip_glow_backward": (c_int, [_P, _P, _P, _P, c_size_t, _P, _P, _P, _P, c_long, POINTER(LayerGrads), _P, c_int,
                                      _P, c_size_t, _P]),
    "glowhip_plan_backward_marks": (c_int, [_P, POINTER(c_int32), POINTER(c_void_p), c_int]),
    "glowhip_optim_step": (c_int, [_P, c_int, c_int, c_float, ctypes.c_double, ctypes.c_double, c_float, c_float, c_int, c_float, c_float, _P, _P, c_int, _P]),
    "glowhip_optim_step_dev": (c_int, [_P, c_int, c_int, _P, ctypes.c_double, ctypes.c_double, c_float, c_float, c_float, c_float, _P, _P, c_int, _P]),
    "glowhip_plan_timing_enable": (c_int, [_P, c_int]),
    "glowhip_plan_timing_read": (c_int, [_P, POINTER(TimingRecord), c_int, POINTER(c_int)]),
}

_lib = None
_lock = threading.Lock()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile libglowhip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC_DIR, "clean"], check=True, capture_output=not verbose)
    proc = subprocess.run(["make", "-C", CSRC_DIR, "-j", str(min(8, os.cpu_count() or 1))],
                          capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("building libglowhip.so failed:\n" + proc.stdout[-4000:] + proc.stderr[-4000:])
    if verbose:
        print(proc.stdout)
    return LIB_PATH


def lib() -> ctypes.CDLL:
    """The loaded library; raises (never falls back) if it is not there."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise GlowHipError(
                        f"{LIB_PATH} not found: build it with `make -C {CSRC_DIR}` (or __graft_entry__.build()). "
                        "There is no CPU/PyTorch fallback for the flow path.")
                handle = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(handle, name)
                    fn.restype, fn.argtypes = res, args
                _lib = handle
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().glowhip_last_error()
        raise GlowHipError(f"libglowhip error {rc}: {msg.decode() if msg else '?'}")


def stream_ptr(device=None) -> c_void_p:
    """hipStream_t of torch's current stream on ``device`` (so kernels order with torch's own work)."""
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t) -> c_void_p:
    return c_void_p(None) if t is None else c_void_p(t.data_ptr())


def require_device_tensor(t: torch.Tensor, what: str = "input", allow_uint8: bool = False) -> torch.Tensor:
    """The HIP path only takes fp32 tensors resident on a GPU (the Glow input also as 8-bit pixels); anything else is an
    error, not a fallback."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{what}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise GlowHipError(f"{what} is on {t.device}: the Glow flow path runs only on a HIP device "
                           "(there is no CPU fallback; use oracle/ for CPU checks in tests)")
    if allow_uint8 and t.dtype == torch.uint8:
        return t.contiguous()
    if t.dtype != torch.float32:
        raise GlowHipError(f"{what}: dtype {t.dtype} unsupported, the path computes in fp32")
    return t.contiguous()
