"""ctypes binding of libumhs_hip.so (include/umhs_hip.h).

There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised.  The
library is built in-tree by ``__graft_entry__.build()`` / ``python -m umhsnerf.build``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 11  # include/umhs_hip.h UMHS_ABI_VERSION: bumped with every signature change
LIB_PATH = os.environ.get("UMHS_LIB_PATH") or os.path.join(_HERE, "libumhs_hip.so")  # override: A/B builds of tools/ab_lib.sh
MAX_STREAMS = 4

_vp, _i64, _i32, _f32 = C.c_void_p, C.c_int64, C.c_int32, C.c_float


class FieldCfg(C.Structure):
    _fields_ = [("n_bands", _i32), ("n_classes", _i32), ("pred_specular", _i32), ("density_only", _i32), ("temperature", _f32)]


PARAM_FIELDS = (
    "base_w0", "base_b0", "base_w1", "base_b1",
    "head_w0", "head_b0", "head_w1", "head_b1", "head_w2", "head_b2",
    "feat_w0", "feat_b0", "feat_w1", "feat_b1", "feat_w2", "feat_b2",
    "dir_w0", "dir_b0", "dir_w1", "dir_b1",
    "endmembers",
)


class FieldParams(C.Structure):
    _fields_ = [(k, _vp) for k in PARAM_FIELDS]


class FieldGrads(C.Structure):
    _fields_ = [(k, _vp) for k in PARAM_FIELDS]


class ValueStreams(C.Structure):
    _fields_ = [("n_streams", _i32), ("k", _i32 * MAX_STREAMS), ("values", _vp * MAX_STREAMS), ("out", _vp * MAX_STREAMS)]


class ValueGrads(C.Structure):
    _fields_ = [("n_streams", _i32), ("k", _i32 * MAX_STREAMS), ("values", _vp * MAX_STREAMS), ("d_out", _vp * MAX_STREAMS),
                ("d_values", _vp * MAX_STREAMS)]


# symbol -> (restype, argtypes); must list every function declared in include/umhs_hip.h
SIGNATURES = {
    "umhs_strerror": (C.c_char_p, [C.c_int]),
    "umhs_abi_version": (C.c_int, []),
    "umhs_positions_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, C.c_int, C.POINTER(_f32), _vp, _vp, _vp, _vp]),
    "umhs_hashgrid_fwd": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, C.c_int, _vp, _i64, _i64, _vp]),
    "umhs_hashgrid_bwd_workspace_bytes": (C.c_size_t, [_i64, C.c_int, C.c_int]),
    "umhs_hashgrid_bwd": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _i64, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_size_t, _vp]),
    "umhs_field_fwd_workspace_bytes": (C.c_size_t, [C.POINTER(FieldCfg)]),
    "umhs_field_fwd": (C.c_int, [C.POINTER(FieldCfg), C.POINTER(FieldParams), _vp, _i64, _i64, _vp, _vp, _vp, _i64,
                                 _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, C.c_int, _vp]),
    "umhs_field_fwd_prepare": (C.c_int, [C.POINTER(FieldCfg), C.POINTER(FieldParams), _vp, C.c_size_t, _vp]),
    "umhs_field_base_fwd": (C.c_int, [C.POINTER(FieldCfg), C.POINTER(FieldParams), _vp, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp,
                                     C.c_size_t, C.c_int, _vp]),
    "umhs_field_heads_fwd_scratch_bytes": (C.c_size_t, [C.POINTER(FieldCfg), _i64, _i64]),
    "umhs_field_heads_fwd_supported": (C.c_int, [C.POINTER(FieldCfg)]),
    "umhs_field_heads_fwd": (C.c_int, [C.POINTER(FieldCfg), C.POINTER(FieldParams), _vp, C.c_int, _vp, _vp, _i64, _vp, _vp, _vp, _i64] + [_vp] * 6
                             + [_vp, C.c_size_t, _vp, C.c_size_t, C.c_int, _vp]),
    "umhs_field_bwd_prepare": (C.c_int, [C.POINTER(FieldCfg), C.POINTER(FieldParams), _vp, C.c_size_t, _vp]),
    "umhs_hashgrid_bwd_prepare": (C.c_int, [_vp, _vp, _i64, C.c_int, C.c_int, C.c_int, _vp, C.c_size_t, _vp]),
    "umhs_hashgrid_bwd_apply": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp,
                                          C.c_size_t, _vp]),
    "umhs_hashgrid_bwd_apply_adam": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp,
                                               C.c_size_t, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _i64, C.c_int, _vp]),
    "umhs_field_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(FieldCfg), _i64]),
    "umhs_field_bwd": (C.c_int, [C.POINTER(FieldCfg), C.POINTER(FieldParams), _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64,
                                 _vp, _vp, _vp, _vp, C.POINTER(FieldGrads), _vp, C.c_size_t, C.c_int, _vp]),
    "umhs_field_bwd_composited_supported": (C.c_int, [C.POINTER(FieldCfg)]),
    "umhs_field_bwd_composited_scratch_bytes": (C.c_size_t, [C.POINTER(FieldCfg), _i64, _i64]),
    "umhs_field_bwd_composited": (C.c_int, [C.POINTER(FieldCfg), C.POINTER(FieldParams), _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _i64,
                                            _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, C.POINTER(FieldGrads), _vp,
                                            C.c_size_t, _vp, C.c_size_t, C.c_int, _vp]),
    "umhs_composite_bwd_dots": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, C.c_int, _vp, _vp]),
    "umhs_pack_info": (C.c_int, [_vp, _i64, _i64, _vp, _vp]),
    "umhs_composite_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, C.POINTER(ValueStreams), _vp, _vp, _vp, _vp]),
    "umhs_composite_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, C.POINTER(ValueGrads), _vp, C.c_int, _vp, _vp]),
    "umhs_accumulate_fwd": (C.c_int, [_vp, _vp, _i64, _i64, C.POINTER(ValueStreams), _vp]),
    "umhs_accumulate_bwd": (C.c_int, [_vp, _vp, _i64, _i64, C.POINTER(ValueGrads), _vp, _vp]),
    "umhs_spec2rgb_fwd": (C.c_int, [_vp, _vp, _i64, C.c_int, _vp, _vp]),
    "umhs_spec2rgb_bwd": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, _vp, C.c_int, _vp]),
    "umhs_tmid_minmax": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "umhs_ray_epilogue_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, C.c_int, C.c_int, _f32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "umhs_loss_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, C.c_int, _f32, _f32, _vp, _vp]),
    "umhs_loss_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, C.c_int, _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "umhs_march_walk_workspace_bytes": (C.c_size_t, [_i64]),
    "umhs_march_walk": (C.c_int, [_vp, _vp, _i64, _vp, C.POINTER(_f32), C.c_int, C.c_int, _f32, _f32, _vp, _vp, _vp, _f32, _vp, C.c_size_t, _vp]),
    "umhs_march_count": (C.c_int, [_vp, _vp, _i64, _vp, C.POINTER(_f32), C.c_int, C.c_int, _f32, _f32, _f32, _f32, _vp, _vp, _vp, _f32, _vp, _vp, C.c_size_t, _vp]),
    "umhs_march_write": (C.c_int, [_vp, _vp, _i64, _vp, C.POINTER(_f32), C.c_int, C.c_int, _f32, _f32, _f32, _f32, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "umhs_march_scratch": (C.c_int, [_vp, _vp, _i64, _vp, C.POINTER(_f32), C.c_int, C.c_int, _f32, _f32, _f32, _f32, _vp, _vp, _vp, _f32, C.c_int, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "umhs_march_compact": (C.c_int, [_vp, _i64, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "umhs_visibility": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _f32, _f32, _vp, _vp]),
    "umhs_visibility_count": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _f32, _f32, _vp, _vp, _vp]),
    "umhs_ray_prefix": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "umhs_sample_midpoints": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "umhs_compact_samples": (C.c_int, [_vp] * 3 + [_i64] + [_vp] * 13),
    "umhs_ray_train_tail_scratch_bytes": (C.c_size_t, []),
    "umhs_ray_train_tail": (C.c_int, [_vp] * 10 + [_i64, C.c_int, C.c_int, _f32, _f32, _f32, C.c_int] + [_vp] * 9 + [C.c_size_t, _vp]),
    "umhs_enc_gather": (C.c_int, [_vp, _vp, _i64, _i64, C.c_int, _vp, _vp]),
    "umhs_pixel_indices": (C.c_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "umhs_raygen": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "umhs_pixel_gather": (C.c_int, [_vp, _vp, C.c_int, _i64, _i64, _i64, C.c_int, _i64, _vp, _vp]),
    "umhs_pixel_metrics": (C.c_int, [_vp, _vp, _i64, C.c_int, _vp, C.c_int, _vp]),
    "umhs_ssim_partials": (_i64, [C.c_int, C.c_int, C.c_int]),
    "umhs_ssim": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _i64, _vp]),
    "umhs_adam_step_rows": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _i64, _f32, _vp]),
    "umhs_adam_step_rows_range": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _f32, _f32, _f32, _f32, _i64, _f32, _i64, _i64, _vp]),
    "umhs_hashgrid_fwd_count": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, C.c_int, _vp, _i64, _i64, _vp, C.c_size_t, _vp]),
    "umhs_hashgrid_bwd_prepare_counted": (C.c_int, [_vp, _vp, _i64, C.c_int, C.c_int, _vp, C.c_size_t, _vp]),
    "umhs_rgb_mlp_bwd_workspace_bytes": (C.c_size_t, [C.c_int, _i64]),
    "umhs_rgb_base_fwd": (C.c_int, [_vp] * 6 + [_i64] + [_vp] * 4),
    "umhs_rgb_head_fwd": (C.c_int, [_vp] * 8 + [_i64] + [_vp] * 2),
    "umhs_rgb_base_bwd": (C.c_int, [_vp] * 8 + [_i64] + [_vp] * 5 + [C.c_int, _vp, C.c_size_t, _vp]),
    "umhs_rgb_head_bwd": (C.c_int, [_vp] * 9 + [_i64] + [_vp] * 7 + [C.c_int, _vp, C.c_size_t, _vp]),
    "umhs_adam_step": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _i64, _f32, _i64, _i64, _vp]),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load libumhs_hip.so (once).  Raises if it has not been built -- there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the UMHS hot path has no CPU fallback. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950)."
            )
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        if l.umhs_abi_version() != ABI_VERSION:
            raise RuntimeError("libumhs_hip.so ABI version mismatch")
        _lib = l
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        raise RuntimeError(f"{what} failed: {lib().umhs_strerror(code).decode()} ({code})")


def ptr(t: Optional[torch.Tensor]):
    """Device pointer of a contiguous fp32/int64 CUDA(HIP) tensor, or NULL."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("umhsnerf HIP ops need tensors on a HIP device (cuda:N); there is no CPU path")
    if not t.is_contiguous():
        raise RuntimeError("umhsnerf HIP ops need contiguous tensors")
    return C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """hipStream_t of torch's current stream on the current device.  Called once per launch: the public
    ``torch.cuda.current_stream().cuda_stream`` costs ~8 us of Python (40 launches = 0.3 ms of a launch-bound step), the raw getter 0.3."""
    if _raw_stream is not None and _cur_device is not None:
        return C.c_void_p(_raw_stream(_cur_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()
