"""ctypes binding of libdeepcharuco_amd.so (the C ABI declared in include/deepcharuco_amd.h).

There is deliberately NO fallback: if the HIP library is missing or a call
fails, an exception is raised.  Nothing in this package computes the hot path
on the CPU or through stock PyTorch operators.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DCX_LIB") or os.path.join(_HERE, "libdeepcharuco_amd.so")   # DCX_LIB: a variant build (kernel sweeps)
CSRC = os.path.join(_HERE, "csrc")

_lib: Optional[C.CDLL] = None


class DcxError(RuntimeError):
    def __init__(self, code: int, where: str):
        self.code = code
        try:
            msg = lib().dcx_error_string(code).decode()
        except Exception:  # pragma: no cover
            msg = "?"
        super().__init__(f"{where} failed with code {code}: {msg}")


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources in-tree for gfx950 (``make`` drives hipcc; works without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j4"]
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=not verbose)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libdeepcharuco_amd.so failed:\n" + res.stdout[-4000:] + res.stderr[-8000:])
    if verbose:
        print(res.stdout)
    return LIB_PATH


_vp, _i, _l, _sz = C.c_void_p, C.c_int, C.c_long, C.c_size_t

# name -> (restype, argtypes); mirrors include/deepcharuco_amd.h one to one
SIGNATURES = {
    "dcx_version": (C.c_char_p, []),
    "dcx_error_string": (C.c_char_p, [_i]),
    "dcx_detector_create": (_i, [C.POINTER(_vp), C.POINTER(_vp), _i, _i]),
    "dcx_detector_destroy": (_i, [_vp]),
    "dcx_refiner_create": (_i, [C.POINTER(_vp), C.POINTER(_vp), _i]),
    "dcx_refiner_destroy": (_i, [_vp]),
    "dcx_detector_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "dcx_refiner_workspace_bytes": (_sz, [_vp, _i]),
    "dcx_bgr2gray": (_i, [_vp, _l, _i, _i, _i, _i, _vp, _vp]),
    "dcx_bgr2gray_legacy14": (_i, [_vp, _l, _i, _i, _i, _i, _vp, _vp]),
    "dcx_pre_image": (_i, [_vp, _vp, _sz, _vp]),
    "dcx_detector_forward": (_i, [_vp, _vp, _l, _i, _vp, _i, _i, _i, _vp, _sz, _vp, _vp, _vp]),
    "dcx_detector_decode": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dcx_pred_to_keypoints": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dcx_label_to_keypoints": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dcx_build_patch_table": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "dcx_extract_patches_u8": (_i, [_vp, _l, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "dcx_extract_patches_f32": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "dcx_refiner_forward": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "dcx_argmax2d": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "dcx_pipeline_workspace_bytes": (_sz, [_vp, _vp, _i, _i, _i, _i]),
    "dcx_infer_batch": (_i, [_vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dcx_conv_layer": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "dcx_nchw_to_c4": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "dcx_c4_to_nchw": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "dcx_set_timing": (_i, [_i]),
    "dcx_get_timing": (_i, []),
    "dcx_profile_enabled": (_i, []),
    "dcx_last_timings": (_i, [C.POINTER(C.c_float)]),
    "dcx_conv_pick_name": (C.c_char_p, [_i] * 8),
    "dcx_conv_pick_name_ups": (C.c_char_p, [_i] * 9),
    "dcx_set_deterministic": (_i, [_i]),
    "dcx_get_deterministic": (_i, []),
    "dcx_calibrate_xcd": (_i, [_i, C.POINTER(C.c_float), _vp]),
    "dcx_set_xcd_weights": (_i, [C.POINTER(C.c_float)]),
    "dcx_get_xcd_weights": (_i, [C.POINTER(C.c_float)]),
    "dcx_set_tail_fence": (_i, [_i]),
    "dcx_get_tail_fence": (_i, []),
    "dcx_profile_enable": (_i, [_i]),
    "dcx_profile_count": (_i, []),
    "dcx_profile_filter": (_i, [_i]),
    "dcx_profile_sample": (_i, [_i]),
    "dcx_profile_fetch": (_i, [C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(C.c_double), C.POINTER(C.c_float), _i]),
    "dcx_profile_kernel_name": (C.c_char_p, [_i]),
    "dcx_profile_clocks": (_i, [C.POINTER(C.c_float), _i]),
    "dcx_profile_probe_words": (_i, [_i, C.POINTER(C.c_ulonglong)]),
}


def lib() -> C.CDLL:
    """Load the library (once). Raises if it has not been built -- there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C {CSRC}`. deepcharuco_amd has no CPU / stock-PyTorch fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def csrc_sha256() -> str:
    """Hash of the kernel sources (csrc/*.hip, *.h): stamps measurements that are only valid for these kernels
    (profiles/pmc_traffic.json)."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(CSRC, name), "rb") as f:
                h.update(name.encode())
                h.update(f.read())
    return h.hexdigest()


def check(rc: int, where: str) -> None:
    if rc != 0:
        raise DcxError(rc, where)


def ptr(t) -> Optional[int]:
    """Device/host pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
