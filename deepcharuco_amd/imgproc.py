"""Host image helpers either side of the hot path (the reference uses cv2 here)."""
from __future__ import annotations

import numpy as np

_cv2 = None          # resolved once: the module, or False when OpenCV is not installed


def _opencv():
    global _cv2
    if _cv2 is None:
        try:
            import cv2  # type: ignore
            _cv2 = cv2
        except ImportError:
            _cv2 = False
    return _cv2


def bgr2gray(img_bgr: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(img, cv2.COLOR_BGR2GRAY) (call site /root/reference/src/inference.py:40).

    Uses OpenCV when it is importable; otherwise OpenCV's published 8-bit fixed-point formula
    gray = (1868*B + 9617*G + 4899*R + 8192) >> 14.
    """
    cv2 = _opencv()
    if cv2:
        return cv2.cvtColor(img_bgr, cv2.COLOR_BGR2GRAY)
    if img_bgr.ndim != 3 or img_bgr.shape[2] != 3 or img_bgr.dtype != np.uint8:
        raise ValueError("expected a (H,W,3) uint8 BGR image")
    acc = img_bgr[..., 0].astype(np.uint32) * np.uint32(1868)
    acc += img_bgr[..., 1].astype(np.uint32) * np.uint32(9617)
    acc += img_bgr[..., 2].astype(np.uint32) * np.uint32(4899)
    acc += np.uint32(8192)
    acc >>= np.uint32(14)
    return acc.astype(np.uint8)


def bgr2gray_device(bgr):
    """Device version (``dcx_bgr2gray``): (B,H,W,3) or (H,W,3) uint8 GPU tensor -> (B,H,W) / (H,W) uint8 gray on the GPU,
    same fixed-point formula as :func:`bgr2gray`'s fallback (bit-identical; the numpy version costs the host more than half
    of a bs=1 ``infer_image`` call)."""
    import torch
    from . import _lib
    if bgr.device.type != "cuda" or bgr.dtype != torch.uint8 or bgr.shape[-1] != 3 or bgr.ndim not in (3, 4):
        raise ValueError("expected a (B,H,W,3) or (H,W,3) uint8 tensor on the GPU")
    x = bgr.contiguous()
    b = 1 if x.ndim == 3 else x.shape[0]
    h, w = x.shape[-3], x.shape[-2]
    gray = torch.empty(x.shape[:-1], dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().dcx_bgr2gray(x.data_ptr(), h * w * 3, w * 3, b, h, w, gray.data_ptr(), _lib.current_stream()),
                   "dcx_bgr2gray")
    return gray
