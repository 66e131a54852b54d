"""Host image helpers either side of the hot path (the reference uses cv2 here)."""
from __future__ import annotations

import numpy as np


def bgr2gray(img_bgr: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(img, cv2.COLOR_BGR2GRAY) (call site /root/reference/src/inference.py:40).

    Uses OpenCV when it is importable; otherwise OpenCV's published 8-bit fixed-point formula
    gray = (1868*B + 9617*G + 4899*R + 8192) >> 14.
    """
    try:
        import cv2  # type: ignore
        return cv2.cvtColor(img_bgr, cv2.COLOR_BGR2GRAY)
    except ImportError:
        pass
    if img_bgr.ndim != 3 or img_bgr.shape[2] != 3 or img_bgr.dtype != np.uint8:
        raise ValueError("expected a (H,W,3) uint8 BGR image")
    b = img_bgr[..., 0].astype(np.int32)
    g = img_bgr[..., 1].astype(np.int32)
    r = img_bgr[..., 2].astype(np.int32)
    return ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)
