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


# cv2.cvtColor(img, COLOR_BGR2GRAY) on 8-bit images: (cB, cG, cR, shift) of gray = (cB*B + cG*G + cR*R + (1 << (shift-1))) >> shift.
#   "opencv4"  -- OpenCV 4.x (the range the reference pins: opencv-contrib-python >= 4.6, < 4.12, requirements.txt:5):
#                 imgproc/src/color.hpp `gray_shift = 15, RY15 = 9798, GY15 = 19235, BY15 = 3735`, used by RGB2Gray<uchar>
#                 (imgproc/src/color_rgb.simd.hpp; its SIMD and scalar branches compute the same integer expression).
#   "legacy14" -- the 14-bit constants `R2Y = 4899, G2Y = 9617, B2Y = 1868` (yuv_shift = 14) that older OpenCV generations also
#                 used for 8-bit gray; in 4.x they remain for 16-bit images and the YUV conversions only.
# The two differ by one gray level on ~0.26 % of colour pixels and never when B = G = R.
BGR2GRAY_VARIANTS = {"opencv4": (3735, 19235, 9798, 15), "legacy14": (1868, 9617, 4899, 14)}
DEFAULT_BGR2GRAY = "opencv4"


def bgr2gray_variant_of_opencv(version: str) -> str:
    """Which fixed-point variant a given ``cv2.__version__`` computes for 8-bit BGR->gray, FROM THE VERSION STRING ALONE: only valid
    for the range the reference pins (>= 4.6, < 4.12: 15-bit).  The constants did not change at the 3 -> 4 boundary -- the 15-bit
    RGB2Gray landed mid-series (around 4.1.x, and was merged into late 3.4.x) -- so for any other version ask the installed module
    itself: :func:`bgr2gray_variant_of_module`."""
    try:
        major = int(str(version).split(".")[0])
    except ValueError:
        return DEFAULT_BGR2GRAY
    return "opencv4" if major >= 4 else "legacy14"


_PROBE = None


def _probe_pixels() -> np.ndarray:
    """A few BGR pixels on which the two fixed-point variants give different gray levels (found once, deterministically)."""
    global _PROBE
    if _PROBE is None:
        g = np.arange(0, 256, 5, dtype=np.uint8)
        cube = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(1, -1, 3)
        dif = bgr2gray_fixed_point(cube, "opencv4") != bgr2gray_fixed_point(cube, "legacy14")
        _PROBE = np.ascontiguousarray(cube[:, dif[0]][:, :16])
    return _PROBE


def bgr2gray_variant_of_module(cv2_module) -> str:
    """Which variant the INSTALLED OpenCV computes: converts 16 pixels on which the variants differ and looks at the answer
    (works for every version, unlike parsing ``cv2.__version__``).  Raises if it matches neither."""
    px = _probe_pixels()
    got = np.asarray(cv2_module.cvtColor(px, cv2_module.COLOR_BGR2GRAY))
    for name in BGR2GRAY_VARIANTS:
        if np.array_equal(got, bgr2gray_fixed_point(px, name)):
            return name
    raise RuntimeError("the installed OpenCV's 8-bit BGR->gray matches neither the 15-bit nor the 14-bit fixed-point formula")


def bgr2gray_fixed_point(img_bgr: np.ndarray, variant: str = DEFAULT_BGR2GRAY) -> np.ndarray:
    """The integer formula itself (never cv2): what the device kernels ``dcx_bgr2gray`` / ``dcx_bgr2gray_legacy14`` compute."""
    if img_bgr.ndim < 3 or img_bgr.shape[-1] != 3 or img_bgr.dtype != np.uint8:
        raise ValueError("expected a (...,H,W,3) uint8 BGR image")
    cb, cg, cr, shift = BGR2GRAY_VARIANTS[variant]
    acc = img_bgr[..., 0].astype(np.uint32) * np.uint32(cb)
    acc += img_bgr[..., 1].astype(np.uint32) * np.uint32(cg)
    acc += img_bgr[..., 2].astype(np.uint32) * np.uint32(cr)
    acc += np.uint32(1 << (shift - 1))
    acc >>= np.uint32(shift)
    return acc.astype(np.uint8)


def bgr2gray(img_bgr: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(img, cv2.COLOR_BGR2GRAY) (call site /root/reference/src/inference.py:40).

    Uses OpenCV when it is importable (what the reference calls); otherwise the 8-bit fixed-point formula of the OpenCV
    generation the reference pins (4.x): gray = (3735*B + 19235*G + 9798*R + 16384) >> 15.
    """
    cv2 = _opencv()
    if cv2:
        return cv2.cvtColor(img_bgr, cv2.COLOR_BGR2GRAY)
    if img_bgr.ndim != 3:
        raise ValueError("expected a (H,W,3) uint8 BGR image")
    return bgr2gray_fixed_point(img_bgr, DEFAULT_BGR2GRAY)


def bgr2gray_device(bgr, variant: str = DEFAULT_BGR2GRAY):
    """Device version (``dcx_bgr2gray``; ``variant="legacy14"``: ``dcx_bgr2gray_legacy14``): (B,H,W,3) or (H,W,3) uint8 GPU
    tensor -> (B,H,W) / (H,W) uint8 gray on the GPU, the same integer formula as :func:`bgr2gray_fixed_point` (bit-identical;
    the numpy version costs the host more than half of a bs=1 ``infer_image`` call)."""
    import torch
    from . import _lib
    if bgr.device.type != "cuda" or bgr.dtype != torch.uint8 or bgr.shape[-1] != 3 or bgr.ndim not in (3, 4):
        raise ValueError("expected a (B,H,W,3) or (H,W,3) uint8 tensor on the GPU")
    if variant not in BGR2GRAY_VARIANTS:
        raise ValueError(f"unknown BGR->gray variant {variant!r}")
    x = bgr.contiguous()
    b = 1 if x.ndim == 3 else x.shape[0]
    h, w = x.shape[-3], x.shape[-2]
    gray = torch.empty(x.shape[:-1], dtype=torch.uint8, device=x.device)
    fn = _lib.lib().dcx_bgr2gray if variant == "opencv4" else _lib.lib().dcx_bgr2gray_legacy14
    with torch.cuda.device(x.device):
        _lib.check(fn(x.data_ptr(), h * w * 3, w * 3, b, h, w, gray.data_ptr(), _lib.current_stream()), "dcx_bgr2gray")
    return gray
