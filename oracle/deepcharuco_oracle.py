"""ORACLE -- test infrastructure, NOT product code.

CPU restatement (plain torch.nn.functional fp32 + numpy, no Lightning / cv2 /
numba) of the reference's detect+refine path, one function per reference
symbol, each citing the reference file:line it follows.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker / the timed CPU baseline.  Nothing under
``deepcharuco_amd/`` imports it: the product path is the HIP library and fails
loudly when that library is missing.

Pinning: ``oracle/make_golden.py`` (run in the build container, where
``/root/reference`` is mounted) imports the reference's own modules with
stubbed third-party packages and asserts that every function here returns
*identical* tensors (``torch.equal``) to the reference's for the same seeded
weights and frames, then writes the fixtures under ``tests/golden``.  The
reference ships no tests or golden vectors of its own, so
those fixtures -- outputs of the reference itself run in the build container --
are what pins this oracle.  ``cv2.cvtColor`` / ``cv2.solvePnP`` are third-party
(OpenCV, not vendored, cv2 absent here): :func:`bgr2gray` restates OpenCV's
published 8-bit fixed-point formula and is "parity unpinned" for that one step.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

TensorDict = Dict[str, torch.Tensor]


def to_torch_state_dict(sd) -> TensorDict:
    return {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in sd.items()}


# ---------------------------------------------------------------- host pre-processing

# (cB, cG, cR, shift) of OpenCV's 8-bit RGB2Gray, by OpenCV generation.  OpenCV is third-party, un-vendored and absent
# here (parity unpinned for this one step); the reference pins opencv-contrib-python >= 4.6, < 4.12 (requirements.txt:5).
#   opencv4:  imgproc/src/color.hpp  gray_shift = 15, BY15 = 3735, GY15 = 19235, RY15 = 9798   (RGB2Gray<uchar>, 4.x)
#   legacy14: imgproc/src/color.hpp  yuv_shift = 14,  B2Y = 1868,  G2Y = 9617,   R2Y = 4899    (8-bit gray before 4.x; in 4.x only
#             16-bit images and YUV)
BGR2GRAY_VARIANTS = {"opencv4": (3735, 19235, 9798, 15), "legacy14": (1868, 9617, 4899, 14)}


def bgr2gray(img_bgr: np.ndarray, variant: str = "opencv4") -> np.ndarray:
    """cv2.cvtColor(img, COLOR_BGR2GRAY) call at inference.py:40.

    OpenCV 8-bit path, fixed point: gray = (B*cB + G*cG + R*cR + (1 << (shift-1))) >> shift; default = the 4.x
    constants the reference's requirements pin: (B*3735 + G*19235 + R*9798 + 16384) >> 15.
    """
    cb, cg, cr, shift = BGR2GRAY_VARIANTS[variant]
    b = img_bgr[..., 0].astype(np.int32)
    g = img_bgr[..., 1].astype(np.int32)
    r = img_bgr[..., 2].astype(np.int32)
    return ((b * cb + g * cg + r * cr + (1 << (shift - 1))) >> shift).astype(np.uint8)


def pre_bgr_image(image: np.ndarray) -> np.ndarray:
    """model_utils.py:46-50 -- (g - 128) / 255 in float32, true division, add channel axis."""
    image = image[..., np.newaxis].astype(np.float32)
    image = (image - 128) / 255
    return image.transpose((2, 0, 1))


# ---------------------------------------------------------------- networks

def _cbr(x: torch.Tensor, sd: TensorDict, conv: str, bn: str, pad: int) -> torch.Tensor:
    """conv3x3 (+bias) -> BatchNorm2d(eval, eps 1e-5) -> ReLU  (net.py:60, refinenet.py:56)."""
    x = F.conv2d(x, sd[f"{conv}.weight"], sd[f"{conv}.bias"], stride=1, padding=pad)
    x = F.batch_norm(x, sd[f"{bn}.running_mean"], sd[f"{bn}.running_var"],
                     sd[f"{bn}.weight"], sd[f"{bn}.bias"], training=False, momentum=0.0, eps=1e-5)
    return F.relu(x)


def detector_forward(sd: TensorDict, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """dcModel.forward net.py:50-80. x (N,1,H,W) f32 -> loc (N,65,H/8,W/8), ids (N,n_ids+1,H/8,W/8)."""
    with torch.no_grad():
        x = _cbr(x, sd, "conv1a", "bn1a", 1)
        x = _cbr(x, sd, "conv1b", "bn1b", 1)
        x = F.max_pool2d(x, 2, 2)
        x = _cbr(x, sd, "conv2a", "bn2a", 1)
        x = _cbr(x, sd, "conv2b", "bn2b", 1)
        x = F.max_pool2d(x, 2, 2)
        x = _cbr(x, sd, "conv3a", "bn3a", 1)
        x = _cbr(x, sd, "conv3b", "bn3b", 1)
        x = F.max_pool2d(x, 2, 2)
        x = _cbr(x, sd, "conv4a", "bn4a", 1)
        x = _cbr(x, sd, "conv4b", "bn4b", 1)
        cpa = _cbr(x, sd, "convPa", "bnPa", 1)
        loc = F.conv2d(cpa, sd["convPb.weight"], sd["convPb.bias"])   # no activation net.py:74
        cda = _cbr(x, sd, "convDa", "bnDa", 1)
        ids = F.conv2d(cda, sd["convDb.weight"], sd["convDb.bias"])   # no activation net.py:77
    return loc, ids


def detector_infer_image(sd: TensorDict, img: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """dcModel.infer_image net.py:82-99: img (1,H,W) -> forward(img[None]) -> (loc, ids)."""
    return detector_forward(sd, img[None])


def refinenet_forward(sd: TensorDict, x: torch.Tensor) -> torch.Tensor:
    """RefineNet.forward refinenet.py:49-83. x (K,1,24,24) -> (K,1,64,64)."""
    with torch.no_grad():
        x = _cbr(x, sd, "conv1a", "bn1a", 0)
        x = _cbr(x, sd, "conv1b", "bn1b", 0)
        x = _cbr(x, sd, "conv2a", "bn2a", 0)
        x = _cbr(x, sd, "conv2b", "bn2b", 0)
        x = F.max_pool2d(x, 2, 2)
        x = _cbr(x, sd, "conv3a", "bn3a", 1)
        x = _cbr(x, sd, "conv3b", "bn3b", 1)
        x = F.interpolate(x, scale_factor=2, mode="nearest")   # UpsamplingNearest2d refinenet.py:19
        x = _cbr(x, sd, "conv4a", "bn4a", 1)
        x = _cbr(x, sd, "conv4b", "bn4b", 1)
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        x = _cbr(x, sd, "conv5a", "bn5a", 1)
        x = _cbr(x, sd, "conv5b", "bn5b", 1)
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        x = _cbr(x, sd, "convPa", "bnPa", 1)
        x = F.conv2d(x, sd["convPb.weight"], sd["convPb.bias"])
    return x


# ---------------------------------------------------------------- post-processing

def pred_argmax(loc_hat: torch.Tensor, ids_hat: torch.Tensor, dust_bin_ids: int):
    """model_utils.py:53-78. argmax over channels (first max wins), loc dust-bin (64) masks ids."""
    ids_argmax = torch.argmax(ids_hat, dim=1)
    loc_argmax = torch.argmax(loc_hat, dim=1)
    ids_argmax = torch.where(loc_argmax == 64, dust_bin_ids, ids_argmax)
    return loc_argmax, ids_argmax


def label_to_keypoints(loc: torch.Tensor, ids: torch.Tensor, dust_bin_ids: int):
    """model_utils.py:91-124. Raster-order (n,y,x) nonzero; xs = 8*cx + loc%8, ys = 8*cy + loc//8."""
    assert loc.ndim == 3 and ids.ndim == 3
    mask = ids != dust_bin_ids
    indices = torch.nonzero(mask, as_tuple=False)
    ids_found = ids[mask]
    region_pixel = loc[mask]
    xs = 8 * indices[:, -1] + (region_pixel % 8)
    ys = 8 * indices[:, -2] + torch.div(region_pixel, 8, rounding_mode="floor")
    return torch.stack((xs, ys), dim=1), ids_found


def pred_to_keypoints(loc_hat: torch.Tensor, ids_hat: torch.Tensor, dust_bin_ids: int):
    """model_utils.py:81-88."""
    assert loc_hat.ndim == 4 and ids_hat.ndim == 4
    la, ia = pred_argmax(loc_hat, ids_hat, dust_bin_ids)
    return label_to_keypoints(la, ia, dust_bin_ids)


def extract_patches(img: torch.Tensor, keypoints: torch.Tensor, patch_size: int = 24) -> torch.Tensor:
    """model_utils.py:19-36. patch[k,i,j] = img[ky-12+i, kx-12+j], 0.0 outside the image."""
    pad = patch_size // 2
    padded = F.pad(img.squeeze(0), (pad, pad, pad, pad), mode="constant", value=0)
    k = keypoints.shape[0]
    ar = torch.arange(patch_size)
    ys = keypoints[:, 1, None] + ar            # (K,P) rows in padded coords
    xs = keypoints[:, 0, None] + ar            # (K,P)
    return padded[ys[:, :, None], xs[:, None, :]].reshape(k, patch_size, patch_size)


def speedy_bargmax2d(x: torch.Tensor) -> torch.Tensor:
    """model_utils.py:39-43. Flat argmax (first max) over the last two dims -> (col,row)."""
    _, idx = torch.max(x.reshape(x.shape[0], -1), dim=1)
    return torch.stack((idx % x.shape[2], idx // x.shape[2]), dim=1)


def refinenet_infer_patches(sd: TensorDict, patches: torch.Tensor, keypoints: torch.Tensor):
    """RefineNet.infer_patches refinenet.py:85-115."""
    assert patches.shape[-2:] == (24, 24)
    if patches.ndim == 3:
        patches = patches.unsqueeze(1)
    loc_hat = refinenet_forward(sd, patches)[:, 0]
    corners = speedy_bargmax2d(loc_hat)
    corners_og = (corners - 32) / 8 + keypoints
    return corners_og, corners


# ---------------------------------------------------------------- glue

def infer_image(img_bgr: np.ndarray, dust_bin_ids: int, sd_dc: TensorDict,
                sd_rn: Optional[TensorDict] = None, gray: Optional[np.ndarray] = None):
    """infer_image inference.py:32-70 (draw_pred=False).

    Returns (K,3) [x,y,id] sorted by id (stable); float64 with RefineNet, int64
    without; ``np.array([])`` when no corner fires (inference.py:51-52).
    ``gray`` lets callers skip the BGR->gray step (measured configs start from
    grayscale frames).
    """
    if gray is None:
        gray = bgr2gray(img_bgr)
    img_gray = torch.tensor(pre_bgr_image(gray))
    loc_hat, ids_hat = detector_infer_image(sd_dc, img_gray)
    kpts, ids_found = pred_to_keypoints(loc_hat, ids_hat, dust_bin_ids)
    if ids_found.shape[0] == 0:
        return np.array([])
    if sd_rn is not None:
        patches = extract_patches(img_gray, kpts)
        kpts, _ = refinenet_infer_patches(sd_rn, patches, kpts)
    kpts = kpts.numpy()
    ids_np = ids_found.numpy()
    return np.array([[k[0], k[1], idx] for k, idx in sorted(zip(kpts, ids_np), key=lambda t: t[1])])


def keypoint_confidences(loc_hat: torch.Tensor, ids_hat: torch.Tensor, dust_bin_ids: int) -> torch.Tensor:
    """Confidences of the key-points ``pred_to_keypoints`` returns, in ITS order (torch.nonzero's raster order): (K,2) float32
    [softmax(loc)[arg-max], softmax(ids)[arg-max]] per firing cell.

    The reference's ``pred_to_keypoints`` (models/model_utils.py:81-88) promises "optionally confidences" in its docstring but
    returns none, and nothing in the reference thresholds on a score; its heads are trained with cross-entropy on exactly these
    logits (models/net.py:136-137), so the probability CE assigns -- ``torch.softmax(logits, dim=1)`` at the arg-max class -- is
    the only confidence the reference's own arithmetic defines.  This is that expression in stock torch on the reference's
    logits; there is no reference output to pin it to beyond the logits themselves (which ARE pinned, make_golden.py)."""
    assert loc_hat.ndim == 4 and ids_hat.ndim == 4
    loc_argmax, ids_argmax = pred_argmax(loc_hat, ids_hat, dust_bin_ids)
    mask = ids_argmax != dust_bin_ids
    p_loc = torch.softmax(loc_hat, dim=1).max(dim=1).values
    p_ids = torch.softmax(ids_hat, dim=1).max(dim=1).values
    return torch.stack([p_loc[mask], p_ids[mask]], dim=1)


def top2_margin(logits: torch.Tensor) -> torch.Tensor:
    """Per-position gap between the largest and second-largest channel (near-tie policy, H1)."""
    top = torch.topk(logits, 2, dim=1).values
    return top[:, 0] - top[:, 1]


def solve_pnp_object_points(keypoints: np.ndarray, col_count: int, row_count: int, square_len: float):
    """Object/image point construction of solve_pnp inference.py:15-26 (everything before cv2.solvePnP)."""
    inn_rc = np.arange(1, row_count)
    inn_cc = np.arange(1, col_count)
    object_points = np.zeros(((col_count - 1) * (row_count - 1), 3), np.float32)
    object_points[:, :2] = np.array(np.meshgrid(inn_rc, inn_cc)).reshape((2, -1)).T * square_len
    image_points = keypoints[:, :2].astype(np.float32)
    return object_points[keypoints[:, 2].astype(int)], image_points
