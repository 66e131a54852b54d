"""HIP-backed mirror of /root/reference/src/models/model_utils.py.

``pre_bgr_image`` stays a host numpy function (it is one in the reference, :46-50); the tensor
functions take GPU tensors and run the library's kernels.  ``corner_sub_pix`` / ``pred_sub_pix``
(cv2.cornerSubPix, training labels only) are out of scope.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from ._handles import check_dev_tensor, require_cuda


def pre_bgr_image(image: np.ndarray) -> np.ndarray:
    """model_utils.py:46-50 (host side, identical arithmetic: float32 subtract, true divide)."""
    image = image[..., np.newaxis].astype(np.float32)
    image = (image - 128) / 255
    return image.transpose((2, 0, 1))


def pre_image_device(gray_u8: torch.Tensor) -> torch.Tensor:
    """Device version of pre_bgr_image: uint8 (...,H,W) on the GPU -> float32 same shape."""
    gray_u8 = check_dev_tensor(gray_u8, gray_u8.device, torch.uint8, "gray")
    out = torch.empty(gray_u8.shape, dtype=torch.float32, device=gray_u8.device)
    with torch.cuda.device(gray_u8.device):
        _lib.check(_lib.lib().dcx_pre_image(gray_u8.data_ptr(), out.data_ptr(), gray_u8.numel(),
                                            _lib.current_stream()), "dcx_pre_image")
    return out


def _decode(loc_hat: torch.Tensor, ids_hat: torch.Tensor, dust_bin_ids: int, want_maps: bool):
    assert loc_hat.ndim == 4 and ids_hat.ndim == 4
    dev = loc_hat.device
    loc_hat = check_dev_tensor(loc_hat, dev, torch.float32, "loc_hat")
    ids_hat = check_dev_tensor(ids_hat, dev, torch.float32, "ids_hat")
    n, n_loc, hc, wc = loc_hat.shape
    n_ids1 = ids_hat.shape[1]
    kmax = hc * wc
    counts = torch.empty((n,), dtype=torch.int32, device=dev)
    rows = torch.empty((n, kmax, 4), dtype=torch.int32, device=dev)
    la = torch.empty((n, hc, wc), dtype=torch.int32, device=dev) if want_maps else None
    ia = torch.empty((n, hc, wc), dtype=torch.int32, device=dev) if want_maps else None
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().dcx_pred_to_keypoints(loc_hat.data_ptr(), ids_hat.data_ptr(), n, n_loc, n_ids1, hc, wc,
                                                    dust_bin_ids, kmax, counts.data_ptr(), rows.data_ptr(),
                                                    _lib.ptr(la), _lib.ptr(ia), _lib.current_stream()),
                   "dcx_pred_to_keypoints")
    return counts, rows, la, ia


def pred_argmax(loc_hat: torch.Tensor, ids_hat: torch.Tensor, dust_bin_ids: int):
    """model_utils.py:53-78 -> (loc_argmax, ids_argmax) int64 (N,Hc,Wc); ids masked by loc == 64."""
    _, _, la, ia = _decode(loc_hat, ids_hat, dust_bin_ids, True)
    return la.to(torch.int64), ia.to(torch.int64)


def pred_to_keypoints(loc_hat: torch.Tensor, ids_hat: torch.Tensor, dust_bin_ids: int):
    """model_utils.py:81-88 -> (kpts (K,2) int64 (x,y), ids (K,) int64), raster order over (n,y,x).

    Like the reference (torch.nonzero, model_utils.py:114) this needs K on the host: one D2H of
    the per-frame counts.  The batched pipeline (inference.infer_batch) avoids that sync.
    """
    counts, rows, _, _ = _decode(loc_hat, ids_hat, dust_bin_ids, False)
    cnt = counts.cpu().tolist()
    parts = [rows[b, :c] for b, c in enumerate(cnt) if c > 0]
    if not parts:
        dev = loc_hat.device
        return (torch.empty((0, 2), dtype=torch.int64, device=dev), torch.empty((0,), dtype=torch.int64, device=dev))
    r = torch.cat(parts, dim=0).to(torch.int64)
    return r[:, 0:2].contiguous(), r[:, 2].contiguous()


def label_to_keypoints(loc: torch.Tensor, ids: torch.Tensor, dust_bin_ids: int):
    """model_utils.py:91-124: class-index maps (N,Hc,Wc) -- what ``pred_argmax`` returns, or a dataset label -- ->
    (kpts (K,2) int64 (x,y), ids (K,) int64) in ``torch.nonzero``'s raster order; ``mask = ids != dust_bin_ids``,
    ``x = 8*ix + loc % 8``, ``y = 8*iy + loc // 8`` (HIP: ``dcx_label_to_keypoints``).

    Where this differs from the reference's function: the maps are processed on the GPU -- CPU tensors (the reference also runs
    this on dataset labels, data.py) are moved to the current GPU and the result comes back on the CPU; class indices must lie in
    [0, 255] (ValueError otherwise: they are packed into one word per cell; the reference's labels are < 65 / <= n_ids);
    a ``dust_bin_ids`` outside [0, 255] can equal no label, so every cell fires, as ``ids != dust_bin_ids`` would say.
    One host synchronisation (the dynamic output shape, as in the reference's ``torch.nonzero``)."""
    assert loc.ndim == 3 and ids.ndim == 3
    if ids.device != loc.device or loc.shape != ids.shape:
        raise ValueError("label_to_keypoints expects two (N,Hc,Wc) label maps on the same device")
    home = loc.device
    dev = home if home.type == "cuda" else require_cuda("cuda")
    loc = loc.to(device=dev, dtype=torch.int64).contiguous()
    ids = ids.to(device=dev, dtype=torch.int64).contiguous()
    n, hc, wc = loc.shape
    kmax = hc * wc
    meta = torch.zeros((n + 1,), dtype=torch.int32, device=dev)          # counts[n] | bad flag: one D2H
    rows = torch.empty((n, kmax, 4), dtype=torch.int32, device=dev)
    codes = torch.empty((n, kmax), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().dcx_label_to_keypoints(loc.data_ptr(), ids.data_ptr(), n, hc, wc, int(dust_bin_ids), kmax, meta.data_ptr(),
                                                     rows.data_ptr(), codes.data_ptr(), meta.data_ptr() + 4 * n, _lib.current_stream()),
                   "dcx_label_to_keypoints")
    m = meta.cpu().tolist()
    cnt = m[:n]
    if m[n]:
        raise ValueError("label_to_keypoints: class indices must lie in [0, 255]")
    parts = [rows[b, :c] for b, c in enumerate(cnt) if c > 0]
    if not parts:
        return (torch.empty((0, 2), dtype=torch.int64, device=home), torch.empty((0,), dtype=torch.int64, device=home))
    r = torch.cat(parts, dim=0).to(device=home, dtype=torch.int64)
    return r[:, 0:2].contiguous(), r[:, 2].contiguous()


def extract_patches(img: torch.Tensor, keypoints: torch.Tensor, patch_size: int = 24) -> torch.Tensor:
    """model_utils.py:19-36: img (1,H,W) normalised f32, keypoints (K,2) int (x,y) -> (K,24,24), zero padded."""
    if patch_size != 24:
        raise ValueError("only the reference's 24x24 patches are supported")
    dev = img.device
    img = check_dev_tensor(img, dev, torch.float32, "img")
    h, w = img.shape[-2:]
    k = keypoints.shape[0]
    patches = torch.empty((k, 24, 24), dtype=torch.float32, device=dev)
    if k == 0:
        return patches
    table = torch.zeros((k, 4), dtype=torch.int32, device=dev)
    table[:, 1:3] = keypoints.to(device=dev, dtype=torch.int32)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().dcx_extract_patches_f32(img.data_ptr(), h, w, table.data_ptr(), None, k,
                                                      patches.data_ptr(), _lib.current_stream()),
                   "dcx_extract_patches_f32")
    return patches


def speedy_bargmax2d(x: torch.Tensor) -> torch.Tensor:
    """model_utils.py:39-43: x (K,h,w) -> (K,2) int64 (col,row) of the first maximum."""
    dev = x.device
    x = check_dev_tensor(x, dev, torch.float32, "x")
    k, h, w = x.shape
    out = torch.empty((k, 2), dtype=torch.int32, device=dev)
    if k:
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().dcx_argmax2d(x.data_ptr(), k, h, w, out.data_ptr(), _lib.current_stream()),
                       "dcx_argmax2d")
    return out.to(torch.int64)
