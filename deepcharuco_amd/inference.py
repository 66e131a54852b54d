"""Drop-in for /root/reference/src/inference.py: ``load_models`` (:73-84), ``infer_image`` (:32-70),
``solve_pnp`` (:15-29) with the reference's signatures, plus the batched ``infer_batch``.

``infer_image`` runs the library's sync-free batch pipeline with B = 1 (one H2D of the gray frame,
one D2H of the packed corner list).  ``infer_image_staged`` follows the reference statement by
statement through the mirrored ``models.*`` functions (same intermediate tensors, same host syncs)
and exists so every mirrored function is exercised end to end; both return identical arrays.
"""
from __future__ import annotations

import ctypes as _ctypes
import warnings
from typing import List, Optional

import numpy as np
import torch

from . import _lib
from .imgproc import _opencv, bgr2gray
from .models._handles import require_cuda
from .models.model_utils import extract_patches, pre_bgr_image, pred_to_keypoints
from .models.net import dcModel, lModel
from .models.refinenet import RefineNet, lRefineNet

__all__ = ["load_models", "infer_image", "infer_image_staged", "infer_batch", "infer_batch_device", "unpack_results", "packed_len",
           "solve_pnp", "solve_pnp_batch", "solve_pnp_submit", "set_deterministic", "InferenceModel", "calibrate_xcd",
           "set_xcd_weights", "get_xcd_weights"]

DEFAULT_KMAX = 64


def load_models(deepc_ckpt: str, refinenet_ckpt: Optional[str] = None, n_ids: int = 16, device="cuda"):
    """inference.py:73-84: checkpoints -> (deepc, refinenet | None), weights resident on ``device``."""
    deepc = lModel.load_from_checkpoint(deepc_ckpt, dcModel=dcModel(n_ids))
    deepc.eval()
    deepc.to(device)
    refinenet = None
    if refinenet_ckpt is not None:
        refinenet = lRefineNet.load_from_checkpoint(refinenet_ckpt, refinenet=RefineNet())
        refinenet.eval()
        refinenet.to(device)
    return deepc, refinenet


def _cv2_solvepnp():
    try:
        import cv2  # type: ignore
    except ImportError as e:  # pragma: no cover - cv2 is absent in the build image
        raise ImportError("solve_pnp needs OpenCV (cv2.solvePnP), which is not installed") from e
    return cv2.solvePnP


def _pnp_points(keypoints, col_count, row_count, square_len):
    """inference.py:20-26: board-frame object points of the detected ids + their image points."""
    inn_rc = np.arange(1, row_count)
    inn_cc = np.arange(1, col_count)
    object_points = np.zeros(((col_count - 1) * (row_count - 1), 3), np.float32)
    object_points[:, :2] = np.array(np.meshgrid(inn_rc, inn_cc)).reshape((2, -1)).T * square_len
    image_points = keypoints[:, :2].astype(np.float32)
    return object_points[keypoints[:, 2].astype(int)], image_points


def solve_pnp(keypoints, col_count, row_count, square_len, camera_matrix, dist_coeffs):
    """inference.py:15-29. Host side (cv2.solvePnP), as in the reference."""
    if keypoints.shape[0] < 4:
        return False, None, None
    object_points_found, image_points = _pnp_points(keypoints, col_count, row_count, square_len)
    return _cv2_solvepnp()(object_points_found, image_points, camera_matrix, dist_coeffs)


_pnp_pools: dict = {}


def _pool(workers: Optional[int]):
    """Process-wide thread pools for the PnP stage (cv2 releases the GIL inside solvePnP), one per worker count."""
    import concurrent.futures as cf
    import os
    n = int(workers) if workers else min(32, os.cpu_count() or 1)
    pool = _pnp_pools.get(n)
    if pool is None:
        pool = _pnp_pools[n] = cf.ThreadPoolExecutor(max_workers=n, thread_name_prefix="dcx-pnp")
    return pool


def solve_pnp_submit(keypoints_list, col_count, row_count, square_len, camera_matrix, dist_coeffs, workers: Optional[int] = None):
    """Asynchronous half of :func:`solve_pnp_batch`: one future per frame (frames with < 4 corners resolve immediately
    to ``(False, None, None)`` like inference.py:16-17).  Raises ImportError at once when OpenCV is missing."""
    import concurrent.futures as cf
    fn = _cv2_solvepnp()
    pool = _pool(workers)
    futs = []
    for kp in keypoints_list:
        kp = np.asarray(kp)
        if kp.ndim != 2 or kp.shape[0] < 4:
            f = cf.Future()
            f.set_result((False, None, None))
        else:
            obj, img = _pnp_points(kp, col_count, row_count, square_len)
            f = pool.submit(fn, obj, img, camera_matrix, dist_coeffs)
        futs.append(f)
    return futs


def solve_pnp_batch(keypoints_list, col_count, row_count, square_len, camera_matrix, dist_coeffs, workers: Optional[int] = None):
    """``solve_pnp`` (inference.py:15-29) for every frame of an ``infer_batch`` result, on host threads, so that the
    one-call-per-frame host stage of the reference's callers (pose_estimation.py:58-63) does not cap a multi-GPU batch
    path: at 8 k frames/s per GPU a serial ~50 us solvePnP per frame would already eat 40 % of one core per GPU.
    Returns a list of ``(ret, rvec, tvec)`` in frame order, each identical to ``solve_pnp`` on that frame."""
    return [f.result() for f in solve_pnp_submit(keypoints_list, col_count, row_count, square_len, camera_matrix,
                                                dist_coeffs, workers)]


# ------------------------------------------------------------------------------------------------

def set_deterministic(enabled: bool = True) -> None:
    """Force the direct convolution kernels for every layer: logits become bit-identical across batch sizes and devices
    (``dcx_set_deterministic``).  Process-global."""
    _lib.check(_lib.lib().dcx_set_deterministic(1 if enabled else 0), "dcx_set_deterministic")
    from .graph import clear_graph_cache
    clear_graph_cache()          # captured graphs froze the kernels chosen under the previous mode


def calibrate_xcd(device="cuda", rounds: int = 8) -> List[float]:
    """Measure the eight XCDs of ``device`` and re-weight the convolution kernels' per-XCD item shares (``dcx_calibrate_xcd``).

    The XCDs of one MI355X run this load 3-6 % apart and a launch ends with its slowest XCD; with shares proportional to the
    measured speeds they finish together.  Speed only -- work items and bits are unchanged.  Set-up code: synchronous (~10 ms,
    0.8 GB of scratch allocated and freed), call it once the GPU is warm (after a few batches), never inside a timed region.
    Graphs captured before keep their old shares, so the graph cache is cleared.  Returns the weights (1.0 = an equal share)."""
    dev = require_cuda(device)
    w = (_ctypes.c_float * 8)()
    with torch.cuda.device(dev):
        torch.cuda.synchronize()
        _lib.check(_lib.lib().dcx_calibrate_xcd(int(rounds), w, _lib.current_stream()), "dcx_calibrate_xcd")
    from .graph import clear_graph_cache
    clear_graph_cache()
    return [float(v) for v in w]


def set_xcd_weights(weights=None, device="cuda") -> None:
    """Per-XCD item shares by hand: 8 relative speeds, ``None`` = equal (``dcx_set_xcd_weights``)."""
    dev = require_cuda(device)
    arr = None if weights is None else (_ctypes.c_float * 8)(*[float(v) for v in weights])
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().dcx_set_xcd_weights(arr), "dcx_set_xcd_weights")
    from .graph import clear_graph_cache
    clear_graph_cache()


def get_xcd_weights(device="cuda") -> List[float]:
    dev = require_cuda(device)
    w = (_ctypes.c_float * 8)()
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().dcx_get_xcd_weights(w), "dcx_get_xcd_weights")
    return [float(v) for v in w]


def _unwrap(deepc, refinenet):
    det = deepc.model if hasattr(deepc, "model") else deepc
    ref = None
    if refinenet is not None:
        ref = refinenet.model if hasattr(refinenet, "model") else refinenet
    return det, ref


PIXEL_FORMATS = {"gray": 0, "opencv4": 1, "legacy14": 2}       # DCX_PIX_GRAY8 / DCX_PIX_BGR8 / DCX_PIX_BGR8_LEGACY14


def packed_len(batch: int, pool: int, conf: bool = False) -> int:
    """int32 words of the packed result of a batch: counts[B] | starts[B] | rows[pool][4] | xy[pool][2] (| conf[pool][2])."""
    return 2 * batch + (8 if conf else 6) * pool


def launch_pipeline(det, ref, frames_ptr: int, b: int, h: int, w: int, bpp: int, pix: int, dust_bin_ids: int, pool: int,
                    ws: torch.Tensor, out_ptr: int, conf: bool = False) -> None:
    """``dcx_infer_batch`` on raw pointers (current device / stream): dense frames at ``frames_ptr``, the packed result
    (``packed_len(b, pool, conf)`` int32 words) at ``out_ptr``.  The pointers only have to be DEVICE-ACCESSIBLE: pinned host
    memory qualifies, which is how the bs=1 hipGraph reads the frame and writes the corner list without copy nodes."""
    counts_p, starts_p, rows_p = out_ptr, out_ptr + 4 * b, out_ptr + 8 * b
    xy_p = rows_p + 16 * pool
    conf_p = xy_p + 8 * pool
    _lib.check(_lib.lib().dcx_infer_batch(det.handle, ref.handle if ref else None, frames_ptr, h * w * bpp, w * bpp, pix, b, h, w,
                                          dust_bin_ids, pool, ws.data_ptr(), ws.numel(), counts_p, starts_p, rows_p,
                                          xy_p if ref else None, conf_p if conf else None, _lib.current_stream()), "dcx_infer_batch")


def infer_batch_device(frames: torch.Tensor, dust_bin_ids: int, deepc, refinenet=None, kmax: int = DEFAULT_KMAX,
                       out: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None, pool: Optional[int] = None,
                       conf: bool = False, bgr_variant: str = "opencv4") -> torch.Tensor:
    """Enqueue detect+refine for a batch of GPU-resident frames; no host synchronisation.

    frames: (B,H,W) uint8 gray, or (B,H,W,3) uint8 BGR (what the reference's callers hold, pose_estimation.py:53-59; the
    ``cv2.cvtColor`` of inference.py:40 then happens inside the first layer's load, ``bgr_variant`` = which fixed-point
    constants, see imgproc.py), contiguous, on the model's GPU.

    The result is the batch's CORNER POOL: like the reference (inference.py:51-57) every firing cell of a frame is refined --
    there is no per-frame cap; the only capacity is the pool of the whole batch, ``pool`` slots (default ``B * kmax``: ``kmax``
    is the AVERAGE number of corners per frame the buffers are sized for).  One flat int32 tensor on the GPU, one allocation so
    that one D2H (or one all-gather) moves everything:
        [0, B)                counts[b]   firing cells of frame b (never truncated)
        [B, 2B)               starts[b]   first pool slot of frame b (frames sit in the pool in the order they finished)
        [2B, 2B + 4 pool)     rows[p]     (x, y, id, cell), frame b's in raster order at p = starts[b] + k
        [.., + 2 pool)        xy[p]       refined (x, y) as float32 bit patterns (if refinenet)
        [.., + 2 pool)        conf[p]     (``conf=True``) soft-max probability of the winning loc / ids class, float32 bits
    Slots >= pool are dropped; ``sum(counts) > pool`` tells the caller to re-run with a larger pool.  Use ``unpack_results``.

    Scratch memory: by default a buffer owned by the detector object and keyed by the current HIP stream (so several
    streams / threads may drive one model pair concurrently, each on its own stream); pass ``ws`` (uint8 GPU tensor of
    at least ``dcx_pipeline_workspace_bytes`` bytes) to manage it yourself.  ``out`` (optional) must be a contiguous
    int32 tensor of exactly ``packed_len(B, pool, conf)`` elements on the model's GPU.
    """
    det, ref = _unwrap(deepc, refinenet)
    dev = det.device
    if (not isinstance(frames, torch.Tensor) or frames.device != dev or frames.dtype != torch.uint8 or not frames.is_contiguous()
            or not (frames.ndim == 3 or (frames.ndim == 4 and frames.shape[3] == 3))):
        raise ValueError("frames must be a contiguous (B,H,W) gray or (B,H,W,3) BGR uint8 tensor on the model's GPU")
    b, h, w = frames.shape[:3]
    bpp = 1 if frames.ndim == 3 else 3
    if bgr_variant not in ("opencv4", "legacy14"):
        raise ValueError(f"unknown BGR->gray variant {bgr_variant!r}")
    pix = PIXEL_FORMATS["gray" if bpp == 1 else bgr_variant]
    if pool is None:
        pool = b * kmax
    if pool <= 0:
        raise ValueError("pool must be positive")
    L = _lib.lib()
    with torch.cuda.device(dev):
        nbytes = L.dcx_pipeline_workspace_bytes(det.handle, ref.handle if ref else None, b, h, w, pool)
        if nbytes == 0:
            raise ValueError("bad batch/shape for dcx_pipeline_workspace_bytes")
        if ws is None:
            ws = det._ws.get("pipe", dev, nbytes)
        elif ws.device != dev or ws.dtype != torch.uint8 or not ws.is_contiguous() or ws.numel() < nbytes:
            raise ValueError(f"ws must be a contiguous uint8 tensor of >= {nbytes} bytes on {dev}")
        n_i32 = packed_len(b, pool, conf)
        if out is None:
            out = torch.empty((n_i32,), dtype=torch.int32, device=dev)
        elif out.device != dev or out.dtype != torch.int32 or out.numel() != n_i32 or not out.is_contiguous():
            raise ValueError(f"out must be a contiguous int32 tensor of {n_i32} elements on {dev}")
        launch_pipeline(det, ref, frames.data_ptr(), b, h, w, bpp, pix, dust_bin_ids, pool, ws, out.data_ptr(), conf)
    return out


def unpack_results(packed: np.ndarray, batch: int, pool: int, refined: bool, conf: bool = False):
    """Host unpack of ``infer_batch_device``'s buffer -> per-frame arrays in infer_image's format.

    Per frame: (K,3) rows [x, y, id] sorted by id (stable w.r.t. raster order, inference.py:68-69);
    float64 when refined, int64 otherwise; ``np.array([])`` when K == 0 (inference.py:51-52).  A frame whose corners did not
    all fit the pool (only possible when ``counts.sum() > pool``) is returned as ``None``: the caller re-runs with a larger
    pool.  Also returns the raw counts; with ``conf=True`` a third value, the per-frame (K,2) float32 arrays
    [p_loc, p_ids] in the same (id-sorted) order."""
    packed = np.asarray(packed, dtype=np.int32)
    counts = packed[:batch]
    head = packed[:2 * batch].tolist()          # counts, starts as Python ints (the per-frame loop below is host-latency code: bs=1 calls)
    rows = packed[2 * batch:2 * batch + 4 * pool].reshape(pool, 4)
    xy = packed[2 * batch + 4 * pool:2 * batch + 6 * pool].view(np.float32).reshape(pool, 2)
    cf = packed[2 * batch + 6 * pool:2 * batch + 8 * pool].view(np.float32).reshape(pool, 2) if conf else None
    res: List[Optional[np.ndarray]] = []
    confs: List[Optional[np.ndarray]] = []
    for b in range(batch):
        k, s0 = head[b], head[batch + b]
        if k == 0:
            res.append(np.array([]))
            confs.append(np.zeros((0, 2), np.float32))
            continue
        if s0 + k > pool:
            res.append(None)
            confs.append(None)
            continue
        rb = rows[s0:s0 + k]
        order = rb[:, 2].argsort(kind="stable")
        a = np.empty((k, 3), np.float64 if refined else np.int64)
        a[:, 0:2] = xy[s0:s0 + k] if refined else rb[:, 0:2]          # (the assignment widens: float32 -> float64 / int32 -> int64)
        a[:, 2] = rb[:, 2]
        res.append(a[order])
        if conf:
            confs.append(cf[s0:s0 + k][order].copy())
    if conf:
        return res, counts.copy(), confs
    return res, counts.copy()


def _to_device_frames(frames, dev):
    """(B,H,W) gray / (B,H,W,3) BGR uint8, host array or GPU tensor -> contiguous GPU tensor."""
    if isinstance(frames, torch.Tensor):       # already on the GPU
        d = frames
    else:
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        d = torch.from_numpy(frames)
    if d.dtype != torch.uint8 or not (d.ndim == 3 or (d.ndim == 4 and d.shape[3] == 3)):
        raise ValueError("expected (B,H,W) uint8 gray frames or (B,H,W,3) uint8 BGR frames")
    return d.to(dev).contiguous()


def infer_batch(frames, dust_bin_ids: int, deepc, refinenet=None, kmax: int = DEFAULT_KMAX, pool: Optional[int] = None,
                conf: bool = False, bgr_variant: str = "opencv4"):
    """Batched infer_image: (B,H,W) uint8 gray frames or (B,H,W,3) uint8 BGR frames (host array or GPU tensor) -> list of B
    keypoint arrays (with ``conf=True``: ``(keypoints, confidences)``, per frame a (K,2) float32 array [p_loc, p_ids]).

    Frames are independent (the reference has no cross-frame state) and the kernel family of every layer depends on the
    layer only, so frame b's corners are bit-identical to ``infer_image`` on that frame alone, whatever the
    batch size.  Every firing cell of every frame is refined, as in the reference (inference.py:51-57): a frame may fire any
    number of cells; only if the WHOLE batch fires more than ``pool`` (default ``B * kmax``) cells is it run a second time, with
    a pool of exactly the size the first pass reported (never silently truncated).  BGR frames are converted on the device with
    the fixed-point formula of the OpenCV generation the reference pins (``bgr_variant``, see imgproc.py)."""
    det, _ = _unwrap(deepc, refinenet)
    d_frames = _to_device_frames(frames, det.device)
    b = d_frames.shape[0]
    if pool is None:
        pool = b * kmax
    while True:
        packed = infer_batch_device(d_frames, dust_bin_ids, deepc, refinenet, pool=pool, conf=conf, bgr_variant=bgr_variant).cpu().numpy()
        out = unpack_results(packed, b, pool, refinenet is not None, conf)
        need = int(out[1].astype(np.int64).sum())
        if need <= pool:
            return (out[0], out[2]) if conf else out[0]
        warnings.warn(f"the batch produced {need} corners > pool={pool}; re-running with pool={need}")
        pool = need


_graph_state = {"enabled": None, "warned": False}
_graph_mod = [None]          # deepcharuco_amd.graph, imported on first use
_cuda_seen = [False]         # require_cuda("cuda") has passed once in this process


def _graphs_enabled() -> bool:
    if _graph_state["enabled"] is None:
        import os
        _graph_state["enabled"] = os.environ.get("DCX_GRAPH", "1") not in ("", "0")
    return bool(_graph_state["enabled"])


def infer_image(img: np.ndarray, dust_bin_ids: int, deepc, refinenet=None, draw_pred: bool = False, device="cuda"):
    """inference.py:32-70. BGR uint8 (H,W,3) -> (keypoints (K,3) [x,y,id] sorted by id, image).

    One frame per call is the reference's own protocol (benchmark.py:37-53), so this path is built for call latency:
    the whole call -- upload, BGR->gray, both networks, decode, download -- is one hipGraph replay per image shape
    (``graph.GraphedPipeline``; ``DCX_GRAPH=0`` keeps the eager launches).  BGR->gray runs through OpenCV on the host when
    cv2 is importable (what the reference calls), otherwise on the device, inside the first layer's load, with the fixed-point
    formula of the OpenCV generation the reference pins.

    ``draw_pred=True`` is a debugging aid and a SLOW path: the reference also draws the UNREFINED detections, which only the
    staged path (``infer_image_staged``: host BGR->gray, no graph, the reference's three host syncs) has at hand; drawing needs
    OpenCV (ImportError without it, unless the frame has no detections: then the copy is returned undrawn, as the reference's
    helper would)."""
    if device != "cuda":           # ("cuda" = the current device; anything else is checked and resolved)
        require_cuda(device)
    elif not _cuda_seen[0]:
        require_cuda(device)
        _cuda_seen[0] = True
    if not isinstance(img, np.ndarray) or img.ndim != 3 or img.shape[2] != 3 or img.dtype != np.uint8:
        raise ValueError("expected a (H,W,3) uint8 BGR image")
    if draw_pred:      # debugging aid: the reference draws the UNREFINED detections too, which only the staged path has at hand
        return infer_image_staged(img, dust_bin_ids, deepc, refinenet, True, device)
    keypoints = None
    frame = bgr2gray(img)[None] if _opencv() else img[None]       # cv2 on the host (what the reference calls), else on the device
    G = _graph_mod[0]
    if G is None:
        from . import graph as G       # (late: graph.py imports this module)
        _graph_mod[0] = G
    graphs_usable, cached_pipeline = G.graphs_usable, G.cached_pipeline
    if _graphs_enabled() and graphs_usable():
        try:
            for attempt in (0, 1):
                pipe = cached_pipeline(dust_bin_ids, deepc, refinenet, img.shape[0], img.shape[1], bgr=frame.ndim == 4)
                try:
                    keypoints = pipe.run(frame)[0]
                    break
                except ReferenceError:
                    # another thread retired this pipeline between the lookup and the replay (a model of the pair was reloaded or
                    # the cache was cleared): look it up / capture it once more, then fall through to the eager launches
                    continue
        except (_lib.DcxError, ValueError):
            raise
        except RuntimeError as e:      # graph capture unavailable: same kernels, launched eagerly
            if not _graph_state["warned"]:
                warnings.warn(f"hipGraph capture failed ({e}); falling back to eager kernel launches")
                _graph_state["warned"] = True
            _graph_state["enabled"] = False
    if keypoints is None:
        keypoints = infer_batch(frame, dust_bin_ids, deepc, refinenet)[0]
    return keypoints, img


def infer_image_staged(img: np.ndarray, dust_bin_ids: int, deepc, refinenet=None, draw_pred: bool = False,
                       device="cuda"):
    """The reference's body (inference.py:40-70) statement by statement on the mirrored functions, including its two drawing
    steps: the detector's key-points in red (radius 3, with ids) before the K = 0 early-out, the refined corners in yellow
    (radius 1) after RefineNet -- each on a copy, so ``draw_pred=True`` never touches the caller's array."""
    require_cuda(device)
    img_gray = bgr2gray(img)
    img_gray = pre_bgr_image(img_gray)
    img_gray = torch.tensor(img_gray, device=device)
    loc_hat, ids_hat = deepc.infer_image(img_gray)
    keypoints, ids_found = pred_to_keypoints(loc_hat, ids_hat, dust_bin_ids)
    if draw_pred:      # nothing to draw on a frame without detections: the reference's helper returns the copy (aruco_utils.py:175)
        img = (draw_inner_corners(img, keypoints.cpu().numpy(), ids_found.cpu().numpy(), radius=3, draw_ids=True, color=(0, 0, 255))
               if ids_found.shape[0] else img.copy())
    if ids_found.shape[0] == 0:
        return np.array([]), img
    if refinenet is not None:
        patches = extract_patches(img_gray, keypoints)
        keypoints, _ = refinenet.infer_patches(patches, keypoints)
    keypoints = keypoints.cpu().numpy()
    ids_found = ids_found.cpu().numpy()
    if draw_pred and refinenet is not None:
        img = draw_inner_corners(img, keypoints, ids_found, draw_ids=False, radius=1, color=(0, 255, 255))
    keypoints = np.array([[k[0], k[1], idx] for k, idx in sorted(zip(keypoints, ids_found), key=lambda x: x[1])])
    return keypoints, img


def draw_inner_corners(img: np.ndarray, corners: np.ndarray, ids: np.ndarray, draw_ids: bool = False, radius: int = 2,
                       color=(0, 0, 255)) -> np.ndarray:
    """What ``infer_image(draw_pred=True)`` draws with (the reference's helper of the same name, aruco_utils.py:135-192): on a COPY
    of the 3-channel image, a thin circle at every corner rounded to the nearest pixel (corners beyond the right / bottom edge are
    skipped) and, if asked, the id in green beside it.  Needs OpenCV (host-side drawing is not part of the GPU path)."""
    try:
        import cv2  # type: ignore
    except ImportError as e:
        raise ImportError("draw_pred=True needs OpenCV for drawing, which is not installed") from e
    assert img.ndim == 3 and img.shape[-1] == 3
    out = img.copy()
    font, thickness = cv2.FONT_HERSHEY_COMPLEX_SMALL, 1
    for corner, idx in zip(corners, ids):
        c = np.round(corner).astype(int)
        if c[0] > out.shape[1] or c[1] > out.shape[0]:
            continue
        cv2.circle(out, (int(c[0]), int(c[1])), radius=radius, color=color, thickness=thickness)
        if draw_ids:
            (tw, th), _ = cv2.getTextSize(str(idx), font, .5, thickness)
            cv2.putText(out, str(idx), (int(c[0]) - tw // 2 - 7, int(c[1]) + th // 2 - 3), font, .45, (0, 255, 0), thickness)
    return out


class InferenceModel:
    """Convenience holder (the north-star's "InferenceModel"): both nets + device, reference call shapes."""

    def __init__(self, deepc_ckpt: str, refinenet_ckpt: Optional[str] = None, n_ids: int = 16, device="cuda"):
        self.n_ids = n_ids
        self.device = device
        self.deepc, self.refinenet = load_models(deepc_ckpt, refinenet_ckpt, n_ids, device)

    def infer_image(self, img: np.ndarray, draw_pred: bool = False):
        return infer_image(img, self.n_ids, self.deepc, self.refinenet, draw_pred, self.device)

    def infer_batch(self, frames: np.ndarray, kmax: int = DEFAULT_KMAX, conf: bool = False):
        """(B,H,W) gray or (B,H,W,3) BGR uint8 frames -> list of keypoint arrays (``conf=True``: also the confidences)."""
        return infer_batch(frames, self.n_ids, self.deepc, self.refinenet, kmax, conf=conf)
