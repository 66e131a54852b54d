"""Synthetic workloads for the measured configurations (BASELINE.json configs).

The published checkpoints are not in the reference mount, so every measured configuration runs numpy-seeded weights
whose ids dust-bin bias is calibrated so that a realistic number of cells fire:

* :func:`calibrate_dustbin` -- shift ``convDb.bias[n_ids]`` so that on average ``per_frame`` cells fire on the given
  frames (cfg2/cfg3/cfg4: data-dependent K with mean 16);
* :func:`select_fixed_k_frames` -- cfg5 asks for exactly 16 corners in EVERY frame ("16-corner batched RefineNet
  patches per frame").  One shared dust-bin bias cannot force that, so the generator keeps drawing seeded candidate
  frames and keeps those on which the detector fires exactly ``k`` cells.  The selection happens in the workload
  generator, never inside the measured pipeline (which stays data-dependent).

Everything here is set-up code that runs outside every timed region; the logits / counts come from the HIP library.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

from . import weights as W
from .inference import infer_batch_device
from .models.net import dcModel

# name -> per-GPU workload of a BASELINE.json config (configs[0] is the CPU plumbing case, not a GPU workload)
PRESETS: Dict[str, dict] = {
    "cfg2": dict(batch=32, height=240, width=320, kmax=64, frames="board", fixed_k=0, baseline_index=1, gpus=1),
    "cfg3": dict(batch=128, height=480, width=640, kmax=64, frames="board", fixed_k=0, baseline_index=2, gpus=1),
    "cfg4": dict(batch=128, height=240, width=320, kmax=64, frames="board", fixed_k=0, baseline_index=3, gpus=8),
    "cfg5": dict(batch=32, height=960, width=1280, kmax=16, frames="board4", fixed_k=16, baseline_index=4, gpus=8),
}


def workload_label(batch: int, height: int, width: int, world: int, fixed_k: int) -> str:
    """Truthful description of what runs; a BASELINE config is named only when shape AND GPU count match it."""
    base = f"bs={batch}/GPU x{world} GPU(s) = {batch * world} frames of {width}x{height}, full detect+refine pipeline"
    if fixed_k:
        base += f", exactly {fixed_k} corners in every frame"
    for name, p in PRESETS.items():
        if (batch, height, width, fixed_k) == (p["batch"], p["height"], p["width"], p["fixed_k"]):
            if world == p["gpus"]:
                return base + f" (BASELINE configs[{p['baseline_index']}])"
            return base + f" (per-GPU load of BASELINE configs[{p['baseline_index']}], which is quoted on {p['gpus']} GPUs)"
    return base + " (custom shape, not a BASELINE config)"


def _logits_on(sd_dc: W.StateDict, frames_dev: torch.Tensor, dev, n_ids: int):
    """HIP detector logits of the frames as host numpy: (loc arg-max (B,Hc,Wc), ids logits (B,n_ids+1,Hc,Wc))."""
    det = dcModel(n_ids, sd_dc, dev)
    loc_parts, ids_parts = [], []
    for i in range(0, frames_dev.shape[0], 32):          # chunked: logits of 32 high-resolution frames are ~200 MB
        out = det.forward_u8(frames_dev[i:i + 32])
        loc_parts.append(out["loc"].cpu().numpy().argmax(1))     # numpy on the host: no stock-PyTorch device operator in this package
        ids_parts.append(out["ids"].cpu().numpy())
    del det
    return np.concatenate(loc_parts), np.concatenate(ids_parts)


def calibrate_dustbin(sd_dc: W.StateDict, frames_dev: torch.Tensor, dev, n_ids: int = 16, per_frame: int = 16,
                      diverse_ids: bool = False) -> W.StateDict:
    """Returns a copy of ``sd_dc`` whose ``convDb.bias[n_ids]`` makes ``per_frame * B`` cells fire on ``frames_dev``
    (HIP detector logits -> host numpy; the (k-th, k+1-th) largest non-dust-bin margins bracket the shift).

    ``diverse_ids``: first equalise ``convDb.bias[0:n_ids]`` per class (``weights.diverse_ids_bias_shift``) so that the firing
    cells carry all the ids of the board instead of the one or two a random-init ids head lets win everywhere.
    (Round 4 had a ``kmax`` argument that stopped counting a frame's cells at the pipeline's per-frame capacity; the pipeline has
    no per-frame capacity any more -- every firing cell is refined -- so the target is simply the number of firing cells.)"""
    sd = {k_: v.copy() for k_, v in sd_dc.items()}
    k = per_frame * frames_dev.shape[0]
    la, ids = _logits_on(sd, frames_dev, dev, n_ids)
    if diverse_ids:
        shift, _ = W.diverse_ids_bias_shift(np.moveaxis(ids, 1, 0), la, n_ids, k)
        sd["convDb.bias"][:n_ids] = (sd["convDb.bias"][:n_ids] + shift).astype(np.float32)
        la, ids = _logits_on(sd, frames_dev, dev, n_ids)      # the detector's own logits with the shifted biases (not ids + shift)
    m = ids[:, :n_ids].max(1) - ids[:, n_ids]
    mm = np.where(la == 64, -1e30, m)
    flat = mm.ravel()
    order = np.argsort(-flat, kind="stable")
    ms = flat[order]
    k = max(1, min(k, flat.size - 1))
    delta = np.float32((ms[k - 1] + ms[k]) / 2)
    sd["convDb.bias"][n_ids] = np.float32(sd["convDb.bias"][n_ids] + delta)
    return sd


def frame_counts(frames_dev: torch.Tensor, dc, dust_bin_ids: int = 16) -> np.ndarray:
    """Firing cells per frame, from the product pipeline itself (detector + decode, no RefineNet)."""
    b = frames_dev.shape[0]
    packed = infer_batch_device(frames_dev, dust_bin_ids, dc, None, pool=1)     # counts are exact whatever the pool
    return packed[:b].cpu().numpy().copy()


def select_fixed_k_frames(kind: str, seed: int, batch: int, height: int, width: int, k: int, dc, dev,
                          dust_bin_ids: int = 16, chunk: int = 32, max_candidates: int = 4096,
                          first: np.ndarray = None) -> Tuple[np.ndarray, List[int]]:
    """``batch`` seeded frames on each of which the detector ``dc`` fires exactly ``k`` cells.

    Candidates are ``synthetic_frames(kind, seed + j, ...)`` for j = 0, 1, ...; returns (frames, the j that were kept).
    ``first`` (optional, (n,H,W) uint8) are frames to place at the front regardless (e.g. a golden fixture frame whose
    count is known to be k)."""
    keep: List[np.ndarray] = [] if first is None else [f for f in first]
    kept_ids: List[int] = [-1] * len(keep)
    j = 0
    seen: List[int] = []
    while len(keep) < batch:
        if j >= max_candidates:
            raise RuntimeError(f"only {len(keep)} of {batch} frames with exactly {k} corners among {j} candidates "
                               f"(corner counts seen: min {min(seen)}, median {int(np.median(seen))}, max {max(seen)})")
        cand = W.synthetic_frames(kind, seed + j, chunk, height, width)
        counts = frame_counts(torch.from_numpy(cand).to(dev), dc, dust_bin_ids)
        seen += counts.tolist()
        for i in np.nonzero(counts == k)[0]:
            if len(keep) < batch:
                keep.append(cand[i])
                kept_ids.append(j + int(i))
        j += chunk
    return np.stack(keep), kept_ids
