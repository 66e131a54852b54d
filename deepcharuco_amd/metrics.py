"""Accuracy harness: the reference's validation metrics over this path's outputs.

Mirrors, with the same names, argument meaning and arithmetic:
  * ``DC_Metrics``         /root/reference/src/models/metrics.py:38-132  (mean worst-case L2 px error per id and the
                           ratio of target ids matched within ``px_margin`` = 3 px);
  * ``Refinenet_Metrics``  metrics.py:135-161 (L2 distance between heat-map arg-max and target arg-max, in 1/8 px);
  * ``label_to_keypoints`` metrics.py:25-35 (the metrics' own copy: key-points as float32);
  * ``compute_l2_distance`` / ``pixel_error``  /root/reference/src/utils.py:6-52 (raw vs refined error report).

The reference derives both metric classes from ``torchmetrics.Metric`` (training-loop plumbing: state registration and
DDP reduction); here they are plain accumulators with the same ``update`` / ``compute`` contract.  What feeds them is
this repo's hot path: ``update`` decodes GPU logits with the HIP decode kernel (``models.model_utils.pred_to_keypoints``)
and takes heat-map arg-maxes with ``speedy_bargmax2d``; ``update_keypoints`` takes ``infer_batch`` results directly.
The per-id matching arithmetic is host-side torch / numpy with the reference's semantics (including its assumption that an id
occurs once per target frame -- a duplicated target id behaves exactly as it does there).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

__all__ = ["DC_Metrics", "Refinenet_Metrics", "label_to_keypoints", "compute_l2_distance", "pixel_error",
           "keypoints_from_results"]


def label_to_keypoints(loc: torch.Tensor, ids: torch.Tensor, dust_bin_ids: int):
    """metrics.py:25-35: label maps (N,Hc,Wc) -> (kpts (K,2) float32 (x,y), ids (K,)), raster order (frame, row, column)."""
    if loc.ndim != 3 or ids.ndim != 3:
        raise AssertionError("label maps must be (N, Hc, Wc)")
    loc, ids = loc.cpu(), ids.cpu()
    n, cy, cx = torch.nonzero(ids != dust_bin_ids, as_tuple=True)       # row-major: the order the reference's masks produce
    cell = loc[n, cy, cx]                                               # 0..63 inside the 8x8 cell: x + 8 y
    kpts = torch.stack((8 * cx + cell % 8, 8 * cy + cell // 8), dim=1).to(torch.float32)
    return kpts, ids[n, cy, cx]


def _pred_to_keypoints(loc_hat: torch.Tensor, ids_hat: torch.Tensor, dust_bin_ids: int):
    """metrics.py:18-22 on the HIP decode kernel: logits (1,C,Hc,Wc) on the GPU -> (kpts float32 (K,2), ids (K,)) on the host."""
    from .models.model_utils import pred_to_keypoints
    k, i = pred_to_keypoints(loc_hat, ids_hat, dust_bin_ids)
    return k.cpu().float(), i.cpu()


class DC_Metrics:
    """metrics.py:38-132.  ``distance``: mean over frames of (sum over target ids of the worst L2 distance between the
    predictions and the target of that id) / (ids found); ``ratio``: mean fraction of target slots whose id is matched
    within ``px_margin`` pixels.  Both accumulate over ``update`` calls (the reference relies on torchmetrics to
    average across steps; ``compute`` returns the running sums exactly as metrics.py:131-132 does)."""

    higher_is_better: Optional[bool] = False

    def __init__(self, dust_bin_ids: int):
        self.distance = torch.tensor(0.)
        self.ratio = torch.tensor(0.)
        self.px_margin = 3
        self.dust_bin_ids = dust_bin_ids

    def reset(self) -> None:
        self.distance = torch.tensor(0.)
        self.ratio = torch.tensor(0.)

    # metrics.py:48-76
    def update(self, preds, target) -> None:
        """preds = (loc_hat (N,65,Hc,Wc), ids_hat (N,n_ids+1,Hc,Wc)) GPU logits; target = (loc (N,Hc,Wc), ids (N,Hc,Wc)) labels."""
        (loc_x, ids_x), (loc_target, ids_target) = preds, target
        bs = loc_x.shape[0]
        pred = [_pred_to_keypoints(loc_x[i].unsqueeze(0), ids_x[i].unsqueeze(0), self.dust_bin_ids) for i in range(bs)]
        tgt = [label_to_keypoints(loc_target[i].unsqueeze(0), ids_target[i].unsqueeze(0), self.dust_bin_ids) for i in range(bs)]
        self._accumulate(pred, tgt)

    def update_keypoints(self, results: Sequence[np.ndarray], target) -> None:
        """``results``: per-frame (K,3) [x, y, id] arrays as returned by ``infer_batch`` / ``infer_image`` (detector-only
        or refined); target as in :meth:`update`, or a list of per-frame (K,3) arrays."""
        pred = [keypoints_from_results(r) for r in results]
        if isinstance(target, tuple):
            loc_target, ids_target = target
            tgt = [label_to_keypoints(loc_target[i].unsqueeze(0), ids_target[i].unsqueeze(0), self.dust_bin_ids)
                   for i in range(len(pred))]
        else:
            tgt = [keypoints_from_results(t) for t in target]
        self._accumulate(pred, tgt)

    def _accumulate(self, pred, tgt) -> None:
        bs = len(pred)
        l2_sum = 0.
        ratio_sum = 0.
        atleast = False
        for (keypoint, id_), (keypoint_target, id_target) in zip(pred, tgt):
            l2_dist = self.compute_l2_distance(keypoint, id_, keypoint_target, id_target)
            ratio = self.compute_ratio(keypoint, id_, keypoint_target, id_target)
            if l2_dist is not None:
                atleast = True
                l2_sum += l2_dist
                ratio_sum += ratio
        if atleast:
            self.distance += l2_sum / bs
            self.ratio += ratio_sum / bs

    def _worst_per_id(self, keypoints, ids, target_keypoints, target_ids):
        """Shared first half of metrics.py:79-129 (the reference's own TODO asks for this merge): for the i-th unique target
        id that both sides contain -> (i, max over predictions of the L2 distance to that id's target)."""
        out = []
        for slot, tid in enumerate(torch.unique(target_ids)):
            p_idx = torch.nonzero(ids == tid).squeeze(1)
            t_idx = torch.nonzero(target_ids == tid).squeeze(1)
            if p_idx.numel() == 0 or t_idx.numel() == 0:
                continue
            pair = torch.cdist(keypoints[p_idx], target_keypoints[t_idx], p=2).squeeze(1)
            out.append((slot, torch.max(pair, dim=0).values))
        return out

    def compute_ratio(self, keypoints, ids, target_keypoints, target_ids):
        """metrics.py:79-102: fraction of the target slots whose id is found with a worst distance < px_margin."""
        n_slots = len(target_ids)
        if n_slots == 0:
            return None
        matches = torch.zeros((n_slots,))
        for slot, worst in self._worst_per_id(keypoints, ids, target_keypoints, target_ids):
            if worst < self.px_margin:          # a duplicated target id makes this ambiguous, exactly as in the reference
                matches[slot] = 1
        return matches.mean()

    def compute_l2_distance(self, keypoints, ids, target_keypoints, target_ids):
        """metrics.py:104-129: sum of the worst distances / number of ids found (at least 1)."""
        n_slots = len(target_ids)
        if n_slots == 0:
            return None
        distances = torch.zeros((n_slots,))
        per_id = self._worst_per_id(keypoints, ids, target_keypoints, target_ids)
        for slot, worst in per_id:
            distances[slot] = worst
        return distances.sum() / max(1, len(per_id))

    def compute(self):
        return self.distance, self.ratio


class Refinenet_Metrics:
    """metrics.py:135-161: accumulated mean L2 distance (in heat-map pixels = 1/8 image px) between the arg-max of the
    predicted 64x64 heat-map and the arg-max of the target heat-map."""

    higher_is_better: Optional[bool] = False

    def __init__(self):
        self.distance = torch.tensor(0.)

    def reset(self) -> None:
        self.distance = torch.tensor(0.)

    @staticmethod
    def _pred_argmax_rc(x: torch.Tensor) -> torch.Tensor:
        """Predicted heat-maps (bs,64,64) ON THE GPU -> (bs,2) (row, col) of the first maximum, by the HIP arg-max kernel."""
        if x.device.type != "cuda":
            raise RuntimeError("Refinenet_Metrics.update expects the predicted heat-maps on the GPU (RefineNet.forward output); "
                               "deepcharuco_amd has no CPU path")
        from .models.model_utils import speedy_bargmax2d
        cr = speedy_bargmax2d(x.float()).cpu()                    # (col, row)
        return torch.stack((cr[:, 1], cr[:, 0]), dim=1)

    @staticmethod
    def _target_argmax_rc(t: torch.Tensor) -> torch.Tensor:
        """Label heat-maps (host data, metrics.py:152-154) -> (bs,2) (row, col)."""
        t = t.cpu()
        d = t.shape[-1]
        m = t.reshape(t.shape[0], -1).argmax(1)
        return torch.stack((torch.div(m, d, rounding_mode="floor"), m % d), dim=1)

    def update(self, preds: torch.Tensor, target: torch.Tensor) -> None:
        loc_x, loc_target = preds, target
        loc_x = loc_x.squeeze(1)
        loc_indices = self._pred_argmax_rc(loc_x).unsqueeze(1)
        target_indices = self._target_argmax_rc(loc_target).unsqueeze(1)
        dist = torch.cdist(loc_indices.float(), target_indices.float(), p=2).squeeze(1)
        self.distance += dist.mean()

    def update_corners(self, corners: torch.Tensor, target: torch.Tensor) -> None:
        """``corners`` (K,2) int (col,row) as returned by ``RefineNet.infer_patches`` (its second output)."""
        c = corners.cpu()
        loc_indices = torch.stack((c[:, 1], c[:, 0]), dim=1).unsqueeze(1)
        target_indices = self._target_argmax_rc(target).unsqueeze(1)
        dist = torch.cdist(loc_indices.float(), target_indices.float(), p=2).squeeze(1)
        self.distance += dist.mean()

    def compute(self):
        return self.distance


def keypoints_from_results(res: np.ndarray) -> Tuple[torch.Tensor, torch.Tensor]:
    """One frame of ``infer_batch`` output ((K,3) [x,y,id], or ``np.array([])``) -> (kpts float32 (K,2), ids int64 (K,))."""
    a = np.asarray(res)
    if a.ndim != 2 or a.shape[0] == 0:
        return torch.zeros((0, 2), dtype=torch.float32), torch.zeros((0,), dtype=torch.int64)
    return torch.from_numpy(a[:, :2].astype(np.float32)), torch.from_numpy(a[:, 2].astype(np.int64))


# ---------------------------------------------------------------------------------------------------------------------
# /root/reference/src/utils.py

def compute_l2_distance(keypoints, ids, target_keypoints, target_ids):
    """utils.py:6-30.  One slot per target key-point; slot r (r = rank of an id among the sorted distinct target ids) holds
    the WORST L2 distance between the key-points carrying that id and its target; slots of ids nobody predicted stay 0, as do
    the trailing slots when ids repeat among the targets.  ``None`` without targets.  Vectorised: one scatter-max over the
    predictions instead of a Python loop over ids (an id that occurs several times among the targets is paired element by
    element like the reference's broadcast, and fails the same way when the counts disagree)."""
    keypoints, target_keypoints = np.asarray(keypoints), np.asarray(target_keypoints)
    ids, target_ids = np.asarray(ids), np.asarray(target_ids)
    out = np.zeros((len(target_ids),))
    if out.size == 0:
        return None
    uniq, first, count = np.unique(target_ids, return_index=True, return_counts=True)
    rank = np.minimum(np.searchsorted(uniq, ids), len(uniq) - 1)
    known = uniq[rank] == ids                                     # predictions whose id exists among the targets
    easy = known & (count[rank] == 1)
    if easy.any():
        worst = np.linalg.norm(keypoints[easy] - target_keypoints[first[rank[easy]]], ord=2, axis=1)
        np.maximum.at(out, rank[easy], worst)
    for r in np.nonzero(count > 1)[0]:                            # repeated target id: numpy pairs rows one to one
        mine = np.nonzero(ids == uniq[r])[0]
        if mine.size:
            out[r] = np.linalg.norm(keypoints[mine] - target_keypoints[target_ids == uniq[r]], ord=2, axis=1).max()
    return out


def pixel_error(kpts_raw, kpts_ref, kpts_target, verbose: bool = True):
    """utils.py:33-52: mean pixel error of the detector's raw key-points and of the RefineNet-refined ones against the
    targets ((K,3) [x,y,id] arrays); ``(None, None)`` when a raw id is not among the target ids.  Returns
    ``(mean raw error, mean refined error)``; ``verbose`` prints a three-line summary."""
    raw_xy, raw_id = kpts_raw[:, :2], kpts_raw[:, 2]
    ref_xy, ref_id = kpts_ref[:, :2], kpts_ref[:, 2]
    tgt_xy, tgt_id = kpts_target[:, :2], kpts_target[:, 2]
    if not np.isin(raw_id, tgt_id).all():
        return None, None
    err = {"raw vs target": compute_l2_distance(raw_xy, raw_id, tgt_xy, tgt_id),
           "refined vs target": compute_l2_distance(ref_xy, ref_id, tgt_xy, tgt_id),
           "refined vs raw": compute_l2_distance(ref_xy, ref_id, raw_xy, raw_id)}
    if verbose:
        print(f"pixel error over {np.unique(raw_id).size} of {tgt_id.size} target corners")
        for label, d in err.items():
            print(f"  {label:<18s} mean {d.mean():.3f} px   max {d.max():.3f} px")
    return err["raw vs target"].mean(), err["refined vs target"].mean()
