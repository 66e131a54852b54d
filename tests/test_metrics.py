"""Accuracy harness: deepcharuco_amd/metrics.py against the values the REFERENCE's own
models/metrics.py / utils.py produced on the seeded cases (tests/golden/metrics_golden.npz, written by
oracle/make_golden.py).  Host-side parts here; the GPU-fed entries (HIP decode / arg-max) are in the -m gpu tests below."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from deepcharuco_amd import metrics as PM
from oracle import deepcharuco_oracle as O
from oracle import metrics_cases as MC


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(GOLDEN, "metrics_golden.npz"))


def _oracle_results(loc, ids):
    res = []
    for b in range(loc.shape[0]):
        k, i = O.pred_to_keypoints(loc[b:b + 1], ids[b:b + 1], 16)
        res.append(np.concatenate([k.numpy(), i.numpy()[:, None]], axis=1) if k.shape[0] else np.array([]))
    return res


def test_dc_metrics_keypoint_entry_matches_reference_values(fx):
    m = PM.DC_Metrics(16)
    for n, seed in enumerate(fx["dc_seeds"]):
        loc, ids, loc_t, ids_t = [torch.from_numpy(a) for a in MC.dc_case(int(seed))]
        m.update_keypoints(_oracle_results(loc, ids), (loc_t, ids_t))
        assert [float(m.distance), float(m.ratio)] == fx["dc_per_update"][n].tolist()
    d, r = m.compute()
    assert float(d) == float(fx["dc_distance"]) and float(r) == float(fx["dc_ratio"])
    # targets given as per-frame (K,3) arrays instead of label maps: same numbers
    m2 = PM.DC_Metrics(16)
    for seed in fx["dc_seeds"]:
        loc, ids, loc_t, ids_t = [torch.from_numpy(a) for a in MC.dc_case(int(seed))]
        tgt = []
        for b in range(loc_t.shape[0]):
            k, i = PM.label_to_keypoints(loc_t[b:b + 1], ids_t[b:b + 1], 16)
            tgt.append(np.concatenate([k.numpy(), i.numpy()[:, None].astype(np.float32)], axis=1) if k.shape[0] else np.array([]))
        m2.update_keypoints(_oracle_results(loc, ids), tgt)
    assert float(m2.distance) == float(fx["dc_distance"]) and float(m2.ratio) == float(fx["dc_ratio"])
    m.reset()
    assert float(m.distance) == 0.0


def test_dc_metrics_edge_cases():
    m = PM.DC_Metrics(16)
    empty_k, empty_i = torch.zeros((0, 2)), torch.zeros((0,), dtype=torch.int64)
    assert m.compute_l2_distance(empty_k, empty_i, empty_k, empty_i) is None          # no targets (metrics.py:108-109)
    assert m.compute_ratio(empty_k, empty_i, empty_k, empty_i) is None
    tk, ti = torch.tensor([[10., 10.], [50., 20.]]), torch.tensor([3, 7])
    assert float(m.compute_l2_distance(empty_k, empty_i, tk, ti)) == 0.0               # nothing found: 0 / max(1, 0)
    assert float(m.compute_ratio(empty_k, empty_i, tk, ti)) == 0.0
    pk, pi = torch.tensor([[10., 13.], [11., 10.], [80., 80.]]), torch.tensor([3, 3, 5])
    assert float(m.compute_l2_distance(pk, pi, tk, ti)) == 3.0                         # worst of the two id-3 detections
    assert float(m.compute_ratio(pk, pi, tk, ti)) == 0.0                               # 3.0 is not < px_margin
    m.update_keypoints([np.array([])], [np.array([])])                                 # frame without targets: no change
    assert float(m.distance) == 0.0 and float(m.ratio) == 0.0


def test_refinenet_metrics_corner_entry_and_gpu_only_preds(fx):
    m = PM.Refinenet_Metrics()
    for n, seed in enumerate(fx["rn_seeds"]):
        heat, target = [torch.from_numpy(a) for a in MC.refinenet_case(int(seed))]
        corners = O.speedy_bargmax2d(heat[:, 0])                 # (col,row): what RefineNet.infer_patches returns
        m.update_corners(corners, target)
        assert float(m.distance) == float(fx["rn_per_update"][n])
    assert float(m.compute()) == float(fx["rn_distance"])
    heat, target = [torch.from_numpy(a) for a in MC.refinenet_case(21)]
    with pytest.raises(RuntimeError, match="no CPU path"):
        PM.Refinenet_Metrics().update(heat, target)              # predicted heat-maps must come from the GPU path


def test_pixel_error_matches_reference_values(fx, capsys):
    raw, ref, tgt, bad = MC.pixel_error_case(int(fx["pe_seed"]))
    e_raw, e_ref = PM.pixel_error(raw, ref, tgt)
    assert "raw vs target" in capsys.readouterr().out           # a summary is printed (own format; the VALUES are the reference's)
    assert e_raw == float(fx["pe_raw"]) and e_ref == float(fx["pe_ref"])
    assert PM.pixel_error(bad, ref, tgt, verbose=False) == (None, None)
    assert np.array_equal(PM.compute_l2_distance(raw[:, :2], raw[:, 2], tgt[:, :2], tgt[:, 2]), fx["pe_l2"])
    assert PM.compute_l2_distance(raw[:, :2], raw[:, 2], tgt[:0, :2], tgt[:0, 2]) is None


# ----------------------------------------------------------------------------------------------- GPU-fed entries

@pytest.mark.gpu
def test_dc_and_refinenet_metrics_from_gpu_tensors(fx):
    dev = torch.device("cuda", 0)
    m = PM.DC_Metrics(16)
    for n, seed in enumerate(fx["dc_seeds"]):
        loc, ids, loc_t, ids_t = [torch.from_numpy(a) for a in MC.dc_case(int(seed))]
        m.update((loc.to(dev), ids.to(dev)), (loc_t, ids_t))     # HIP decode kernel
        assert [float(m.distance), float(m.ratio)] == fx["dc_per_update"][n].tolist()
    r = PM.Refinenet_Metrics()
    for n, seed in enumerate(fx["rn_seeds"]):
        heat, target = [torch.from_numpy(a) for a in MC.refinenet_case(int(seed))]
        r.update(heat.to(dev), target)                           # HIP arg-max kernel
        assert float(r.distance) == float(fx["rn_per_update"][n])


@pytest.mark.gpu
def test_accuracy_harness_over_infer_batch_outputs():
    """End to end: the harness fed with infer_batch results.  Targets = the ORACLE's refined corners, so the refined HIP
    output must score distance 0 / ratio 1 on every frame with unique ids, and pixel_error's raw-vs-refined report must
    equal the one computed from the oracle's own raw / refined outputs."""
    from deepcharuco_amd import weights as W
    from deepcharuco_amd.inference import infer_batch
    from deepcharuco_amd.models.net import dcModel, lModel
    from deepcharuco_amd.models.refinenet import RefineNet, lRefineNet
    dev = torch.device("cuda", 0)
    frames = W.synthetic_frames("board", 640, 8, 120, 160)
    sd_dc = W.synthetic_state_dict("detector", 41)
    t = O.to_torch_state_dict(sd_dc)
    x = torch.from_numpy(np.stack([O.pre_bgr_image(f) for f in frames]))
    loc, ids = O.detector_forward(t, x)
    mg = ids[:, :16].max(1).values - ids[:, 16]
    mg = torch.where(loc.argmax(1) == 64, torch.tensor(-1e30), mg).flatten().sort(descending=True).values
    sd_dc["convDb.bias"][16] += np.float32((mg[8 * 5 - 1] + mg[8 * 5]) / 2)       # ~5 corners per frame
    sd_rn = W.synthetic_state_dict("refinenet", 42)
    dc, rn = lModel(dcModel(16, sd_dc, dev)), lRefineNet(RefineNet(sd_rn, dev))
    got_ref = infer_batch(frames, 16, dc, rn)
    got_raw = infer_batch(frames, 16, dc, None)
    t_dc, t_rn = O.to_torch_state_dict(sd_dc), O.to_torch_state_dict(sd_rn)
    exp_ref = [O.infer_image(None, 16, t_dc, t_rn, gray=f) for f in frames]
    exp_raw = [O.infer_image(None, 16, t_dc, None, gray=f) for f in frames]
    def dedup(a):      # the reference's metric assumes an id occurs once per TARGET frame: keep each id's first corner
        _, first = np.unique(a[:, 2], return_index=True)
        return a[np.sort(first)]
    have = [b for b in range(8) if exp_ref[b].ndim == 2]
    assert len(have) >= 5
    targets = [dedup(exp_ref[b]) for b in have]
    vals = {}
    for name, res in (("hip_ref", got_ref), ("oracle_ref", exp_ref), ("hip_raw", got_raw), ("oracle_raw", exp_raw)):
        m = PM.DC_Metrics(16)
        m.update_keypoints([res[b] for b in have], targets)
        vals[name] = (float(m.distance), float(m.ratio))
    assert vals["hip_ref"] == vals["oracle_ref"] and vals["hip_raw"] == vals["oracle_raw"]
    assert vals["hip_raw"][0] > 0.0 and 0.0 < vals["hip_ref"][1] <= 1.0
    one = [b for b in have if len(set(exp_ref[b][:, 2])) == exp_ref[b].shape[0]]      # frames whose ids are unique
    m = PM.DC_Metrics(16)
    m.update_keypoints([got_ref[b] for b in one], [exp_ref[b] for b in one])
    assert len(one) >= 1 and float(m.distance) == 0.0 and float(m.ratio) == 1.0
    b = one[0]
    with contextlib.redirect_stdout(io.StringIO()):
        a = PM.pixel_error(got_raw[b].astype(np.float64), got_ref[b], exp_ref[b])
        e = PM.pixel_error(exp_raw[b].astype(np.float64), exp_ref[b], exp_ref[b])
    assert a == e and a[1] == 0.0
