"""CPU: the oracle restatement reproduces the fixtures generated from the reference itself
(oracle/make_golden.py), stage by stage.  Integer outputs exact; logits within fp32 re-ordering
noise (the reference's own oneDNN result moves by ~2e-6 between 1 and 8 threads)."""
import numpy as np
import torch

from oracle import deepcharuco_oracle as O

LOGIT_ATOL = 2e-5


def test_pre_bgr_lut():
    import os
    from conftest import GOLDEN
    lut = np.load(os.path.join(GOLDEN, "pre_bgr_lut.npz"))["lut"]
    got = O.pre_bgr_image(np.arange(256, dtype=np.uint8).reshape(16, 16)).reshape(256)
    assert got.dtype == np.float32 and np.array_equal(got, lut)


def test_bgr2gray_formula():
    import os
    from conftest import GOLDEN
    d = np.load(os.path.join(GOLDEN, "bgr2gray_formula.npz"))
    assert np.array_equal(O.bgr2gray(d["bgr"]), d["gray"])
    assert np.array_equal(O.bgr2gray(d["bgr"], "legacy14"), d["gray_legacy14"])
    # pixels on which the OpenCV 4.x (15-bit) and the older 14-bit constants DISAGREE: each variant gives its own answer
    dif = d["bgr_differ"]
    assert dif.shape == (16, 32, 3) and not (d["gray_differ"] == d["gray_differ_legacy14"]).any()
    assert np.abs(d["gray_differ"].astype(int) - d["gray_differ_legacy14"].astype(int)).max() == 1
    assert np.array_equal(O.bgr2gray(dif), d["gray_differ"]) and np.array_equal(O.bgr2gray(dif, "legacy14"), d["gray_differ_legacy14"])
    # default = the generation the reference pins (opencv-contrib-python >= 4.6, < 4.12): gray_shift 15, 3735 / 19235 / 9798
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [10, 200, 30]]], np.uint8)
    assert O.bgr2gray(px).tolist() == [[29, 150, 76, 255, (10 * 3735 + 200 * 19235 + 30 * 9798 + 16384) >> 15]]
    g = np.arange(256, dtype=np.uint8).reshape(16, 16)
    for v in O.BGR2GRAY_VARIANTS:
        cb, cg, cr, sh = O.BGR2GRAY_VARIANTS[v]
        assert cb + cg + cr == 1 << sh
        assert np.array_equal(O.bgr2gray(np.repeat(g[..., None], 3, 2), v), g)   # gray in -> same gray out, either variant


def test_detector_stages(golden):
    fx = golden.fx
    sd = O.to_torch_state_dict(golden.sd_dc)
    x = torch.tensor(O.pre_bgr_image(golden.frame))
    loc, ids = O.detector_infer_image(sd, x)
    assert loc.shape == (1, 65, golden.meta["H"] // 8, golden.meta["W"] // 8)
    assert ids.shape == (1, golden.n_ids + 1, golden.meta["H"] // 8, golden.meta["W"] // 8)
    if "loc_logits" in fx:
        assert np.abs(loc[0].numpy() - fx["loc_logits"]).max() <= LOGIT_ATOL
        assert np.abs(ids[0].numpy() - fx["ids_logits"]).max() <= LOGIT_ATOL
    la, ia = O.pred_argmax(loc, ids, golden.n_ids)
    safe = fx["loc_margin"] > 1e-4
    assert np.array_equal(la[0].numpy()[safe], fx["loc_argmax"].astype(np.int64)[safe])
    kpts, ids_found = O.pred_to_keypoints(loc, ids, golden.n_ids)
    assert np.array_equal(kpts.numpy(), fx["kpts"]) and np.array_equal(ids_found.numpy(), fx["ids_found"])


def test_patches_and_refinenet(golden):
    fx = golden.fx
    x = torch.tensor(O.pre_bgr_image(golden.frame))
    kpts = torch.from_numpy(fx["kpts"])
    patches = O.extract_patches(x, kpts)
    assert np.array_equal(patches[:2].numpy(), fx["patches_first2"])
    assert np.allclose(patches.double().sum((1, 2)).numpy(), fx["patch_sums"], rtol=0, atol=1e-9)
    bp = O.extract_patches(x, torch.from_numpy(fx["border_kpts"]))
    assert np.array_equal(bp.numpy(), fx["border_patches"])
    sd = O.to_torch_state_dict(golden.sd_rn)
    heat = O.refinenet_forward(sd, patches[:, None])
    assert heat.shape == (kpts.shape[0], 1, 64, 64)
    assert np.abs(heat[:2, 0].numpy() - fx["heat_first2"]).max() <= LOGIT_ATOL
    cog, c = O.refinenet_infer_patches(sd, patches, kpts)
    assert np.array_equal(c.numpy(), fx["corners"]) and np.array_equal(cog.numpy(), fx["corners_og"])
    assert cog.dtype == torch.float32 and c.dtype == torch.int64


def test_infer_image_end_to_end(golden):
    fx = golden.fx
    sd_dc, sd_rn = O.to_torch_state_dict(golden.sd_dc), O.to_torch_state_dict(golden.sd_rn)
    a = O.infer_image(golden.bgr, golden.n_ids, sd_dc, sd_rn)
    assert a.dtype == np.float64 and np.array_equal(a, fx["final_rn"])
    b = O.infer_image(golden.bgr, golden.n_ids, sd_dc, None)
    assert b.dtype == np.int64 and np.array_equal(b, fx["final_norn"])
    assert np.all(np.diff(a[:, 2]) >= 0)   # sorted by id


def test_no_corner_returns_empty(golden_tiny):
    sd = {k: v.copy() for k, v in golden_tiny.sd_dc.items()}
    sd["convDb.bias"][golden_tiny.n_ids] = np.float32(1e4)
    out = O.infer_image(golden_tiny.bgr, golden_tiny.n_ids, O.to_torch_state_dict(sd),
                        O.to_torch_state_dict(golden_tiny.sd_rn))
    assert out.shape == (0,) and out.dtype == np.float64


def test_solve_pnp_points():
    """The fixture holds what the REFERENCE's own solve_pnp (inference.py:15-29) handed to a recording cv2.solvePnP
    (oracle/make_golden.py) -- several boards incl. non-square ones, int and float key-points, unsorted / repeated ids."""
    import os
    from conftest import GOLDEN
    d = np.load(os.path.join(GOLDEN, "solve_pnp_points.npz"))
    assert int(d["n_cases"]) >= 5
    for i in range(int(d["n_cases"])):
        cols, rows, sq = d[f"board{i}"]
        objp, imgp = O.solve_pnp_object_points(d[f"kp{i}"], int(cols), int(rows), float(sq))
        assert objp.dtype == np.float32 and imgp.dtype == np.float32
        assert np.array_equal(objp, d[f"objp{i}"]) and np.array_equal(imgp, d[f"imgp{i}"]), i
