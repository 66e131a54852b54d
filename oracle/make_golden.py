"""ORACLE tooling -- generates tests/golden/*.npz from the REFERENCE ITSELF.

Runs only in the build container (needs /root/reference).  It imports the
reference's own hot-path modules (src/inference.py, src/models/*.py) with the
four absent third-party packages stubbed (pytorch_lightning, torchmetrics,
numba, cv2 -- none of which take part in the hot-path arithmetic except
cv2.cvtColor, see oracle/deepcharuco_oracle.py), feeds them seeded synthetic
weights and frames, and

  1. asserts oracle/deepcharuco_oracle.py returns identical tensors
     (torch.equal) for every hot-path function  -> pins the oracle;
  2. writes the reference's outputs as small fixtures under tests/golden
     (data only: inputs are regenerated from seeds and guarded by SHA-256).

Usage:  python oracle/make_golden.py            (from the repo root)
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
sys.path.insert(0, REPO)

from deepcharuco_amd import weights as W  # noqa: E402
from oracle import deepcharuco_oracle as O  # noqa: E402


# ------------------------------------------------------------------ reference import

def import_reference():
    sys.dont_write_bytecode = True
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = torch.nn.Module
    sys.modules["pytorch_lightning"] = pl

    tm = types.ModuleType("torchmetrics")

    class Metric(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def add_state(self, name, default, dist_reduce_fx=None):
            setattr(self, name, default)
    tm.Metric = Metric
    sys.modules["torchmetrics"] = tm

    nb = types.ModuleType("numba")
    nb.njit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
    nb.prange = range
    sys.modules["numba"] = nb

    cv2 = types.ModuleType("cv2")
    cv2.COLOR_BGR2GRAY = 6
    cv2.TERM_CRITERIA_EPS = 2
    cv2.TERM_CRITERIA_COUNT = 1
    cv2.aruco = types.SimpleNamespace()

    def cvtColor(img, code):
        assert code == cv2.COLOR_BGR2GRAY
        return O.bgr2gray(img)
    cv2.cvtColor = cvtColor

    # recording stand-in for cv2.solvePnP: the reference's solve_pnp (inference.py:15-29) is CALLED and what it hands to
    # OpenCV (object points, image points, camera matrix, distortion) is captured -- the PnP solve itself is third-party
    cv2.solvePnP_calls = []

    def solvePnP(object_points, image_points, camera_matrix, dist_coeffs):
        cv2.solvePnP_calls.append((np.array(object_points, copy=True), np.array(image_points, copy=True),
                                   camera_matrix, dist_coeffs))
        return True, np.zeros((3, 1)), np.zeros((3, 1))
    cv2.solvePnP = solvePnP
    sys.modules["cv2"] = cv2

    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "models"))
    os.chdir(REF)
    import inference as ref_inf          # noqa
    from models import net as ref_net    # noqa
    from models import refinenet as ref_rn  # noqa
    from models import model_utils as ref_mu  # noqa
    return ref_inf, ref_net, ref_rn, ref_mu


def metrics_golden(outdir):
    """Accuracy-harness fixture: the REFERENCE's DC_Metrics / Refinenet_Metrics (models/metrics.py:38-161, torchmetrics
    stubbed) and utils.pixel_error (utils.py:33-52) on the seeded cases of oracle/metrics_cases.py.  The product
    restatement (deepcharuco_amd/metrics.py) is asserted equal here through its key-point entry (the logits entry
    needs the GPU decode and is compared with these values by the -m gpu test)."""
    import contextlib
    import io
    from models import metrics as ref_metrics   # noqa  (sys.path / cwd set by import_reference)
    import utils as ref_utils                   # noqa
    from oracle import metrics_cases as MC
    from deepcharuco_amd import metrics as PM

    fx = {}
    ref_dc, my_dc = ref_metrics.DC_Metrics(16), PM.DC_Metrics(16)
    per_update = []
    for seed in (11, 12):
        loc, ids, loc_t, ids_t = [torch.from_numpy(a) for a in MC.dc_case(seed)]
        ref_dc.update((loc, ids), (loc_t, ids_t))
        per_update.append([float(ref_dc.distance), float(ref_dc.ratio)])
        # product harness fed with the key-points the oracle decodes from the same logits
        res = []
        for b in range(loc.shape[0]):
            k, i = O.pred_to_keypoints(loc[b:b + 1], ids[b:b + 1], 16)
            res.append(np.concatenate([k.numpy(), i.numpy()[:, None]], axis=1) if k.shape[0] else np.array([]))
        my_dc.update_keypoints(res, (loc_t, ids_t))
        assert float(my_dc.distance) == float(ref_dc.distance) and float(my_dc.ratio) == float(ref_dc.ratio), seed
    d, r = ref_dc.compute()
    fx["dc_seeds"] = np.array([11, 12])
    fx["dc_per_update"] = np.array(per_update, np.float64)
    fx["dc_distance"], fx["dc_ratio"] = np.float64(d), np.float64(r)
    assert 0.0 < float(r) < 2.0 and float(d) > 0.0

    ref_rn = ref_metrics.Refinenet_Metrics()
    rn_updates = []
    for seed in (21, 22):
        heat, target = [torch.from_numpy(a) for a in MC.refinenet_case(seed)]
        ref_rn.update(heat, target)
        rn_updates.append(float(ref_rn.distance))
    fx["rn_seeds"] = np.array([21, 22])
    fx["rn_per_update"] = np.array(rn_updates, np.float64)
    fx["rn_distance"] = np.float64(ref_rn.compute())

    raw, ref, tgt, bad = MC.pixel_error_case(31)
    with contextlib.redirect_stdout(io.StringIO()):
        e_raw, e_ref = ref_utils.pixel_error(raw, ref, tgt)
        none_case = ref_utils.pixel_error(bad, ref, tgt)
        m_raw, m_ref = PM.pixel_error(raw, ref, tgt)
    assert none_case == (None, None) and PM.pixel_error(bad, ref, tgt, verbose=False) == (None, None)
    assert (m_raw, m_ref) == (e_raw, e_ref)
    fx["pe_seed"] = np.array(31)
    fx["pe_raw"], fx["pe_ref"] = np.float64(e_raw), np.float64(e_ref)
    fx["pe_l2"] = ref_utils.compute_l2_distance(raw[:, :2], raw[:, 2], tgt[:, :2], tgt[:, 2])
    assert np.array_equal(fx["pe_l2"], PM.compute_l2_distance(raw[:, :2], raw[:, 2], tgt[:, :2], tgt[:, 2]))
    np.savez_compressed(os.path.join(outdir, "metrics_golden.npz"), **fx)
    print(f"[golden] metrics: DC distance {float(d):.6f} ratio {float(r):.6f}; RefineNet distance {float(fx['rn_distance']):.6f}; "
          f"pixel_error raw {e_raw:.6f} ref {e_ref:.6f}; product harness == reference")


def ref_models(ref_net, ref_rn, sd_dc, sd_rn, n_ids):
    dc = ref_net.lModel(ref_net.dcModel(n_ids))
    missing = dc.model.load_state_dict(O.to_torch_state_dict(sd_dc), strict=False)
    assert all(k.endswith("num_batches_tracked") for k in missing.missing_keys), missing
    assert not missing.unexpected_keys
    dc.eval()
    rn = ref_rn.lRefineNet(ref_rn.RefineNet())
    missing = rn.model.load_state_dict(O.to_torch_state_dict(sd_rn), strict=False)
    assert all(k.endswith("num_batches_tracked") for k in missing.missing_keys), missing
    rn.eval()
    return dc, rn


def calibrate_dustbin(dc, sd_dc, frame_u8, n_ids, target):
    """Shift convDb.bias[n_ids] so that exactly `target` cells fire on this frame."""
    x = torch.tensor(O.pre_bgr_image(frame_u8))
    loc, ids = dc.infer_image(x)
    la = loc.argmax(1)[0]
    m = (ids[0, :n_ids].max(0).values - ids[0, n_ids])
    m = torch.where(la == 64, torch.tensor(-1e30), m).flatten().sort(descending=True).values
    delta = float((m[target - 1] + m[target]) / 2)
    new_bias = np.float32(sd_dc["convDb.bias"][n_ids] + np.float32(delta))
    sd_dc["convDb.bias"][n_ids] = new_bias
    with torch.no_grad():
        dc.model.convDb.bias[n_ids] = float(new_bias)
    return float(new_bias)


def equalise_ids(dc, sd_dc, frame_u8, n_ids, target):
    """Per-class shift of convDb.bias[0:n_ids] (weights.diverse_ids_bias_shift on the REFERENCE's logits) so that the `target`
    strongest cells of this frame carry many distinct ids (net.py:48,76-77; model_utils.py:72-77)."""
    x = torch.tensor(O.pre_bgr_image(frame_u8))
    loc, ids = dc.infer_image(x)
    shift, _ = W.diverse_ids_bias_shift(ids[0].numpy(), loc.argmax(1)[0].numpy(), n_ids, target)
    sd_dc["convDb.bias"][:n_ids] = (sd_dc["convDb.bias"][:n_ids] + shift).astype(np.float32)
    with torch.no_grad():
        dc.model.convDb.bias[:n_ids] = torch.from_numpy(sd_dc["convDb.bias"][:n_ids].copy())


CASES = [
    # name, weight seed, frame kind, frame seed, H, W, target K, keep full logits
    dict(name="tiny_noise_64x96", wseed=3, kind="noise", fseed=5, H=64, W=96, K=6, full=True),
    dict(name="noise_240x320", wseed=1234, kind="noise", fseed=0, H=240, W=320, K=16, full=True),
    dict(name="board_240x320", wseed=1234, kind="board", fseed=1, H=240, W=320, K=16, full=False),
    dict(name="board_480x640", wseed=7, kind="board", fseed=2, H=480, W=640, K=16, full=False),
    # BASELINE configs[4] resolution; K forced to exactly 16 by the top-16 non-dust-bin margins (calibrate_dustbin)
    dict(name="board4_960x1280", wseed=91, kind="board4", fseed=900, H=960, W=1280, K=16, full=False),
    # ids head equalised per class on the reference's logits: >= 12 of the 16 ids fire on this one frame (the other fixtures'
    # random-init ids heads fire 1-3 distinct ids); full logits kept so the 17-way arg-max is checked on diverse winners
    dict(name="diverse_ids_240x320", wseed=7, kind="board", fseed=2, H=240, W=320, K=16, full=True, diverse=True),
    # The reference's ONLY real input: the 320x240 colour photo its benchmark times (src/benchmark.py:34-35,
    # src/reference/samples_test/IMG_7412.png), read with Pillow (RGB -> BGR = what cv2.imread returns), through the reference's
    # infer_image with two synthetic weight sets (the published checkpoints are not in the mount).  The u8 image travels inside the
    # fixture (data); its gray version is what the stubbed cv2.cvtColor (= the oracle's OpenCV-4.x fixed-point formula) produced.
    dict(name="img7412_240x320", wseed=1234, kind="img7412", fseed=0, H=240, W=320, K=16, full=True),
    dict(name="img7412_diverse_240x320", wseed=7, kind="img7412", fseed=0, H=240, W=320, K=16, full=False, diverse=True, min_ids=8),
]
IMG7412 = "/root/reference/src/reference/samples_test/IMG_7412.png"


def load_img7412():
    """(240,320,3) uint8 BGR, as cv2.imread(SAMPLE_IMAGE) (benchmark.py:35) returns it: PNG decoding is lossless, so Pillow's RGB
    planes reversed are the same bytes."""
    from PIL import Image
    rgb = np.asarray(Image.open(IMG7412).convert("RGB"))
    assert rgb.shape == (240, 320, 3) and rgb.dtype == np.uint8
    return np.ascontiguousarray(rgb[..., ::-1])
N_IDS = 16


def main():
    torch.manual_seed(0)
    ref_inf, ref_net, ref_rn, ref_mu = import_reference()
    outdir = os.path.join(REPO, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    index = {"torch": torch.__version__, "numpy": np.__version__,
             "threads": torch.get_num_threads(), "cases": []}

    # pre_bgr_image on every u8 value (IEEE division check for the device normaliser)
    lut = ref_mu.pre_bgr_image(np.arange(256, dtype=np.uint8).reshape(16, 16)).reshape(256)
    assert np.array_equal(lut, O.pre_bgr_image(np.arange(256, dtype=np.uint8).reshape(16, 16)).reshape(256))
    np.savez_compressed(os.path.join(outdir, "pre_bgr_lut.npz"), lut=lut.astype(np.float32))

    # label_to_keypoints (model_utils.py:91-124) on label maps of its own: also cells whose id fires while loc == 64
    g = torch.Generator().manual_seed(11)
    for (n_, hc_, wc_, dust_) in ((1, 30, 40, 16), (3, 7, 9, 16), (2, 12, 5, 3)):
        loc_m = torch.randint(0, 65, (n_, hc_, wc_), generator=g)
        ids_m = torch.where(torch.rand((n_, hc_, wc_), generator=g) < 0.8, torch.tensor(dust_), torch.randint(0, 17, (n_, hc_, wc_), generator=g))
        rk, ri = ref_mu.label_to_keypoints(loc_m, ids_m, dust_)
        ok_, oi_ = O.label_to_keypoints(loc_m, ids_m, dust_)
        assert torch.equal(rk.to(torch.int64), ok_.to(torch.int64)) and torch.equal(ri, oi_) and rk.shape[0] > 0

    for c in CASES:
        name = c["name"]
        sd_dc = W.synthetic_state_dict("detector", c["wseed"], N_IDS)
        sd_rn = W.synthetic_state_dict("refinenet", c["wseed"] + 1)
        dc, rn = ref_models(ref_net, ref_rn, sd_dc, sd_rn, N_IDS)
        photo = None
        if c["kind"] == "img7412":
            photo = load_img7412()
            import cv2 as cv2_stub_
            frame = cv2_stub_.cvtColor(photo, cv2_stub_.COLOR_BGR2GRAY)     # what inference.py:40 computes (stub = oracle formula)
            assert not np.array_equal(frame, photo[..., 1])                  # a real colour image, not gray x3
        else:
            frame = W.synthetic_frames(c["kind"], c["fseed"], 1, c["H"], c["W"])[0]
        if c.get("diverse"):
            equalise_ids(dc, sd_dc, frame, N_IDS, c["K"])
        dust_bias = calibrate_dustbin(dc, sd_dc, frame, N_IDS, c["K"])
        tsd_dc, tsd_rn = O.to_torch_state_dict(sd_dc), O.to_torch_state_dict(sd_rn)

        # ---- reference run, stage by stage
        gray_f = ref_mu.pre_bgr_image(frame)
        x = torch.tensor(gray_f)
        loc, ids = dc.infer_image(x)
        la, ia = ref_mu.pred_argmax(loc, ids, N_IDS)
        kpts, ids_found = ref_mu.pred_to_keypoints(loc, ids, N_IDS)
        assert kpts.shape[0] == c["K"], (name, kpts.shape)
        if c.get("diverse"):
            assert len(set(ids_found.tolist())) >= c.get("min_ids", 12), (name, ids_found)
        patches = ref_mu.extract_patches(x, kpts)
        with torch.no_grad():
            heat = rn(patches[:, None])
        corners_og, corners = rn.infer_patches(patches, kpts)
        bgr = photo if photo is not None else np.repeat(frame[..., None], 3, axis=2)
        final_rn, img_out = ref_inf.infer_image(bgr, N_IDS, dc, rn, draw_pred=False, device="cpu")
        assert img_out is bgr
        final_norn, _ = ref_inf.infer_image(bgr, N_IDS, dc, None, draw_pred=False, device="cpu")

        # ---- oracle must be identical
        o_loc, o_ids = O.detector_infer_image(tsd_dc, x)
        assert torch.equal(o_loc, loc) and torch.equal(o_ids, ids), f"{name}: detector logits differ"
        o_la, o_ia = O.pred_argmax(o_loc, o_ids, N_IDS)
        assert torch.equal(o_la, la) and torch.equal(o_ia, ia)
        o_k, o_i = O.pred_to_keypoints(o_loc, o_ids, N_IDS)
        assert torch.equal(o_k, kpts) and torch.equal(o_i, ids_found)
        o_p = O.extract_patches(x, kpts)
        assert torch.equal(o_p, patches), f"{name}: patches differ"
        o_heat = O.refinenet_forward(tsd_rn, patches[:, None])
        assert torch.equal(o_heat, heat), f"{name}: refinenet heat-map differs"
        o_cog, o_c = O.refinenet_infer_patches(tsd_rn, patches, kpts)
        assert torch.equal(o_cog, corners_og) and torch.equal(o_c, corners)
        o_final = O.infer_image(bgr, N_IDS, tsd_dc, tsd_rn)
        assert o_final.dtype == final_rn.dtype and np.array_equal(o_final, final_rn)
        o_final2 = O.infer_image(bgr, N_IDS, tsd_dc, None)
        assert o_final2.dtype == final_norn.dtype and np.array_equal(o_final2, final_norn)

        # border key-points for extract_patches
        bk = torch.tensor([[0, 0], [c["W"] - 1, c["H"] - 1], [5, c["H"] - 3], [c["W"] - 2, 7]])
        bpatches = ref_mu.extract_patches(x, bk)
        assert torch.equal(O.extract_patches(x, bk), bpatches)
        assert float(bpatches[0, 12, 12]) == float(x[0, 0, 0])  # centre pixel = key-point pixel

        hm = heat[:, 0].reshape(heat.shape[0], -1)
        hm_top = torch.topk(hm, 2, dim=1).values
        fx = dict(
            meta=json.dumps(dict(name=name, wseed=c["wseed"], kind=c["kind"], fseed=c["fseed"],
                                 H=c["H"], W=c["W"], n_ids=N_IDS, K=c["K"])),
            dust_bias=np.float32(dust_bias),
            convDb_bias=sd_dc["convDb.bias"].astype(np.float32).copy(),       # the whole ids-head bias the case ran with
            distinct_ids=np.array(len(set(ids_found.tolist()))),
            sha_dc=W.state_dict_sha256(sd_dc, "detector", N_IDS),
            sha_rn=W.state_dict_sha256(sd_rn, "refinenet"),
            sha_frame=W.frames_sha256(frame),
            loc_argmax=la[0].numpy().astype(np.int8),
            ids_argmax=ia[0].numpy().astype(np.int8),
            ids_argmax_raw=ids.argmax(1)[0].numpy().astype(np.int8),
            loc_margin=O.top2_margin(loc)[0].numpy(),
            ids_margin=O.top2_margin(ids)[0].numpy(),
            kpts=kpts.numpy(), ids_found=ids_found.numpy(),
            border_kpts=bk.numpy(), border_patches=bpatches.numpy(),
            patches_first2=patches[:2].numpy(),
            patch_sums=patches.double().sum((1, 2)).numpy(),
            corners=corners.numpy(), corners_og=corners_og.numpy(),
            heat_margin=(hm_top[:, 0] - hm_top[:, 1]).numpy(),
            heat_first2=heat[:2, 0].numpy(),
            final_rn=final_rn, final_norn=final_norn,
        )
        if photo is not None:
            fx["bgr_image"] = photo
        if c["full"]:
            fx["loc_logits"] = loc[0].numpy()
            fx["ids_logits"] = ids[0].numpy()
        np.savez_compressed(os.path.join(outdir, f"{name}.npz"), **fx)

        # ---- K == 0 behaviour (inference.py:51-52): dust-bin always wins
        if name.startswith("tiny"):
            sd0 = {k: v.copy() for k, v in sd_dc.items()}
            sd0["convDb.bias"][N_IDS] = np.float32(1e4)
            dc0, _ = ref_models(ref_net, ref_rn, sd0, sd_rn, N_IDS)
            empty, _ = ref_inf.infer_image(bgr, N_IDS, dc0, rn, draw_pred=False, device="cpu")
            o_empty = O.infer_image(bgr, N_IDS, O.to_torch_state_dict(sd0), tsd_rn)
            assert empty.shape == (0,) and empty.dtype == np.float64
            assert o_empty.shape == (0,) and o_empty.dtype == np.float64

        index["cases"].append(dict(name=name, K=int(kpts.shape[0]),
                                   min_loc_margin=float(fx["loc_margin"].min()),
                                   min_ids_margin=float(fx["ids_margin"].min()),
                                   min_heat_margin=float(fx["heat_margin"].min())))
        print(f"[golden] {name}: K={kpts.shape[0]} dust_bias={dust_bias:.6f} "
              f"min margins loc {fx['loc_margin'].min():.2e} ids {fx['ids_margin'].min():.2e} "
              f"heat {fx['heat_margin'].min():.2e}; oracle == reference")

    # BGR -> gray restatement on colour pixels (cv2 absent: parity unpinned, formulas only).  `gray` = the OpenCV 4.x 8-bit
    # formula (15-bit constants; the range the reference pins), `gray_legacy14` = the older 14-bit one; `bgr_differ` are pixels
    # on which the two DISAGREE (gray-replicated input can never tell them apart), with both answers.
    rng = np.random.default_rng(99)
    bgr = rng.integers(0, 256, (16, 24, 3), dtype=np.uint8)
    pool = rng.integers(0, 256, (400000, 3), dtype=np.uint8)
    dif = pool[O.bgr2gray(pool, "opencv4") != O.bgr2gray(pool, "legacy14")]
    assert 500 < dif.shape[0] < 2000, dif.shape            # ~0.26 % of random colour pixels
    dif = dif[:512].reshape(16, 32, 3)
    # hand-checked anchors: (B,G,R) -> 15-bit / 14-bit
    assert int(O.bgr2gray(np.array([[[255, 0, 0]]], np.uint8))[0, 0]) == (255 * 3735 + 16384) >> 15 == 29
    assert int(O.bgr2gray(np.array([[[0, 255, 0]]], np.uint8))[0, 0]) == (255 * 19235 + 16384) >> 15 == 150
    assert int(O.bgr2gray(np.array([[[0, 0, 255]]], np.uint8))[0, 0]) == (255 * 9798 + 16384) >> 15 == 76
    np.savez_compressed(os.path.join(outdir, "bgr2gray_formula.npz"), bgr=bgr, gray=O.bgr2gray(bgr),
                        gray_legacy14=O.bgr2gray(bgr, "legacy14"), bgr_differ=dif, gray_differ=O.bgr2gray(dif),
                        gray_differ_legacy14=O.bgr2gray(dif, "legacy14"))

    # solve_pnp (inference.py:15-29): the REFERENCE's function is called with a recording cv2.solvePnP; the fixture holds
    # the arguments it handed to OpenCV for several boards (square 5x5, the demo's 5x5 at another scale, non-square 4x7 /
    # 7x4, float and int key-point arrays, unsorted ids, repeated ids) plus the < 4 points short-circuit
    import cv2 as cv2_stub
    cam = np.array([[600.0, 0, 160], [0, 600.0, 120], [0, 0, 1]])
    dist = np.zeros(5)
    rng = np.random.default_rng(2024)
    pnp_cases = [
        (np.array([[10.5, 20.25, 3], [100.0, 50.0, 0], [30.0, 31.0, 15], [7.0, 8.0, 9]]), 5, 5, 0.01),
        (np.array([[1.125, 2.0, 0], [3.0, 4.5, 5], [5.0, 6.0, 10], [7.0, 8.0, 15], [9.0, 1.0, 7], [2.0, 2.0, 7]]), 5, 5, 0.035),
        (np.array([[12, 40, 17], [200, 31, 2], [77, 78, 9], [5, 6, 0], [319, 239, 11]], dtype=np.int64), 4, 7, 0.02),
        (np.array([[12, 40, 17], [200, 31, 2], [77, 78, 9], [5, 6, 0], [319, 239, 11]], dtype=np.int64), 7, 4, 0.02),
        (np.concatenate([rng.uniform(0, 320, (16, 2)), rng.permutation(16)[:, None].astype(np.float64)], axis=1), 5, 5, 0.01),
    ]
    fxp = dict(n_cases=np.array(len(pnp_cases)), cam=cam, dist=dist)
    for i, (kp, cols, rows, sq) in enumerate(pnp_cases):
        del cv2_stub.solvePnP_calls[:]
        ret = ref_inf.solve_pnp(kp, cols, rows, sq, cam, dist)
        assert ret[0] is True and len(cv2_stub.solvePnP_calls) == 1
        objp, imgp, cam_seen, dist_seen = cv2_stub.solvePnP_calls[0]
        assert cam_seen is cam and dist_seen is dist and objp.dtype == np.float32 and imgp.dtype == np.float32
        o_objp, o_imgp = O.solve_pnp_object_points(kp, cols, rows, sq)                      # pins the oracle's restatement
        assert o_objp.dtype == objp.dtype and np.array_equal(o_objp, objp) and np.array_equal(o_imgp, imgp)
        fxp[f"kp{i}"], fxp[f"board{i}"] = kp, np.array([cols, rows, sq], np.float64)
        fxp[f"objp{i}"], fxp[f"imgp{i}"] = objp, imgp
    del cv2_stub.solvePnP_calls[:]
    assert ref_inf.solve_pnp(pnp_cases[0][0][:3], 5, 5, 0.01, cam, dist) == (False, None, None)   # inference.py:16-17
    assert not cv2_stub.solvePnP_calls
    # legacy names (first case) kept for the host-thread PnP tests
    fxp["kp"], fxp["objp"], fxp["imgp"] = fxp["kp0"], fxp["objp0"], fxp["imgp0"]
    np.savez_compressed(os.path.join(outdir, "solve_pnp_points.npz"), **fxp)
    print(f"[golden] solve_pnp: {len(pnp_cases)} calls of the reference's own solve_pnp recorded; oracle == reference")

    metrics_golden(outdir)

    with open(os.path.join(outdir, "index.json"), "w") as f:
        json.dump(index, f, indent=1)
    print("[golden] wrote", outdir)


if __name__ == "__main__":
    main()
