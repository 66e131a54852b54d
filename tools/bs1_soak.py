#!/usr/bin/env python3
"""Soak of the one-frame path (the split-position kernels, hipGraph replay): every result of N infer_image calls on the reference's
photo is compared with the reference's own answer (fixture), and single frames of other sizes / contents with the batched path
(other kernels of the same families: must be bit-identical).   usage: python tools/bs1_soak.py [calls] [quick]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepcharuco_amd import weights as W, workload as WL  # noqa: E402
from deepcharuco_amd.inference import infer_batch, infer_image  # noqa: E402
from deepcharuco_amd.models.net import dcModel, lModel  # noqa: E402
from deepcharuco_amd.models.refinenet import RefineNet, lRefineNet  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
quick = len(sys.argv) > 2 and sys.argv[2] == "quick"
dev = torch.device("cuda", 0)
bad = 0
for fxn in ("img7412_240x320.npz", "img7412_diverse_240x320.npz"):
    fx = np.load(os.path.join(ROOT, "tests", "golden", fxn))
    meta = json.loads(str(fx["meta"]))
    sd = W.synthetic_state_dict("detector", meta["wseed"], meta["n_ids"])
    for k in ("convDb.bias",):
        if (k.replace(".", "_")) in fx.files:
            sd[k] = fx[k.replace(".", "_")].astype(np.float32).copy()
    if "convDb_weight" in fx.files:
        sd["convDb.weight"] = fx["convDb_weight"].astype(np.float32).copy()
    dc = lModel(dcModel(16, sd, dev))
    rn = lRefineNet(RefineNet(W.synthetic_state_dict("refinenet", meta["wseed"] + 1), dev))
    bgr = np.ascontiguousarray(fx["bgr_image"])
    exp = fx["final_rn"]
    kp, _ = infer_image(bgr, 16, dc, rn, device="cuda")
    if not (kp.shape == exp.shape and np.array_equal(kp, exp)):
        print(f"{fxn}: the weights of this fixture are not reproduced by this tool's recipe -- comparing every call with the FIRST call instead")
        exp = kp.copy()
    t0 = time.time()
    nb = 0
    for i in range(calls):
        kp, _ = infer_image(bgr, 16, dc, rn, device="cuda")
        if kp.shape != exp.shape or not np.array_equal(kp, exp):
            nb += 1
    print(f"{fxn}: {calls} graph replays, {nb} results differ, {calls / (time.time() - t0):.0f} calls/s, {exp.shape[0]} corners")
    bad += nb
# other single frames: infer_image (small-launch kernels, graph) vs the same frame inside a batch of 8 (bs=32-style kernels)
rn = lRefineNet(RefineNet(W.synthetic_state_dict("refinenet", 1235), dev))
for (h, w) in (((240, 320), (120, 160), (250, 330)) if quick else ((240, 320), (480, 640), (120, 160), (250, 330), (960, 1280))):
    frames = W.synthetic_frames("board", 77, 8, h, w)
    # dust-bin bias calibrated to ~16 firing cells per frame at this size (so that RefineNet, too, runs its one-frame launches)
    sd = WL.calibrate_dustbin(W.synthetic_state_dict("detector", 1234), torch.from_numpy(frames).to(dev), dev, diverse_ids=True)
    dc = lModel(dcModel(16, sd, dev))
    batch = infer_batch(frames, 16, dc, rn)
    nb = 0
    for rep in range(3):
        for b in range(8):
            bgr = np.repeat(frames[b][:, :, None], 3, axis=2)
            kp, _ = infer_image(bgr, 16, dc, rn, device="cuda")
            if kp.shape != batch[b].shape or not np.array_equal(kp, batch[b]):
                nb += 1
    print(f"{h}x{w}: 24 single-frame calls vs the batched path: {nb} differ (corners per frame {[int(x.shape[0]) if x.ndim == 2 else 0 for x in batch]})")
    bad += nb
print("SOAK", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
