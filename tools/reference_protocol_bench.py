#!/usr/bin/env python3
"""The reference's own measurement (/root/reference/src/benchmark.py:29-53) on the drop-in API:
load_models(ckpt, ckpt) -> 5 warm-up + N timed infer_image(img_bgr, n_ids, deepc, refinenet, draw_pred=False)
calls on ONE 320x240 image, fps = N / elapsed (host clock, includes BGR->gray, H2D, both nets, D2H, sort).
Synthetic seeded checkpoints are written in Lightning's format first (the published ones are not in the mount)."""
import argparse, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcharuco_amd import weights as W
from deepcharuco_amd.inference import load_models, infer_image


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-n", type=int, default=500)
    args = ap.parse_args()
    d = tempfile.mkdtemp()
    sd = W.synthetic_state_dict("detector", 1234)
    sd["convDb.bias"][16] += np.float32(3.8)      # ~16 corners on this frame (same calibration idea as bench.py)
    W.save_lightning_style_checkpoint(os.path.join(d, "dc.ckpt"), sd)
    W.save_lightning_style_checkpoint(os.path.join(d, "rn.ckpt"), W.synthetic_state_dict("refinenet", 1235))
    deepc, refinenet = load_models(os.path.join(d, "dc.ckpt"), os.path.join(d, "rn.ckpt"), n_ids=16, device="cuda")
    gray = W.synthetic_frames("board", 1000, 1, 240, 320)[0]
    img = np.repeat(gray[..., None], 3, axis=2)
    for _ in range(5):
        kp, _ = infer_image(img, 16, deepc, refinenet, draw_pred=False, device="cuda")
    t = time.time()
    for _ in range(args.n):
        kp, _ = infer_image(img, 16, deepc, refinenet, draw_pred=False, device="cuda")
    dt = time.time() - t
    print(f"reference protocol (bs=1 infer_image loop, {args.n} calls): {args.n / dt:.1f} fps, "
          f"{1e3 * dt / args.n:.3f} ms/call, {kp.shape[0]} corners/frame")
    # where the time goes on the host
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(100):
        infer_image(img, 16, deepc, refinenet, draw_pred=False, device="cuda")
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)


if __name__ == "__main__":
    main()
