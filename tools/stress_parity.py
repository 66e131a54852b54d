"""Parity stress at scale (not part of the test suite): >= 20,000 seeded frames over four resolutions through the HIP path vs
the oracle, with the near-tie structure made explicit.

For EVERY 8x8 cell the oracle decides, the top-2 margin of its 65-way `loc` and its 17-way `ids` logits is bucketed, and the
HIP / oracle arg-max disagreements are counted per bucket; the same for the 4,096-way RefineNet heat-map arg-max of the
corners the oracle finds (every 4th frame), and every 4th frame is also compared end to end (`infer_batch` row arrays).
The largest |HIP logit - oracle logit| is recorded per resolution: the arg-max policy of tests/test_gpu_parity.py (MARGIN)
must be >= 2x that number, and this table is the evidence that nothing disagrees above it.

Round 5 -- FOUR comparisons against the same oracle pass (batched, N threads), bucketed by ITS top-2 margins:
  hip_default   the product path (Winograd families)                     -- what ships
  hip_direct    the product path in deterministic mode (direct family: every multiply-add of the layers as written)
                                                                          -- what Winograd costs in parity is (hip_default - hip_direct)
  oracle_1thr   the SAME oracle code on the SAME tensors with torch.set_num_threads(1)
                                                                          -- the reference's own noise floor between thread counts
  oracle_bs1    the oracle one frame per call (the reference's real protocol, inference.py:32-70) vs the batched pass, same threads
                                                                          -- the reference's own noise floor between batch sizes
"statistical parity" is defensible exactly as far as hip_default is at or below the oracle_* columns.

The oracle side (frame rendering, torch-CPU detector / RefineNet) runs in worker processes; the main process owns the GPU and
compares on the device.      usage (MI355X):   python tools/stress_parity.py [frames=20000] [workers=14] [threads=16]
Writes gpurun_out/stress_parity_summary.json (copy to profiles/) and prints the table."""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N_IDS = 16
EDGES = [0.0, 1e-7, 1e-6, 3e-6, 1e-5, 3e-5, 1e-4, 1e-3, float("inf")]
SHM = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
# resolution -> (share of the frame budget, frames per chunk)
PLAN = {(64, 96): (0.40, 256), (120, 160): (0.30, 128), (240, 320): (0.25, 48), (480, 640): (0.05, 12)}


def oracle_chunk(spec):
    """Worker: render the chunk's frames, run the oracle, write everything the GPU side needs to one .npz in shared memory."""
    import torch
    cid, h, w, n, wseed, threads, do_f64, do_bs1 = spec
    torch.set_num_threads(threads)
    from deepcharuco_amd import weights as W
    from oracle import deepcharuco_oracle as O
    frames = np.concatenate([W.synthetic_frames("noise", 700000 + 1000 * cid, n // 2, h, w),
                             W.synthetic_frames("board", 800000 + 1000 * cid, n - n // 2, h, w)])
    sd_dc = W.synthetic_state_dict("detector", wseed, N_IDS)
    sd_rn = W.synthetic_state_dict("refinenet", wseed + 1)
    x = torch.from_numpy(np.stack([O.pre_bgr_image(f) for f in frames]))          # (n,1,h,w)
    t_dc = O.to_torch_state_dict(sd_dc)
    loc, ids = O.detector_forward(t_dc, x)
    la = loc.argmax(1)
    if cid % 3 == 1:
        # a third of the chunks: ids-head biases equalised per class on the oracle's logits, so the firing cells carry all 16 ids
        shift, _ = W.diverse_ids_bias_shift(np.moveaxis(ids.numpy(), 1, 0), la.numpy(), N_IDS, 12 * n)
        sd_dc["convDb.bias"][:N_IDS] = (sd_dc["convDb.bias"][:N_IDS] + shift).astype(np.float32)
        loc, ids = O.detector_forward(O.to_torch_state_dict(sd_dc), x)
        la = loc.argmax(1)
    # dust-bin bias so that ~12 cells per frame fire (same rule as workload.calibrate_dustbin, on the oracle's logits) ...
    m = (ids[:, :N_IDS].max(1).values - ids[:, N_IDS])
    m = torch.where(la == 64, torch.tensor(-1e30), m).flatten().sort(descending=True).values
    k = 12 * n
    if cid % 2 == 0:
        delta = np.float32((m[k - 1] + m[k]) / 2)
    else:
        # ... every second chunk: NOT the midpoint of the gap -- the threshold lands within +-2e-4 of a cell's own margin, so that
        # the fire / no-fire decision is sampled where it is close
        rng = np.random.default_rng(cid)
        delta = np.float32(float(m[k - 1 + int(rng.integers(-3, 4))]) + rng.uniform(-2e-4, 2e-4))
    sd_dc["convDb.bias"][N_IDS] = np.float32(sd_dc["convDb.bias"][N_IDS] + delta)
    t_dc = O.to_torch_state_dict(sd_dc)
    loc, ids = O.detector_forward(t_dc, x)
    t_rn = O.to_torch_state_dict(sd_rn)
    # every 4th frame: the whole path (key-points, patches, heat-maps, final rows)
    sub = list(range(0, n, 4))
    kp_all, fr_idx, heat_idx, heat_margin, finals = [], [], [], [], []
    for b in sub:
        kp, idf = O.pred_to_keypoints(loc[b:b + 1], ids[b:b + 1], N_IDS)
        finals.append(O.infer_image(None, N_IDS, t_dc, t_rn, gray=frames[b]))
        if kp.shape[0] == 0:
            continue
        patches = O.extract_patches(x[b], kp)
        heat = O.refinenet_forward(t_rn, patches[:, None])[:, 0].reshape(kp.shape[0], -1)
        top = torch.topk(heat, 2, dim=1)
        kp_all.append(kp.numpy()); fr_idx.append(np.full(kp.shape[0], b)); heat_idx.append(top.indices[:, 0].numpy())
        heat_margin.append((top.values[:, 0] - top.values[:, 1]).numpy())
    cat = lambda l, dt: np.concatenate(l).astype(dt) if l else np.zeros((0,), dt)
    kp_cat = np.concatenate(kp_all).astype(np.int64) if kp_all else np.zeros((0, 2), np.int64)
    kf_cat = cat(fr_idx, np.int64)

    def heat_argmax_on(kp_np, kf_np):
        """flat arg-max of the oracle's heat-maps on GIVEN key-points (the N-thread pass's), under the current thread count"""
        out = []
        for b in np.unique(kf_np):
            k = torch.from_numpy(kp_np[kf_np == b])
            heat = O.refinenet_forward(t_rn, O.extract_patches(x[int(b)], k)[:, None])[:, 0].reshape(k.shape[0], -1)
            out.append(heat.argmax(1).numpy())
        return cat(out, np.int64)

    # ---- the reference against ITSELF (1): one frame per call, same thread count (inference.py:32-70 is per frame) -- every 4th frame
    if do_bs1:
        loc_b1 = torch.cat([O.detector_forward(t_dc, x[b:b + 1])[0] for b in sub])
        ids_b1 = torch.cat([O.detector_forward(t_dc, x[b:b + 1])[1] for b in sub])
    else:
        loc_b1 = ids_b1 = torch.zeros(0)
    # ---- exact arithmetic: the same graph in float64 on the same float32 inputs and weights ("truth" for BOTH fp32 evaluations)
    # (in slices of <= 1.3 M pixels: float64 activations of a whole chunk would be 2x the fp32 pass's memory in every worker)
    if do_f64:
        t64 = {k_: v.double() for k_, v in t_dc.items()}
        step = max(1, (1 << 20) * 5 // 4 // (h * w))
        parts = [O.detector_forward(t64, x[i:i + step].double()) for i in range(0, n, step)]
        loc64, ids64 = torch.cat([p_[0] for p_ in parts]), torch.cat([p_[1] for p_ in parts])
        del parts, t64
    else:
        loc64 = ids64 = torch.zeros(0)
    # ---- the reference against ITSELF (2): the same tensors, one thread
    torch.set_num_threads(1)
    loc_1t, ids_1t = O.detector_forward(t_dc, x)
    heat_idx_1t = heat_argmax_on(kp_cat, kf_cat) if kp_cat.shape[0] else np.zeros((0,), np.int64)
    finals_1t = [O.infer_image(None, N_IDS, t_dc, t_rn, gray=frames[b]) for b in sub]
    torch.set_num_threads(threads)
    path = os.path.join(SHM, f"dcx_stress_{os.getpid()}_{cid}.npz")
    np.savez(path, frames=frames, loc=loc.numpy(), ids=ids.numpy(), convDb_bias=sd_dc["convDb.bias"].astype(np.float32),
             kp=kp_cat, kp_frame=kf_cat,
             heat_idx=cat(heat_idx, np.int64), heat_margin=cat(heat_margin, np.float32), sub=np.array(sub),
             finals=np.array(finals, dtype=object), wseed=wseed, cid=cid,
             loc_1t=loc_1t.numpy(), ids_1t=ids_1t.numpy(), heat_idx_1t=heat_idx_1t, finals_1t=np.array(finals_1t, dtype=object),
             loc_b1=loc_b1.numpy(), ids_b1=ids_b1.numpy(), loc64=loc64.numpy(), ids64=ids64.numpy())
    return path


def run(total=20000, workers=14, threads=16, do_f64=True, do_bs1=True, do_direct=True, plan=None, verbose=True,
        summary_name="stress_parity_summary.json"):
    """The whole comparison; returns the summary dict (also written to gpurun_out/<summary_name>).  tests/test_gpu_parity.py runs a
    bounded slice of it (no float64 / one-frame-per-call / direct-family columns) with hard gates; `main` the full table."""
    import torch
    plan = PLAN if plan is None else plan
    specs, cid = [], 0
    for (h, w), (share, per) in plan.items():
        for _ in range(int(np.ceil(total * share / per))):
            specs.append((cid, h, w, per, 5000 + 7 * (cid % 61), threads, do_f64, do_bs1))        # 61 different weight sets
            cid += 1
    specs.sort(key=lambda s: -s[1] * s[2] * s[3])                                   # biggest chunks first
    from deepcharuco_amd import weights as W
    from deepcharuco_amd.inference import infer_batch
    from deepcharuco_amd.models.model_utils import extract_patches, pre_bgr_image
    from deepcharuco_amd.models.net import dcModel, lModel
    from deepcharuco_amd.models.refinenet import RefineNet, lRefineNet
    from deepcharuco_amd.inference import set_deterministic
    dev = torch.device("cuda", 0)
    edges = torch.tensor(EDGES[1:-1], device=dev)
    nb = len(EDGES) - 1
    zero = lambda: {"cells": np.zeros(nb, np.int64), "disagree": np.zeros(nb, np.int64)}
    COLS = ("hip_default", "hip_direct", "oracle_1thr", "oracle_bs1")
    # second group: everybody against EXACT arithmetic (the oracle's graph in float64 on the same fp32 inputs / weights), bucketed by
    # the float64 pass's margins -- how far is each fp32 evaluation from the truth they both approximate?
    COLS64 = ("oracle_f32_vs_f64", "hip_default_vs_f64", "hip_direct_vs_f64")
    col = {c: {"stats": {"loc": zero(), "ids": zero(), "heat": zero(), "fire": zero()}, "max_abs_logit_diff": 0.0, "sum_abs": 0.0,
               "count": 0, "cells_decided_differently": 0, "e2e_frames": 0, "e2e_bad": 0, "e2e_corners": 0, "frames": 0} for c in COLS + COLS64}
    ids_hist = np.zeros(N_IDS, np.int64)
    per_res = {}
    frames_done = chunks_done = 0
    t0 = time.time()

    def compare_logits(c, g_loc, g_ids, o_loc, o_ids):
        """one column: candidate logits (g_*) against the N-thread batched oracle's (o_*), bucketed by the ORACLE's margins"""
        C = col[c]
        C["frames"] += g_loc.shape[0]
        for name, g, o in (("loc", g_loc, o_loc), ("ids", g_ids, o_ids)):
            diff = (g.to(o.dtype) - o).abs()
            C["max_abs_logit_diff"] = max(C["max_abs_logit_diff"], float(diff.max()))
            C["sum_abs"] += float(diff.sum()); C["count"] += diff.numel()
            top = torch.topk(o, 2, dim=1).values
            bucket = torch.bucketize((top[:, 0] - top[:, 1]).flatten(), edges.to(o.dtype), right=True)
            bad = (g.argmax(1) != o.argmax(1)).flatten()
            C["stats"][name]["cells"] += torch.bincount(bucket, minlength=nb).cpu().numpy()
            C["stats"][name]["disagree"] += torch.bincount(bucket[bad], minlength=nb).cpu().numpy()
        # the decision the reference takes per cell: (fires?, id, 8x8 offset)
        o_la, o_ia, g_la, g_ia = o_loc.argmax(1), o_ids.argmax(1), g_loc.argmax(1), g_ids.argmax(1)
        fire_o, fire_g = (o_la != 64) & (o_ia != N_IDS), (g_la != 64) & (g_ia != N_IDS)
        C["cells_decided_differently"] += int(((fire_o != fire_g) | (fire_o & ((o_la != g_la) | (o_ia != g_ia)))).sum())
        # fire / no-fire by the distance of the oracle's decision from its threshold (cells whose loc head fires)
        fm = (o_ids[:, :N_IDS].max(1).values - o_ids[:, N_IDS]).abs()[o_la != 64]
        bucket = torch.bucketize(fm, edges.to(fm.dtype), right=True)
        C["stats"]["fire"]["cells"] += torch.bincount(bucket, minlength=nb).cpu().numpy()
        C["stats"]["fire"]["disagree"] += torch.bincount(bucket[(fire_o != fire_g)[o_la != 64]], minlength=nb).cpu().numpy()
        return fire_o, o_ia

    def compare_heat(c, idx, z):
        bucket = np.digitize(z["heat_margin"], EDGES[1:-1], right=False)
        col[c]["stats"]["heat"]["cells"] += np.bincount(bucket, minlength=nb)
        col[c]["stats"]["heat"]["disagree"] += np.bincount(bucket[idx != z["heat_idx"]], minlength=nb)

    def compare_e2e(c, res, finals):
        for a, e in zip(res, finals):
            e = np.asarray(e, dtype=np.float64) if np.asarray(e).size else np.array([])
            a = np.asarray(a, dtype=np.float64) if np.asarray(a).size else np.array([])
            col[c]["e2e_frames"] += 1
            col[c]["e2e_corners"] += 0 if e.ndim == 1 else e.shape[0]
            col[c]["e2e_bad"] += not (a.shape == e.shape and np.array_equal(a, e))

    label = [f"[{EDGES[i]:g}, {EDGES[i + 1]:g})" for i in range(nb)]

    def summarize():
        """the JSON summary of what has been compared so far (also written every 25 chunks: a cut-off run keeps its evidence)"""
        res = {}
        for c, C in col.items():
            res[c] = {"frames": C["frames"], "max_abs_logit_diff": C["max_abs_logit_diff"],
                      "mean_abs_logit_diff": C["sum_abs"] / max(1, C["count"]), "cells_decided_differently": C["cells_decided_differently"],
                      "histogram": {k: {"decided": v["cells"].tolist(), "differs": v["disagree"].tolist()} for k, v in C["stats"].items()},
                      "end_to_end": {"frames": C["e2e_frames"], "corners": C["e2e_corners"], "mismatched_frames": C["e2e_bad"]}}
        out = {"frames": frames_done, "seconds": round(time.time() - t0, 1), "per_resolution": per_res, "oracle_threads": threads,
               "reference_pass": f"oracle, all frames of a chunk in one batch, {threads} threads",
               "columns": {"hip_default": "product path (Winograd families) vs the reference pass",
                           "hip_direct": "product path, deterministic mode (direct family) vs the reference pass",
                           "oracle_1thr": "the oracle itself with torch.set_num_threads(1) vs the reference pass",
                           "oracle_bs1": "the oracle one frame per call (every 4th frame; logits only) vs the reference pass",
                           "oracle_f32_vs_f64": "the reference pass itself vs EXACT arithmetic (the oracle's graph in float64 on the same fp32 inputs / weights)",
                           "hip_default_vs_f64": "product path vs exact arithmetic", "hip_direct_vs_f64": "product path, direct family, vs exact arithmetic"},
               "firing_cells_per_id": ids_hist.tolist(),
               "weight_sets": "61 seeds; a third of the chunks with the ids-head biases equalised per class (all 16 ids fire); every second chunk "
                              "with the dust-bin threshold within +-2e-4 of a cell's own margin (fire/no-fire sampled where it is close)",
               "buckets": label, "results": res}
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", summary_name), "w") as f:
            json.dump(out, f, indent=1)
        return out

    ctx = mp.get_context("spawn")
    with ctx.Pool(workers) as pool:
        for path in pool.imap_unordered(oracle_chunk, specs):
            z = np.load(path, allow_pickle=True)
            os.remove(path)
            frames = z["frames"]
            n, h, w = frames.shape
            sd_dc = W.synthetic_state_dict("detector", int(z["wseed"]), N_IDS)
            sd_dc["convDb.bias"] = z["convDb_bias"].astype(np.float32).copy()      # equalised ids biases (a third of the chunks) + dust-bin
            sd_rn = W.synthetic_state_dict("refinenet", int(z["wseed"]) + 1)
            det, ref = dcModel(N_IDS, sd_dc, dev), RefineNet(sd_rn, dev)
            d_frames = torch.from_numpy(frames).to(dev)
            o_loc, o_ids = torch.from_numpy(z["loc"]).to(dev), torch.from_numpy(z["ids"]).to(dev)
            r = per_res.setdefault(f"{w}x{h}", {"frames": 0})
            r["frames"] += n
            kp, kf, sub = z["kp"], z["kp_frame"], z["sub"]
            patches = None
            if kp.shape[0]:
                xs = torch.stack([torch.from_numpy(pre_bgr_image(frames[b])) for b in sub]).to(dev)       # (S,1,h,w)
                pos = {int(b): i for i, b in enumerate(sub)}
                patches = torch.cat([extract_patches(xs[pos[int(b)]], torch.from_numpy(kp[kf == b]).to(dev)) for b in np.unique(kf)])
            if do_f64:
                t_loc, t_ids = torch.from_numpy(z["loc64"]).to(dev), torch.from_numpy(z["ids64"]).to(dev)
                compare_logits("oracle_f32_vs_f64", o_loc, o_ids, t_loc, t_ids)
            for c, direct in ((("hip_default", False), ("hip_direct", True)) if do_direct else (("hip_default", False),)):
                set_deterministic(direct)
                got = det.forward_u8(d_frames)
                if do_f64:
                    compare_logits(c + "_vs_f64", got["loc"], got["ids"], t_loc, t_ids)
                fire_o, o_ia = compare_logits(c, got["loc"], got["ids"], o_loc, o_ids)
                if c == "hip_default":
                    ids_hist += torch.bincount(o_ia[fire_o], minlength=N_IDS + 1)[:N_IDS].cpu().numpy()
                if patches is not None:          # RefineNet on the oracle's key-points (every 4th frame)
                    _, cor = ref.infer_patches(patches, torch.from_numpy(kp).to(dev))
                    compare_heat(c, (cor[:, 1] * 64 + cor[:, 0]).cpu().numpy(), z)
                compare_e2e(c, infer_batch(frames[sub], N_IDS, lModel(det), lRefineNet(ref), kmax=128), z["finals"])     # end to end (every 4th frame)
            if do_direct:
                set_deterministic(False)
            # the reference against itself: one thread; one frame per call
            compare_logits("oracle_1thr", torch.from_numpy(z["loc_1t"]).to(dev), torch.from_numpy(z["ids_1t"]).to(dev), o_loc, o_ids)
            if kp.shape[0]:
                compare_heat("oracle_1thr", z["heat_idx_1t"], z)
            compare_e2e("oracle_1thr", z["finals_1t"], z["finals"])
            if do_bs1:
                si = torch.from_numpy(np.asarray(sub)).to(dev)
                compare_logits("oracle_bs1", torch.from_numpy(z["loc_b1"]).to(dev), torch.from_numpy(z["ids_b1"]).to(dev), o_loc[si], o_ids[si])
            frames_done += n
            if verbose:
                print(f"[{time.time() - t0:6.0f}s] {frames_done:6d} frames  ({w}x{h} x{n})  arg-max disagreements (loc+ids / heat / e2e frames):  " +
                  "  ".join(f"{c} {int(col[c]['stats']['loc']['disagree'].sum() + col[c]['stats']['ids']['disagree'].sum())}/"
                            f"{int(col[c]['stats']['heat']['disagree'].sum())}/{col[c]['e2e_bad']}" for c in COLS), flush=True)
            del det, ref
            chunks_done += 1
            if chunks_done % 25 == 0:
                summarize()
    return summarize()


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    total = int(argv[0]) if len(argv) > 0 else 20000
    workers = int(argv[1]) if len(argv) > 1 else 14
    threads = int(argv[2]) if len(argv) > 2 else 16
    if "--slice" in sys.argv:
        # the bounded slice tests/test_gpu_parity.py::test_stress_parity_slice gates on: product path + the oracle's own 1-thread
        # noise floor against the N-thread reference pass; no float64 / one-frame-per-call / direct-family columns
        out = run(total, workers, threads, do_f64=False, do_bs1=False, do_direct=False, verbose=False, summary_name="stress_parity_slice.json")
        r = out["results"]
        print(json.dumps({c: {"frames": r[c]["frames"], "max_abs_logit_diff": r[c]["max_abs_logit_diff"],
                              "cells_decided_differently": r[c]["cells_decided_differently"], "end_to_end": r[c]["end_to_end"]}
                          for c in ("hip_default", "oracle_1thr")}))
        return
    out = run(total, workers, threads)
    COLS = ("hip_default", "hip_direct", "oracle_1thr", "oracle_bs1")
    COLS64 = ("oracle_f32_vs_f64", "hip_default_vs_f64", "hip_direct_vs_f64")
    label, nb, frames_done = out["buckets"], len(out["buckets"]), out["frames"]
    col = out["results"]
    print(f"\n{frames_done} frames; reference pass = oracle batched at {threads} threads; every column is compared with IT")
    print("firing cells per id:", out["firing_cells_per_id"])
    print(f"{'':24s}" + "".join(f"{c:>16s}" for c in COLS))
    print(f"{'max |logit diff|':24s}" + "".join(f"{col[c]['max_abs_logit_diff']:16.3e}" for c in COLS))
    print(f"{'mean |logit diff|':24s}" + "".join(f"{col[c]['mean_abs_logit_diff']:16.3e}" for c in COLS))
    print(f"{'cells decided differently':24s}" + "".join(f"{col[c]['cells_decided_differently']:16d}" for c in COLS))
    print(f"{'e2e frames differing':24s}" + "".join(f"{str(col[c]['end_to_end']['mismatched_frames']) + '/' + str(col[c]['end_to_end']['frames']):>16s}" for c in COLS))
    print("\nagainst EXACT arithmetic (float64 evaluation of the same graph; margins of the float64 pass):")
    print(f"{'':24s}" + "".join(f"{c:>22s}" for c in COLS64))
    print(f"{'max |logit diff|':24s}" + "".join(f"{col[c]['max_abs_logit_diff']:22.3e}" for c in COLS64))
    print(f"{'mean |logit diff|':24s}" + "".join(f"{col[c]['mean_abs_logit_diff']:22.3e}" for c in COLS64))
    print(f"{'loc+ids arg-max flips':24s}" + "".join(f"{sum(col[c]['histogram']['loc']['differs']) + sum(col[c]['histogram']['ids']['differs']):22d}" for c in COLS64))
    print(f"{'cells decided differently':24s}" + "".join(f"{col[c]['cells_decided_differently']:22d}" for c in COLS64))
    for name, title in (("loc", "loc 65-way arg-max"), ("ids", "ids 17-way arg-max"), ("heat", "RefineNet 4096-way arg-max"), ("fire", "fire / no-fire (by |ids max - dust-bin|)")):
        print(f"\n{title}: decided by the reference pass | differs in column")
        print(f"{'oracle margin':>22s} | {'decided':>12s} |" + "".join(f"{c:>14s}" for c in COLS))
        for i in range(nb):
            print(f"{label[i]:>22s} | {col['hip_default']['histogram'][name]['decided'][i]:12d} |" +
                  "".join(f"{col[c]['histogram'][name]['differs'][i]:14d}" for c in COLS))


if __name__ == "__main__":
    main()
