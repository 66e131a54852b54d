#!/usr/bin/env python3
"""Where a bs=1 infer_image call's time goes (the reference's own protocol, src/benchmark.py:37-53), from a rocprofv3 kernel trace.

  rocprofv3 --kernel-trace -d out -o bs1 -- python tools/bs1_timeline.py run [calls]      (on the MI355X box)
  python tools/bs1_timeline.py analyze out/.../bs1_results.db                              (anywhere)

`run` replays the captured hipGraph `calls` times on the reference's benchmark image (tests/golden/img7412_240x320.npz).
`analyze` cuts the kernel records into calls (a call starts with the detector's conv1a) and prints, per call (median over the steady
calls): the time inside kernels, the gaps between consecutive kernels of the call (node-to-node latency of the graph), the gap
between the call's last kernel and the next call's first (host: sync, unpack, staging copy, graph launch), and the per-kernel table."""
import json
import os
import sqlite3
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(calls):
    import numpy as np
    import torch
    from deepcharuco_amd import weights as W
    from deepcharuco_amd.inference import infer_image
    from deepcharuco_amd.models.net import dcModel, lModel
    from deepcharuco_amd.models.refinenet import RefineNet, lRefineNet
    dev = torch.device("cuda", 0)
    fx = np.load(os.path.join(ROOT, "tests", "golden", "img7412_240x320.npz"))
    meta = json.loads(str(fx["meta"]))
    sd_dc = W.synthetic_state_dict("detector", meta["wseed"], meta["n_ids"])
    sd_dc["convDb.bias"] = fx["convDb_bias"].astype(np.float32).copy()
    dc = lModel(dcModel(16, sd_dc, dev))
    rn = lRefineNet(RefineNet(W.synthetic_state_dict("refinenet", meta["wseed"] + 1), dev))
    bgr = np.ascontiguousarray(fx["bgr_image"])
    for _ in range(10):
        kp, _ = infer_image(bgr, 16, dc, rn, device="cuda")
    t0 = time.time()
    for _ in range(calls):
        kp, _ = infer_image(bgr, 16, dc, rn, device="cuda")
    el = time.time() - t0
    print(f"bs1: {calls / el:.1f} calls/s, {1e6 * el / calls:.1f} us/call, corners {kp.shape[0]}, same as the reference's: {np.array_equal(kp, fx['final_rn'])}")


def analyze(path):
    import numpy as np
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = list(cur.execute(f"select {name_col}, start, end from kernels order by start"))
    calls, curc = [], None
    for n, s, e in rows:
        if n.startswith("void dcx_conv1_kernel") or n.startswith("dcx_conv1_kernel"):
            if curc:
                calls.append(curc)
            curc = []
        if curc is not None:
            curc.append((n, s, e))
    if curc:
        calls.append(curc)
    sizes = [len(c) for c in calls]
    typical = int(np.median(sizes))
    steady = [i for i in range(20, len(calls) - 1) if len(calls[i]) == typical and len(calls[i + 1]) == typical]
    busy = np.array([sum(e - s for _, s, e in calls[i]) for i in steady]) / 1e3
    intra = np.array([sum(calls[i][k + 1][1] - calls[i][k][2] for k in range(typical - 1)) for i in steady]) / 1e3
    inter = np.array([calls[i + 1][0][1] - calls[i][-1][2] for i in steady]) / 1e3
    period = np.array([calls[i + 1][0][1] - calls[i][0][1] for i in steady]) / 1e3
    print(f"# {path}: {len(calls)} calls, {typical} kernels per call, {len(steady)} steady calls analysed (us, median [min .. max])")
    for nm, a in (("call period (first kernel to next call's first kernel)", period), ("inside kernels", busy),
                  ("gaps between the call's kernels (graph node to node)", intra), ("last kernel -> next call's first kernel (host)", inter)):
        print(f"  {nm:62s} {np.median(a):8.1f}  [{a.min():8.1f} .. {a.max():8.1f}]")
    print(f"  {'per kernel of a call, in launch order':62s}   dur_us   gap_before_us")
    for k in range(typical):
        d = np.median([calls[i][k][2] - calls[i][k][1] for i in steady]) / 1e3
        g = np.median([calls[i][k][1] - calls[i][k - 1][2] for i in steady]) / 1e3 if k else float("nan")
        print(f"    {k:2d} {calls[steady[0]][k][0][:100]:100s} {d:8.2f} {g:8.2f}")


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "analyze":
        analyze(sys.argv[2])
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 300)
