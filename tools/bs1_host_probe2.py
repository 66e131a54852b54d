#!/usr/bin/env python3
"""Where the host time of one GraphedPipeline.run goes: the body of the ROUND-5 form of run() (device context manager,
torch.cuda.current_stream().synchronize(), generic unpack) with a timer after every statement, beside the product's run() and
infer_image on the same pipeline (profiles/experiments/r06_bs1_split_positions.txt 5.).   usage: python tools/bs1_host_probe2.py"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from deepcharuco_amd import weights as W
from deepcharuco_amd.graph import cached_pipeline
from deepcharuco_amd.inference import infer_image, unpack_results
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
for _ in range(50): infer_image(bgr, 16, dc, rn, device="cuda")
pipe = cached_pipeline(16, dc, rn, 240, 320, bgr=True)
frames = bgr[None]
N = 3000
T = {}
def acc(k, t0):
    t1 = time.perf_counter_ns(); T[k] = T.get(k, 0) + (t1 - t0); return t1
tot0 = time.perf_counter_ns()
for _ in range(N):
    t = time.perf_counter_ns()
    with pipe._lock:
        t = acc("lock", t)
        graph, in_np, out_np = pipe.graph, pipe._in_np, pipe._out_np
        ok = pipe._models_unchanged()
        t = acc("models_unchanged", t)
        bad = frames.shape != in_np.shape or frames.dtype != np.uint8
        t = acc("shape check", t)
        np.copyto(in_np, frames)
        t = acc("copyto", t)
        with torch.cuda.device(pipe.dev):
            t = acc("device ctx enter", t)
            graph.replay()
            t = acc("replay", t)
            cs = torch.cuda.current_stream()
            t = acc("current_stream", t)
            cs.synchronize()
            t = acc("synchronize", t)
        t = acc("device ctx exit", t)
        res, counts = unpack_results(out_np, pipe.batch, pipe.pool, True)
        t = acc("unpack", t)
        need = int(counts.sum(dtype=np.int64))
        t = acc("need", t)
    t = acc("unlock", t)
tot = time.perf_counter_ns() - tot0
print(f"inline run(): {tot / N / 1e3:.1f} us per call")
for k, v in T.items(): print(f"  {k:20s} {v / N / 1e3:7.2f} us")
t0 = time.perf_counter_ns()
for _ in range(N): pipe.run(frames)
print(f"pipe.run: {(time.perf_counter_ns() - t0) / N / 1e3:.1f} us")
t0 = time.perf_counter_ns()
for _ in range(N): infer_image(bgr, 16, dc, rn, device="cuda")
print(f"infer_image: {(time.perf_counter_ns() - t0) / N / 1e3:.1f} us")
