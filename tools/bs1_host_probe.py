#!/usr/bin/env python3
"""Host-side cost of one bs=1 infer_image call (the reference's protocol, src/benchmark.py:37-53), piece by piece: staging copy,
graph launch, completion wait (stream.synchronize / event.query spin / hipStreamQuery spin), unpack.   usage: python tools/bs1_host_probe.py"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepcharuco_amd import weights as W  # noqa: E402
from deepcharuco_amd.graph import cached_pipeline  # noqa: E402
from deepcharuco_amd.inference import infer_image, unpack_results  # noqa: E402
from deepcharuco_amd.models.net import dcModel, lModel  # noqa: E402
from deepcharuco_amd.models.refinenet import RefineNet, lRefineNet  # noqa: E402

dev = torch.device("cuda", 0)
fx = np.load(os.path.join(ROOT, "tests", "golden", "img7412_240x320.npz"))
meta = json.loads(str(fx["meta"]))
sd_dc = W.synthetic_state_dict("detector", meta["wseed"], meta["n_ids"])
sd_dc["convDb.bias"] = fx["convDb_bias"].astype(np.float32).copy()
dc = lModel(dcModel(16, sd_dc, dev))
rn = lRefineNet(RefineNet(W.synthetic_state_dict("refinenet", meta["wseed"] + 1), dev))
bgr = np.ascontiguousarray(fx["bgr_image"])
for _ in range(20):
    infer_image(bgr, 16, dc, rn, device="cuda")
pipe = cached_pipeline(16, dc, rn, 240, 320, bgr=True)
N = 2000
frames = bgr[None]


def timeit(fn, n=N):
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return 1e6 * (time.perf_counter() - t0) / n


hip = C.CDLL("libamdhip64.so")
print(f"infer_image total                    {timeit(lambda: infer_image(bgr, 16, dc, rn, device='cuda')):8.1f} us")
print(f"pipe.run total                       {timeit(lambda: pipe.run(frames)):8.1f} us")
print(f"np.copyto(pinned, frame)             {timeit(lambda: np.copyto(pipe._in_np, frames)):8.1f} us")
torch.cuda.synchronize()
s = pipe.stream


def replay_sync():
    pipe.graph.replay()
    torch.cuda.current_stream().synchronize()


print(f"graph.replay + stream.synchronize    {timeit(replay_sync):8.1f} us")
t_launch = []


def replay_only():
    t0 = time.perf_counter()
    pipe.graph.replay()
    t_launch.append(time.perf_counter() - t0)
    torch.cuda.current_stream().synchronize()


timeit(replay_only)
print(f"  graph.replay() host call alone     {1e6 * np.median(t_launch):8.1f} us")
cs = torch.cuda.current_stream()
ev = torch.cuda.Event()


def replay_evspin():
    pipe.graph.replay()
    ev.record(cs)
    while not ev.query():
        pass


print(f"graph.replay + event.query spin      {timeit(replay_evspin):8.1f} us")
sp = C.c_void_p(cs.cuda_stream)
hip.hipStreamQuery.argtypes = [C.c_void_p]


def replay_sqspin():
    pipe.graph.replay()
    while hip.hipStreamQuery(sp) != 0:
        pass


print(f"graph.replay + hipStreamQuery spin   {timeit(replay_sqspin):8.1f} us")
out = pipe._out_np
print(f"unpack_results(B=1)                  {timeit(lambda: unpack_results(out, 1, pipe.pool, True)):8.1f} us")
print(f"with torch.cuda.device(dev): pass    {timeit(lambda: torch.cuda.device(dev).__enter__()):8.1f} us")
from deepcharuco_amd.graph import graphs_usable  # noqa: E402
print(f"graphs_usable()                      {timeit(graphs_usable):8.1f} us")
print(f"cached_pipeline lookup               {timeit(lambda: cached_pipeline(16, dc, rn, 240, 320, bgr=True)):8.1f} us")
