#!/usr/bin/env python3
"""Why does FrameStream(compute_streams=2) not reach what two plain streams reach (bench.py cfg2_two_batches_in_flight)?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcharuco_amd import _lib, weights as W, workload as WL
from deepcharuco_amd.inference import infer_batch_device, launch_pipeline, packed_len
from deepcharuco_amd.stream import FrameStream
from deepcharuco_amd.models.net import dcModel, lModel
from deepcharuco_amd.models.refinenet import RefineNet, lRefineNet
dev = torch.device("cuda", 0)
B, NB = 32, 60
frames = W.synthetic_frames("board", 1000, B, 240, 320)
sd = WL.calibrate_dustbin(W.synthetic_state_dict("detector", 1234), torch.from_numpy(frames).to(dev), dev)
dc, rn = lModel(dcModel(16, sd, dev)), lRefineNet(RefineNet(W.synthetic_state_dict("refinenet", 1235), dev))

def fs_run(name, **kw):
    fs = FrameStream(16, dc, rn, batch=B, height=240, width=320, **kw)
    list(fs.run([frames] * 4))
    best = 0
    for _ in range(3):
        t = time.time()
        n = sum(len(item[1]) for item in fs.run([frames] * NB))
        best = max(best, n / (time.time() - t))
    print(f"{name:70s} {best:8.0f} fps", flush=True)

def plain(name, nstreams, h2d):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    pin = [torch.from_numpy(frames).pin_memory() for _ in range(nstreams)]
    # every stream's device buffer holds the frames (until round 5 these were torch.empty(): with h2d = False the kernels ran on
    # whatever the allocator handed out -- stream 1's batches had no corners at all, which is where "2 streams, no H2D: 11,855 fps" came from)
    d = [torch.from_numpy(frames).to(dev) for _ in range(nstreams)]
    n = packed_len(B, B * 64)
    out = [torch.empty((n,), dtype=torch.int32, device=dev) for _ in range(nstreams)]
    host = [torch.empty((n,), dtype=torch.int32).pin_memory() for _ in range(nstreams)]
    def step(i):
        k = i % nstreams
        with torch.cuda.stream(streams[k]):
            if h2d == "kernel":       # (needs the dcx_upload_u8 entry of profiles/experiments/r05_two_streams_with_uploads.md)
                _lib.check(_lib.lib().dcx_upload_u8(pin[k].data_ptr(), d[k].data_ptr(), d[k].numel(), _lib.current_stream()), "upload")
            elif h2d:
                d[k].copy_(pin[k], non_blocking=True)
            if h2d == "zerocopy":       # the kernels read the frames straight out of pinned host memory
                det, ref = dc.model, rn.model
                nb = _lib.lib().dcx_pipeline_workspace_bytes(det.handle, ref.handle, B, 240, 320, B * 64)
                launch_pipeline(det, ref, pin[k].data_ptr(), B, 240, 320, 1, 0, 16, B * 64, det._ws.get("pipe", dev, nb), out[k].data_ptr())
            else:
                infer_batch_device(d[k], 16, dc, rn, out=out[k], pool=B * 64)
            host[k].copy_(out[k], non_blocking=True)
    for i in range(6): step(i)
    torch.cuda.synchronize()
    best, enq = 0, 0
    for _ in range(3):
        t = time.time()
        for i in range(NB): step(i)
        te = time.time() - t                      # host time to ENQUEUE all batches (a host that blocks in here cannot run ahead)
        torch.cuda.synchronize()
        tot = time.time() - t
        if B * NB / tot > best:
            best, enq = B * NB / tot, te / tot
    print(f"{name:70s} {best:8.0f} fps   (host enqueue loop = {100 * enq:.0f} % of the wall time)", flush=True)

def host_synced(name, nstreams, ahead=2):
    """uploads on a copy stream `ahead` batches early; the HOST waits for the upload's event before it enqueues the batch's
    kernels, so the compute streams carry no copy command and no cross-stream event wait"""
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    copy = torch.cuda.Stream()
    ring = nstreams + ahead
    pin = [torch.from_numpy(frames).pin_memory() for _ in range(ring)]
    d = [torch.empty((B, 240, 320), dtype=torch.uint8, device=dev) for _ in range(ring)]
    ev_up = [torch.cuda.Event() for _ in range(ring)]
    ev_free = [torch.cuda.Event() for _ in range(ring)]
    n = packed_len(B, B * 64)
    out = [torch.empty((n,), dtype=torch.int32, device=dev) for _ in range(ring)]
    host = [torch.empty((n,), dtype=torch.int32).pin_memory() for _ in range(ring)]
    used = [False] * ring
    def upload(j):
        r = j % ring
        if used[r]:
            ev_free[r].synchronize()           # host-side too: the slot's previous batch has finished reading it
        with torch.cuda.stream(copy):
            d[r].copy_(pin[r], non_blocking=True)
            ev_up[r].record(copy)
    def run(total):
        for j in range(min(ahead, total)):
            upload(j)
        for i in range(total):
            if i + ahead < total:
                upload(i + ahead)
            r = i % ring
            ev_up[r].synchronize()
            with torch.cuda.stream(streams[i % nstreams]):
                infer_batch_device(d[r], 16, dc, rn, out=out[r], pool=B * 64)
                ev_free[r].record(streams[i % nstreams])
                host[r].copy_(out[r], non_blocking=True)
            used[r] = True
        torch.cuda.synchronize()
    run(8)
    best = 0
    for _ in range(3):
        t = time.time()
        run(NB)
        best = max(best, B * NB / (time.time() - t))
    print(f"{name:70s} {best:8.0f} fps", flush=True)

plain("plain loop, 1 stream, no H2D", 1, False)
plain("plain loop, 2 streams, no H2D (= bench two_batches_in_flight)", 2, False)
plain("plain loop, 2 streams, H2D on the same stream", 2, True)
plain("plain loop, 1 stream, H2D copy", 1, True)
plain("plain loop, 1 stream, zero-copy frames (pinned host memory)", 1, "zerocopy")
plain("plain loop, 2 streams, zero-copy frames (pinned host memory)", 2, "zerocopy")
host_synced("copy stream 2 ahead, HOST waits for the upload, 2 compute streams", 2)
host_synced("copy stream 2 ahead, HOST waits for the upload, 1 compute stream", 1)
host_synced("copy stream 3 ahead, HOST waits for the upload, 2 compute streams", 2, ahead=3)
fs_run("FrameStream depth=3 compute_streams=2", depth=3, compute_streams=2)
fs_run("FrameStream depth=3 compute_streams=2 h2d_on_compute", depth=3, compute_streams=2, h2d_on_compute=True)
fs_run("FrameStream depth=4 compute_streams=2 h2d_on_compute", depth=4, compute_streams=2, h2d_on_compute=True)
fs_run("FrameStream depth=2 compute_streams=2 h2d_on_compute", depth=2, compute_streams=2, h2d_on_compute=True)
fs_run("FrameStream depth=2 compute_streams=1", depth=2, compute_streams=1)
fs_run("FrameStream depth=2 compute_streams=1 h2d_on_compute", depth=2, compute_streams=1, h2d_on_compute=True)
