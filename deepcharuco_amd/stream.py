"""Asynchronous, double-buffered caller of the batch path.

The reference's callers feed one frame at a time and block on every stage
(/root/reference/src/pose_estimation.py:52-89, src/benchmark.py:45-53).  ``FrameStream`` keeps
``depth`` batches in flight: while batch i runs the HIP pipeline on the compute stream, batch i+1 is
copied host->device from pinned memory on a copy stream and batch i-1's packed corner list is
copied back; the host only ever waits on the oldest batch's completion event.  Results are the same
arrays ``infer_batch`` returns (frames are independent).  With ``pnp=dict(col_count=, row_count=, square_len=,
camera_matrix=, dist_coeffs=)`` the pipeline gets the reference callers' last stage too (pose_estimation.py:61-63):
when a batch is retired its frames' ``solve_pnp`` calls are submitted to host threads and run while the GPU works on
the next batches; ``run`` then yields ``(ticket, results, poses)``.
"""
from __future__ import annotations

import time
import warnings
from typing import Iterable, Iterator, List, Optional, Tuple

import numpy as np
import torch

from .inference import DEFAULT_KMAX, infer_batch, infer_batch_device, packed_len, solve_pnp_submit, unpack_results


class FrameStream:
    def __init__(self, dust_bin_ids: int, deepc, refinenet=None, batch: int = 32, height: int = 240,
                 width: int = 320, kmax: int = DEFAULT_KMAX, depth: int = 2, pnp: Optional[dict] = None,
                 compute_streams: int = 1, bgr: bool = False, h2d_on_compute: Optional[bool] = None):
        """``bgr=True``: the stream is fed (n,H,W,3) BGR frames, as the reference's callers hold them (pose_estimation.py:53-59);
        the colour conversion of inference.py:40 happens on the device inside the first layer's load.  ``kmax``: the AVERAGE
        number of corners per frame the buffers are sized for -- the corner pool of a batch holds ``batch * kmax`` corners and a
        single frame may use any share of it."""
        det = deepc.model if hasattr(deepc, "model") else deepc
        self.dev = det.device
        self.dust_bin_ids, self.deepc, self.refinenet = dust_bin_ids, deepc, refinenet
        self.batch, self.h, self.w, self.kmax, self.depth = batch, height, width, kmax, depth
        self.pool = batch * kmax
        self.bgr = bool(bgr)
        # upload on the batch's own compute stream instead of the shared copy stream: default with several compute streams (+5 % in
        # that mode); with one compute stream the separate copy stream is 0.7 % ahead (profiles/experiments/r05_two_streams_with_uploads.md)
        self.h2d_on_compute = compute_streams > 1 if h2d_on_compute is None else bool(h2d_on_compute)
        self.pnp = pnp
        if not (1 <= compute_streams <= depth):
            raise ValueError("compute_streams must be between 1 and depth")
        n_out = packed_len(batch, self.pool)
        shape = (batch, height, width, 3) if self.bgr else (batch, height, width)
        with torch.cuda.device(self.dev):
            self.copy_stream = torch.cuda.Stream()
            # compute_streams = 2: consecutive batches run the pipeline on alternating streams.  It gains nothing: rounds 2-5 quoted
            # "+7 ... +15 % when the frames are already in HBM", but that comparison gave the second stream lighter batches; with
            # equal work two streams are 1-4 % SLOWER than one (profiles/experiments/r05_batches_in_flight_equal_work.txt): every
            # conv launch already fills both workgroup slots of every CU.  The default stays 1; the option stays for callers whose
            # batches are small enough to leave CUs empty.  The pipeline scratch is per (model, stream)
            self.compute = [torch.cuda.Stream() for _ in range(compute_streams)] if compute_streams > 1 else None
            self.pin_in = [torch.empty(shape, dtype=torch.uint8).pin_memory() for _ in range(depth)]
            self.dev_in = [torch.empty(shape, dtype=torch.uint8, device=self.dev) for _ in range(depth)]
            self.dev_out = [torch.empty((n_out,), dtype=torch.int32, device=self.dev) for _ in range(depth)]
            self.pin_out = [torch.empty((n_out,), dtype=torch.int32).pin_memory() for _ in range(depth)]
            self.ev_h2d = [torch.cuda.Event() for _ in range(depth)]
            self.ev_free = [torch.cuda.Event() for _ in range(depth)]     # compute finished reading dev_in[s]
            self.ev_done = [torch.cuda.Event() for _ in range(depth)]
        self._pending: List[Optional[Tuple[int, int, np.ndarray]]] = [None] * depth   # (ticket, n_frames, host frames)
        self._ticket = 0

    def _collect(self, slot: int):
        ticket, n, frames = self._pending[slot]
        self._pending[slot] = None
        self.ev_done[slot].synchronize()
        res, counts = unpack_results(self.pin_out[slot].numpy(), self.batch, self.pool, self.refinenet is not None)
        res = res[:n]
        need = int(counts.astype(np.int64).sum())
        if need > self.pool:         # rare: the batch fired more cells than its pool holds -> exact re-run with the pool it asked for
            warnings.warn(f"a batch produced {need} corners > pool={self.pool} (batch x kmax); re-running it with pool={need}")
            res = infer_batch(frames, self.dust_bin_ids, self.deepc, self.refinenet, pool=need)
        if self.pnp is not None:     # host stage: futures now, resolved when the batch is handed out
            return ticket, res, solve_pnp_submit(res, **self.pnp)
        return ticket, res

    def submit(self, frames_gray: np.ndarray):
        """Enqueue one batch (n <= batch frames; gray (n,H,W), or BGR (n,H,W,3) for a ``bgr=True`` stream). Returns the (ticket,
        results) of the batch that had to be retired to make room, or None; with a PnP stage the tuple carries a third element,
        the per-frame ``solve_pnp`` futures."""
        n = frames_gray.shape[0]
        want = (self.h, self.w, 3) if self.bgr else (self.h, self.w)
        if n > self.batch or tuple(frames_gray.shape[1:]) != want or frames_gray.dtype != np.uint8:
            raise ValueError(f"frames must be (n<=batch, {', '.join(map(str, want))}) uint8")
        slot = self._ticket % self.depth
        retired = self._collect(slot) if self._pending[slot] is not None else None
        self.pin_in[slot][:n].numpy()[...] = frames_gray
        if n < self.batch:
            self.pin_in[slot][n:].zero_()
        with torch.cuda.device(self.dev):
            compute = torch.cuda.current_stream() if self.compute is None else self.compute[self._ticket % len(self.compute)]
            if not self.h2d_on_compute:
                with torch.cuda.stream(self.copy_stream):
                    self.copy_stream.wait_event(self.ev_free[slot])         # previous user of dev_in[slot] is done
                    self.dev_in[slot].copy_(self.pin_in[slot], non_blocking=True)
                    self.ev_h2d[slot].record(self.copy_stream)
            with torch.cuda.stream(compute):
                if self.h2d_on_compute:
                    compute.wait_event(self.ev_free[slot])
                    self.dev_in[slot].copy_(self.pin_in[slot], non_blocking=True)
                else:
                    compute.wait_event(self.ev_h2d[slot])
                infer_batch_device(self.dev_in[slot], self.dust_bin_ids, self.deepc, self.refinenet, out=self.dev_out[slot],
                                   pool=self.pool)
                self.ev_free[slot].record(compute)
                self.pin_out[slot].copy_(self.dev_out[slot], non_blocking=True)
                self.ev_done[slot].record(compute)
        self._pending[slot] = (self._ticket, n, frames_gray)
        self._ticket += 1
        return retired

    @staticmethod
    def _resolve(r):
        return r if len(r) == 2 else (r[0], r[1], [f.result() for f in r[2]])

    def flush(self) -> Iterator[Tuple]:
        order = sorted((p[0], s) for s, p in enumerate(self._pending) if p is not None)
        for _, slot in order:
            yield self._resolve(self._collect(slot))

    def run(self, batches: Iterable[np.ndarray]) -> Iterator[Tuple]:
        """batches: iterable of (n,H,W) uint8 arrays -> (ticket, per-frame keypoint arrays[, per-frame (ret, rvec, tvec)])
        in submission order.  The PnP futures of a retired batch are resolved one submission later, so they run on the
        host threads while the next batch is being staged and the GPU is busy."""
        if self.pnp is None:                  # nothing to overlap: hand every retired batch out at once
            for fr in batches:
                r = self.submit(fr)
                if r is not None:
                    yield r
            yield from self.flush()
            return
        held = None
        for fr in batches:
            r = self.submit(fr)
            if held is not None:
                yield self._resolve(held)
                held = None
            if r is not None:
                held = r
        if held is not None:
            yield self._resolve(held)
        yield from self.flush()


class ResidentStream:
    """Pipelined caller for batches that are ALREADY in HBM (camera frames decoded on the GPU, a previous kernel's output, the
    measured configuration of bench.py): ``depth`` batches are enqueued ahead of the host, on one HIP stream of their own or on
    ``compute_streams`` alternating ones.

    What it buys is a host that never blocks on the batch it just enqueued: ``submit`` returns at once, the packed corner lists
    of a batch arrive in pinned memory while the next batches run, and unpacking them overlaps GPU work (bench.py's timed loop
    is this class with ``raw=True``).  ``compute_streams`` > 1 puts consecutive batches on alternating HIP streams; frames are
    independent (inference.py:32-70 keeps no cross-frame state), every batch keeps its own launch order and its own scratch
    (the pipeline workspace is per (model, stream)), so the results are bit-identical to ``infer_batch`` either way
    (tests/test_gpu_parity.py::test_resident_stream_*).  At bs=32 320x240 a second stream does NOT raise throughput (-1 ... -4 %
    with equal work in every batch, profiles/experiments/r05_batches_in_flight_equal_work.txt: one batch's launches already
    occupy both workgroup slots of every CU) -- the default is one; more only pays for batches too small to fill the chip.

    ``depth`` output slots rotate (default ``compute_streams + 2``): ``submit`` blocks only on the batch submitted ``depth``
    submissions earlier, so the GPU always has work queued while the host retires / unpacks an older batch.  ``submit(frames)``: ``frames`` = contiguous (n <= batch, H, W) gray or (n, H, W, 3) BGR uint8
    tensor on the model's GPU, produced on the CURRENT stream (the compute stream waits for an event recorded there); the caller
    must leave it untouched until that batch has been handed out.  Returns the ``(ticket, results)`` of the batch that had to be
    retired to make room, or None.  ``raw=True``: results are the packed int32 host arrays (``unpack_results`` layout) instead of
    per-frame key-point arrays -- no host work besides the event wait (a batch that had to be re-run comes back laid out for
    ``pool = sum(counts)``, which the buffer's own first ``n`` words give).  Overflow of a batch's corner pool (``batch * kmax``
    corners for the whole batch, no per-frame cap) is handled as everywhere else: that batch is run once more with the pool the
    first pass reported."""

    def __init__(self, dust_bin_ids: int, deepc, refinenet=None, batch: int = 32, height: int = 240, width: int = 320,
                 kmax: int = DEFAULT_KMAX, compute_streams: int = 1, depth: Optional[int] = None, bgr: bool = False,
                 raw: bool = False, timing: bool = False):
        """``timing=True``: every batch is bracketed by a timing-enabled event pair on its compute stream; ``gpu_ms`` then
        holds, per retired batch, the time from the moment the stream reached the batch to its last byte in pinned memory
        (bench.py's step_breakdown).  The host-side counters ``host_enqueue_s`` (time spent inside ``submit`` launching work) and
        ``host_wait_s`` (time blocked on the oldest batch's completion event) are always kept."""
        det = deepc.model if hasattr(deepc, "model") else deepc
        self.dev = det.device
        self.dust_bin_ids, self.deepc, self.refinenet = dust_bin_ids, deepc, refinenet
        self.batch, self.h, self.w, self.kmax = batch, height, width, kmax
        self.pool = batch * kmax
        self.bgr, self.raw, self.timing = bool(bgr), bool(raw), bool(timing)
        if compute_streams < 1:
            raise ValueError("compute_streams must be >= 1")
        self.depth = depth = compute_streams + 2 if depth is None else depth
        if depth < compute_streams:
            raise ValueError("depth must be >= compute_streams")
        n_out = packed_len(batch, self.pool)
        with torch.cuda.device(self.dev):
            self.compute = [torch.cuda.Stream() for _ in range(compute_streams)]
            self.dev_out = [torch.empty((n_out,), dtype=torch.int32, device=self.dev) for _ in range(depth)]
            self.pin_out = [torch.empty((n_out,), dtype=torch.int32).pin_memory() for _ in range(depth)]
            self.ev_in = [torch.cuda.Event() for _ in range(depth)]
            self.ev_start = [torch.cuda.Event(enable_timing=True) for _ in range(depth)] if self.timing else None
            self.ev_done = [torch.cuda.Event(enable_timing=self.timing) for _ in range(depth)]
        self._pending: List[Optional[Tuple[int, torch.Tensor, torch.cuda.Stream]]] = [None] * depth     # (ticket, device frames, its stream)
        self._ticket = 0
        self.gpu_ms: List[float] = []
        self.host_enqueue_s = 0.0
        self.host_wait_s = 0.0

    def reset_stats(self) -> None:
        self.gpu_ms = []
        self.host_enqueue_s = self.host_wait_s = 0.0

    def _collect(self, slot: int):
        ticket, frames, compute = self._pending[slot]
        self._pending[slot] = None
        n = frames.shape[0]
        t0 = time.perf_counter()
        self.ev_done[slot].synchronize()
        self.host_wait_s += time.perf_counter() - t0
        if self.timing:
            self.gpu_ms.append(self.ev_start[slot].elapsed_time(self.ev_done[slot]))
        packed = self.pin_out[slot].numpy()[:packed_len(n, self.pool)]
        need = int(packed[:n].astype(np.int64).sum())
        if need > self.pool:
            # rare: the batch fired more cells than its pool holds -> exact re-run with the pool it asked for, on the batch's OWN
            # compute stream (behind whatever was submitted after it: the caller's stream is not touched and not synchronised),
            # handed out through pinned memory like every other batch; the host waits for that one event only
            warnings.warn(f"a batch produced {need} corners > pool={self.pool} (batch x kmax); re-running it with pool={need}")
            n_big = packed_len(n, need)
            with torch.cuda.device(self.dev):
                host = torch.empty((n_big,), dtype=torch.int32).pin_memory()
                done = torch.cuda.Event()
                with torch.cuda.stream(compute):
                    big = infer_batch_device(frames, self.dust_bin_ids, self.deepc, self.refinenet, pool=need)
                    host.copy_(big, non_blocking=True)
                    done.record(compute)
            t0 = time.perf_counter()
            done.synchronize()
            self.host_wait_s += time.perf_counter() - t0
            packed = host.numpy()
            if self.raw:
                return ticket, packed
            return ticket, unpack_results(packed, n, need, self.refinenet is not None)[0]
        if self.raw:
            return ticket, packed.copy()
        return ticket, unpack_results(packed, n, self.pool, self.refinenet is not None)[0]

    def submit(self, frames: torch.Tensor):
        want = (self.h, self.w, 3) if self.bgr else (self.h, self.w)
        if (not isinstance(frames, torch.Tensor) or frames.device != self.dev or frames.dtype != torch.uint8
                or not frames.is_contiguous() or tuple(frames.shape[1:]) != want or not (1 <= frames.shape[0] <= self.batch)):
            raise ValueError(f"frames must be a contiguous (1 <= n <= batch, {', '.join(map(str, want))}) uint8 tensor on {self.dev}")
        n = frames.shape[0]
        slot = self._ticket % self.depth
        retired = self._collect(slot) if self._pending[slot] is not None else None
        t0 = time.perf_counter()
        n_out = packed_len(n, self.pool)
        with torch.cuda.device(self.dev):
            compute = self.compute[self._ticket % len(self.compute)]
            self.ev_in[slot].record(torch.cuda.current_stream())     # whatever produced `frames` was enqueued on the caller's stream
            with torch.cuda.stream(compute):
                compute.wait_event(self.ev_in[slot])
                if self.timing:
                    self.ev_start[slot].record(compute)
                infer_batch_device(frames, self.dust_bin_ids, self.deepc, self.refinenet, out=self.dev_out[slot][:n_out],
                                   pool=self.pool)
                self.pin_out[slot][:n_out].copy_(self.dev_out[slot][:n_out], non_blocking=True)
                self.ev_done[slot].record(compute)
        self._pending[slot] = (self._ticket, frames, compute)
        self._ticket += 1
        self.host_enqueue_s += time.perf_counter() - t0
        return retired

    def flush(self) -> Iterator[Tuple]:
        order = sorted((p[0], s) for s, p in enumerate(self._pending) if p is not None)
        for _, slot in order:
            yield self._collect(slot)

    def run(self, batches: Iterable[torch.Tensor]) -> Iterator[Tuple]:
        """batches: iterable of device tensors -> (ticket, results) in submission order."""
        for fr in batches:
            r = self.submit(fr)
            if r is not None:
                yield r
        yield from self.flush()
