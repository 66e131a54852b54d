"""hipGraph replay of one fixed-shape detect+refine call.

The reference's own protocol is one frame per ``infer_image`` call (/root/reference/src/benchmark.py:37-53,
pose_estimation.py:58-59): ~28 small kernel launches whose host-side enqueue cost, plus the host BGR->gray conversion,
is a large part of a 0.5-0.7 ms call on MI355X.  ``GraphedPipeline`` captures the whole call once per shape --
H2D of the frame from pinned memory, BGR->gray on the device (``dcx_bgr2gray``), the sync-free pipeline
(``dcx_infer_batch``), D2H of the packed corner list -- into ONE hipGraph (``torch.cuda.CUDAGraph`` = hipGraph on ROCm)
and replays it per call: one launch from the host instead of ~30.  Results are those of ``infer_batch`` (same kernels,
same order); a call whose frames fire more cells than the captured corner pool holds falls back to the eager path, which
re-runs with a pool of the right size.  BGR frames are converted inside the first layer's load (``DCX_PIX_BGR8``), so the graph
holds no separate colour-conversion node.  A pipeline owns its pinned / device buffers and its stream.  Locking is PER DEVICE:
replays on one GPU are serialised by that GPU's lock (concurrent ``infer_image`` callers never share staging buffers
mid-flight), replays on different GPUs of one process run concurrently; a CAPTURE (and the destruction of a graph) takes the
locks of all devices, because capturing while another thread replays hung the HIP runtime in testing -- callers that want
concurrency on one GPU across threads use ``infer_batch_device`` on their own streams.  The graph freezes the kernel
choice made at capture time, so the cache key also carries the library's process-global mode (``dcx_get_deterministic``) and
graphs are bypassed while per-stage timing / per-launch profiling is on (their hipEvents would be frozen into or out of it).
"""
from __future__ import annotations

import os
import threading
import time
import warnings
import weakref
from typing import List, Optional

import numpy as np
import torch

from . import _lib
from .inference import DEFAULT_KMAX, PIXEL_FORMATS, infer_batch, launch_pipeline, packed_len, unpack_results


class _DeviceLocks:
    """One re-entrant lock per GPU.  ``device(i)``: the lock a replay on GPU i holds.  ``all()``: context manager that takes every
    device's lock in ascending order (capture, graph destruction, cache surgery) -- a single total order, so a thread holding
    one device's lock and another one taking all of them cannot deadlock (a replay never waits for a second lock)."""

    def __init__(self):
        self._locks = {}
        self._guard = threading.Lock()
        self._all = threading.RLock()           # serialises the all-device holders among themselves (also covers devices not seen yet)

    def device(self, index: int) -> "threading.RLock":
        with self._guard:
            lk = self._locks.get(index)
            if lk is None:
                lk = self._locks[index] = threading.RLock()
            return lk

    class _All:
        def __init__(self, owner):
            self.owner, self.held = owner, []

        def __enter__(self):
            # With back-off: a thread that already holds ONE device's lock (a replay whose garbage collection runs a model
            # destructor, which comes here) may be waiting for `_all` while the holder of `_all` waits for that device -- so a
            # holder that cannot get a device lock within 50 ms lets go of everything and starts over.
            n = max(torch.cuda.device_count() if torch.cuda.is_available() else 0, 1)
            while True:
                self.owner._all.acquire()
                for i in range(n):
                    lk = self.owner.device(i)
                    if not lk.acquire(timeout=0.05):
                        break
                    self.held.append(lk)
                else:
                    return self
                for lk in reversed(self.held):
                    lk.release()
                self.held = []
                self.owner._all.release()
                time.sleep(0.001)

        def __exit__(self, *exc):
            for lk in reversed(self.held):
                lk.release()
            self.held = []
            self.owner._all.release()
            return False

    def all(self):
        return _DeviceLocks._All(self)

    def try_all(self) -> Optional[list]:
        """Non-blocking attempt at every device's lock -> the held locks (release with ``release_all``) or None.  For callers that
        must never wait (model destructors, which may run from the garbage collector inside any critical section)."""
        if not self._all.acquire(blocking=False):
            return None
        held = []
        n = max(torch.cuda.device_count() if torch.cuda.is_available() else 0, 1)
        for i in range(n):
            lk = self.device(i)
            if not lk.acquire(blocking=False):
                for h in reversed(held):
                    h.release()
                self._all.release()
                return None
            held.append(lk)
        return held

    def release_all(self, held: list) -> None:
        for lk in reversed(held):
            lk.release()
        self._all.release()


_locks = _DeviceLocks()


class GraphedPipeline:
    def __init__(self, dust_bin_ids: int, deepc, refinenet=None, batch: int = 1, height: int = 240, width: int = 320,
                 kmax: int = DEFAULT_KMAX, bgr: bool = True, zero_copy: Optional[int] = None):
        det = deepc.model if hasattr(deepc, "model") else deepc
        ref = None if refinenet is None else (refinenet.model if hasattr(refinenet, "model") else refinenet)
        self.dev = det.device
        self.dust_bin_ids = dust_bin_ids
        # weak references to the inner models: the pipeline lives in a cache ON the detector, a strong reference back would be a
        # cycle that only the cyclic GC frees (~25 MB of workspace + pinned buffers per shape); with weak ones the graphs die with
        # the model by reference count
        self._det, self._ref = weakref.ref(det), (None if ref is None else weakref.ref(ref))
        # the C handles the graph's kernel arguments point into: a replay first checks that the models still own exactly these
        # (a reload / .to() / destruction frees the weights the captured launches read)
        self._det_handle = det.handle.value
        self._ref_handle = None if ref is None else ref.handle.value
        self.batch, self.h, self.w, self.kmax, self.bgr = batch, height, width, kmax, bgr
        self.pool = batch * kmax
        self.zero_copy = int(os.environ.get("DCX_GRAPH_ZEROCOPY", "2")) if zero_copy is None else int(zero_copy)
        self._lock = _locks.device(self.dev.index)
        self.graph = None
        L = _lib.lib()
        shape = (batch, height, width, 3) if bgr else (batch, height, width)
        n_out = packed_len(batch, self.pool)
        with torch.cuda.device(self.dev):
            self.pin_in = torch.empty(shape, dtype=torch.uint8).pin_memory()
            self.dev_in = torch.empty(shape, dtype=torch.uint8, device=self.dev)
            self.out_dev = torch.empty((n_out,), dtype=torch.int32, device=self.dev)
            self.pin_out = torch.empty((n_out,), dtype=torch.int32).pin_memory()
            nbytes = L.dcx_pipeline_workspace_bytes(det.handle, ref.handle if ref else None, batch, height, width, self.pool)
            if nbytes == 0:
                raise ValueError("bad batch/shape for dcx_pipeline_workspace_bytes")
            self.ws = torch.empty((nbytes,), dtype=torch.uint8, device=self.dev)
            self.stream = torch.cuda.Stream()
            self.pin_in.zero_()
            with torch.cuda.stream(self.stream):          # eager warm-up: per-kernel attributes, lazy module loading
                for _ in range(2):
                    self._enqueue()
            self.stream.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            # thread_local: other threads may allocate / launch eager work while this one captures
            with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="thread_local"):
                self._enqueue()
        self._in_np = self.pin_in.numpy()
        self._out_np = self.pin_out.numpy()
        self._dev_index = self.dev.index
        # views of the packed result for the one-frame unpack (inference.unpack_results' layout: counts, starts, rows, xy)
        o, b, pool = self._out_np, batch, self.pool
        self._rows_np = o[2 * b:2 * b + 4 * pool].reshape(pool, 4)
        self._xy_np = o[2 * b + 4 * pool:2 * b + 6 * pool].view(np.float32).reshape(pool, 2)
        with _cache_lock:
            _live.add(self)           # every pipeline that exists, cached or evicted-but-still-held (drop_graphs_of_* close them all)

    def _models_unchanged(self) -> bool:
        det = self._det()
        if det is None or det._handle is None or det._handle.value != self._det_handle:
            return False
        if self._ref is not None:
            ref = self._ref()
            if ref is None or ref._handle is None or ref._handle.value != self._ref_handle:
                return False
        return True

    @property
    def deepc(self):
        det = self._det()
        if det is None:      # ReferenceError, not RuntimeError: infer_image treats RuntimeError as "capture failed, go eager"
            raise ReferenceError("the detector this graph was captured with has been destroyed")
        return det

    @property
    def refinenet(self):
        if self._ref is None:
            return None
        ref = self._ref()
        if ref is None:
            raise ReferenceError("the RefineNet this graph was captured with has been destroyed")
        return ref

    def _enqueue(self) -> None:
        # BGR frames: the conversion of inference.py:40 happens in the first layer's load (DCX_PIX_BGR8), no separate kernel.
        # zero_copy bit 0: the kernels read the frame straight out of the pinned host buffer (conv1a of both nets are the only
        # readers: 77-230 KB over the host link instead of a copy node + a device read); bit 1: the tail / finalize kernels
        # write the corner list straight into pinned host memory (1.6 KB) instead of a device buffer + a copy node.
        det, ref = self.deepc, self.refinenet
        bpp = 3 if self.bgr else 1
        src = self.pin_in if self.zero_copy & 1 else self.dev_in
        dst = self.pin_out if self.zero_copy & 2 else self.out_dev
        if not self.zero_copy & 1:
            self.dev_in.copy_(self.pin_in, non_blocking=True)
        launch_pipeline(det, ref, src.data_ptr(), self.batch, self.h, self.w, bpp, PIXEL_FORMATS["opencv4" if self.bgr else "gray"],
                        self.dust_bin_ids, self.pool, self.ws, dst.data_ptr())
        if not self.zero_copy & 2:
            self.pin_out.copy_(self.out_dev, non_blocking=True)

    def run(self, frames: np.ndarray) -> List[np.ndarray]:
        """frames: (B,H,W,3) BGR or (B,H,W) gray uint8 host array (as configured) -> list of B keypoint arrays."""
        with self._lock:                                      # one replay at a time per GPU (a capture holds every GPU's lock)
            graph, in_np, out_np = self.graph, self._in_np, self._out_np
            if graph is None:
                raise ReferenceError("this GraphedPipeline has been closed (its models were released)")
            if not self._models_unchanged():      # reloaded / moved / destroyed since the capture: the graph points at freed weights
                self.close()
                raise ReferenceError("the models this graph was captured with have been released or reloaded")
            if frames.shape != in_np.shape or frames.dtype != np.uint8:
                raise ValueError(f"expected uint8 frames of shape {in_np.shape}, got {frames.dtype} {frames.shape}")
            np.copyto(in_np, frames)
            if _current_device() == self._dev_index:          # (the context manager costs 2 us of a 300 us call)
                graph.replay()
                _sync_current_stream(self._dev_index)
            else:
                with torch.cuda.device(self.dev):
                    graph.replay()
                    _sync_current_stream(self._dev_index)
            if self.batch == 1:
                # unpack_results for ONE frame on pre-built views (same values, same dtypes, same stable sort by id: inference.py:68-69)
                need, s0 = out_np[:2].tolist()
                if need == 0:
                    res = [np.array([])]
                elif s0 + need <= self.pool:
                    rb = self._rows_np[s0:s0 + need]
                    refined = self._ref is not None
                    a = np.empty((need, 3), np.float64 if refined else np.int64)
                    a[:, 0:2] = self._xy_np[s0:s0 + need] if refined else rb[:, 0:2]
                    a[:, 2] = rb[:, 2]
                    res = [a[rb[:, 2].argsort(kind="stable")]]
                else:
                    res = [None]
            else:
                res, counts = unpack_results(out_np, self.batch, self.pool, self.refinenet is not None)
                need = int(counts.sum(dtype=np.int64))
            if need > self.pool:                              # rare: more corners than the captured pool -> exact eager re-run
                warnings.warn(f"the call produced {need} corners > the captured graph's pool={self.pool}; re-running eagerly with pool={need}")
                res = infer_batch(frames, self.dust_bin_ids, self.deepc, self.refinenet, pool=need)
        return res

    def close(self) -> None:
        """Retire the pipeline: later ``run`` calls raise ReferenceError.  Never blocks and never destroys anything itself -- the
        hipGraph and the buffers move to a graveyard that is emptied only by a thread holding the locks of ALL devices
        (``_drain``), so hipGraphExecDestroy / hipFree can not overlap a replay or a capture, whichever thread drops the last
        reference (``__del__`` comes here too) and whatever that thread is in the middle of."""
        graph = getattr(self, "graph", None)
        if graph is None:
            return
        self.graph = None
        _graveyard.append((graph, self.ws, self.dev_in, self.out_dev, self.pin_in, self.pin_out, self.stream))
        self.ws = self.dev_in = self.out_dev = self.pin_in = self.pin_out = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- host-latency helpers of the one-frame path (a call is ~0.3 ms: every microsecond of interpreter work is 0.3 % of it) ----------
_hip = None


def _sync_current_stream(dev_index: int) -> None:
    """hipStreamSynchronize on torch's CURRENT stream of the device (what ``torch.cuda.current_stream().synchronize()`` does, without
    building a Stream object: 3.7 -> ~1 us per call, tools/bs1_host_probe2.py).  Falls back to the public API if torch's raw-stream
    accessor or the HIP runtime's symbol is not where this build of torch has them."""
    global _hip
    if _hip is False:
        torch.cuda.current_stream(dev_index).synchronize()
        return
    try:
        raw = torch._C._cuda_getCurrentRawStream(dev_index)
        if _hip is None:
            import ctypes
            lib = ctypes.CDLL("libamdhip64.so")            # the runtime torch has already loaded (same SONAME: same handle)
            lib.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
            lib.hipStreamSynchronize.restype = ctypes.c_int
            _hip = lib
        if _hip.hipStreamSynchronize(raw) == 0:
            return
    except (AttributeError, OSError):
        pass
    # the shortcut is not usable in this process (no raw-stream accessor, no symbol, or a runtime that does not know torch's stream
    # handle): the public API from here on -- it also raises a proper error if the stream itself is in trouble
    _hip = False
    torch.cuda.current_stream(dev_index).synchronize()


def _current_device() -> int:
    try:
        return torch._C._cuda_getDevice()
    except AttributeError:
        return torch.cuda.current_device()


_CACHE_MAX = 8        # per detector: graphs pin ~25 MB of workspace per 320x240 shape
_cache_lock = threading.RLock()         # re-entrant: dropping a pipeline may run model destructors that come back here
_graph_lock = _locks.all                # `with _graph_lock():` = every device's lock (capture, destruction, cache surgery)
_graveyard: list = []                   # hipGraphs / buffers of closed pipelines; list.append / pop are atomic
_caches: "weakref.WeakSet" = weakref.WeakSet()     # the per-detector caches that exist (for clear_graph_cache())
_live: "weakref.WeakSet" = weakref.WeakSet()       # every GraphedPipeline that exists (cached, or evicted but still held by a thread)


def _drain() -> None:
    """Destroy what closed pipelines left behind.  Call with ``_graph_lock()`` held: no replay or capture is in flight anywhere."""
    while _graveyard:
        _graveyard.pop()


class _Cache(dict):
    __slots__ = ("__weakref__",)
    __hash__ = object.__hash__          # identity: lives in a WeakSet


def graphs_usable() -> bool:
    """False while per-stage timing or per-launch profiling is on: their hipEvent records must run eagerly."""
    L = _lib.lib()
    return not (L.dcx_get_timing() or L.dcx_profile_enabled())


def _retire(victims) -> None:
    for v in victims:
        if v is not None:
            v.close()


def clear_graph_cache(deepc=None) -> None:
    """Drop the captured graphs (and their pinned / workspace buffers) of one detector, or of all of them."""
    with _graph_lock():                 # lock order is always device locks -> _cache_lock
        victims = []
        with _cache_lock:
            if deepc is not None:
                det = deepc.model if hasattr(deepc, "model") else deepc
                caches = [getattr(det, "_graph_cache", None)]
            else:
                caches = list(_caches)
            for c in caches:
                if c:
                    victims.extend(c.values())
                    c.clear()
        _retire(victims)
        del victims
        _drain()


def _try_drain() -> None:
    """Empty the graveyard if nobody is replaying or capturing right now; never waits (see ``drop_graphs_of_detector``)."""
    if not _graveyard:
        return
    held = _locks.try_all()
    if held is not None:
        try:
            _drain()
        finally:
            _locks.release_all(held)


def _close_pipelines_of(match) -> None:
    """Retire every pipeline -- cached or evicted-but-still-held -- for which ``match(pipeline)`` holds.  Takes ONLY the cache lock
    (re-entrant) and never waits for a device lock: this runs from model destructors, i.e. possibly from the garbage collector in
    the middle of any critical section of this module (a destructor that took the device locks while its thread already held the
    cache lock deadlocked against a capturing thread: ADVICE r5).  ``close`` itself is lock-free; what it leaves behind is destroyed
    by the next holder of all device locks (a capture, ``clear_graph_cache``) or right here if nobody is replaying.  A thread that
    is about to replay a retired pipeline is stopped by ``run``'s own check of the models' C handles."""
    with _cache_lock:
        victims = [p for p in list(_live) if match(p)]
        for c in list(_caches):
            for k in [k for k, p in c.items() if p in victims]:
                c.pop(k, None)
    _retire(victims)
    del victims
    _try_drain()


def drop_graphs_of_detector(det) -> None:
    """Called when a detector releases its C handle: its graphs hold pointers into the freed weights."""
    _close_pipelines_of(lambda p: p._det() is det or p._det() is None)


def drop_graphs_of_refiner(ref) -> None:
    """Called when a RefineNet releases its C handle (reload / ``to(device)`` / destruction): graphs captured with it hold
    pointers into the freed weights."""
    _close_pipelines_of(lambda p: p._ref is not None and (p._ref() is ref or p._ref() is None))


def cached_pipeline(dust_bin_ids: int, deepc, refinenet, height: int, width: int, bgr: bool,
                    kmax: int = DEFAULT_KMAX) -> Optional[GraphedPipeline]:
    """One graph per (model pair, shape, library mode) for ``infer_image``, shared by all threads.  The cache lives ON the detector
    object (a small LRU) and its pipelines refer back to the models weakly, so graphs die with the model (by reference count)
    instead of pinning it in a module-global table, and a re-allocated model can never alias a cached graph of a freed one.
    A hit costs one dictionary operation under the cache lock; only a miss (capture) takes the locks of all devices.  An evicted
    pipeline that another thread is still holding keeps working until that thread lets go of it."""
    det = deepc.model if hasattr(deepc, "model") else deepc
    ref = None if refinenet is None else (refinenet.model if hasattr(refinenet, "model") else refinenet)
    key = (None if ref is None else (id(ref), ref.handle.value), dust_bin_ids, height, width, bgr, kmax,
           int(_lib.lib().dcx_get_deterministic()))

    def lookup():
        cache = getattr(det, "_graph_cache", None)
        if cache is None:
            cache = det._graph_cache = _Cache()
            _caches.add(cache)
        p = cache.pop(key, None)
        if p is not None and p.graph is None:                 # retired (a model was released / reloaded): a miss, capture again
            p = None
        if p is not None:
            cache[key] = p                                    # most recently used last
        return cache, p

    with _cache_lock:
        cache, p = lookup()
    if p is not None:
        return p
    with _graph_lock():                                       # capture excludes every replay on every GPU (and other captures)
        _drain()
        with _cache_lock:
            cache, p = lookup()                               # another thread may have captured it meanwhile
        if p is None:
            p = GraphedPipeline(dust_bin_ids, deepc, refinenet, 1, height, width, kmax, bgr)
            evicted = []
            with _cache_lock:
                while len(cache) >= _CACHE_MAX:
                    evicted.append(cache.pop(next(iter(cache))))
                cache[key] = p
            del evicted           # not closed: a thread that still holds one finishes its call (it stays in _live, so a model
                                  # release still retires it); __del__ retires it afterwards
            _drain()
    return p
