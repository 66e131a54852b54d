"""hipGraph replay of one fixed-shape detect+refine call.

The reference's own protocol is one frame per ``infer_image`` call (/root/reference/src/benchmark.py:37-53,
pose_estimation.py:58-59): ~28 small kernel launches whose host-side enqueue cost, plus the host BGR->gray conversion,
is a large part of a 0.5-0.7 ms call on MI355X.  ``GraphedPipeline`` captures the whole call once per shape --
H2D of the frame from pinned memory, BGR->gray on the device (``dcx_bgr2gray``), the sync-free pipeline
(``dcx_infer_batch``), D2H of the packed corner list -- into ONE hipGraph (``torch.cuda.CUDAGraph`` = hipGraph on ROCm)
and replays it per call: one launch from the host instead of ~30.  Results are those of ``infer_batch`` (same kernels,
same order); a frame that fires more than ``kmax`` cells falls back to the eager path, which re-runs with a larger
capacity.  A pipeline owns its pinned / device buffers and its stream; capture and replay are serialised process-wide by one
lock (``_graph_lock``): concurrent ``infer_image`` callers never share staging buffers mid-flight, and no hipGraph is captured
while another thread replays one (capturing and replaying from several threads at once hung the HIP runtime in testing) --
callers that want concurrency across threads use ``infer_batch_device`` on their own streams.  The graph freezes the kernel
choice made at capture time, so the cache key also carries the library's process-global mode (``dcx_get_deterministic``) and
graphs are bypassed while per-stage timing / per-launch profiling is on (their hipEvents would be frozen into or out of it).
"""
from __future__ import annotations

import threading
import weakref
from typing import List, Optional

import numpy as np
import torch

from . import _lib
from .inference import DEFAULT_KMAX, infer_batch, infer_batch_device, unpack_results
from .sharding import packed_len


class GraphedPipeline:
    def __init__(self, dust_bin_ids: int, deepc, refinenet=None, batch: int = 1, height: int = 240, width: int = 320,
                 kmax: int = DEFAULT_KMAX, bgr: bool = True):
        det = deepc.model if hasattr(deepc, "model") else deepc
        ref = None if refinenet is None else (refinenet.model if hasattr(refinenet, "model") else refinenet)
        self.dev = det.device
        self.dust_bin_ids = dust_bin_ids
        # weak references to the inner models: the pipeline lives in a cache ON the detector, a strong reference back would be a
        # cycle that only the cyclic GC frees (~25 MB of workspace + pinned buffers per shape); with weak ones the graphs die with
        # the model by reference count
        self._det, self._ref = weakref.ref(det), (None if ref is None else weakref.ref(ref))
        self.batch, self.h, self.w, self.kmax, self.bgr = batch, height, width, kmax, bgr
        L = _lib.lib()
        shape = (batch, height, width, 3) if bgr else (batch, height, width)
        n_out = packed_len(batch, kmax)
        with torch.cuda.device(self.dev):
            self.pin_in = torch.empty(shape, dtype=torch.uint8).pin_memory()
            self.dev_in = torch.empty(shape, dtype=torch.uint8, device=self.dev)
            self.gray = torch.empty((batch, height, width), dtype=torch.uint8, device=self.dev) if bgr else self.dev_in
            self.out_dev = torch.empty((n_out,), dtype=torch.int32, device=self.dev)
            self.pin_out = torch.empty((n_out,), dtype=torch.int32).pin_memory()
            nbytes = L.dcx_pipeline_workspace_bytes(det.handle, ref.handle if ref else None, batch, height, width, kmax)
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

    @property
    def deepc(self):
        det = self._det()
        if det is None:
            raise RuntimeError("the detector this graph was captured with has been destroyed")
        return det

    @property
    def refinenet(self):
        if self._ref is None:
            return None
        ref = self._ref()
        if ref is None:
            raise RuntimeError("the RefineNet this graph was captured with has been destroyed")
        return ref

    def _enqueue(self) -> None:
        self.dev_in.copy_(self.pin_in, non_blocking=True)
        if self.bgr:
            _lib.check(_lib.lib().dcx_bgr2gray(self.dev_in.data_ptr(), self.h * self.w * 3, self.w * 3, self.batch, self.h,
                                               self.w, self.gray.data_ptr(), _lib.current_stream()), "dcx_bgr2gray")
        infer_batch_device(self.gray, self.dust_bin_ids, self.deepc, self.refinenet, self.kmax, out=self.out_dev, ws=self.ws)
        self.pin_out.copy_(self.out_dev, non_blocking=True)

    def run(self, frames: np.ndarray) -> List[np.ndarray]:
        """frames: (B,H,W,3) BGR or (B,H,W) gray uint8 host array (as configured) -> list of B keypoint arrays."""
        if frames.shape != self._in_np.shape or frames.dtype != np.uint8:
            raise ValueError(f"expected uint8 frames of shape {self._in_np.shape}, got {frames.dtype} {frames.shape}")
        with _graph_lock:                                     # one capture / replay at a time per process
            np.copyto(self._in_np, frames)
            with torch.cuda.device(self.dev):
                self.graph.replay()
                torch.cuda.current_stream().synchronize()
            res, counts = unpack_results(self._out_np, self.batch, self.kmax, self.refinenet is not None)
            if int(counts.max()) > self.kmax:                 # rare: capacity exceeded -> exact eager re-run
                gray = frames if not self.bgr else self.gray.cpu().numpy()
                res = infer_batch(gray, self.dust_bin_ids, self.deepc, self.refinenet, kmax=self.kmax)
        return res


_CACHE_MAX = 8        # per detector: graphs pin ~25 MB of workspace per 320x240 shape
_cache_lock = threading.RLock()         # re-entrant: dropping a pipeline may run model destructors that come back here
_graph_lock = threading.RLock()          # serialises hipGraph capture AND replay (+ the synchronise that follows) process-wide
_caches: "weakref.WeakSet" = weakref.WeakSet()     # the per-detector caches that exist (for clear_graph_cache())


class _Cache(dict):
    __slots__ = ("__weakref__",)
    __hash__ = object.__hash__          # identity: lives in a WeakSet


def graphs_usable() -> bool:
    """False while per-stage timing or per-launch profiling is on: their hipEvent records must run eagerly."""
    L = _lib.lib()
    return not (L.dcx_get_timing() or L.dcx_profile_enabled())


def clear_graph_cache(deepc=None) -> None:
    """Drop the captured graphs (and their pinned / workspace buffers) of one detector, or of all of them."""
    with _graph_lock:                   # destruction (hipGraphExecDestroy, hipFree of the workspace) never overlaps a replay / capture
        victims = []                    # ... but happens outside _cache_lock; lock order is always _graph_lock -> _cache_lock
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
        del victims


def drop_graphs_of_detector(det) -> None:
    """Called when a detector releases its C handle: its graphs hold pointers into the freed weights."""
    with _graph_lock:
        victims = []
        with _cache_lock:
            cache = getattr(det, "_graph_cache", None)
            if cache:
                victims = list(cache.values())
                cache.clear()
        del victims


def drop_graphs_of_refiner(ref) -> None:
    """Called when a RefineNet releases its C handle (reload / ``to(device)`` / destruction): graphs captured with it hold
    pointers into the freed weights."""
    with _graph_lock:
        victims = []
        with _cache_lock:
            for c in list(_caches):
                for k in [k for k in c if k[0] is not None and k[0][0] == id(ref)]:
                    victims.append(c.pop(k, None))
        del victims


def cached_pipeline(dust_bin_ids: int, deepc, refinenet, height: int, width: int, bgr: bool,
                    kmax: int = DEFAULT_KMAX) -> Optional[GraphedPipeline]:
    """One graph per (model pair, shape, library mode) for ``infer_image``, shared by all threads.  The cache lives ON the detector
    object (a small LRU) and its pipelines refer back to the models weakly, so graphs die with the model (by reference count)
    instead of pinning it in a module-global table, and a re-allocated model can never alias a cached graph of a freed one."""
    det = deepc.model if hasattr(deepc, "model") else deepc
    ref = None if refinenet is None else (refinenet.model if hasattr(refinenet, "model") else refinenet)
    key = (None if ref is None else (id(ref), ref.handle.value), dust_bin_ids, height, width, bgr, kmax,
           int(_lib.lib().dcx_get_deterministic()))
    with _graph_lock:                                         # capture excludes every replay (and other captures)
        with _cache_lock:
            cache = getattr(det, "_graph_cache", None)
            if cache is None:
                cache = det._graph_cache = _Cache()
                _caches.add(cache)
            p = cache.pop(key, None)
        if p is None:
            p = GraphedPipeline(dust_bin_ids, deepc, refinenet, 1, height, width, kmax, bgr)
        evicted = []
        with _cache_lock:
            while len(cache) >= _CACHE_MAX:
                evicted.append(cache.pop(next(iter(cache))))
            cache[key] = p
        del evicted                                           # destroyed under _graph_lock, outside _cache_lock
    return p
