"""HIP-backed mirror of /root/reference/src/models/net.py (dcModel :9-99, lModel :118-162).

Only the inference surface exists: ``forward`` / ``__call__`` / ``infer_image`` / ``eval`` /
``to``.  Training steps (net.py:130-162) are out of scope.
"""
from __future__ import annotations

import ctypes as C
import sys
from typing import Optional, Tuple

import torch

from .. import _lib
from ..weights import StateDict, state_dict_from_checkpoint
from ._handles import Workspace, check_dev_tensor, require_cuda, tensor_pointer_array


class dcModel:
    """DeepCharuco detector (net.py:9-80) running as hand-written gfx950 kernels."""

    def __init__(self, n_ids: int, state_dict: Optional[StateDict] = None, device="cuda"):
        self.n_ids = n_ids
        self._handle = None
        self._device: Optional[torch.device] = None
        self._ws = Workspace()
        self._sd = None
        if state_dict is not None:
            self.load_state_dict(state_dict, device)

    # -- weights ------------------------------------------------------------------------------
    def load_state_dict(self, state_dict: StateDict, device="cuda") -> "dcModel":
        dev = require_cuda(device)
        arr, keep = tensor_pointer_array(state_dict, "detector", self.n_ids)
        self._release()
        h = C.c_void_p()
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().dcx_detector_create(C.byref(h), arr, len(keep), self.n_ids), "dcx_detector_create")
        self._handle, self._device, self._sd = h, dev, state_dict
        return self

    def to(self, device) -> "dcModel":
        dev = require_cuda(device)
        if self._sd is not None and dev != self._device:
            self.load_state_dict(self._sd, dev)
        return self

    def eval(self) -> "dcModel":   # BN is always evaluated with running statistics (inference.py:75)
        return self

    def parameters(self):
        raise NotImplementedError("weights live in the HIP library's packed layout; use state_dict from weights.py")

    @property
    def handle(self) -> C.c_void_p:
        if self._handle is None:
            raise RuntimeError("dcModel has no weights loaded (call load_state_dict / load_models)")
        return self._handle

    @property
    def device(self) -> torch.device:
        if self._device is None:
            raise RuntimeError("dcModel has no weights loaded")
        return self._device

    def _release(self):
        if self._handle is not None:
            try:
                # hipGraphs captured with this handle's weights (graph.py).  sys.modules, not an import: this runs from __del__,
                # possibly during interpreter shutdown, where an import can fail -- and the handle must be freed regardless
                g = sys.modules.get("deepcharuco_amd.graph")
                if g is not None:
                    g.drop_graphs_of_detector(self)          # device locks first, then the cache lock (same order as everywhere)
            finally:
                _lib.lib().dcx_detector_destroy(self._handle)
                self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # -- forward (net.py:50-80) ---------------------------------------------------------------
    def forward(self, x: torch.Tensor):
        """x (N,1,H,W) normalised f32 on the GPU -> {'loc': (N,65,H/8,W/8), 'ids': (N,n_ids+1,H/8,W/8)}."""
        dev = self.device
        x = check_dev_tensor(x, dev, torch.float32, "x")
        if x.ndim != 4 or x.shape[1] != 1:
            raise ValueError(f"expected (N,1,H,W), got {tuple(x.shape)}")
        n, _, h, w = x.shape
        L = _lib.lib()
        with torch.cuda.device(dev):
            nbytes = L.dcx_detector_workspace_bytes(self.handle, n, h, w)
            ws = self._ws.get("det", dev, nbytes)
            loc = torch.empty((n, 65, h // 8, w // 8), dtype=torch.float32, device=dev)
            ids = torch.empty((n, self.n_ids + 1, h // 8, w // 8), dtype=torch.float32, device=dev)
            _lib.check(L.dcx_detector_forward(self.handle, None, 0, 0, x.data_ptr(), n, h, w, ws.data_ptr(),
                                              ws.numel(), loc.data_ptr(), ids.data_ptr(), _lib.current_stream()),
                       "dcx_detector_forward")
        return {"loc": loc, "ids": ids}

    __call__ = forward

    def forward_u8(self, frames: torch.Tensor):
        """frames (N,H,W) uint8 gray on the GPU; normalisation fused into conv1a. Same outputs as forward."""
        dev = self.device
        frames = check_dev_tensor(frames, dev, torch.uint8, "frames")
        n, h, w = frames.shape
        L = _lib.lib()
        with torch.cuda.device(dev):
            nbytes = L.dcx_detector_workspace_bytes(self.handle, n, h, w)
            ws = self._ws.get("det", dev, nbytes)
            loc = torch.empty((n, 65, h // 8, w // 8), dtype=torch.float32, device=dev)
            ids = torch.empty((n, self.n_ids + 1, h // 8, w // 8), dtype=torch.float32, device=dev)
            _lib.check(L.dcx_detector_forward(self.handle, frames.data_ptr(), h * w, w, None, n, h, w, ws.data_ptr(),
                                              ws.numel(), loc.data_ptr(), ids.data_ptr(), _lib.current_stream()),
                       "dcx_detector_forward")
        return {"loc": loc, "ids": ids}

    def infer_image(self, img: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """net.py:82-99: img (1,H,W) -> (loc, ids) with a leading batch axis of 1."""
        out = self.forward(img[None])
        return out["loc"], out["ids"]


class lModel:
    """Mirror of the Lightning wrapper (net.py:118-128): ``.model`` + ``infer_image``."""

    def __init__(self, dcModel: dcModel):  # noqa: N803  (keyword name used by the reference, inference.py:74)
        self.model = dcModel

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path: str, dcModel: dcModel, map_location=None, **_):  # noqa: N803
        sd = state_dict_from_checkpoint(checkpoint_path, "detector", dcModel.n_ids)
        dcModel._sd = sd
        if map_location is not None:
            dcModel.load_state_dict(sd, map_location)
        return cls(dcModel)

    def forward(self, x):
        return self.model(x)

    __call__ = forward

    def infer_image(self, img):
        return self.model.infer_image(img)

    def eval(self):
        self.model.eval()
        return self

    def to(self, device):
        if self.model._sd is None:
            raise RuntimeError("no weights loaded")
        self.model.load_state_dict(self.model._sd, device)
        return self
