"""HIP-backed mirror of /root/reference/src/models/refinenet.py (RefineNet :9-115, lRefineNet :134-145)."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from .. import _lib
from ..weights import StateDict, state_dict_from_checkpoint
from ._handles import Workspace, check_dev_tensor, require_cuda, tensor_pointer_array


class RefineNet:
    def __init__(self, state_dict: Optional[StateDict] = None, device="cuda"):
        self._handle = None
        self._device: Optional[torch.device] = None
        self._ws = Workspace()
        self._sd = None
        if state_dict is not None:
            self.load_state_dict(state_dict, device)

    def load_state_dict(self, state_dict: StateDict, device="cuda") -> "RefineNet":
        dev = require_cuda(device)
        arr, keep = tensor_pointer_array(state_dict, "refinenet", 16)
        self._release()
        h = C.c_void_p()
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().dcx_refiner_create(C.byref(h), arr, len(keep)), "dcx_refiner_create")
        self._handle, self._device, self._sd = h, dev, state_dict
        return self

    def to(self, device) -> "RefineNet":
        dev = require_cuda(device)
        if self._sd is not None and dev != self._device:
            self.load_state_dict(self._sd, dev)
        return self

    def eval(self) -> "RefineNet":
        return self

    @property
    def handle(self) -> C.c_void_p:
        if self._handle is None:
            raise RuntimeError("RefineNet has no weights loaded")
        return self._handle

    @property
    def device(self) -> torch.device:
        if self._device is None:
            raise RuntimeError("RefineNet has no weights loaded")
        return self._device

    def _release(self):
        if self._handle is not None:
            import sys
            g = sys.modules.get("deepcharuco_amd.graph")     # only if a hipGraph may have been captured with this handle
            if g is not None:
                g.drop_graphs_of_refiner(self)
            _lib.lib().dcx_refiner_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _run(self, patches: torch.Tensor, want_heat: bool, keypoints: Optional[torch.Tensor] = None):
        dev = self.device
        patches = check_dev_tensor(patches, dev, torch.float32, "patches")
        k = patches.shape[0]
        L = _lib.lib()
        corners = torch.empty((k, 2), dtype=torch.int32, device=dev)
        heat = torch.empty((k, 1, 64, 64), dtype=torch.float32, device=dev) if want_heat else None
        xy = table = None
        if keypoints is not None:   # corners_og = (corners - 32) / 8 + keypoints computed by the finalize kernel
            if keypoints.dtype.is_floating_point or keypoints.dtype == torch.bool:
                raise TypeError("keypoints must be an integer tensor (the detector's pixel coordinates, as "
                                "pred_to_keypoints returns them); float keypoints would be truncated")
            xy = torch.empty((k, 2), dtype=torch.float32, device=dev)
            table = torch.zeros((k, 4), dtype=torch.int32, device=dev)
            table[:, 1:3] = keypoints.to(device=dev, dtype=torch.int32)
            table[:, 3] = torch.arange(k, dtype=torch.int32, device=dev)
        if k == 0:
            return corners, heat, xy
        with torch.cuda.device(dev):
            nbytes = L.dcx_refiner_workspace_bytes(self.handle, k)
            ws = self._ws.get("ref", dev, nbytes)
            _lib.check(L.dcx_refiner_forward(self.handle, patches.data_ptr(), k, None, _lib.ptr(table), ws.data_ptr(),
                                             ws.numel(), corners.data_ptr(), _lib.ptr(xy), _lib.ptr(heat),
                                             _lib.current_stream()),
                       "dcx_refiner_forward")
        return corners, heat, xy

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """refinenet.py:49-83: x (K,1,24,24) -> heat-map logits (K,1,64,64)."""
        if x.ndim != 4 or tuple(x.shape[1:]) != (1, 24, 24):
            raise ValueError(f"expected (K,1,24,24), got {tuple(x.shape)}")
        return self._run(x, True)[1]

    __call__ = forward

    def infer_patches(self, patches: torch.Tensor, keypoints: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """refinenet.py:85-115 -> (corners_og (K,2) float32, corners (K,2) int64 as (col,row))."""
        assert patches.shape[-2:] == (24, 24)
        corners32, _, corners_og = self._run(patches.reshape(-1, 24, 24), False, keypoints)
        return corners_og, corners32.to(torch.int64)


class lRefineNet:
    def __init__(self, refinenet: RefineNet):
        self.model = refinenet

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path: str, refinenet: RefineNet, map_location=None, **_):
        sd = state_dict_from_checkpoint(checkpoint_path, "refinenet")
        refinenet._sd = sd
        if map_location is not None:
            refinenet.load_state_dict(sd, map_location)
        return cls(refinenet)

    def forward(self, x):
        return self.model(x)

    __call__ = forward

    def infer_patches(self, patches, keypoints):
        return self.model.infer_patches(patches, keypoints)

    def eval(self):
        return self

    def to(self, device):
        if self.model._sd is None:
            raise RuntimeError("no weights loaded")
        self.model.load_state_dict(self.model._sd, device)
        return self
