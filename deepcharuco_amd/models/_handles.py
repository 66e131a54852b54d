"""Shared plumbing for the model mirrors: device checks, handle creation, workspace cache."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple

import numpy as np
import torch

from .. import _lib
from ..weights import StateDict, state_dict_keys, validate_state_dict


_cuda_checked: dict = {}       # device argument -> resolved device, for arguments that name an index (a per-call cost on the one-frame path)


def require_cuda(device) -> torch.device:
    try:
        hit = _cuda_checked.get(device)
    except TypeError:           # unhashable argument: the slow path decides
        hit = None
    if hit is not None:
        return hit
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError(
            f"deepcharuco_amd runs on MI355X (device 'cuda'); got device={device!r}. "
            "There is no CPU fallback -- use the reference implementation on CPU.")
    if not torch.cuda.is_available():
        raise RuntimeError("deepcharuco_amd needs a visible ROCm GPU (torch.cuda.is_available() is False)")
    if dev.index is None:
        return torch.device("cuda", torch.cuda.current_device())      # follows the current device: not memoised
    try:
        _cuda_checked[device] = dev
    except TypeError:
        pass
    return dev


def tensor_pointer_array(sd: StateDict, kind: str, n_ids: int):
    """Host float32 arrays in state_dict_keys() order -> (ctypes void* array, keep-alive list)."""
    validate_state_dict(sd, kind, n_ids)
    keep = [np.ascontiguousarray(sd[k], dtype=np.float32) for k in state_dict_keys(kind, n_ids)]
    arr = (C.c_void_p * len(keep))(*[a.ctypes.data for a in keep])
    return arr, keep


class Workspace:
    """Grow-only device scratch buffers (torch uint8 tensors), one per (tag, device, HIP stream).

    Owned by a model object, keyed by the stream that is current when it is requested: two streams (or two threads on
    two streams) driving the same model never share activations, and a buffer is only ever used -- and, when it has to
    grow, released -- on the stream it was allocated on, so torch's stream-ordered allocator keeps the old block alive
    until the kernels already queued on that stream are done with it."""

    def __init__(self):
        self._buf: Dict[Tuple[str, int, int], torch.Tensor] = {}

    def get(self, tag: str, device: torch.device, nbytes: int) -> torch.Tensor:
        key = (tag, device.index, torch.cuda.current_stream(device).cuda_stream)
        buf = self._buf.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = None
            self._buf.pop(key, None)
            buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
            self._buf[key] = buf
        return buf


def check_dev_tensor(t: torch.Tensor, device: torch.device, dtype, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if t.device.type != "cuda":
        raise RuntimeError(f"{name} must live on the GPU (got {t.device}); deepcharuco_amd has no CPU path")
    if device is not None and device.index is not None and t.device != device:
        raise RuntimeError(f"{name} lives on {t.device} but the model's weights are on {device}")
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()
