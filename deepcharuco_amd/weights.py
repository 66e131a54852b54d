"""Layer tables, checkpoint reader and numpy-seeded synthetic weights.

The layer tables restate the reference's module definitions:
  detector  -> /root/reference/src/models/net.py:23-48   (dcModel.__init__)
  refinenet -> /root/reference/src/models/refinenet.py:21-47 (RefineNet.__init__)

A "state dict" here is a plain ``dict[str, np.ndarray(float32)]`` using the
reference's own key names (``conv1a.weight``, ``bn1a.running_var`` ...), i.e.
the Lightning checkpoint's ``state_dict`` with the ``model.`` prefix stripped
(prefix comes from ``lModel.model = dcModel`` net.py:121 / refinenet.py:137).

The pre-trained checkpoints are not shipped with the reference mount, so every
parity fixture and the benchmark use :func:`synthetic_state_dict`, a
deterministic ``np.random.default_rng(seed)`` generator that both the fixture
generator (in the build container) and the GPU box can re-run.
"""
from __future__ import annotations

import hashlib
from typing import Dict, List, NamedTuple, Optional

import numpy as np

StateDict = Dict[str, np.ndarray]


class ConvSpec(NamedTuple):
    name: str          # conv module name, e.g. "conv1b"
    bn: Optional[str]  # following BatchNorm2d module name or None (raw 1x1 heads)
    cin: int
    cout: int
    ksize: int         # 3 or 1
    pad: int           # conv padding
    pool: bool = False  # MaxPool2d(2,2) applied after BN+ReLU of this layer
    ups: bool = False   # UpsamplingNearest2d(x2) applied after BN+ReLU of this layer


def detector_specs(n_ids: int = 16) -> List[ConvSpec]:
    """net.py:23-48 in forward order (net.py:60-77)."""
    return [
        ConvSpec("conv1a", "bn1a", 1, 64, 3, 1),
        ConvSpec("conv1b", "bn1b", 64, 64, 3, 1, pool=True),
        ConvSpec("conv2a", "bn2a", 64, 64, 3, 1),
        ConvSpec("conv2b", "bn2b", 64, 64, 3, 1, pool=True),
        ConvSpec("conv3a", "bn3a", 64, 128, 3, 1),
        ConvSpec("conv3b", "bn3b", 128, 128, 3, 1, pool=True),
        ConvSpec("conv4a", "bn4a", 128, 128, 3, 1),
        ConvSpec("conv4b", "bn4b", 128, 128, 3, 1),
        ConvSpec("convPa", "bnPa", 128, 256, 3, 1),
        ConvSpec("convPb", None, 256, 65, 1, 0),
        ConvSpec("convDa", "bnDa", 128, 256, 3, 1),
        ConvSpec("convDb", None, 256, n_ids + 1, 1, 0),
    ]


def refinenet_specs() -> List[ConvSpec]:
    """refinenet.py:21-47 in forward order (refinenet.py:56-81)."""
    return [
        ConvSpec("conv1a", "bn1a", 1, 64, 3, 0),
        ConvSpec("conv1b", "bn1b", 64, 64, 3, 0),
        ConvSpec("conv2a", "bn2a", 64, 128, 3, 0),
        ConvSpec("conv2b", "bn2b", 128, 128, 3, 0, pool=True),
        ConvSpec("conv3a", "bn3a", 128, 128, 3, 1),
        ConvSpec("conv3b", "bn3b", 128, 128, 3, 1, ups=True),
        ConvSpec("conv4a", "bn4a", 128, 128, 3, 1),
        ConvSpec("conv4b", "bn4b", 128, 128, 3, 1, ups=True),
        ConvSpec("conv5a", "bn5a", 128, 64, 3, 1),
        ConvSpec("conv5b", "bn5b", 64, 64, 3, 1, ups=True),
        ConvSpec("convPa", "bnPa", 64, 64, 3, 1),
        ConvSpec("convPb", None, 64, 1, 1, 0),
    ]


def specs_for(kind: str, n_ids: int = 16) -> List[ConvSpec]:
    if kind == "detector":
        return detector_specs(n_ids)
    if kind == "refinenet":
        return refinenet_specs()
    raise ValueError(f"unknown model kind {kind!r}")


BN_EPS = 1e-5  # torch.nn.BatchNorm2d default, never overridden in the reference


def state_dict_keys(kind: str, n_ids: int = 16) -> List[str]:
    """Tensor names in the order the C-ABI ``dcx_*_create`` calls expect them."""
    keys: List[str] = []
    for s in specs_for(kind, n_ids):
        keys += [f"{s.name}.weight", f"{s.name}.bias"]
        if s.bn:
            keys += [f"{s.bn}.weight", f"{s.bn}.bias",
                     f"{s.bn}.running_mean", f"{s.bn}.running_var"]
    return keys


def synthetic_state_dict(kind: str, seed: int, n_ids: int = 16) -> StateDict:
    """Deterministic random weights with non-trivial BN statistics.

    He-normal conv weights, small biases, BN gamma~U(0.5,1.5), beta~N(0,.1),
    running_mean~N(0,.1), running_var~U(0.5,1.5).
    Draw order is fixed: per layer weight, bias, then the four BN tensors.
    """
    rng = np.random.default_rng([seed, 0 if kind == "detector" else 1])
    sd: StateDict = {}
    for s in specs_for(kind, n_ids):
        fan_in = s.cin * s.ksize * s.ksize
        w = rng.standard_normal((s.cout, s.cin, s.ksize, s.ksize), dtype=np.float32)
        sd[f"{s.name}.weight"] = (w * np.float32(np.sqrt(2.0 / fan_in))).astype(np.float32)
        sd[f"{s.name}.bias"] = (rng.standard_normal(s.cout, dtype=np.float32)
                                * np.float32(0.05)).astype(np.float32)
        if s.bn:
            sd[f"{s.bn}.weight"] = rng.uniform(0.5, 1.5, s.cout).astype(np.float32)
            sd[f"{s.bn}.bias"] = (rng.standard_normal(s.cout, dtype=np.float32)
                                  * np.float32(0.1)).astype(np.float32)
            sd[f"{s.bn}.running_mean"] = (rng.standard_normal(s.cout, dtype=np.float32)
                                          * np.float32(0.1)).astype(np.float32)
            sd[f"{s.bn}.running_var"] = rng.uniform(0.5, 1.5, s.cout).astype(np.float32)
    return sd


def diverse_ids_bias_shift(ids_logits: np.ndarray, loc_argmax: np.ndarray, n_ids: int, k: int = 16, iters: int = 400):
    """Per-class shift of ``convDb.bias[0:n_ids]`` that makes the ``k`` strongest firing cells carry as many DISTINCT ids as
    possible.  Random-init ids heads (net.py:48 ``convDb``) let one or two classes win every cell, so fixtures built on them
    exercise the 17-way arg-max on near-constant winners; this equalises, per class, the best margin over the dust-bin among the
    cells the class wins (damped fixed-point iteration, best iterate kept).

    ids_logits (n_ids+1, ...cells) and loc_argmax (...cells) of ONE weight set on some frames -- whoever calls this decides whose
    logits they are (oracle/make_golden.py: the reference's).  Returns (shift float32 [n_ids], distinct ids among the top k)."""
    live = (np.asarray(loc_argmax).reshape(-1) != 64)
    z = np.asarray(ids_logits, dtype=np.float64).reshape(n_ids + 1, -1)[:, live]
    mz = z[:n_ids] - z[n_ids][None]
    b = -mz.max(1)
    best_key, best_b, best_d = (-1, -1.0), b.copy(), 0
    for _ in range(iters):
        zz = mz + b[:, None]
        win, m = zz.argmax(0), zz.max(0)
        order = np.argsort(-m, kind="stable")
        topk = order[:k]
        d = len(set(win[topk].tolist()))
        # The fixed point of this iteration sits ON decision boundaries (a class's best cell is one it barely wins, and the k-th / k+1-th
        # margins meet), which would plant exact ties into the fixtures.  Keep the iterate with the most distinct ids whose firing
        # cells are decided by a healthy margin: class top-2 gap and distance of the k-th / (k+1)-th margins, both capped at 1e-3.
        part = np.partition(zz[:, topk], n_ids - 2, axis=0)
        gap = float(min((part[n_ids - 1] - part[n_ids - 2]).min(), m[order[k - 1]] - m[order[min(k, len(order) - 1)]], 1e-3))
        if (d, gap) > best_key:
            best_key, best_b, best_d = (d, gap), b.copy(), d
        top = np.array([m[win == c].max() if (win == c).any() else m.min() for c in range(n_ids)])
        b = b - 0.15 * (top - np.median(top))
    return (best_b - best_b.mean()).astype(np.float32), int(best_d)


def state_dict_sha256(sd: StateDict, kind: str, n_ids: int = 16) -> str:
    h = hashlib.sha256()
    for k in state_dict_keys(kind, n_ids):
        h.update(np.ascontiguousarray(sd[k], dtype=np.float32).tobytes())
    return h.hexdigest()


def validate_state_dict(sd: StateDict, kind: str, n_ids: int = 16) -> None:
    for s in specs_for(kind, n_ids):
        w = sd[f"{s.name}.weight"]
        if tuple(w.shape) != (s.cout, s.cin, s.ksize, s.ksize):
            raise ValueError(f"{kind}.{s.name}.weight has shape {tuple(w.shape)}, "
                             f"expected {(s.cout, s.cin, s.ksize, s.ksize)}")
        if tuple(sd[f"{s.name}.bias"].shape) != (s.cout,):
            raise ValueError(f"{kind}.{s.name}.bias has wrong shape")
        if s.bn:
            for t in ("weight", "bias", "running_mean", "running_var"):
                if tuple(sd[f"{s.bn}.{t}"].shape) != (s.cout,):
                    raise ValueError(f"{kind}.{s.bn}.{t} has wrong shape")


def state_dict_from_checkpoint(path: str, kind: str, n_ids: int = 16, allow_unsafe: Optional[bool] = None) -> StateDict:
    """Lightning-free reader for the reference's ``.ckpt`` files.

    ``lModel.load_from_checkpoint`` (inference.py:74,80) reads a ``torch.save``d
    dict whose ``state_dict`` keys are ``model.<layer>.<tensor>``; plain
    ``torch.save(module.state_dict())`` files (no prefix, no wrapper) load too.
    ``num_batches_tracked`` and the torchmetrics states are ignored.

    The file is read with ``weights_only=True`` (tensors and plain containers only).  A checkpoint that pickles other
    objects (e.g. hyper-parameters holding custom classes) is refused unless the caller opts in with
    ``allow_unsafe=True`` or ``DCX_ALLOW_UNSAFE_CKPT=1`` -- a full unpickle executes code from the file, so it is never
    the silent fallback; a missing or truncated file raises its own error and is not retried.
    """
    import os
    import pickle
    import torch
    if allow_unsafe is None:
        allow_unsafe = os.environ.get("DCX_ALLOW_UNSAFE_CKPT", "0") not in ("", "0")
    try:
        blob = torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        if not allow_unsafe:
            raise RuntimeError(
                f"{path} needs a full (unsafe) unpickle: {str(e).splitlines()[0]}  -- if you trust the file pass "
                "allow_unsafe=True or set DCX_ALLOW_UNSAFE_CKPT=1") from e
        blob = torch.load(path, map_location="cpu", weights_only=False)
    raw = blob.get("state_dict", blob) if isinstance(blob, dict) else blob
    sd: StateDict = {}
    for k, v in raw.items():
        if k.startswith("model."):
            k = k[len("model."):]
        if k.endswith("num_batches_tracked") or not hasattr(v, "detach"):
            continue
        sd[k] = v.detach().to(torch.float32).cpu().numpy()
    missing = [k for k in state_dict_keys(kind, n_ids) if k not in sd]
    if missing:
        raise KeyError(f"checkpoint {path} lacks {kind} tensors: {missing[:4]}...")
    sd = {k: sd[k] for k in state_dict_keys(kind, n_ids)}
    validate_state_dict(sd, kind, n_ids)
    return sd


def save_lightning_style_checkpoint(path: str, sd: StateDict, full: bool = False, pl_version: str = "2.1.0") -> None:
    """Write ``sd`` the way Lightning would (``state_dict`` + ``model.`` prefix).

    Used by tests and the benchmark to exercise :func:`load_models` end to end with synthetic weights.  ``full=True``
    writes every top-level entry a ``Trainer.fit`` checkpoint of the reference's training scripts carries
    (train.py:37-50: one ``ModelCheckpoint(monitor="val_loss", save_top_k=10)`` callback, Adam from
    net.py:160-162 / refinenet.py:176-179) in the layout of ``pytorch_lightning`` ``pl_version`` (1.9.x and 2.1.x, the two
    pins of the reference's requirement files): ``epoch``, ``global_step``, ``pytorch-lightning_version``,
    ``state_dict``, ``loops``, ``callbacks`` (keyed by the callback's ``state_key`` string, holding paths and score
    tensors), ``optimizer_states`` (Adam moments per parameter), ``lr_schedulers``, and for 2.x ``hparams_name`` /
    ``hyper_parameters``.  Only tensors and plain containers, i.e. what ``torch.load(weights_only=True)`` accepts.
    """
    import collections
    import torch
    state = collections.OrderedDict()
    for k, v in sd.items():
        state[f"model.{k}"] = torch.from_numpy(np.ascontiguousarray(v))
        if k.endswith("running_var"):
            state[f"model.{k}".replace("running_var", "num_batches_tracked")] = torch.tensor(369700 if full else 0)
    out = {"state_dict": state, "epoch": 0, "global_step": 0}
    if full:
        steps, epochs = 369700, 100
        ck = f"/home/user/deepcharuco/src/tb_logs/ckpts/longrun-epoch={epochs - 1}-step={steps}.ckpt"

        def progress(n):
            return {"total": {"ready": n, "completed": n, "started": n, "processed": n},
                    "current": {"ready": n % 3697, "completed": n % 3697, "started": n % 3697, "processed": n % 3697}}
        loops = {
            "fit_loop": {"state_dict": {}, "epoch_loop.state_dict": {"_batches_that_stepped": steps},
                         "epoch_loop.batch_progress": dict(progress(steps), is_last_batch=True),
                         "epoch_loop.scheduler_progress": progress(0),
                         "epoch_loop.automatic_optimization.state_dict": {},
                         "epoch_loop.automatic_optimization.optim_progress": {"optimizer": {"step": progress(steps), "zero_grad": progress(steps)}},
                         "epoch_loop.manual_optimization.state_dict": {},
                         "epoch_loop.val_loop.state_dict": {}, "epoch_loop.val_loop.batch_progress": dict(progress(40), is_last_batch=True),
                         "epoch_progress": progress(epochs)},
            "validate_loop": {"state_dict": {}, "batch_progress": dict(progress(0), is_last_batch=False)},
            "test_loop": {"state_dict": {}, "batch_progress": dict(progress(0), is_last_batch=False)},
            "predict_loop": {"state_dict": {}, "batch_progress": progress(0)},
        }
        cb_key = ("ModelCheckpoint{'monitor': 'val_loss', 'mode': 'min', 'every_n_train_steps': 0, 'every_n_epochs': 1, "
                  "'train_time_interval': None}")
        if pl_version.startswith("1."):
            cb_key = cb_key[:-1] + ", 'save_on_train_epoch_end': None}"
        params = [k for k in state if k.endswith((".weight", ".bias"))]       # Adam sees learnable tensors only
        opt_state = {}
        for i, k in enumerate(params):
            opt_state[i] = {"step": torch.tensor(float(steps)), "exp_avg": torch.zeros_like(state[k]),
                            "exp_avg_sq": torch.full_like(state[k], 1e-6)}
        group = {"lr": 0.005, "betas": (0.9, 0.999), "eps": 1e-08, "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(params)))}
        out.update({
            "epoch": epochs - 1, "global_step": steps, "pytorch-lightning_version": pl_version, "loops": loops,
            "callbacks": {cb_key: {"monitor": "val_loss", "best_model_score": torch.tensor(0.0123), "best_model_path": ck,
                                   "current_score": torch.tensor(0.0125), "dirpath": ck.rsplit("/", 1)[0],
                                   "best_k_models": {ck: torch.tensor(0.0123), ck.replace("99", "98"): torch.tensor(0.0131)},
                                   "kth_best_model_path": ck.replace("99", "98"), "kth_value": torch.tensor(0.0131),
                                   "last_model_path": ""}},
            "optimizer_states": [{"state": opt_state, "param_groups": [group]}],
            "lr_schedulers": [],
        })
        if not pl_version.startswith("1."):
            out.update({"hparams_name": "kwargs", "hyper_parameters": {"lr": 0.005, "n_ids": 16, "tags": ["longrun"]}})
    torch.save(out, path)


# --------------------------------------------------------------------------
# synthetic frames


def synthetic_frames(kind: str, seed: int, batch: int, height: int, width: int) -> np.ndarray:
    """Seeded uint8 grayscale frames, shape (batch, height, width).

    ``noise``: iid uniform 0..255.  ``board``: a perspective-warped checkerboard
    over a smooth background plus mild noise, so activations have the spatial
    structure (edges, flat regions) of a real ChArUco frame.  ``board4``: the
    same scene rendered at (height/4, width/4), enlarged x4 (nearest) with fresh
    per-pixel noise -- 10x cheaper to generate at 1280x960, used where hundreds
    of high-resolution candidates are needed (fixed-K selection, cfg5).
    Frame ``i`` of a batch depends only on ``(kind, seed + i, height, width)``.
    """
    kinds = {"noise": 0, "board": 1, "board4": 2}
    if kind not in kinds:
        raise ValueError(f"unknown frame kind {kind!r}")
    if kind == "board4" and (height % 4 or width % 4):
        raise ValueError("board4 frames need height and width divisible by 4")
    out = np.empty((batch, height, width), np.uint8)
    for i in range(batch):
        rng = np.random.default_rng([seed + i, height, width, kinds[kind]])
        if kind == "noise":
            out[i] = rng.integers(0, 256, (height, width), dtype=np.uint8)
        elif kind == "board":
            out[i] = _board_frame(rng, height, width)
        else:
            small = _board_frame(rng, height // 4, width // 4).astype(np.int16)
            big = np.repeat(np.repeat(small, 4, axis=0), 4, axis=1)
            big += rng.integers(-6, 7, (height, width), dtype=np.int16)
            out[i] = np.clip(big, 0, 255).astype(np.uint8)
    return out


def _board_frame(rng: np.random.Generator, h: int, w: int) -> np.ndarray:
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    # random homography close to a rotation+scale about the image centre
    ang = rng.uniform(-0.6, 0.6)
    sc = rng.uniform(0.5, 0.9) * min(h, w)
    cx, cy = w / 2 + rng.uniform(-0.1, 0.1) * w, h / 2 + rng.uniform(-0.1, 0.1) * h
    px, py = rng.uniform(-0.25, 0.25, 2) / max(h, w)
    xr, yr = xs - cx, ys - cy
    den = 1.0 + px * xr + py * yr
    u = (np.cos(ang) * xr + np.sin(ang) * yr) / den / sc + 0.5
    v = (-np.sin(ang) * xr + np.cos(ang) * yr) / den / sc + 0.5
    inside = (u >= 0) & (u < 1) & (v >= 0) & (v < 1)
    squares = ((np.floor(u * 5) + np.floor(v * 5)) % 2).astype(np.float64)
    # low-frequency background
    bg = 110 + 60 * np.sin(xs / w * rng.uniform(2, 6) + rng.uniform(0, 6)) \
             * np.cos(ys / h * rng.uniform(2, 6) + rng.uniform(0, 6))
    lo, hi = rng.uniform(10, 60), rng.uniform(180, 250)
    img = np.where(inside, lo + (hi - lo) * squares, bg)
    img = img + rng.normal(0, 4.0, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def frames_sha256(frames: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(frames).tobytes()).hexdigest()
