"""CPU: host-side logic of the product package and the C-ABI surface (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import REPO
from deepcharuco_amd import weights as W


def test_library_loads_and_exports_every_declared_symbol():
    from deepcharuco_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    handle = ctypes.CDLL(_lib.LIB_PATH)
    header = open(os.path.join(REPO, "include", "deepcharuco_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(dcx_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(handle, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.SIGNATURES), "ctypes signature table and header disagree"
    lib = _lib.lib()
    assert b"gfx950" in lib.dcx_version()
    assert b"DCX_E_SHAPE" in lib.dcx_error_string(-2)


def test_null_arguments_are_rejected_without_a_gpu():
    from deepcharuco_amd import _lib
    lib = _lib.lib()
    assert lib.dcx_pre_image(None, None, 10, None) == -1
    assert lib.dcx_argmax2d(None, 1, 2, 2, None, None) == -1
    assert lib.dcx_detector_workspace_bytes(None, 1, 240, 320) == 0
    h = ctypes.c_void_p()
    assert lib.dcx_detector_create(ctypes.byref(h), None, 64, 16) == -1


def test_cpu_device_is_refused_loudly():
    from deepcharuco_amd.models.net import dcModel
    from deepcharuco_amd.inference import infer_image
    with pytest.raises(RuntimeError, match="no CPU fallback|no CPU path|MI355X"):
        dcModel(16).load_state_dict(W.synthetic_state_dict("detector", 0), device="cpu")
    with pytest.raises(RuntimeError):
        infer_image(np.zeros((16, 16, 3), np.uint8), 16, None, None, device="cpu")


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(REPO, "deepcharuco_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f"{f} imports the oracle"
                assert "F.conv2d" not in src and "torch.nn.functional" not in src, f"{f} uses stock torch ops"


def test_state_dict_layout_and_checkpoint_roundtrip(tmp_path):
    sd = W.synthetic_state_dict("detector", 11, 16)
    assert len(W.state_dict_keys("detector", 16)) == 64 and len(W.state_dict_keys("refinenet")) == 68
    n_params = sum(v.size for k, v in sd.items() if "running" not in k)
    assert n_params == 1_242_002   # detector parameter count
    rn = W.synthetic_state_dict("refinenet", 11)
    assert sum(v.size for k, v in rn.items() if "running" not in k) == 999_233
    p = str(tmp_path / "dc.ckpt")
    W.save_lightning_style_checkpoint(p, sd)
    back = W.state_dict_from_checkpoint(p, "detector", 16)
    assert list(back) == W.state_dict_keys("detector", 16)
    assert all(np.array_equal(back[k], sd[k]) for k in sd)
    with pytest.raises(KeyError):
        W.state_dict_from_checkpoint(p, "refinenet")
    assert W.state_dict_sha256(sd, "detector") == W.state_dict_sha256(W.synthetic_state_dict("detector", 11), "detector")
    assert W.state_dict_sha256(sd, "detector") != W.state_dict_sha256(W.synthetic_state_dict("detector", 12), "detector")


def test_synthetic_frames_are_per_frame_deterministic():
    a = W.synthetic_frames("board", 5, 3, 64, 96)
    b = W.synthetic_frames("board", 6, 2, 64, 96)
    assert a.dtype == np.uint8 and a.shape == (3, 64, 96)
    assert np.array_equal(a[1], b[0]) and np.array_equal(a[2], b[1])
    assert a.std() > 20


def test_unpack_results_sorting_and_dtypes():
    from deepcharuco_amd.inference import unpack_results
    b, kmax = 3, 4
    packed = np.zeros(b + b * kmax * 4 + b * kmax * 2, np.int32)
    packed[:b] = [3, 0, 9]   # frame 2 overflows kmax
    rows = packed[b:b + b * kmax * 4].reshape(b, kmax, 4)
    rows[0, :3] = [[8, 9, 5, 1], [16, 17, 2, 2], [24, 25, 5, 3]]
    rows[2, :4] = [[1, 1, 0, 0], [2, 2, 0, 1], [3, 3, 0, 2], [4, 4, 0, 3]]
    xy = packed[b + b * kmax * 4:].view(np.float32).reshape(b, kmax, 2)
    xy[0, :3] = [[8.5, 9.25], [16.125, 17], [24, 25.5]]
    res, counts = unpack_results(packed, b, kmax, True)
    assert res[0].dtype == np.float64
    assert np.array_equal(res[0], [[16.125, 17, 2], [8.5, 9.25, 5], [24, 25.5, 5]])   # stable by id
    assert res[1].shape == (0,) and res[1].dtype == np.float64
    assert res[2].shape == (4, 3) and counts.tolist() == [3, 0, 9]
    res, _ = unpack_results(packed, b, kmax, False)
    assert res[0].dtype == np.int64 and np.array_equal(res[0], [[16, 17, 2], [8, 9, 5], [24, 25, 5]])


def test_bgr2gray_host():
    from deepcharuco_amd.imgproc import bgr2gray
    d = np.load(os.path.join(REPO, "tests", "golden", "bgr2gray_formula.npz"))
    assert np.array_equal(bgr2gray(d["bgr"]), d["gray"])


def test_solve_pnp_short_circuit():
    from deepcharuco_amd.inference import solve_pnp
    assert solve_pnp(np.zeros((3, 3)), 5, 5, 0.01, None, None) == (False, None, None)


def test_shard_range_partition():
    from deepcharuco_amd.sharding import shard_range
    for n in (0, 1, 7, 32, 1024):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
