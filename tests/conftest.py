import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")
CASES = ["tiny_noise_64x96", "noise_240x320", "board_240x320", "board_480x640", "board4_960x1280", "diverse_ids_240x320",
         # the reference's only real input (src/benchmark.py:34-35: IMG_7412.png, a 320x240 colour photo) under two weight sets
         "img7412_240x320", "img7412_diverse_240x320"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when no GPU is visible, e.g. `pytest tests` in the build container
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


class GoldenCase:
    """A fixture file + the regenerated (SHA-checked) weights and frame it was produced from."""

    def __init__(self, name):
        from deepcharuco_amd import weights as W
        self.name = name
        self.fx = np.load(os.path.join(GOLDEN, f"{name}.npz"))
        self.meta = json.loads(str(self.fx["meta"]))
        m = self.meta
        self.n_ids = m["n_ids"]
        self.sd_dc = W.synthetic_state_dict("detector", m["wseed"], m["n_ids"])
        if "convDb_bias" in self.fx:          # the whole ids-head bias (per-class equalisation of the diverse-ids case)
            self.sd_dc["convDb.bias"] = self.fx["convDb_bias"].astype(np.float32).copy()
        self.sd_dc["convDb.bias"][m["n_ids"]] = self.fx["dust_bias"]
        self.sd_rn = W.synthetic_state_dict("refinenet", m["wseed"] + 1)
        self._bgr = None
        if "bgr_image" in self.fx:            # a real colour image travels inside the fixture; gray = the 8-bit fixed-point formula
            from deepcharuco_amd.imgproc import bgr2gray_fixed_point
            self._bgr = np.ascontiguousarray(self.fx["bgr_image"])
            self.frame = bgr2gray_fixed_point(self._bgr)
        else:
            self.frame = W.synthetic_frames(m["kind"], m["fseed"], 1, m["H"], m["W"])[0]
        assert W.state_dict_sha256(self.sd_dc, "detector", m["n_ids"]) == str(self.fx["sha_dc"]), \
            "regenerated detector weights differ from the ones the fixture was made with"
        assert W.state_dict_sha256(self.sd_rn, "refinenet") == str(self.fx["sha_rn"])
        assert W.frames_sha256(self.frame) == str(self.fx["sha_frame"])

    @property
    def bgr(self):
        if self._bgr is not None:
            return self._bgr
        return np.repeat(self.frame[..., None], 3, axis=2)


_cache = {}


@pytest.fixture(params=CASES)
def golden(request):
    if request.param not in _cache:
        _cache[request.param] = GoldenCase(request.param)
    return _cache[request.param]


@pytest.fixture
def golden_tiny():
    if "tiny_noise_64x96" not in _cache:
        _cache["tiny_noise_64x96"] = GoldenCase("tiny_noise_64x96")
    return _cache["tiny_noise_64x96"]
