"""The N>1 product path with REAL kernels: ranks launched with torch.distributed.run run
`sharding.infer_frames_sharded` / `infer_batches_sharded` through the HIP pipeline and rank 0 compares every frame with
the oracle.  gloo: two ranks share the one visible GPU; nccl (RCCL): one GPU per rank, skipped on a 1-GPU box."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(backend, world, tmp_path):
    out = str(tmp_path / f"sharded_{backend}.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("DCX_FORCE_CFG", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(REPO, "tests", "sharded_gpu_worker.py"), backend, out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    v = json.load(open(out))
    rep = os.path.join(REPO, "gpurun_out")
    os.makedirs(rep, exist_ok=True)
    with open(os.path.join(rep, f"sharded_{backend}_report.json"), "w") as f:
        json.dump(v, f, indent=1)
    return v


def _launch8(mode, tmp_path, timeout=700):
    """Eight ranks; gloo on a 1-GPU box (the ranks time-slice the GPU), RCCL when eight GPUs are visible."""
    backend = "nccl" if torch.cuda.device_count() >= 8 else "gloo"
    out = str(tmp_path / f"world8_{mode}.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    env.pop("DCX_FORCE_CFG", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(REPO, "tests", "sharded_world8_worker.py"), mode, backend, out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0:
        import glob
        errs = "".join(open(f).read()[-3000:] for f in sorted(glob.glob(out + ".rank*.err"))[:2])
        raise AssertionError(errs + r.stdout[-1500:] + r.stderr[-2500:])
    v = json.load(open(out))
    rep = os.path.join(REPO, "gpurun_out")
    os.makedirs(rep, exist_ok=True)
    with open(os.path.join(rep, f"sharded_world8_{mode}_{backend}_report.json"), "w") as f:
        json.dump(v, f, indent=1)
    return v


def test_world8_cfg4_full_size_every_rank_checked(tmp_path):
    """BASELINE configs[3] at its true world size: 1,024 frames of 320x240 over EIGHT ranks (128 each) through
    infer_batches_sharded, two batches in flight, plus the ragged 1,021-frame split; frames of every rank's shard vs the oracle."""
    v = _launch8("cfg4", tmp_path)
    assert v["world"] == 8 and v["frames"] == 1024 and v["results_returned"] == [1024, 1024, 1021]
    assert [b - a for a, b in v["split"]] == [128] * 8
    assert [b - a for a, b in v["split_ragged"]] == [128] * 5 + [127] * 3
    assert v["mismatched"] == 0 and v["mismatched_per_rank"] == [0] * 8 and v["frames_checked"] >= 8 * 10
    assert v["corners_checked"] > 500
    # round 5: the pool is sized for 64 corners per frame on average and has no per-frame cap: the busiest of the 1,024 frames
    # (some fire more than 64 cells) is complete and identical to the oracle; pools that are far too small are repeated collectively
    assert v["pool_per_rank"] == 128 * 64 and v["busiest_frame"]["identical"] and v["busiest_frame"]["corners"] == v["max_corners"]
    assert v["overflow_rerun_identical"]


def test_world8_cfg5_reduced_every_rank_checked(tmp_path):
    """BASELINE configs[4] reduced in batch only: 8 ranks x 4 frames of 1280x960, exactly 16 corners per frame, kmax = 16."""
    v = _launch8("cfg5", tmp_path)
    assert v["world"] == 8 and v["frames"] == 32 and v["all_frames_have_16"] and v["corners_per_frame_seen"] == [16]
    assert v["mismatched"] == 0 and v["mismatched_per_rank"] == [0] * 8 and v["frames_checked"] >= 16


def test_world8_cfg5_full_size_every_rank_checked(tmp_path):
    """BASELINE configs[4] at its FULL size: 256 frames of 1280x960 over eight ranks (32 each; 8 x 12.6 GB of workspace on the
    one 288 GB GPU), exactly 16 corners in every frame, kmax = 16, two batches in flight; three frames of EVERY rank's shard vs the
    oracle."""
    v = _launch8("cfg5full", tmp_path, timeout=1500)
    assert v["world"] == 8 and v["frames"] == 256 and v["all_frames_have_16"] and v["corners_per_frame_seen"] == [16]
    assert v["mismatched"] == 0 and v["mismatched_per_rank"] == [0] * 8 and v["frames_checked"] == 24 and v["corners_checked"] == 24 * 16


def test_two_ranks_on_one_gpu_gloo_real_kernels(tmp_path):
    v = _launch("gloo", 2, tmp_path)
    assert v["world"] == 2 and v["split"] == [[0, 6], [6, 11]]
    assert v["mismatched_single"] == 0 and v["mismatched_pipelined"] == 0 and v["mismatched_bgr"] == 0 and v["one_frame_ok"] and v["corners"] > 50


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank; this box has one")
def test_two_ranks_rccl_real_kernels(tmp_path):
    v = _launch("nccl", 2, tmp_path)
    assert v["mismatched_single"] == 0 and v["mismatched_pipelined"] == 0 and v["mismatched_bgr"] == 0 and v["one_frame_ok"]


def test_one_rank_rccl_real_kernels(tmp_path):
    """RCCL itself on the 1-GPU box: a process group of ONE rank runs the same product path (all_gather_into_tensor on the
    side stream, double-buffered, barrier) -- what the 8-GPU job does per rank, minus the peers."""
    v = _launch("nccl", 1, tmp_path)
    assert v["world"] == 1 and v["mismatched_single"] == 0 and v["mismatched_pipelined"] == 0 and v["mismatched_bgr"] == 0 and v["one_frame_ok"] and v["corners"] > 50


def test_bench_n_gt_1_code_path_with_one_rccl_rank(tmp_path):
    """bench.py --force-dist: the N>1 code path of the benchmark (nccl process group, communicator warm-up, side-stream gather,
    barrier + MAX all-reduce around the timed region, parity of frames taken out of the gathered buffer) with one rank."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.pop("DCX_FORCE_CFG", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--force-dist", "--steps", "4", "--warmup", "2",
                        "--no-cpu-baseline", "--no-extras", "--parity-frames", "4"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["gather_overlapped"] is True and line["parity"]["mismatched_frames"] == 0 and line["parity"]["frames_checked"] >= 4
    assert line["n_gpus"] == 1 and line["value"] > 1000


_LAUNCHER_VARS = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT",
                  "TORCHELASTIC_RUN_ID", "DCX_FORCE_CFG")


def _bare_env():
    return {k: v for k, v in os.environ.items() if k not in _LAUNCHER_VARS}


def test_bench_self_launches_its_ranks_from_a_bare_shell():
    """`python bench.py --gpus 2 ...` exactly as the driver starts it (no RANK / WORLD_SIZE / MASTER_* in the environment):
    bench.py starts its own two ranks under torch.distributed.run, rank 0 prints ONE JSON line with n_gpus = 2, frames of BOTH
    ranks' timed batches are identical to the oracle, exit code 0.  gloo because the box has one GPU (RCCL when it has two)."""
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", backend, "--steps", "2",
                        "--warmup", "1", "--no-extras"], env=_bare_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["ranks"]["world_size"] == 2 and line["ranks"]["distinct_processes"] == 2 and line["ranks"]["backend"] == backend
    assert line["parity"]["mismatched_frames"] == 0 and line["parity"]["frames_checked"] >= 10     # 8 of rank 0 + >= 2 of rank 1
    assert line["gather_overlapped"] in (True, False) and line["value"] > 1000


def test_bench_refuses_rccl_with_fewer_gpus_than_ranks():
    """Never a silent gloo: --gpus N with the default backend (RCCL) and fewer than N visible GPUs is a clear error."""
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0",
                        "--no-extras"], env=_bare_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "GPU(s) visible" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_single_gpu_line_has_no_rank_block():
    """--gpus 1 stays a single in-process run: no process group, no `ranks` block, n_gpus = 1."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--no-extras", "--no-cpu-baseline"], env=_bare_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and "ranks" not in line and "gather_overlapped" not in line and "roofline" in line
