#!/usr/bin/env bash
# bs=1 reference protocol (bench.py's bs1_reference_protocol, 1,500 calls) for library variants built by tools/sweep_build.sh,
# two interleaved rounds on one box.   usage: tools/ab_bs1.sh name [name ...]
for r in 1 2; do
  for v in "$@"; do
    DCX_LIB=$GRAFT_REPO_ROOT/build_variants/lib_$v.so python - "$v" <<'PY' 2>/dev/null
import sys, torch
sys.path.insert(0, ".")
import bench as Bn
cx = Bn.Ctx(); cx.dev = torch.device("cuda", 0); torch.cuda.set_device(0)
r = Bn.bs1_reference_protocol(cx, n_iter=1500)
print(f"{sys.argv[1]:10s} bs1 {r['value']:8.1f} calls/s  {r['ms_per_call']:.4f} ms  parity mismatches {r['parity']['mismatched_frames']}", flush=True)
PY
  done
done
