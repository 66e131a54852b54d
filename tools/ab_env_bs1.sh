#!/usr/bin/env bash
# A/B of environment switches on the reference's bs=1 protocol (bench.py's bs1_reference_protocol, 1,500 calls), interleaved rounds on
# one box.   usage: tools/ab_env_bs1.sh "DCX_W2HS=0" "DCX_W2HS=1" [...]      (each argument: one or more VAR=value, space separated)
for r in 1 2 3; do
  for v in "$@"; do
    env $v python - "$v" <<'PY' 2>/dev/null
import sys, torch
sys.path.insert(0, ".")
import bench as Bn
cx = Bn.Ctx(); cx.dev = torch.device("cuda", 0); torch.cuda.set_device(0)
r = Bn.bs1_reference_protocol(cx, n_iter=1500)
print(f"{sys.argv[1]:24s} bs1 {r['value']:8.1f} calls/s  {r['ms_per_call']:.4f} ms  parity mismatches {r['parity']['mismatched_frames']}", flush=True)
PY
  done
done
