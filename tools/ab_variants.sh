#!/usr/bin/env bash
# A/B of library variants built by tools/sweep_build.sh (build_variants/lib_<name>.so), alternating, on one box:
#   the reference's bs=1 protocol (bench.py's bs1_reference_protocol) and the bs=32 step.   usage: tools/ab_variants.sh nameA nameB [rounds]
A=$1; B=$2; R=${3:-2}
for r in $(seq $R); do
  for v in $A $B; do
    DCX_LIB=$GRAFT_REPO_ROOT/build_variants/lib_$v.so python - "$v" <<'PY' 2>/dev/null
import json, sys, time, numpy as np, torch
sys.path.insert(0, ".")
import bench as Bn
cx = Bn.Ctx(); cx.dev = torch.device("cuda", 0); torch.cuda.set_device(0)
r = Bn.bs1_reference_protocol(cx, n_iter=1500)
print(sys.argv[1], "bs1", r["value"], "calls/s", r["ms_per_call"], "ms  parity mismatches", r["parity"]["mismatched_frames"], flush=True)
PY
    DCX_LIB=$GRAFT_REPO_ROOT/build_variants/lib_$v.so python bench.py --no-extras --no-cpu-baseline --steps 60 --parity-frames 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', 'bs32', d['value'], r['frac'], r['shader_clock_ghz'], r['e2e_executed_frac'], {k.split('<')[2][:14] if k.count('<')>1 else k: (v['ms_per_step'], v['frac']) for k,v in r['per_kernel'].items() if '8, 8' in k})"
  done
done
