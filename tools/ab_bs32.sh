#!/usr/bin/env bash
# A/B of library variants (tools/sweep_build.sh) on the bs=32 step, alternating rounds on one box.   usage: tools/ab_bs32.sh name [name ...]
for r in 1 2 3; do
  for v in "$@"; do
    DCX_LIB=$GRAFT_REPO_ROOT/build_variants/lib_$v.so python bench.py --no-extras --no-cpu-baseline --steps 60 --parity-frames 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', 'bs32', d['value'], 'fps', d['ms_per_step'], 'ms  frac', r['frac'], 'clk', r['shader_clock_ghz'], 'e2e', r['e2e_executed_frac'], 'parity', d['parity']['mismatched_frames'])"
  done
done
