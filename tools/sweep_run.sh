for v in "$@"; do
  DCX_LIB=$GRAFT_REPO_ROOT/build_variants/lib_$v.so python bench.py --no-extras --no-cpu-baseline --steps 60 --parity-frames 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', d['value'], r['frac'], r['shader_clock_ghz'], r['e2e_executed_frac'], {k.split('<')[2][:14] if k.count('<')>1 else k: v['frac'] for k,v in r['per_kernel'].items()})"
done
