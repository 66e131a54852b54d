set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; export TMPDIR=/tmp
cd $R
python -m pytest tests -m gpu -x -q -k "conv or golden or soak or parity_128" 2>&1 | tail -3
for m in 1 0 1 0; do DCX_XCD_WALK=$m python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5 > $O/xcd_$m.json 2>/dev/null; python tools/bench_summary.py $O/xcd_$m.json | head -4; done
cd /tmp
for m in 1 0; do
  rm -rf $O/prof_xcd${m}_fetch $O/prof_xcd${m}_write
  DCX_XCD_WALK=$m timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/prof_xcd${m}_fetch -o fetch -- python $R/bench.py --no-extras --no-cpu-baseline --steps 4 --warmup 2 --no-profile > $O/rocprof_xcd${m}_fetch.log 2>&1
done
rm -rf $O/prof_calib; timeout 120 rocprofv3 --pmc FETCH_SIZE -d $O/prof_calib -o calib -- $R/tools/ubench/fetch_calib > $O/calib.log 2>&1
grep "known" $O/calib.log
