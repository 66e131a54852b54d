#!/usr/bin/env bash
# Regenerates profiles/rNN_* on an MI355X box (run from the repo root through gpurun):
#   gpurun --timeout 1500 -- 'bash tools/run_profiles.sh 01'
# then, back in the build container:   bash tools/run_profiles.sh 01 summarize
# Counter passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc is never combined with
# tracing other than the implicit kernel dispatch records), as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
set -u
R=${1:-01}
MODE=${2:-collect}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
if [ "$MODE" = collect ]; then
    mkdir -p "$OUT"; export TMPDIR=/tmp
    (cd "$ROOT" && timeout 700 python bench.py --steps 50 --warmup 5 > "$OUT/bench.log" 2>&1; echo "bench exit $?" >> "$OUT/bench.log")
    cd /tmp
    rm -rf "$OUT"/prof_r${R}*
    B="python $ROOT/bench.py --no-cpu-baseline --no-extras"
    timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_r${R}" -o r${R} -- $B --steps 10 --warmup 3 > "$OUT/rocprof.log" 2>&1
    timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/prof_r${R}_fetch" -o fetch -- $B --steps 4 --warmup 2 --no-profile > "$OUT/rocprof_fetch.log" 2>&1
    timeout 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/prof_r${R}_write" -o write -- $B --steps 4 --warmup 2 --no-profile > "$OUT/rocprof_write.log" 2>&1
    timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
        -d "$OUT/prof_r${R}_sq" -o sq -- $B --steps 4 --warmup 2 --no-profile > "$OUT/rocprof_sq.log" 2>&1
    timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE \
        -d "$OUT/prof_r${R}_lds" -o lds -- $B --steps 4 --warmup 2 --no-profile > "$OUT/rocprof_lds.log" 2>&1
    # FETCH_SIZE calibration on known byte counts (tools/ubench/fetch_calib.hip) and the flat-walk A/B of the item order
    # (the binary is git-ignored: a fresh checkout has none -- build it here rather than lose the calibration pass)
    [ -x "$ROOT/tools/ubench/fetch_calib" ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 "$ROOT/tools/ubench/fetch_calib.hip" -o "$ROOT/tools/ubench/fetch_calib"
    timeout 120 rocprofv3 --pmc FETCH_SIZE -d "$OUT/prof_r${R}_calib" -o calib -- $ROOT/tools/ubench/fetch_calib > "$OUT/calib.log" 2>&1
    DCX_XCD_WALK=0 timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/prof_r${R}_fetch_flat" -o fetch -- $B --steps 4 --warmup 2 --no-profile > "$OUT/rocprof_fetch_flat.log" 2>&1
    # the other single-GPU configs: per-kernel times (their roofline blocks are in bench.log's other_configs)
    timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/prof_r${R}_cfg3" -o cfg3 -- $B --config cfg3 --steps 4 --warmup 2 --no-profile > "$OUT/rocprof_cfg3.log" 2>&1
    timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/prof_r${R}_cfg5" -o cfg5 -- $B --config cfg5 --steps 4 --warmup 2 --no-profile > "$OUT/rocprof_cfg5.log" 2>&1
    # bs=1 (the reference's own protocol): per-kernel times
    timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/prof_r${R}_bs1" -o bs1 -- python $ROOT/tools/bs1_loop.py 60 1 > "$OUT/rocprof_bs1.log" 2>&1
    timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE \
        -d "$OUT/prof_r${R}_lds_bs1" -o lds -- python $ROOT/tools/bs1_loop.py 60 1 > "$OUT/rocprof_lds_bs1.log" 2>&1
    (cd "$ROOT" && timeout 200 python tools/stream_bench.py > "$OUT/stream_bench.log" 2>&1)
    (cd "$ROOT" && timeout 200 python tools/layer_table.py > "$OUT/layer_table.log" 2>&1)
    tail -1 "$OUT/bench.log"
else
    cd "$ROOT"
    python tools/make_pmc_traffic.py gpurun_out/prof_r${R}_fetch/fetch_results.db gpurun_out/prof_r${R}_write/write_results.db gpurun_out/prof_r${R}_calib/calib_results.db > profiles/pmc_traffic.json
    python tools/rocprof_pmc_summary.py gpurun_out/prof_r${R}_calib/calib_results.db calib > profiles/r${R}_pmc_fetch_calibration.txt
    grep "known bytes" gpurun_out/calib.log >> profiles/r${R}_pmc_fetch_calibration.txt
    python tools/rocprof_pmc_summary.py gpurun_out/prof_r${R}_fetch_flat/fetch_results.db dcx_ > profiles/r${R}_pmc_fetch_size_flat_walk.txt
    python tools/rocprof_summary.py gpurun_out/prof_r${R}_bs1/bs1_results.db | cut -c1-220 > profiles/r${R}_kernel_stats_bs1.txt
    python tools/rocprof_summary.py gpurun_out/prof_r${R}/r${R}_results.db | cut -c1-220 > profiles/r${R}_kernel_stats.txt
    python tools/rocprof_summary.py gpurun_out/prof_r${R}_cfg3/cfg3_results.db | cut -c1-220 > profiles/r${R}_kernel_stats_cfg3.txt
    python tools/rocprof_summary.py gpurun_out/prof_r${R}_cfg5/cfg5_results.db | cut -c1-220 > profiles/r${R}_kernel_stats_cfg5.txt
    python tools/rocprof_pmc_summary.py gpurun_out/prof_r${R}_fetch/fetch_results.db dcx_ > profiles/r${R}_pmc_fetch_size.txt
    python tools/rocprof_pmc_summary.py gpurun_out/prof_r${R}_write/write_results.db dcx_ > profiles/r${R}_pmc_write_size.txt
    python tools/rocprof_pmc_summary.py gpurun_out/prof_r${R}_sq/sq_results.db dcx_conv > profiles/r${R}_pmc_sq.txt
    python tools/rocprof_pmc_summary.py gpurun_out/prof_r${R}_lds/lds_results.db dcx_conv > profiles/r${R}_pmc_lds_valu.txt
    python tools/rocprof_pmc_summary.py gpurun_out/prof_r${R}_lds_bs1/lds_results.db dcx_conv > profiles/r${R}_pmc_lds_valu_bs1.txt
    grep '^{' gpurun_out/bench.log > profiles/r${R}_bench_n1.json
    grep -v "^\[\|Warning\|warn" gpurun_out/stream_bench.log > profiles/r${R}_stream_bench.txt
    cp gpurun_out/layer_table.log profiles/r${R}_layer_table.txt
    ls -la profiles/
fi
