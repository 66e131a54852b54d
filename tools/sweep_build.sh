#!/usr/bin/env bash
# Kernel-tuning aid: build variants of the library with different -D overrides into build_variants/ (git-ignored, but
# shipped to the GPU box), then run each with DCX_LIB=<variant>.  usage: tools/sweep_build.sh name "-DFOO=1 -DBAR=2" [name flags ...]
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/deepcharuco_amd/csrc
mkdir -p "$ROOT/build_variants"
make -C "$CS" -j8 > /dev/null
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -pragma-unroll-threshold=200000"
while [ $# -ge 2 ]; do
    name=$1; extra=$2; shift 2
    (
      /opt/rocm/bin/hipcc $FLAGS $extra -Rpass-analysis=kernel-resource-usage -c "$CS/dcx_conv_mfma.hip" -o "$ROOT/build_variants/$name.o" 2> "$ROOT/build_variants/$name.log"
      /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 "$ROOT/build_variants/$name.o" "$CS/dcx_misc.o" "$CS/dcx_tail.o" "$CS/dcx_api.o" -o "$ROOT/build_variants/lib_$name.so"
      rm -f "$ROOT/build_variants/$name.o"
      echo "$name: w2h $(grep -A8 'wino2h_kernelI12DcxWino2hCfgILi8ELi16ELb1' "$ROOT/build_variants/$name.log" | grep -oE 'VGPRs: [0-9]+|ScratchSize \[bytes/lane\]: [0-9]+' | paste -sd' ') | w2p $(grep -A8 'wino2p_kernelI12DcxWino2pCfgILi8ELi16ELi2' "$ROOT/build_variants/$name.log" | grep -oE 'VGPRs: [0-9]+|ScratchSize \[bytes/lane\]: [0-9]+|Occupancy \[waves/SIMD\]: [0-9]+' | paste -sd' ')"
    ) &
done
wait
