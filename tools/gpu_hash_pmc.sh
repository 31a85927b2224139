#!/bin/bash
# Run ON THE GPU BOX (via gpurun): SQ counters of the level-0 phrase naming kernel (k_for_each_agg) in ONE build of <reads> reads.
# Usage: tools/gpu_hash_pmc.sh <out-dir under gpurun_out> [reads] [genome]
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; READS=${2:-6622517}; GENOME=${3:-33000000}
mkdir -p "$OUT"
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_ACTIVE_INST_LDS \
    -d "$OUT/pmc1" -o p --output-format csv -- python3 "$R/tools/gpu_one_build.py" $READS $GENOME > "$OUT/pmc1.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD \
    -d "$OUT/pmc2" -o p --output-format csv -- python3 "$R/tools/gpu_one_build.py" $READS $GENOME > "$OUT/pmc2.log" 2>&1
for P in 1 2; do
  F=$(find "$OUT/pmc$P" -name '*counter_collection.csv' | head -1)
  echo "== pass $P" >> "$OUT/summary.txt"
  [ -n "$F" ] && python3 "$R/tools/pmc_summary.py" "$F" k_for_each_agg k_start_bits >> "$OUT/summary.txt" 2>&1
  rm -rf "$OUT/pmc$P"
done
cat "$OUT/summary.txt"
