#!/bin/bash
# Run ON THE GPU BOX (via gpurun): SQ counters of the radix-sort kernels alone (tools/sortbench.hip), one build per argument.
# Usage: tools/gpu_sortbench_pmc.sh <out-dir under gpurun_out> <n> <kbits> <skew> <binary> [<binary> ...]
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; N=$2; KB=$3; SK=$4
shift 4
mkdir -p "$OUT"
cd /tmp
for B in "$@"; do
  T=$(basename "$B")
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_ACTIVE_INST_LDS \
      -d "$OUT/pmc1_$T" -o p --output-format csv -- "$R/$B" $N $KB 1 $SK > "$OUT/pmc1_$T.log" 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD \
      -d "$OUT/pmc2_$T" -o p --output-format csv -- "$R/$B" $N $KB 1 $SK > "$OUT/pmc2_$T.log" 2>&1
  for P in 1 2; do
    F=$(find "$OUT/pmc${P}_$T" -name '*counter_collection.csv' | head -1)
    echo "== $T skew $SK pass $P" >> "$OUT/summary.txt"
    [ -n "$F" ] && python3 "$R/tools/pmc_summary.py" "$F" k_rs_ >> "$OUT/summary.txt" 2>&1
    rm -rf "$OUT/pmc${P}_$T"
  done
done
cat "$OUT/summary.txt"
