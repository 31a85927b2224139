#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + PMC passes of the bench command.
# Usage: tools/collect_profiles.sh <tag>      -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t --output-format csv -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.log"
rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o f --output-format csv -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o w --output-format csv -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_write.log"
python3 "$R/tools/summarize_profiles.py" "$OUT" "$TAG"
