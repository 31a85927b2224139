#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + PMC passes of the bench command (the 10 GB headline build).
# Usage: tools/collect_profiles.sh <tag> [extra bench args]     -> gpurun_out/prof_<tag>/...
# The program itself follows `--` (python3 bench.py): no env/bash hop between rocprofv3 and the process that touches the GPU.
# PMC passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass) and carry no trace options.
set -u
TAG=${1:-r03}
shift || true
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t --output-format csv -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-cli "$@" > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.log"
rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o f --output-format csv -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-cli "$@" > /dev/null 2> "$OUT/pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o w --output-format csv -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-cli "$@" > /dev/null 2> "$OUT/pmc_write.log"
python3 "$R/tools/summarize_profiles.py" "$OUT" "$TAG" 2
ls -la "$OUT"
