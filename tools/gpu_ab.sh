#!/bin/bash
# Run ON THE GPU BOX: the bench command once per variant, same box, same process layout.  A variant is "name:ENV=VAL,ENV=VAL" (dev: prefix ->
# the development library tools/_build/libgrlbwt_dev.so).  Writes gpurun_out/<tag>/bench_<name>.json + sites_<name>.txt and a summary line each.
#   tools/gpu_ab.sh r06b base: two:GRLBWT_ASM_TWO_PASS=1 ipt3:dev:GRLBWT_DEV_XS_IPT=3
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
for v in "$@"; do
    name=${v%%:*}; rest=${v#*:}
    envs=()
    if [[ $rest == dev:* ]]; then rest=${rest#dev:}; envs+=("GRLBWT_HIP_LIB=$R/tools/_build/libgrlbwt_dev.so"); fi
    IFS=',' read -ra kv <<< "$rest"
    for e in "${kv[@]}"; do [[ -n $e ]] && envs+=("$e"); done
    env "${envs[@]}" GRLBWT_BENCH_DETAIL=2 python3 "$R/bench.py" --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-cli --no-extra > "$OUT/bench_$name.json" 2> "$OUT/sites_$name.txt"
    python3 - "$OUT/bench_$name.json" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
g = d["roofline_groups"]
print("%-10s %8.2f ms  md5 %s  AB %.1f C %.1f hash %.1f dict %.1f  kernels %.1f  peak %.1f GB" % (sys.argv[2], d["ms_per_step"], d["image"]["md5"][:8],
      g["induce_AB"]["kernel_ms_total"], g["induce_C"]["kernel_ms_total"], g["hash_emit"]["kernel_ms_total"], g["dict_stage"]["kernel_ms_total"], d["kernel_ms_total"],
      d["device_memory"]["peak_live_bytes"] / 1e9))
PY
done
