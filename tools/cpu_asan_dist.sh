#!/bin/bash
# CPU only: the collection-level flow (engine logic over the serial test stand-in of the primitives, gloo transport) under
# AddressSanitizer, 2-4 ranks, every form of the dictionary stage (sharded by owner with carried records / with the record round
# trip, gathered, long-phrase levels, large-group refinement).  Prints one line per run; any "asan=" other than 0 is a finding.
#   tools/cpu_asan_dist.sh [out-dir]
set -u
cd "$(dirname "$0")/.."
OUT=${1:-/tmp/asan_out}
mkdir -p "$OUT"
make -s -C tests/hostsim asan 2>&1 | grep -i " error" && exit 1
ASAN=$(gcc -print-file-name=libasan.so)
L=$PWD/tests/hostsim/_build/libgrlbwt_sim_asan.so
P=29730
run() {   # world case VAR=value...
  W=$1; CASE=$2; shift 2
  P=$((P + 1))
  LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0 timeout 900 env GRLBWT_QUIET_ENV=1 "$@" python -m torch.distributed.run --nnodes=1 --nproc-per-node $W \
      --master-addr 127.0.0.1 --master-port $P tests/dist_worker.py $L gloo $CASE "$OUT" > "$OUT/run.log" 2>&1
  echo "world=$W $CASE $* rc=$? asan=$(grep -c 'ERROR: AddressSanitizer' "$OUT/run.log")"
}
S="GRLBWT_DIST_SHARDED_DICT_MIN=1 GRLBWT_DIST_SHARDED_DICT_MIN_SYMS=0"
run 4 reads GRLBWT_DIST_SHARDED_DICT_MIN_SYMS=0
run 4 tokens GRLBWT_DIST_SHARDED_DICT_MIN_SYMS=0
run 2 reads $S
run 3 uniform $S GRLBWT_DIST_REC_ROUND_TRIP=1
run 3 longruns $S
run 4 samechar GRLBWT_DIST_SHARDED_DICT_MIN_SYMS=0
run 3 tokens GRLBWT_DIST_GATHERED_DICT=1
run 3 tiny $S
run 2 repetitive $S GRLBWT_SEG_CAP=1
run 3 reads $S GRLBWT_RUN_KEYS_MIN=6
run 3 reads GRLBWT_DIST_REPLICATED_PREBWT=1 GRLBWT_DIST_GATHERED_DICT=1
# (positions as (owner, offset): every rank's part padded by 2^31 positions, the global numbering passes 2^32)
run 4 tokens $S GRLBWT_TEST_DICT_PART_PAD=2147483648 GRLBWT_RUN_KEYS_MIN=1073741824
# (no failure-injection case here: with libasan preloaded into python the first C++ throw of the library trips ASAN's own
# __cxa_throw interceptor check -- the failure paths are covered by tests/test_dist_gloo.py without the sanitizer)
