#!/bin/bash
# lease script: which combination of transport / pool makes the world-1 collection-level build fail
ulimit -c 0
cd "$(dirname "$0")/.."
LIB=$(python -c "import __graft_entry__ as g; print(g.build_hip())")
mkdir -p gpurun_out/dd
run() { # name backend env...
  name=$1; backend=$2; shift 2
  env "$@" python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29701 tests/dist_worker.py $LIB $backend illumina_dev:${READS:-6622517}:${GENOME:-33000000} gpurun_out/dd > gpurun_out/dd/$name.log 2>&1
  echo "$name rc=$? $(cat gpurun_out/dd/illumina_dev.rank0.md5 2>/dev/null) $(grep -o 'grlbwt error.*' gpurun_out/dd/$name.log | head -1)"
  rm -f gpurun_out/dd/illumina_dev.rank0.md5
}
GRLBWT_POOL_CLASSIC=1 python bench.py --workload illumina --reads ${READS:-6622517} --read-len 150 --genome ${GENOME:-33000000} --steps 2 --no-extra --no-cpu-baseline > gpurun_out/dd/single_classic.json 2> gpurun_out/dd/single_classic.err; echo "single classic rc=$? $(cut -c1-160 gpurun_out/dd/single_classic.json)"
run nccl nccl A=1
run gloo_arena gloo-cuda A=1
run gloo_classic gloo-cuda GRLBWT_POOL_CLASSIC=1
run nccl_sync nccl GRLBWT_DIST_SYNC_COLLECTIVES=1
