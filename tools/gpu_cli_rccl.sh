#!/bin/bash
# lease script: the CLI's collection-level mode (in-library RCCL transport): one rank, two ranks on one device (if RCCL allows), 1 GB
ulimit -c 0
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build_hip(); g.build_cli()"
CLI=grlbwt_amd/bin/grlbwt
mkdir -p gpurun_out/cli
python -m pytest tests/test_cli.py -x -q -m gpu 2>&1 | tail -5
echo "== two ranks on one device"
GRLBWT_CLI_SAME_DEVICE=1 timeout 300 $CLI tests/golden/test_byte_alphabet.txt --gpus 2 -o gpurun_out/cli/two > gpurun_out/cli/two.log 2>&1; echo "rc=$? $(md5sum gpurun_out/cli/two.rl_bwt 2>/dev/null)"; tail -5 gpurun_out/cli/two.log
echo "== 1 GB, one rank over RCCL vs plain"
python - <<'PY'
import torch, sys
sys.path.insert(0, ".")
from grlbwt_amd import workloads
t = workloads.sampled_reads_torch(6622517, 150, 33000000, seed=20260003, device="cuda:0")
open("/tmp/ill1g.txt", "wb").write(t.cpu().numpy().tobytes())
PY
$CLI /tmp/ill1g.txt -o /tmp/a > gpurun_out/cli/plain.log 2>&1; echo "plain rc=$? $(grep grlbwt-timing gpurun_out/cli/plain.log)"
GRLBWT_CLI_FORCE_RCCL=1 $CLI /tmp/ill1g.txt -o /tmp/b > gpurun_out/cli/rccl1.log 2>&1; echo "rccl rc=$? $(grep grlbwt-timing gpurun_out/cli/rccl1.log)"
md5sum /tmp/a.rl_bwt /tmp/b.rl_bwt
