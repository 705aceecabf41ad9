#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
timeout -k 10 900 python3 -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "strip_mined or launch_policy" 2>&1 | tail -5 &&
for n in 1000 2000 5000; do
  echo "== synth_real($n) default" && timeout -k 10 300 python3 -u tools/real_trace.py -n $n 2>&1 | tail -1 &&
  echo "== synth_real($n) PC_LONG_PRIORITY=0" && PC_LONG_PRIORITY=0 timeout -k 10 300 python3 -u tools/real_trace.py -n $n 2>&1 | tail -1 || exit 1
done
timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --verify-pairs 0 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.readlines()[-1]); print('bench', r['ms_per_step'], r['stage_ms'])" &&
bash tools/r04/tl_real1000.sh 1000
