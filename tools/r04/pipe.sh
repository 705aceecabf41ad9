#!/bin/bash
# The pipelined form of the strip-mined kernel: parity (strip test under every PC_PIPE setting), then the real-collection-shaped fill
# at three sizes with and without it
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
timeout -k 10 900 python3 -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "strip_mined" 2>&1 | tail -5 &&
for n in 1000 2000 5000; do
  echo "== synth_real($n) default" && timeout -k 10 300 python3 -u tools/real_trace.py -n $n 2>&1 | tail -1 &&
  echo "== synth_real($n) PC_PIPE=0" && PC_PIPE=0 timeout -k 10 300 python3 -u tools/real_trace.py -n $n 2>&1 | tail -1 || exit 1
done
