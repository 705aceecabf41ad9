#!/bin/bash
# Host-side launch order / stream count on the real-collection-shaped fills and the benchmark's (device code untouched)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
run() { # label, env...
  label=$1; shift
  for n in 1000 2000 5000; do
    echo -n "$label synth_real($n): "; env "$@" timeout -k 10 300 python3 -u tools/real_trace.py -n $n 2>&1 | tail -1 | python3 -c "import sys,ast; r=ast.literal_eval(sys.stdin.read()); print(round(r['ms_align'],2))" || exit 1
  done
  for n in 2000 5000; do
    echo -n "$label synth($n,5000): "; env "$@" timeout -k 10 300 python3 -u tools/real_trace.py -n $n --synth 5000 2>&1 | tail -1 | python3 -c "import sys,ast; r=ast.literal_eval(sys.stdin.read()); print(round(r['ms_align'],2))" || exit 1
  done
}
run default X=1
run small_first_2048 PC_ALIGN_SMALL_FIRST=2048
run small_first_8192 PC_ALIGN_SMALL_FIRST=8192
run streams4 PC_ALIGN_STREAMS=4
run streams4_small_first PC_ALIGN_STREAMS=4 PC_ALIGN_SMALL_FIRST=2048
run streams12 PC_ALIGN_STREAMS=12
