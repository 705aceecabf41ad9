#!/bin/bash
# r04 (second session): the variant chooser's constants once more, now that the cell is 6 % cheaper (relative cost of the step prologue up)
set -u
mkdir -p gpurun_out; OUT=gpurun_out/r04_choose_sweep2.txt; : > $OUT
export TMPDIR=/tmp
run() { echo "$1 $(env $2 python3 tools/quick_bench.py -n 3000 --steps 3 2>&1 | grep 'step 2' | sed 's/.*align \([0-9.]*\).*GCUPS \([0-9.]*\)/align \1 GCUPS \2/')" | tee -a $OUT; }
run "default" PC_DUMMY=1
for c0 in 0.3 0.8 1.5; do for c1 in 0.3 0.535 0.8; do run "c0=$c0 c1=$c1" "PC_CHOOSE_C0=$c0 PC_CHOOSE_C1=$c1"; done; done
for cell in 0.88 0.91 0.97 1.0; do run "cell=$cell" "PC_CHOOSE_CELL=$cell"; done
run "default" PC_DUMMY=1
