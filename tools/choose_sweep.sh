#!/bin/bash
# Sweep of the variant chooser's cost-model constants (pc_nw_choose_variant) on synth(3000,5000) -m peq:
#   gpurun -- 'bash tools/choose_sweep.sh TAG'  ->  gpurun_out/TAG_choose_sweep.txt
TAG=${1:-x}
export TMPDIR=/tmp
: > gpurun_out/${TAG}_choose_sweep.txt
for c0 in 0.0 0.3 0.6 1.0; do for c1 in 0.3 0.535 0.8; do
  echo "c0=$c0 c1=$c1 $(PC_CHOOSE_C0=$c0 PC_CHOOSE_C1=$c1 python tools/quick_bench.py -n 3000 --steps 3 2>&1 | grep 'step 2' | sed 's/.*align \([0-9.]*\).*GCUPS \([0-9.]*\)/align \1 GCUPS \2/')" >> gpurun_out/${TAG}_choose_sweep.txt
done; done
cat gpurun_out/${TAG}_choose_sweep.txt
