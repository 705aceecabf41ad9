#!/bin/bash
# the set-metric half of tools/profile_round.sh, again (its first run stopped at a use-after-loan in tools/set_metric_bench.py that the r04 loan guard caught)
set -u
TAG=r04_final; OUT=gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
rm -rf $OUT/${TAG}_trace_set $OUT/${TAG}_pmc_set_fetch $OUT/${TAG}_pmc_set_write
python3 tools/set_metric_bench.py --sizes 2000,5000,20000 --out $OUT/${TAG}_set_metric_sweep.json > $OUT/${TAG}_set_metric_sweep.log 2>&1 || echo "sweep failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_set -o s -- python3 tools/set_metric_bench.py --sizes 2000,20000 --steps 5 --check 0 > $OUT/${TAG}_trace_set.log 2>&1 || echo "set trace failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_set_fetch -o f -- python3 tools/set_metric_bench.py --sizes 20000 --steps 1 --check 0 > $OUT/${TAG}_pmc_set_fetch.log 2>&1 || echo "set fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_set_write -o w -- python3 tools/set_metric_bench.py --sizes 20000 --steps 1 --check 0 > $OUT/${TAG}_pmc_set_write.log 2>&1 || echo "set write failed"
tail -4 $OUT/${TAG}_set_metric_sweep.log | cut -c1-250
du -sh $OUT/${TAG}_trace_set $OUT/${TAG}_pmc_set_fetch $OUT/${TAG}_pmc_set_write
