#!/bin/bash
# r04, second session: the whole GPU suite on the final tree, the bench line (quotes profiles/traffic.json), long genes at 9,000 / 20,000
set -u
OUT=gpurun_out/r04_final_y; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
timeout -k 10 1000 python3 -u -m pytest tests -q -m gpu --durations=8 > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc $rc" | tee $OUT/summary.txt; tail -14 $OUT/pytest.log | tee -a $OUT/summary.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python3 bench.py > $OUT/bench_peq5000.json 2> $OUT/bench_peq5000.err; cut -c1-400 $OUT/bench_peq5000.json
timeout -k 10 250 python3 -u tools/long_gene_bench.py --lens 9000,20000 --pairs 2048 --variants 0 --check 3 2>&1 | grep --line-buffered -v amdgpu | tee $OUT/long_gene_bench_9000_20000.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
