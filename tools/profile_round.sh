#!/bin/bash
# All of a round's measurement records in one GPU-box call (rocprofv3 needs the program itself after "--").
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r02_c'
# then, back in the container:  python tools/collect_profiles.py r02_c
# Writes under gpurun_out/<tag>_*; nothing here is timed against anything else on the box.
set -u
TAG=${1:-r02}
OUT=gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
QUICK="--cpu-seconds 0 --verify-pairs 0"
echo "== bench (default)"; python3 bench.py > $OUT/${TAG}_bench_peq5000.json 2> $OUT/${TAG}_bench_peq5000.err || echo "bench failed"
echo "== kernel trace, peq N=5000"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_peq -o t -- python3 bench.py --steps 3 --warmup 1 $QUICK > $OUT/${TAG}_trace_peq.log 2>&1 || echo "trace failed"
echo "== PMC FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 0 $QUICK > $OUT/${TAG}_pmc_fetch.log 2>&1 || echo "fetch failed"
echo "== PMC WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o w -- python3 bench.py --steps 1 --warmup 0 $QUICK > $OUT/${TAG}_pmc_write.log 2>&1 || echo "write failed"
echo "== PMC VALU"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_valu -o v -- python3 bench.py --steps 1 --warmup 0 $QUICK > $OUT/${TAG}_pmc_valu.log 2>&1 || echo "valu failed"
echo "== set metrics: sweep (checked), then traced"
python3 tools/set_metric_bench.py --sizes 2000,5000,20000 --out $OUT/${TAG}_set_metric_sweep.json > $OUT/${TAG}_set_metric_sweep.log 2>&1 || echo "sweep failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_set -o s -- python3 tools/set_metric_bench.py --sizes 2000,20000 --steps 5 --check 0 > $OUT/${TAG}_trace_set.log 2>&1 || echo "set trace failed"
echo "== set metrics: PMC at N=20000"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_set_fetch -o f -- python3 tools/set_metric_bench.py --sizes 20000 --steps 1 --check 0 > $OUT/${TAG}_pmc_set_fetch.log 2>&1 || echo "set fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_set_write -o w -- python3 tools/set_metric_bench.py --sizes 20000 --steps 1 --check 0 > $OUT/${TAG}_pmc_set_write.log 2>&1 || echo "set write failed"
find $OUT -name "${TAG}_*" -maxdepth 1 | sort
du -sh $OUT/${TAG}_* | tail -20
