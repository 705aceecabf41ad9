#!/bin/bash
# Everything a round's profiles/<round>/final/ holds besides what tools/profile_round.sh takes (bench line, kernel traces, PMC passes,
# set-metric sweep): the K4 benches by bucket size and by column-gene length, other sizes, the 8-rank shard rehearsed on one GPU, the
# slice scaling, the real-collection shape, the N = 10,000 pipeline, what --gpus N costs, the multi-GPU bench rehearsals (ranks and
# --route process), the randomised stress and -- PART=b -- the big fill.  One GPU-box call per part (each < 20 min):
#   gpurun --timeout 1190 -- 'bash tools/final_records.sh r05 a'      then b, then c
#   python tools/collect_profiles.py r05_final r05/final              (in the container: profile_round's part)
#   cp gpurun_out/r05_final_x/* profiles/r05/final/
# (replaces tools/r04/{final2..6,extras_a,extras_b,last,last2,suite}.sh)
set -u
TAG=${1:-r05}; PART=${2:-a}
OUT=gpurun_out/${TAG}_final_x; mkdir -p $OUT
t() { timeout -k 10 "$@"; }
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
if [ "$PART" = a ]; then
  bash tools/profile_round.sh ${TAG}_final 2>&1 | tail -3
  python3 tools/summarize_profile.py k4span gpurun_out/${TAG}_final_trace_peq $OUT/bench_peq5000_k4_span.txt gpurun_out/${TAG}_final_bench_peq5000.json | tail -4
  for n in 2000 10000 20000; do t 300 python3 -u tools/quick_bench.py -n $n --steps 3 --check 3000 2>&1 | grep --line-buffered -E "step 2|oracle" >> $OUT/sizes.txt; done; cut -c1-200 $OUT/sizes.txt
  t 200 python3 -u tools/shard_balance.py 5000 8 peq balanced 2>&1 | grep --line-buffered -v amdgpu > $OUT/shard_rehearsal_8ranks.txt; tail -2 $OUT/shard_rehearsal_8ranks.txt
  t 200 python3 -u tools/slice_scaling.py 5000 2>&1 | grep --line-buffered -v amdgpu > $OUT/slice_scaling.txt; tail -2 $OUT/slice_scaling.txt
elif [ "$PART" = b ]; then
  t 300 python3 -u tools/bucket_size_bench.py --lens 100,207,420,800 2>&1 | grep --line-buffered -v amdgpu > $OUT/bucket_size_bench.txt; tail -5 $OUT/bucket_size_bench.txt
  t 300 python3 -u tools/long_gene_bench.py --lens 2000,4000,4500,5000,9000,20000 --pairs 2048 --variants 0,32,48,-1 --check 4 2>&1 | grep --line-buffered -v amdgpu > $OUT/long_gene_bench.txt; tail -7 $OUT/long_gene_bench.txt
  t 400 python3 -u tools/real_shape.py -n 5000 --also 1000,2000 --out $OUT/real_shape.json > $OUT/real_shape.txt 2>&1; tail -3 $OUT/real_shape.txt | cut -c1-400
  t 400 python3 -u tools/pipeline_time.py 10000 peq > $OUT/pipeline_10000.txt 2>&1; tail -3 $OUT/pipeline_10000.txt | cut -c1-300
  t 300 python3 -u tools/launch_cost.py -n 5000 --ranks 1,2,4 --out $OUT/launch_cost.txt 2>&1 | tail -3 | cut -c1-200
  # bench.py by itself with N > 1: two ranks sharing this GPU over the gloo rehearsal transport, and one process driving two contexts
  PC_BENCH_BACKEND=gloo t 300 python3 bench.py --gpus 2 --genomes 2000 --cpu-seconds 0 > $OUT/bench_rehearsal_2ranks_spawned.json 2> $OUT/bench_rehearsal_2ranks_spawned.err; cut -c1-300 $OUT/bench_rehearsal_2ranks_spawned.json
  PC_BENCH_DEVICE_IDS=0,0 t 300 python3 bench.py --gpus 2 --route process --genomes 2000 --cpu-seconds 0 > $OUT/bench_rehearsal_2devices_one_process.json 2> $OUT/bench_rehearsal_2devices_one_process.err; cut -c1-300 $OUT/bench_rehearsal_2devices_one_process.json
  t 500 python3 -u tools/big_fill.py 2>&1 | grep --line-buffered -v amdgpu | tee $OUT/big_fill.txt | cut -c1-400
else
  t 1000 python3 -u -m pytest tests -q -m gpu --durations=12 > $OUT/gpu_suite.txt 2>&1; echo "pytest rc $?"; tail -16 $OUT/gpu_suite.txt
  t 700 python3 -u tools/stress_random.py 20261006 2000 > $OUT/stress.txt 2>&1; tail -3 $OUT/stress.txt
fi
find $OUT -maxdepth 1 -type f | sort
