#!/bin/bash
# r04 final records beside tools/profile_round.sh: the tools' outputs under gpurun_out/r04_final_x/
set -u
OUT=gpurun_out/r04_final_x; mkdir -p $OUT
t() { timeout -k 10 "$@"; }
t 400 python3 -u tools/pipeline_time.py 10000 peq > $OUT/pipeline_10000.txt 2>&1; tail -3 $OUT/pipeline_10000.txt
bash tools/rehearse_ranks.sh r04_final_x/k 4 2000 > $OUT/rehearse.txt 2>&1; tail -4 $OUT/rehearse.txt | cut -c1-300
t 700 python3 -u tools/stress_random.py 20261006 2000 > $OUT/stress.txt 2>&1; tail -3 $OUT/stress.txt
# the bench line once more, now that profiles/traffic.json carries this device code's PMC record (roofline.traffic)
t 400 python3 bench.py > $OUT/bench_peq5000.json 2> $OUT/bench_peq5000.err; cut -c1-300 $OUT/bench_peq5000.json
