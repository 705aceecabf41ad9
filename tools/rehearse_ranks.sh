#!/bin/bash
# bench.py under the driver's launcher with RANKS ranks sharing this box's one GPU (gloo transport: the gather / reduce is
# staged through host memory), in both split modes.  A correctness + bookkeeping record of the N > 1 path, not a timing.
#   gpurun -- 'bash tools/rehearse_ranks.sh r03_k 4 2000'
TAG=${1:-x}; RANKS=${2:-4}; N=${3:-2000}
for mode in pairs alignments; do
  PC_BENCH_BACKEND=gloo PHAMCLUST_DIST_MODE=$mode python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $RANKS --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
    bench.py --gpus $RANKS --genomes $N --steps 2 --warmup 1 --verify-pairs 5000 2> gpurun_out/${TAG}_rehearsal_${mode}.err | grep '^{' > gpurun_out/${TAG}_bench_rehearsal_${RANKS}ranks_${mode}.json
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/${TAG}_bench_rehearsal_${RANKS}ranks_${mode}.json'))
print('$mode', d['n_gpus'], d['verified'], d['stage_ms'], d['shards'], d['config']['dist_mode'])"
done
