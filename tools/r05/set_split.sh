#!/bin/bash
# Where a 64 x 64 sparse tile's time goes (r05): release kernel against builds with the epilogue stubbed (epi1: plain store of the
# integer pair, epi2: no stores), the probes skipped, or both.  Variants: tools/build_variant.py x_<name> --only pc_pairs.hip -DS6X_...
set -u
OUT=gpurun_out/r05_set_split
mkdir -p $OUT
python3 tools/set_time.py --sizes 2000,20000 --metrics jc,pocp,af --check 0 --variants=base,x_epi1,x_epi2,x_noprobe,x_noprobe_epi2 --env PC_SET_KERNEL=sparse64 > $OUT/split.jsonl 2> $OUT/split.err
cat $OUT/split.jsonl
