#!/bin/bash
# r04, second session: the round's records re-taken on the device code with the one-instruction re-tags
# (tools/profile_round.sh r04_final, then the tools beside it -> gpurun_out/r04_final_x/)
set -u
rm -rf gpurun_out/r04_final_* 2>/dev/null
bash tools/profile_round.sh r04_final 2>&1 | tail -3
bash tools/r04/extras_a.sh
