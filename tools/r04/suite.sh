#!/bin/bash
# The whole GPU suite, progress into gpurun_out/suite.log as it goes (a run piped into tail looks hung to the box's watchdog)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
timeout -k 10 1100 python3 -u -m pytest tests -x -v -m gpu > gpurun_out/suite.log 2>&1
rc=$?
tail -6 gpurun_out/suite.log
exit $rc
