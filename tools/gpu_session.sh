#!/bin/bash
# Developer helper (GPU box): run a list of steps, each with its own time limit and log under gpurun_out/<tag>/; a step that
# is KILLED (timeout) or dies by a signal ends the session (no further GPU step after a kill); an ordinary non-zero exit
# (failed assertions) is recorded and the session goes on.
# usage: tools/gpu_session.sh <tag> "<limit_s> <name> <command...>" ...
tag=$1; shift
o=gpurun_out/$tag; mkdir -p $o
export TMPDIR=/tmp
for spec in "$@"; do
  limit=${spec%% *}; rest=${spec#* }; name=${rest%% *}; cmd=${rest#* }
  echo "== $name (limit ${limit}s): $cmd" | tee -a $o/session.log
  t0=$(date +%s)
  timeout -k 10 $limit bash -c "$cmd" > $o/$name.log 2>&1
  rc=$?
  echo "   rc=$rc  $(( $(date +%s) - t0 ))s" | tee -a $o/session.log
  tail -n 6 $o/$name.log | cut -c1-300
  if [ $rc -ge 124 ]; then echo "step $name was killed or died by a signal (rc=$rc): stopping the session" | tee -a $o/session.log; exit $rc; fi
done
exit 0
