#!/bin/bash
# is the per-process slow mode a stream -> pipe placement accident?  N fresh processes: hand-off latencies + step time
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
X="GPK_LIBRARY=$root/gpflow_amd/libgpk_exp.so"
for i in $(seq 1 12); do
  env $X GPK_STREAM_SELFTEST=1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-gpr --no-train --no-extras --no-other 2> /tmp/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('run $i ms_per_step=%.4f' % d['ms_per_step'], end='  ')"
  grep "hand-off" /tmp/err.txt
done
