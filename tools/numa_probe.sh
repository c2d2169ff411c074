#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
lscpu | grep -E "NUMA|Model name|^CPU\(s\)|Thread|Socket" 
for d in /sys/class/drm/card*/device; do echo "$d numa_node=$(cat $d/numa_node 2>/dev/null) local_cpulist=$(cat $d/local_cpulist 2>/dev/null)"; done
nproc; cat /proc/self/status | grep Cpus_allowed_list
run() { taskset -c $1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-gpr --no-train --no-extras --no-other 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('cpus $1 ms_per_step=%.4f' % d['ms_per_step'])"; }
n=$(nproc)
run 0-$((n-1))
run 0
run $((n/4))
run $((n/2))
run $((3*n/4))
run $((n-1))
run 0-$((n/2-1))
run $((n/2))-$((n-1))
