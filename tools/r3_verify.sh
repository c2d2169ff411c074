#!/bin/bash
# round 3: full GPU suite, the default bench line, the world-size-1 RCCL self-test (A/B against the same run without it)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q > $out/r3v_pytest.log 2>&1; echo "pytest rc=$?" >> $out/r3v_pytest.log )
( time timeout 600 python bench.py > $out/r3v_bench.json 2> $out/r3v_bench.err ) 2> $out/r3v_bench.time
for rep in 1 2; do
  timeout 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-gpr --no-train --no-extras --no-other 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('plain          ms_per_step=%.4f' % d['ms_per_step'])"
  timeout 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-gpr --no-train --no-extras --no-other --rccl-selftest 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('rccl-selftest  ms_per_step=%.4f' % d['ms_per_step'])"
done > $out/r3v_rccl_selftest.log 2>&1
