#!/bin/bash
# final evidence of round 3 on the final product code: full GPU suite, default bench line, rocprofv3 stats + PMC passes
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r03_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_pytest.log )
timeout 600 python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
bash tools/profile_round.sh r03 > gpurun_out/r03_profile_round.log 2>&1
