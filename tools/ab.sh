#!/bin/bash
# A/B runs of bench.py under environment toggles, all on the same box:  tools/ab.sh "VAR=1" "VAR2=3 VAR3=4" ...
for cfg in "" "$@"; do
  for rep in 1 2; do
    env $cfg python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('cfg=[$cfg] rep=$rep steps/s=%.1f ms=%.3f gemmTF=%.1f gpr_ms=%.2f' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d.get('gpr_cholesky',{}).get('ms_total',0)))"
  done
done
