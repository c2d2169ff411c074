#!/bin/bash
# Same-box A/B of the GPR leg (N = 16384) under the tunables of the experimental library:  tools/ab_gpr.sh "GPK_NBO=1024" ...
root=${GRAFT_REPO_ROOT:-$(pwd)}
run() {
  env $1 timeout 120 python $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-extras --no-other 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d['gpr_cholesky']; t=g['trailing_update_roofline']
print('cfg=[$1] gpr_ms=%.2f chol_TF=%.1f trailing_TF=%.2f (%.3f) chipwide=%.1f predict_ms=%.1f cached_ms=%.1f kb_ms=%.3f' % (g['ms_total'], g['cholesky_gflops_incl_build_and_tail']/1e3, t['achieved'], t['frac'], t['phase_chipwide']['achieved'], g['predict']['ms_total'], g['predict']['cached_posterior_ms'], g['kernel_build_full_ms']))" || echo "cfg=[$1] FAILED"
}
run ""
for cfg in "$@"; do run "GPK_LIBRARY=$root/gpflow_amd/libgpk_exp.so $cfg"; done
