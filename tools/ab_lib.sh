#!/bin/bash
# same-box A/B between the product library and other builds of it:  tools/ab_lib.sh <lib.so>...   (AB_REPS, AB_FLAGS as tools/r5_ab_all.sh)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
run() {
  for rep in $(seq 1 ${AB_REPS:-2}); do
    env $1 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline ${AB_FLAGS:---no-train --no-extras} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d.get('other_workloads',{})
g=lambda k: o.get(k,{}).get('ms_per_step',0)
s='cfg=[$1] rep=$rep cm=%.3f c3=%.3f c4_8192=%.3f c4_1024=%.3f c5sh=%.3f c5sep=%.3f unwh=%.3f qdiag=%.3f' % (d['ms_per_step'], g('c3'), g('c4_shard_8192'), g('c4_shard_1024'), g('c5_shared'), g('c5_separate'), g('cm_unwhitened'), g('cm_q_diag'))
if 'train_step' in d: s+=' train=%.3f' % d['train_step']['ms_per_step']
if 'gpr_cholesky' in d:
    G=d['gpr_cholesky']; s+=' gpr=%.2f trail=%.4f' % (G.get('ms_total',0), G.get('trailing_update_roofline',{}).get('frac',0))
print(s)" || echo "cfg=[$1] rep=$rep FAILED"
  done
}
run ""
for lib in "$@"; do run "GPK_LIBRARY=$root/$lib"; done
