#!/bin/bash
# same-box A/B of the training step under environment settings of the host layer:  tools/ab_train_env.sh "ENV=.." "ENV=.." ...  (first: none)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
for rep in $(seq 1 ${AB_REPS:-2}); do
  for cfg in "" "$@"; do
    env $cfg timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gpr --no-extras --no-other 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('cfg=[$cfg] rep=$rep cm=%.3f train=%.3f train_elbo=%.9e' % (d['ms_per_step'], d['train_step']['ms_per_step'], d['train_step']['last_elbo']))" || echo "cfg=[$cfg] FAILED"
  done
done
