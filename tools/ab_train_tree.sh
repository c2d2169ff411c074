#!/bin/bash
# same-box A/B of the training step between this tree and a second tree (old_tree/: `git archive <rev>` + the built library):
#   tools/ab_train_tree.sh [reps]
root=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in $(seq 1 ${1:-2}); do
  for t in old_tree .; do
    (cd $root/$t && timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gpr --no-extras --no-other 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('tree=$t rep=$rep cm=%.3f train=%.3f elbo=%.9e train_elbo=%.9e' % (d['ms_per_step'], d['train_step']['ms_per_step'], d['last_elbo'], d['train_step']['last_elbo']))") || echo "tree=$t FAILED"
  done
done
