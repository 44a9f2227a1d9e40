#!/bin/bash
# compare kernel variants selected by environment switches: one line per variant (step / kernel ms from bench.py)
set -e
for v in ${VARIANTS:-"BASE=1" "BASE=2"}; do
  echo "== $v"
  env $v python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels_rank0']
print('ms/step %.4f ' % d['ms_per_step'] + ' '.join('%s %.3f' % (n, v['avg_ms']) for n, v in sorted(k.items())))"
done
