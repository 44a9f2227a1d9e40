#!/bin/bash
# Same-box A/B of one environment switch on the bench step: alternates VAR=A / VAR=B, REPS times each, one line per run
# (median / min / max ms of three timed windows).  usage: VAR=CONSENRICH_AMD_TAIL_EARLY A=1 B=0 [REPS=4] [ARGS="--steps 30"] bash scripts/ab_env.sh
REPS=${REPS:-4}
ARGS=${ARGS:-"--steps 30"}
for i in $(seq $REPS); do
  for val in "$A" "$B"; do
    env $VAR=$val python3 bench.py $ARGS --no-cpu-baseline --no-extras 2>/dev/null | VAL="$val" VARN="$VAR" python3 -c '
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("%s=%s  ms/step %.4f (min %.4f max %.4f)" % (os.environ["VARN"], os.environ["VAL"], d["ms_per_step"], d["ms_per_step_min"], d["ms_per_step_max"]))'
  done
done
