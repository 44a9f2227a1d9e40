#!/bin/bash
# sweep on one box: values VALS of the environment variable VAR, ROUNDS rounds; step time and state chain of the bench workload
VAR=${VAR:-CONSENRICH_AMD_SB_BINS}; VALS=${VALS:-"24576 20480"}; ROUNDS=${ROUNDS:-4}
for i in $(seq 1 $ROUNDS); do
  for x in $VALS; do
    echo -n "$VAR=$x  "
    env $VAR=$x CFGS=-1,-1,-1 timeout -k 10 200 python3 scripts/tune.py 2>&1 | tail -1 | grep -o "ms/step [0-9.]* (profiled [0-9.]*)\|'fwd_state_chain': [0-9.]*" | paste -sd' '
  done
done
