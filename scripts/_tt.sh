for d in 1 8 32 128; do
  export CONSENRICH_AMD_TREND_TOL_DIV=$d
  echo "DIV=$d"
  CONSENRICH_AMD_WARMSTART=0 python3 scripts/_wsd.py 2>&1 | tail -1
  CFGS='-1,-1,-1;96,96,80;112,112,96' python3 scripts/tune.py | cut -c1-130
done
