for sb in 4096 6144 8192 10240 12288 14336 20480 24576; do
  echo "SB_BINS=$sb"; CONSENRICH_AMD_SB_BINS=$sb CFGS=-1,-1,-1 python3 scripts/tune.py
done
