cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_q; mkdir -p $OUT
timeout 300 python tools/gpu/tune_ab.py --rounds 3 --steps 5 base wgrad_per_cu=1 wgrad_per_cu=3 > $OUT/ab.log 2>&1; tail -1 $OUT/ab.log
