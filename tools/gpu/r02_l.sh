# rotating wave priority among co-resident workgroups (even progress instead of staggered finishing?)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_l; mkdir -p $OUT
timeout 300 python tools/gpu/tune_ab.py --rounds 3 --steps 5 base prio_rot=1 prio_rot=2 prio_rot=3 > $OUT/ab.log 2>&1; tail -1 $OUT/ab.log
timeout 200 python tools/gpu/tune_ab.py --rounds 2 --steps 8 --eval-only base prio_rot=1 prio_rot=2 prio_rot=3 > $OUT/ab_eval.log 2>&1; tail -1 $OUT/ab_eval.log
UNIPOSE_SYNC_WGRAD=1 timeout 300 python tools/gpu/tune_ab.py --rounds 2 --steps 5 base prio_rot=1 prio_rot=2 > $OUT/ab_sync.log 2>&1; tail -1 $OUT/ab_sync.log
