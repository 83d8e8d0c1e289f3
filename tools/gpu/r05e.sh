# round 5, GPU call E: headline bench with the row-wise batched re-pack in the step + steady-state kernel statistics (exclusive)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_e; mkdir -p $OUT
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-stock-baseline > $OUT/bench.log 2> $OUT/bench.err; echo "bench exit $?"
tail -1 $OUT/bench.log > $OUT/bench.json
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
r=d["roofline"]
print("value", d["value"], "ms", d["ms_per_step"], "loss", d["loss"], r["kernel"], r["achieved"], r["frac"], "excl", r.get("exclusive",{}).get("frac"), "alt", d.get("alt_math",{}).get("value"))
for o in d.get("other_configs", []): print("other", o["value"], o["ms_per_step"])
PY
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-stock-baseline --no-profile --no-alt-math --no-other-configs"
UNIPOSE_SYNC_WGRAD=1 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_sync -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/$OUT/prof_sync.log 2>&1
UNIPOSE_SYNC_WGRAD=1 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_736x -o bench -- python $GRAFT_REPO_ROOT/bench.py --size 736 --batch 16 --math bf16s $ARGS > $GRAFT_REPO_ROOT/$OUT/prof_736x.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $(find $OUT/prof_sync -name "*.db" | head -1) --steady 2 > $OUT/kernel_stats_exclusive_steady.txt 2>&1
python tools/rocprof_summary.py $(find $OUT/prof_736x -name "*.db" | head -1) --steady 2 > $OUT/kernel_stats_736_bf16s_exclusive_steady.txt 2>&1
find $OUT -name "*.db" -delete
head -45 $OUT/kernel_stats_exclusive_steady.txt | cut -c1-150
grep -i "pack\|# " $OUT/kernel_stats_736_bf16s_exclusive_steady.txt | cut -c1-150
