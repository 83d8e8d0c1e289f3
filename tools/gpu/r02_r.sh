cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_r; mkdir -p $OUT
VENDOR_BUDGET_S=240 timeout 420 python tools/gpu/vendor_shapes.py > $OUT/vendor_shapes.txt 2>$OUT/vendor_shapes.err; echo "exit $?"
cat $OUT/vendor_shapes.txt; tail -3 $OUT/vendor_shapes.err
