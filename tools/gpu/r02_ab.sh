cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_ab; mkdir -p $OUT
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
grep -o "SQ_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|TCC_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*\|TA_[A-Z_0-9]*" $OUT/avail.txt | sort -u | tr '\n' ' ' | head -c 6000
