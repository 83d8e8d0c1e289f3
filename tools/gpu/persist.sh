#!/bin/bash
# persistent stream-K A/B (see tools/gpu/persist_ab.py); results under gpurun_out/persist/
# usage: persist.sh <timeout s> <rounds> <steps> <grids> [variants tpw:xcd:grid,...] [csv]
mkdir -p gpurun_out/persist
CSV=""
if [ -n "$6" ]; then CSV="--csv gpurun_out/persist/csv"; export UNIPOSE_SYNC_WGRAD=1; fi
timeout ${1:-170} python tools/gpu/persist_ab.py --rounds ${2:-2} --steps ${3:-4} --grids "${4}" --variants "${5}" $CSV > gpurun_out/persist/ab.log 2>&1
tail -4 gpurun_out/persist/ab.log
