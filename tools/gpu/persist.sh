#!/bin/bash
# persistent stream-K A/B (see tools/gpu/persist_ab.py); results under gpurun_out/persist/
mkdir -p gpurun_out/persist
timeout ${1:-170} python tools/gpu/persist_ab.py --rounds ${2:-2} --steps ${3:-4} --grids ${4:-0} > gpurun_out/persist/ab.log 2>&1
tail -5 gpurun_out/persist/ab.log
