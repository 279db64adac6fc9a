#!/bin/bash
# A/B of the park pool on the bench frame (config 3): kernel time and stage statistics per setting.
set -e
mkdir -p gpurun_out/ab
for park in -1 16 64 128; do
  echo "== park $park" | tee -a gpurun_out/ab/park.log
  python scripts/one_frame.py --park $park --launches 3 >> gpurun_out/ab/park.log 2>&1
  python scripts/one_frame.py --park $park --launches 1 --counters >> gpurun_out/ab/park.log 2>&1
done
cat gpurun_out/ab/park.log
