#!/bin/bash
# kernel time of the bench frame (config 3) over the lane-scheduling thresholds; one line per setting
for y in ${YIELDS:-36 40 44 48 52}; do for r in ${REFILLS:-8 12 16 20}; do
  echo -n "yield $y refill $r: "; python scripts/one_frame.py --launches 2 --yield-lanes $y --refill-lanes $r 2>/dev/null | tail -1
done; done
