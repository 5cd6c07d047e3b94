#!/bin/bash
# GPU box: S3 score per-call time against the touch distance of the loader (EPG_S3_AHEAD, 0 = no touches)
for d in 0 2 4 6 8 12 16 24; do
  echo -n "EPG_S3_AHEAD=$d  "; EPG_S3_AHEAD=$d python tools/s3_score_probe.py --dbg 0
done
