#!/bin/bash
# the P3 tower's weight gradient (2 x 128^2, 256 -> 256) on the own kernel's tile shapes against the library's (us per call):
# 0 = 128x128 (the rule), 524288 = 128x64, 1048576 = 64x128, 131072 = 64x64; JDET_CONV_WGRAD_DEEP_MID=2: two steps ahead on the mid tiles
cd ${GRAFT_REPO_ROOT:-/root/repo}
for e in "JDET_CONV_WGRAD_DEEP_MID=0" "JDET_CONV_WGRAD_DEEP_MID=2"; do
  echo "== $e"; env $e python scripts/conv_wgrad_timing.py 0 524288 1048576 131072 2>&1 | grep -v Warn | head -2
done
