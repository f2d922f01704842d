#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4soak; mkdir -p $O; cd $R
timeout 1500 python tools/soak.py --epochs 30 > $O/soak.txt 2>&1; echo "rc $?" >> $O/soak.txt
