#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_tp
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tp -o tp -- python3 $R/examples/rising_bubble_3d.py 64 4 4 > $O/tp.log 2>&1
cd $R && python3 scripts/kstats.py gpurun_out/r06_tp/tp 16; grep "s wall" $O/tp.log | tail -3
