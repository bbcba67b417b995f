#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
R=$PWD
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_dev/res -o res -- python3 $R/scripts/dev/res_lazy_bench.py > /dev/null 2>&1
cd $R && python3 scripts/kstats.py gpurun_out/r06_dev/res 14
