#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06
mkdir -p $O
for i in 1 2; do
python bench.py --config cavity --cells 32 --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain   ', d['ms_per_step'], d['ms_per_step_min'], d['roofline']['kernel_ms'])"
python bench.py --config cavity --cells 32 --no-cpu-baseline --steps 200 --warmup 20 --through-comm 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('through ', d['ms_per_step'], d['ms_per_step_min'], d['roofline']['kernel_ms'], d.get('phase_ms_rank0'))"
done 2>&1 | tee $O/through_comm_q4_32.log
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/tc32 -o tc32 -- python3 $GRAFT_REPO_ROOT/bench.py --config cavity --cells 32 --no-cpu-baseline --steps 200 --warmup 20 --through-comm > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python3 scripts/kstats.py $O/tc32 10
