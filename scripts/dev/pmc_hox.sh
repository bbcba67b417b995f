#!/bin/bash
# counters of the Q4/Q3 x-marching kernel (bench.py --config cavity): kernel trace + two SQ passes
# usage (GPU box): bash scripts/dev/pmc_hox.sh <tag>
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bench_cavity_q4" -o bench_cavity_q4 -- python3 $R/bench.py --config cavity --no-cpu-baseline > "$O/bench_cavity_q4.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d "$O/pmc_q4_sq1" -o pmc_q4_sq1 -- python3 $R/bench.py --config cavity --steps 5 --warmup 2 --no-cpu-baseline > "$O/pmc_q4_sq1.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$O/pmc_q4_sq2" -o pmc_q4_sq2 -- python3 $R/bench.py --config cavity --steps 5 --warmup 2 --no-cpu-baseline > "$O/pmc_q4_sq2.log" 2>&1
# the same kernel without a state stream (LIN_MODE 2: diagnostic run of the explicit scheme on the same box)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d "$O/pmc_q4ns_sq1" -o pmc_q4ns_sq1 -- python3 $R/bench.py --config cavity --steps 5 --warmup 2 --no-cpu-baseline --linearization "coupled velocity explicit" > "$O/pmc_q4ns_sq1.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$O/pmc_q4ns_sq2" -o pmc_q4ns_sq2 -- python3 $R/bench.py --config cavity --steps 5 --warmup 2 --no-cpu-baseline --linearization "coupled velocity explicit" > "$O/pmc_q4ns_sq2.log" 2>&1
cd $R
for n in pmc_q4_sq1 pmc_q4_sq2 pmc_q4ns_sq1 pmc_q4ns_sq2; do python3 scripts/pmc_summary.py $O/$n > $O/$n.txt 2>&1; done
python3 scripts/kstats.py $O/bench_cavity_q4 8
cat $O/pmc_q4_sq1.txt $O/pmc_q4_sq2.txt $O/pmc_q4ns_sq1.txt $O/pmc_q4ns_sq2.txt | grep -A9 "ns_hox_kernel" | head -80
