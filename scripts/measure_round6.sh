#!/bin/bash
# The measurements behind the round-6 entries of DESIGN.md / profiles/ (GPU box):
#   gpurun --timeout 2400 -- 'bash scripts/measure_round6.sh r06'
# writes gpurun_out/<tag>/...; scripts/collect_profiles.py <tag> copies the summaries into profiles/ and updates
# profiles/pmc_traffic.json (the entries of kernels this script does not profile are kept).
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
prof() { local name=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$name" -o "$name" -- "$@" > "$O/$name.log" 2>&1; }
pmc() { local name=$1 ctr=$2; shift 2; rocprofv3 --pmc $ctr --output-format csv -d "$O/$name" -o "$name" -- "$@" > "$O/$name.log" 2>&1; }
prof bench_n1 python3 $R/bench.py
prof bench_n1_streaming python3 $R/bench.py --variant 4 --no-cpu-baseline
prof bench_cavity_q4 python3 $R/bench.py --config cavity
prof ops python3 $R/scripts/bench_ops.py
# HBM traffic of the Q2/Q1 kernel in both modes: separate counter passes (MI355X_MICROARCH.md)
pmc pmc_q2_fetch FETCH_SIZE python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline
pmc pmc_q2_write WRITE_SIZE python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline
pmc pmc_q2s_fetch FETCH_SIZE python3 $R/bench.py --variant 4 --steps 5 --warmup 2 --no-cpu-baseline
pmc pmc_q2s_write WRITE_SIZE python3 $R/bench.py --variant 4 --steps 5 --warmup 2 --no-cpu-baseline
pmc pmc_q2_sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline
pmc pmc_q2_sq2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline
pmc pmc_q4_fetch FETCH_SIZE python3 $R/bench.py --config cavity --steps 5 --warmup 2 --no-cpu-baseline
pmc pmc_q4_write WRITE_SIZE python3 $R/bench.py --config cavity --steps 5 --warmup 2 --no-cpu-baseline
cd $R
python3 bench.py > "$O/bench_n1_plain.log" 2>&1
python3 bench.py --cells 160 --no-cpu-baseline > "$O/bench_160cubed.log" 2>&1
python3 bench.py --cells 256 --steps 20 --warmup 3 --no-cpu-baseline > "$O/bench_256cubed_one_gpu.log" 2>&1
python3 bench.py --config cavity > "$O/bench_cavity_q4_plain.log" 2>&1
python3 bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline > "$O/bench_2ranks_one_gpu.log" 2>&1
python3 bench.py --gpus 2 --config cavity --steps 10 --warmup 2 --no-cpu-baseline > "$O/bench_cavity_2ranks_one_gpu.log" 2>&1
python3 bench.py --through-comm --steps 30 --warmup 5 --no-cpu-baseline > "$O/bench_through_comm_q2_128.log" 2>&1
python3 bench.py --config cavity --cells 32 --steps 50 --warmup 5 --no-cpu-baseline > "$O/bench_cavity_q4_32.log" 2>&1
python3 bench.py --config cavity --cells 32 --through-comm --steps 50 --warmup 5 --no-cpu-baseline > "$O/bench_through_comm_q4_32.log" 2>&1
python3 scripts/bench_ho.py > "$O/bench_ho.log" 2>&1
echo done > "$O/done"
