#!/bin/bash
# All measurements behind DESIGN.md / profiles/ of one round, on the GPU box:
#   gpurun --timeout 3000 -- 'bash scripts/measure_round.sh r04'
# writes gpurun_out/<tag>/<name>/*_kernel_stats.csv + <name>.log; scripts/collect_profiles.py copies the
# summaries into profiles/ and builds profiles/pmc_traffic.json.
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
prof() { # name, program ...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$name" -o "$name" -- "$@" > "$O/$name.log" 2>&1
}
pmc() { # name, counters, program ...
  local name=$1 ctr=$2; shift 2
  rocprofv3 --pmc $ctr --output-format csv -d "$O/$name" -o "$name" -- "$@" > "$O/$name.log" 2>&1
}
prof bench_n1 python3 $R/bench.py
prof bench_cavity_q4 python3 $R/bench.py --config cavity
prof bench_160cubed python3 $R/bench.py --cells 160 --no-cpu-baseline
prof ops python3 $R/scripts/bench_ops.py
prof two_phase_step python3 $R/scripts/time_two_phase.py 64 4
prof beltrami64_step python3 $R/scripts/time_beltrami_step.py 64 3
# HBM traffic of the two dominant kernels: separate counter passes (MI355X_MICROARCH.md)
pmc pmc_q2_fetch FETCH_SIZE python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline
pmc pmc_q2_write WRITE_SIZE python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline
pmc pmc_q4_fetch FETCH_SIZE python3 $R/bench.py --config cavity --steps 5 --warmup 2 --no-cpu-baseline
pmc pmc_q4_write WRITE_SIZE python3 $R/bench.py --config cavity --steps 5 --warmup 2 --no-cpu-baseline
pmc pmc_q4_sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" python3 $R/bench.py --config cavity --steps 5 --warmup 2 --no-cpu-baseline
pmc pmc_q4_sq2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" python3 $R/bench.py --config cavity --steps 5 --warmup 2 --no-cpu-baseline
# calibration of the FETCH_SIZE counter for 8-B-per-lane reads (the access width of ns_ho_kernel)
hipcc --offload-arch=gfx950 -O3 $R/scripts/dev/fetch_probe.hip -o /tmp/fetch_probe > "$O/fetch_probe_build.log" 2>&1
pmc pmc_fetch_probe FETCH_SIZE /tmp/fetch_probe
# the two stencil kernels alone (built here: hipcc ... scripts/dev/stencil_probe.hip / div_probe.hip -o scripts/dev/build/...)
for p in stencil_probe div_probe; do
  [ -x $R/scripts/dev/build/$p ] && $R/scripts/dev/build/$p > "$O/$p.log" 2>&1
done
[ -x $R/scripts/dev/build/stencil_probe ] && $R/scripts/dev/build/stencil_probe 161 321 >> "$O/stencil_probe.log" 2>&1
[ -x $R/scripts/dev/build/div_probe ] && $R/scripts/dev/build/div_probe 64 128 >> "$O/div_probe.log" 2>&1
# no rocprof: the plain bench lines and the large meshes
cd $R
python3 bench.py > "$O/bench_n1_plain.log" 2>&1
python3 bench.py --cells 256 --steps 20 --warmup 3 --no-cpu-baseline > "$O/bench_256cubed_one_gpu.log" 2>&1
python3 bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline > "$O/bench_2ranks_one_gpu.log" 2>&1
python3 bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline --src-consistent > "$O/bench_2ranks_one_gpu_src_consistent.log" 2>&1
python3 bench.py --gpus 2 --config cavity --steps 10 --warmup 2 --no-cpu-baseline > "$O/bench_cavity_2ranks_one_gpu.log" 2>&1
# fixed overhead of the distributed path on one rank (world = 1 through adaflo_ns_vmult_distributed, phased schedule forced)
python3 bench.py --through-comm --steps 30 --warmup 5 --no-cpu-baseline > "$O/bench_through_comm_q2_128.log" 2>&1
python3 bench.py --config cavity --cells 32 --steps 50 --warmup 5 --no-cpu-baseline > "$O/bench_cavity_q4_32.log" 2>&1
python3 bench.py --config cavity --cells 32 --through-comm --steps 50 --warmup 5 --no-cpu-baseline > "$O/bench_through_comm_q4_32.log" 2>&1
python3 scripts/bench_ho.py > "$O/bench_ho.log" 2>&1
# calibration of FETCH_SIZE for ns_hox_kernel: the same pass without the state stream (diagnostic build, then the product
# build again); state bytes are known, so (FETCH - FETCH_nostate) / state bytes is the counter's unit for the LDS-DMA reads
bash scripts/exp_ho.sh "-DHOX_EXP=4" > "$O/exp_nostate_build.log" 2>&1
cd /tmp
pmc pmc_q4_fetch_nostate FETCH_SIZE python3 $R/bench.py --config cavity --steps 5 --warmup 2 --no-cpu-baseline
cd $R
bash scripts/exp_ho.sh "-DHOX_EXP=0" >> "$O/exp_nostate_build.log" 2>&1
echo done > "$O/done"
