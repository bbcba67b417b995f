export ADAFLO_BENCH_NOCHECK=1
cd $GRAFT_REPO_ROOT
# usage: exp_q2.sh "<defines>" ...   e.g. "-DQ2_EXP=4 -DQ2_RING=18"
# (development builds: only the instantiations of the headline benchmark, see Q2_FAST_BUILD)
for e in "$@"; do
  hipcc -c adaflo_amd/csrc/ns_q2.hip -o adaflo_amd/lib/ns_q2.o -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-gpu-rdc -DQ2_FAST_BUILD $e || continue
  hipcc -shared -o adaflo_amd/lib/libadaflo_hip.so adaflo_amd/lib/*.o --offload-arch=gfx950 -fno-gpu-rdc
  python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline ${BENCH_ARGS} > /tmp/exp_out.txt 2>&1
  grep "^PROF launch 41" /tmp/exp_out.txt
  tail -1 /tmp/exp_out.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('exp [$e] GDoF/s %.2f  ms/step %.4f (min %.4f)  kernel_ms %.4f' % (d['value']/1e3, d['ms_per_step'], d.get('ms_per_step_min',0), d['roofline']['kernel_ms']))"
done
