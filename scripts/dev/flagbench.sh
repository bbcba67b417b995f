cd "${GRAFT_REPO_ROOT}"
b() { python bench.py --no-cpu-baseline --steps 100 --warmup 10 $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; }
b product_cavity "--config cavity"
ADAFLO_LIB_PATH=$PWD/adaflo_amd/lib/variants/lib_hox_nopostsched.so b hox_nopostsched_cavity "--config cavity"
b product_cavity "--config cavity"
echo "--- residuals product"; python scripts/dev/res_lazy_bench.py 2>&1 | grep "^{" | cut -c1-150 | head -2
echo "--- residuals q2 nopostsched"; ADAFLO_LIB_PATH=$PWD/adaflo_amd/lib/variants/lib_q2_nopostsched.so python scripts/dev/res_lazy_bench.py 2>&1 | grep "^{" | cut -c1-150 | head -2
echo "--- level set product"; python scripts/bench_ops.py ls 2>&1 | grep "^{" | cut -c1-120 | head -12
echo "--- level set q1 nopostsched"; ADAFLO_LIB_PATH=$PWD/adaflo_amd/lib/variants/lib_q1_nopostsched.so python scripts/bench_ops.py ls 2>&1 | grep "^{" | cut -c1-120 | head -12
