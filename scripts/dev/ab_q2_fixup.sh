#!/bin/bash
# A/B of the Q2/Q1 seam fix-up kernel (straight-line vs the loop form, -DQ2_FIXUP_LOOP): kernel averages by rocprofv3 --stats
cd $GRAFT_REPO_ROOT
export ADAFLO_BENCH_NOCHECK=1
for e in "" "-DQ2_FIXUP_LOOP"; do
  hipcc -c adaflo_amd/csrc/ns_q2.hip -o adaflo_amd/lib/ns_q2.o -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-gpu-rdc -DQ2_FAST_BUILD $e || continue
  hipcc -shared -o adaflo_amd/lib/libadaflo_hip.so adaflo_amd/lib/*.o --offload-arch=gfx950 -fno-gpu-rdc
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ab_prof && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_prof -o ab -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 3 --no-cpu-baseline > /tmp/ab.log 2>&1)
  echo "variant [$e]"; python3 -c "
import csv
for r in csv.DictReader(open('/tmp/ab_prof/ab_kernel_stats.csv')):
    if 'q2_seam_fixup' in r['Name'] or 'ns_q2_kernel' in r['Name']:
        print('   %-40s calls %4s  avg %9.1f us' % (r['Name'].replace('adaflo_hip::(anonymous namespace)::','')[:40], r['Calls'], float(r['AverageNs'])/1e3))
"
done
# leave the DEFAULT library behind: the loop above ends on the non-default fix-up form and a reduced instantiation set
hipcc -c adaflo_amd/csrc/ns_q2.hip -o adaflo_amd/lib/ns_q2.o -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-gpu-rdc \
  && hipcc -shared -o adaflo_amd/lib/libadaflo_hip.so adaflo_amd/lib/*.o --offload-arch=gfx950 -fno-gpu-rdc
