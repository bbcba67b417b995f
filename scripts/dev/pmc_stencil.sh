# development: PMC passes on the stencil probe (one counter set per run)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03/stencil_pmc; mkdir -p $O
P=$R/scripts/dev/build/stencil_probe
run() { name=$1; ctr=$2; timeout 120 rocprofv3 --pmc $ctr --output-format csv -d $O/$name -o $name -- $P > $O/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
run sq2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES"
run tcp "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum"
python3 - <<PY
import csv, glob, collections
for name in ("fetch","write","sq1","sq2","tcp"):
    fs = glob.glob("$O/%s/**/*counter_collection.csv" % name, recursive=True)
    if not fs: print(name, "no csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(fs[0])):
        agg[row["Kernel_Name"][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in agg.items():
        print(name, k, {c: (sum(v)/len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
