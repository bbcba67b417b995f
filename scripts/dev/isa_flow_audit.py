"""development (round 6): list register-allocator copies (v_accvgpr_write / v_accvgpr_read / v_mov / scratch_store) that hipcc placed in
the FLOW block of a divergent if / else -- between the block's label and the s_andn2_saveexec that flips EXEC to the else side --,
i.e. that execute under the THEN mask only.  Harmless when the value is live for the then lanes only; fatal when it is live for
all lanes and re-loaded behind the join (the k = 5 extrapolating residual of the x-marching kernel).
   usage: isa_flow_audit.py <unit, e.g. ns_hox> [hipcc flags...]     (prints one line per kernel that has such copies)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
unit, flags = sys.argv[1], sys.argv[2:]
work = tempfile.mkdtemp(prefix="isa_audit_")
base = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-fno-gpu-rdc"]
subprocess.check_call(["hipcc", "-c", os.path.join(ROOT, "adaflo_amd/csrc", unit + ".hip"), "-o", "ref.o", "-save-temps"] + base + flags,
                      cwd=work, stderr=subprocess.DEVNULL)
lines = open(os.path.join(work, unit + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
COPY = re.compile(r"\s+(v_accvgpr_write_b32|v_accvgpr_read_b32|v_mov_b32_e32|v_mov_b64_e32|scratch_store_dword\w*|scratch_load_dword\w*) ")
kernel, label, found, total = None, None, {}, 0
for i, l in enumerate(lines):
    if l.startswith("_Z") and ":" in l.split(";")[0]:
        kernel = l.split(":")[0]
        label = None
    elif re.match(r"^\.LBB\d+_\d+:", l):
        label = i
    elif kernel and label is not None and re.match(r"\s+s_andn2_saveexec_b64 (s\[\d+:\d+\]), \1", l):
        copies = [lines[q].strip() for q in range(label + 1, i) if COPY.match(lines[q])]
        if copies:
            found.setdefault(kernel, []).append((label, copies))
            total += len(copies)
        label = None
    elif re.match(r"\s+s_and_saveexec_b64 ", l):
        label = None  # (an if starts in this block: what follows is ordinary THEN code, not a Flow block)
for k, blocks in found.items():
    dem = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    print("%s\n   %d Flow block(s) with copies under the THEN mask: %s" % (dem[:150], len(blocks), "; ".join(
        "%d copies (%s ...)" % (len(c), c[0]) for _, c in blocks)))
print("%s: %d kernels, %d copies in Flow blocks ahead of the EXEC flip" % (unit, len(found), total))
