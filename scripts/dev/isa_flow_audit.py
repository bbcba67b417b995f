"""development (round 6): the build-time audit of adaflo_amd/isa_audit.py for ONE unit with extra hipcc flags -- lists the kernels in
which hipcc placed register-allocator copies (v_accvgpr_write / _read, v_mov, scratch spills) in the FLOW block of a divergent
if / else, ahead of the s_andn2_saveexec that flips EXEC: they execute under the THEN mask.
   usage: isa_flow_audit.py <unit, e.g. ns_hox> [hipcc flags...]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from adaflo_amd import build, isa_audit  # noqa: E402

unit, flags = sys.argv[1], sys.argv[2:]
work = tempfile.mkdtemp(prefix="isa_audit_")
listing = os.path.join(work, unit + ".s")
subprocess.check_call(["hipcc", "-S", "--cuda-device-only", os.path.join(build.CSRC, unit + ".hip"), "-o", listing] + build.FLAGS + flags,
                      stderr=subprocess.DEVNULL)
found = isa_audit.flow_block_copies(open(listing, errors="replace").read())
for k, blocks in found.items():
    dem = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    print("%s\n   %d Flow block(s) with copies under the THEN mask: %s" % (dem[:150], len(blocks), "; ".join(
        "%d copies (%s ...)" % (len(c), c[0]) for _, c in blocks)))
print("%s: %s" % (unit, isa_audit.summarize(found)))
