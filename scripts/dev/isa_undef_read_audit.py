"""development: vector registers of one kernel of a hipcc -S listing that are READ before their first WRITE in program
order (function inputs excepted).  In structured code -- forward branches around divergent regions, backward branches only
for loops -- program order approximates dominance, so such a read sees whatever the register held: a value of the previous
loop iteration, or garbage in the first one.
  usage: isa_undef_read_audit.py file.s 'kernel-name-substring (demangled)'"""
import re
import subprocess
import sys


def kernel_body(path, needle):
    lines = open(path).read().split("\n")
    for i, l in enumerate(lines):
        if l.startswith("_Z") and ":" in l:
            dem = subprocess.run(["c++filt", l.split(":")[0]], capture_output=True, text=True).stdout
            if needle in dem:
                body = []
                for m in lines[i + 1:]:
                    if m.startswith(".Lfunc_end"):
                        return body
                    body.append(m)
    raise SystemExit("kernel not found")


REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), r) for r in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


NO_DST = ("ds_write", "global_store", "buffer_store", "scratch_store", "global_load_lds", "s_", "v_cmp", "v_cmpx", "v_writelane",
          "v_readlane", "v_readfirstlane", "ds_swizzle_placeholder")
written, first_bad = {("v", 0)}, {}
for ln, l in enumerate(kernel_body(sys.argv[1], sys.argv[2])):
    t = l.strip()
    if not t or t.startswith((";", ".")):
        continue
    t = t.split(";")[0].strip()
    parts = t.split(None, 1)
    if len(parts) < 2:
        continue
    op, ops = parts
    oplist = [o.strip() for o in ops.split(",")]
    if op.startswith("v_writelane"):
        dst, src = regs(oplist[0]), set()          # (writes one lane: counts as a definition of the spill register)
    elif op.startswith(NO_DST):
        dst, src = set(), regs(ops)
        if op.startswith(("v_readlane", "v_readfirstlane", "v_cmp")):
            src = regs(",".join(oplist[1:]))
    else:
        dst, src = regs(oplist[0]), regs(",".join(oplist[1:]))
        if op.startswith(("v_fmac", "v_mac", "v_accvgpr_write")) or "_dpp" in op and False:
            pass
        if op.startswith(("v_fmac", "v_mac")):
            src |= dst                              # accumulator is read
    for r in sorted(src):
        if r not in written and r not in first_bad:
            first_bad[r] = (ln, t)
    written |= dst
print("registers read before their first write in program order:", len(first_bad))
for r, (ln, t) in sorted(first_bad.items(), key=lambda kv: kv[1][0]):
    print("   %s%d  at line %d: %s" % (r[0], r[1], ln, t[:110]))
