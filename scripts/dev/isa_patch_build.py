"""development (round 6): build a variant library whose code object for ONE translation unit comes from an EDITED assembly
listing -- the tool that tells a pipeline hazard (goes away when wait states are inserted) from a wrong instruction stream
(stays).  Steps: hipcc -save-temps (device .s, host .s with the embedded fat binary) -> edit the device .s -> cc1as -> lld ->
clang-offload-bundler -> host .s with .incbin of the new fat binary -> object -> adaflo_amd/lib/variants/lib_<tag>.so.
   usage: isa_patch_build.py <tag> <unit> <edit> '<demangled kernel substring>' [hipcc flags...]
   edits: poison (every vector register that is dead at the header of the kernel's largest loop -- first access in the loop body
          is a write -- gets a value that names it, 2 (1 + N / 256) as the high word of a double, at the top of every iteration:
          neutral for a correct instruction stream; in a wrong one a stale read shows WHICH register it read) | none | nop-valu (s_nop 7 behind every VALU instruction of the kernel, asm statements untouched) |
          nop-dpp (s_nop 7 ahead of and behind every DPP move) | nop-agpr (s_nop 7 around v_accvgpr_* and v_readlane / v_writelane) |
          flow-spills (register-allocator copies in the Flow block of a divergent if / else moved behind the join)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin"
tag, unit, edit, needle = sys.argv[1:5]
flags = sys.argv[5:]
work = tempfile.mkdtemp(prefix="isa_patch_")
base = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-fno-gpu-rdc"]
subprocess.check_call(["hipcc", "-c", os.path.join(ROOT, "adaflo_amd/csrc", unit + ".hip"), "-o", "ref.o", "-save-temps"] + base + flags,
                      cwd=work, stderr=subprocess.DEVNULL)
dev_s = os.path.join(work, unit + "-hip-amdgcn-amd-amdhsa-gfx950.s")
host_s = os.path.join(work, unit + "-host-x86_64-unknown-linux-gnu.s")
lines = open(dev_s).read().split("\n")
start = None
for i, l in enumerate(lines):
    if l.startswith("_Z") and ":" in l:
        dem = subprocess.run(["c++filt", l.split(":")[0]], capture_output=True, text=True).stdout
        if needle in dem:
            start = i
            break
assert start is not None, "kernel not found"
out, in_asm, n_ins, inside = lines[:start + 1], False, 0, True
NOP = "\ts_nop 7"
for l in lines[start + 1:]:
    if inside and l.startswith(".Lfunc_end"):
        inside = False
    t = l.strip()
    if not inside:
        out.append(l)
        continue
    if "#ASMSTART" in t:
        in_asm = True
    if "#ASMEND" in t:
        in_asm = False
        out.append(l)
        continue
    op = t.split()[0] if t and not t.startswith((";", ".")) else ""
    pre = post = False
    if not in_asm and op:
        if edit == "nop-valu":
            post = op.startswith("v_")
        elif edit == "nop-dpp":
            pre = post = "_dpp" in t
        elif edit == "nop-agpr":
            pre = post = op.startswith(("v_accvgpr", "v_readlane", "v_writelane", "v_readfirstlane"))
    if pre:
        out.append(NOP)
        n_ins += 1
    out.append(l)
    if post:
        out.append(NOP)
        n_ins += 1
if edit == "poison":
    REG = re.compile(r"\b(v)(\d+)\b|\b(v)\[(\d+):(\d+)\]")

    def regs(tok):
        r = set()
        for m in REG.finditer(tok):
            if m.group(1):
                r.add(int(m.group(2)))
            else:
                r.update(range(int(m.group(4)), int(m.group(5)) + 1))
        return r
    body = out[start + 1:]
    # the largest loop: label L ... backward branch to L
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    best = None
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i and (best is None or i - labels[m.group(1)] > best[1] - best[0]):
            best = (labels[m.group(1)], i)
    lo, hi = best
    first = {}
    no_dst = ("ds_write", "global_store", "buffer_store", "scratch_store", "global_load_lds", "s_", "v_cmp", "v_readlane",
              "v_readfirstlane", "v_writelane")
    for l in body[lo:hi]:
        t = l.strip().split(";")[0].strip()
        if not t or t.startswith("."):
            continue
        parts = t.split(None, 1)
        if len(parts) < 2:
            continue
        op, ops = parts
        ol = ops.split(",")
        if op.startswith(no_dst):
            dst, src = set(), regs(ops)
        else:
            dst, src = regs(ol[0]), regs(",".join(ol[1:]))
            if op.startswith(("v_fmac", "v_mac")) or "_dpp" in t:
                src |= dst
        for r in src:
            first.setdefault(r, "r")
        for r in dst:
            first.setdefault(r, "w")
    dead = sorted(r for r, k in first.items() if k == "w")
    ins = ["\tv_mov_b32_e32 v%d, 0x%x" % (r, 0x40000000 | (r << 12)) for r in dead]
    out = out[:start + 1] + body[:lo + 1] + ins + body[lo + 1:]
    print("poison: loop of %d lines, %d registers dead at its header" % (hi - lo, len(dead)))
if edit == "flow-spills":
    # Round 6, second half: register-allocator copies (v_accvgpr_write / v_mov) that sit in the FLOW block of a divergent
    # if / else -- between the block's label and the s_andn2_saveexec that flips EXEC to the else side -- execute under the
    # THEN mask only; a value that is live for all lanes and re-loaded after the join is lost for the else lanes.  Move such
    # copies behind the s_or_b64 exec, exec that closes the region (checked: sources not written, destinations not touched
    # in between).
    REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")

    def regs(tok):
        r = set()
        for m in REG.finditer(tok):
            if m.group(1):
                r.add(m.group(1) + m.group(2))
            else:
                r.update(m.group(3) + str(x) for x in range(int(m.group(4)), int(m.group(5)) + 1))
        return r
    body = out[start + 1:]
    moved, i = 0, 0
    while i < len(body):
        m = re.match(r"\s+s_andn2_saveexec_b64 (s\[\d+:\d+\]), (s\[\d+:\d+\])", body[i])
        if not m or m.group(1) != m.group(2):
            i += 1
            continue
        lab = i - 1
        while lab >= 0 and not re.match(r"^\.LBB\d+_\d+:", body[lab]):
            lab -= 1
        close = None
        for j in range(i + 1, min(i + 400, len(body))):
            if re.match(r"\s+s_or_b64 exec, exec, " + re.escape(m.group(1)), body[j]):
                close = j
                break
        if lab < 0 or close is None:
            i += 1
            continue
        cand = [q for q in range(lab + 1, i) if re.match(r"\s+(v_accvgpr_write_b32|v_mov_b32_e32|v_mov_b64_e32) ", body[q])]
        take = []
        for q in cand:
            ops = body[q].split(";")[0].split(None, 1)[1].split(",")
            dst, src = regs(ops[0]), regs(",".join(ops[1:]))
            ok = True
            for r in range(lab + 1, close + 1):
                if r == q:
                    continue
                t = body[r].split(";")[0].strip()
                if not t or t.startswith("."):
                    continue
                parts = t.split(None, 1)
                if len(parts) < 2:
                    continue
                allr = regs(parts[1])
                d2 = set() if parts[0].startswith(("global_store", "ds_write", "s_", "v_cmp", "v_readlane", "v_writelane")) else regs(parts[1].split(",")[0])
                if (r > q and (d2 & src)) or (allr & dst and r not in cand):
                    ok = False
                    break
            if ok:
                take.append(q)
        if take:
            lines = [body[q] for q in take]
            for q in sorted(take, reverse=True):
                del body[q]
            pos = close - len(take) + 1
            body[pos:pos] = lines
            moved += len(take)
            print("flow-spills: block at line %d: moved %d of %d copies behind the join" % (lab, len(take), len(cand)))
        i = close + 1
    out = out[:start + 1] + body
    print("flow-spills: %d instructions moved" % moved)
open(os.path.join(work, "dev_mod.s"), "w").write("\n".join(out))
print("edit %s: %d s_nop inserted" % (edit, n_ins))
run = lambda cmd: subprocess.check_call(cmd, cwd=work)
run([LLVM + "/clang", "-cc1as", "-triple", "amdgcn-amd-amdhsa", "-filetype", "obj", "-target-cpu", "gfx950", "-mrelocation-model", "pic",
     "-o", "dev_mod.o", "dev_mod.s"])
run([LLVM + "/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-plugin-opt=-amdgpu-internalize-symbols",
     "-plugin-opt=mcpu=gfx950", "-o", "dev_mod.out", "dev_mod.o"])
run([LLVM + "/clang-offload-bundler", "-type=o", "-bundle-align=4096",
     "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", "-input=dev_mod.out",
     "-output=mod.hipfb"])
h = open(host_s).read().split("\n")
for i, l in enumerate(h):
    if l.startswith("\t.asciz\t\"__CLANG_OFFLOAD_BUNDLE__"):
        label = h[i - 1].rstrip(":")
        h[i] = "\t.incbin \"%s\"" % os.path.join(work, "mod.hipfb")
        assert h[i + 1].startswith("\t.size\t" + label)
        h[i + 1] = "\t.size\t%s, %d" % (label, os.path.getsize(os.path.join(work, "mod.hipfb")))
        break
else:
    raise SystemExit("embedded fat binary not found in the host listing")
open(os.path.join(work, "host_mod.s"), "w").write("\n".join(h))
vdir = os.path.join(ROOT, "adaflo_amd/lib/variants")
os.makedirs(vdir, exist_ok=True)
obj = os.path.join(vdir, "%s_%s.o" % (unit, tag))
run([LLVM + "/clang", "-c", "host_mod.s", "-o", obj])
objs = [os.path.join(ROOT, "adaflo_amd/lib", f) for f in sorted(os.listdir(os.path.join(ROOT, "adaflo_amd/lib")))
        if f.endswith(".o") and f != unit + ".o"]
lib = os.path.join(vdir, "lib_%s.so" % tag)
subprocess.check_call(["hipcc", "-shared", "-o", lib] + objs + [obj, "--offload-arch=gfx950", "-fno-gpu-rdc"])
os.remove(obj)
print(lib)
