"""Build-time audit of the device listings (round 6).  hipcc 7.2 can place register-allocator copies -- AGPR spills
(v_accvgpr_write / _read), v_mov copies, scratch spills -- of values that are live for ALL lanes into the FLOW block of a
divergent if / else: between the block's label (the target of the s_cbranch_execz that skips the THEN part) and the
s_andn2_saveexec that flips EXEC to the ELSE side.  There they execute under the THEN mask; lanes of the else side re-load
stale registers behind the join.  This is what made the Q5/Q4 extrapolating residual store to wild addresses in rounds 5 and
6 (the same listing with the copies moved behind the join is exact).  `flow_block_copies` finds the
pattern in a gfx950 assembly listing; adaflo_amd/build.py runs it over every unit it compiles and refuses to link a library
with a hit (scripts/dev/isa_flow_audit.py: the same for one unit with extra flags)."""
import re

COPY = re.compile(r"\s+(v_accvgpr_write_b32|v_accvgpr_read_b32|v_mov_b32_e32|v_mov_b64_e32|scratch_store_dword\w*|scratch_load_dword\w*) ")
LABEL = re.compile(r"^\.LBB\d+_\d+:")
# the two forms of the flip to the ELSE side hipcc emits: s_andn2_saveexec sX, sX  |  s_or_saveexec sX, sX ... s_xor exec, exec, sX
# (what stands between the block's label and the first of these runs under the THEN mask)
FLIP = re.compile(r"\s+s_(andn2|or)_saveexec_b64 (s\[\d+:\d+\]), \2")
IF = re.compile(r"\s+s_and_saveexec_b64 ")


def flow_block_copies(listing):
    """{kernel symbol: [(line of the Flow block's label, [copy instructions])]} of an assembly listing (text)"""
    found, kernel, label = {}, None, None
    lines = listing.split("\n")
    for i, l in enumerate(lines):
        if l.startswith("_Z") and ":" in l.split(";")[0]:
            kernel, label = l.split(":")[0], None
        elif l.startswith(".Lfunc_end"):
            kernel = None
        elif kernel is None:
            continue
        elif LABEL.match(l):
            label = i
        elif IF.match(l):
            label = None      # an `if` starts in this block: what follows is ordinary THEN code, not a Flow block
        elif label is not None and FLIP.match(l):
            copies = [lines[q].split(";")[0].strip() for q in range(label + 1, i) if COPY.match(lines[q])]
            if copies:
                found.setdefault(kernel, []).append((label, copies))
            label = None
    return found


def summarize(found):
    return {"kernels": len(found), "copies": sum(len(c) for blocks in found.values() for _, c in blocks),
            "symbols": sorted(found)}


def scratch_use(remarks):
    """{"max_scratch": bytes per lane, "max_scratch_kernel": symbol, "kernels_with_scratch": n} from the text hipcc prints with
    -Rpass-analysis=kernel-resource-usage.  A parity-green change can send a 512-register kernel to kilobytes of scratch (round
    6: a run-time choice between two counted waits in the Q2/Q1 residual -- 3-5 KB, 12x slower, every test green): the build
    records the figure and tests/test_isa_audit.py bounds it."""
    worst, name, n, cur = 0, None, 0, None
    for line in remarks.split("\n"):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            continue
        m = re.search(r"remark:\s+ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and cur:
            v = int(m.group(1))
            n += v > 0
            if v > worst:
                worst, name = v, cur
    return {"max_scratch": worst, "max_scratch_kernel": name, "kernels_with_scratch": n}
