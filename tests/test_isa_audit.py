"""adaflo_amd/isa_audit.py: the build-time audit for register-allocator copies that hipcc places in the Flow block of a
divergent if / else, ahead of the EXEC flip -- where they execute under the THEN mask (the cause of the
wild stores of the Q5/Q4 extrapolating residual in rounds 5 and 6).  CPU only: the detector on synthetic listings, and the
record the build leaves for every unit of the library."""
import glob
import json
import os

from adaflo_amd import build, isa_audit

BAD = """
_ZN6kernelE:                            ; @_ZN6kernelE
\tv_mov_b32_e32 v237, v0
\ts_and_saveexec_b64 vcc, s[4:5]
\ts_xor_b64 s[4:5], exec, vcc
\ts_cbranch_execz .LBB0_2
\tglobal_store_dwordx4 v[2:3], v[10:13], off
.LBB0_2:                                ; %Flow
\tv_accvgpr_write_b32 a12, v237
\tv_accvgpr_write_b32 a3, v236
\ts_andn2_saveexec_b64 s[4:5], s[4:5]
\tv_fmac_f64_e32 v[16:17], s[18:19], v[0:1]
\ts_or_b64 exec, exec, s[4:5]
\tv_accvgpr_read_b32 v192, a12
\ts_endpgm
.Lfunc_end0:
"""
# the same copies behind the join; and ordinary THEN code (no skip label between the `if` and the flip)
GOOD = BAD.replace("\tv_accvgpr_write_b32 a12, v237\n\tv_accvgpr_write_b32 a3, v236\n", "").replace(
    "\ts_or_b64 exec, exec, s[4:5]\n", "\ts_or_b64 exec, exec, s[4:5]\n\tv_accvgpr_write_b32 a12, v237\n\tv_accvgpr_write_b32 a3, v236\n")
THEN_CODE = """
_ZN6kernelE:
.LBB0_1:
\tv_mov_b32_e32 v20, s96
\ts_and_saveexec_b64 s[10:11], s[8:9]
\ts_xor_b64 s[10:11], exec, s[10:11]
\tv_mov_b32_e32 v5, s33
\ts_andn2_saveexec_b64 s[10:11], s[10:11]
\tv_lshl_add_u32 v5, v21, 3, s60
\ts_or_b64 exec, exec, s[10:11]
.Lfunc_end0:
"""


# the other form of the flip hipcc emits: s_or_saveexec sX, sX ... s_xor_b64 exec, exec, sX
BAD2 = BAD.replace("\ts_andn2_saveexec_b64 s[4:5], s[4:5]\n", "\ts_or_saveexec_b64 s[4:5], s[4:5]\n\ts_xor_b64 exec, exec, s[4:5]\n")


def test_detector_on_synthetic_listings():
    assert isa_audit.summarize(isa_audit.flow_block_copies(BAD2))["copies"] == 2
    hit = isa_audit.flow_block_copies(BAD)
    assert list(hit) == ["_ZN6kernelE"] and len(hit["_ZN6kernelE"]) == 1
    assert hit["_ZN6kernelE"][0][1] == ["v_accvgpr_write_b32 a12, v237", "v_accvgpr_write_b32 a3, v236"]
    assert isa_audit.summarize(hit) == {"kernels": 1, "copies": 2, "symbols": ["_ZN6kernelE"]}
    assert isa_audit.flow_block_copies(GOOD) == {}
    assert isa_audit.flow_block_copies(THEN_CODE) == {}


def test_every_unit_of_the_library_was_audited_clean():
    """build() compiles each unit's device listing next to its object and records the audit; it refuses to link on a hit"""
    build.build()                                   # (no-op when the library is current)
    report = json.load(open(build.AUDIT))
    units = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(build.CSRC, "*.hip")))
    assert sorted(u for u in report if u in units) == units
    assert all(report[u]["copies"] == 0 for u in units), {u: report[u] for u in units if report[u]["copies"]}


# bytes of scratch per lane of the worst kernel of each unit on the tree of round 6 (a ratchet: raise a figure knowingly).  The
# sweep kernels are written for (near) zero scratch; fdm.hip and one generic level-set kernel carry small arrays there
SCRATCH_CEILING = {"fdm": 640, "ls_kernels": 1344, "ns_hox": 192, "ns_q2": 128, "q1_sweep": 192,
                   "ns_ho": 4096, "ns_hop": 4096}   # (the superseded kernels, compiled only with ADAFLO_BUILD_VARIANTS=1: not policed)


def test_no_kernel_of_the_library_lives_on_scratch():
    """the build records the largest scratch frame per unit (hipcc -Rpass-analysis=kernel-resource-usage).  Kilobytes in a
    sweep kernel mean a register-allocation accident that parity tests do not see (round 6: 3-5 KB in the Q2/Q1 residual
    after a two-line change, 12x slower, every test green)"""
    build.build()
    report = json.load(open(build.AUDIT))
    units = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(build.CSRC, "*.hip")))
    assert all("max_scratch" in report[u] for u in units)
    over = {u: (report[u]["max_scratch"], report[u]["max_scratch_kernel"]) for u in units
            if report[u]["max_scratch"] > SCRATCH_CEILING.get(u, 0)}
    assert not over, over


def test_scratch_parser():
    text = ("x.hpp:1:1: remark: Function Name: _ZN1aE [-Rpass-analysis=kernel-resource-usage]\n"
            "x.hpp:1:1: remark:     ScratchSize [bytes/lane]: 0 [-Rpass-analysis=kernel-resource-usage]\n"
            "x.hpp:1:1: remark: Function Name: _ZN1bE [-Rpass-analysis=kernel-resource-usage]\n"
            "x.hpp:1:1: remark:     ScratchSize [bytes/lane]: 3676 [-Rpass-analysis=kernel-resource-usage]\n")
    assert isa_audit.scratch_use(text) == {"max_scratch": 3676, "max_scratch_kernel": "_ZN1bE", "kernels_with_scratch": 1}
