#!/usr/bin/env python3
"""advection operator / right-hand side of the level set on the sweep structure: streamed evaluated_convection vs.
velocity evaluated from the nodal field (rows of scripts/bench_ops.py for two mesh sizes)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import bench_ops  # noqa: E402

for nc in ((40, 40, 80), (64, 64, 128)):
    bench_ops.ls_case(4, nc, only=("ls_advect_rhs", "ls_advect_vmult", "ls_advect_vmult_nodal", "ls_reinit_vmult", "ls_reinit_vmult_nodal"))
