"""bench.py contract: helpers on the CPU; on the GPU box the self-launch of `--gpus 2` (gloo dry run when the
box has one GPU, RCCL otherwise) and the cavity configuration (BASELINE configs[4])."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from adaflo_amd import parallel  # noqa: E402


def test_hashed_source_is_partition_independent():
    """the source vector is a function of the GLOBAL DoF index: bricks of any partition see the same
    global vector and replicas of interface DoFs agree"""
    k, cells = 2, [3, 2, 2]
    whole = parallel.BrickPartition((1, 1, 1), 0, [6, 2, 2], [0] * 3, [1] * 3)
    g = bench.hashed_uniform(torch, bench.global_dof_index(torch, whole, k, 3, "cpu"), 0).reshape(5, 5, 13, 3)
    for rank in range(2):
        part = parallel.BrickPartition((2, 1, 1), rank, cells, [0] * 3, [1] * 3)
        loc = bench.hashed_uniform(torch, bench.global_dof_index(torch, part, k, 3, "cpu"), 0).reshape(5, 5, 7, 3)
        assert torch.equal(loc, g[:, :, 6 * rank:6 * rank + 7])
    assert -1.0 <= float(g.min()) and float(g.max()) < 1.0 and abs(float(g.mean())) < 0.1
    assert bench.b_alg_per_cell(2) == 2992 and bench.b_alg_per_cell(4) == 15504


def test_beltrami_interpolant_uses_gauss_lobatto_nodes():
    from adaflo_amd import BrickMesh, beltrami
    from adaflo_amd.navier_stokes import node_coordinates
    mesh = BrickMesh([2, 3, 2], [0, 0, 0], [1, 1, 3])
    for k in (2, 4):
        got = bench.beltrami_nodal(torch, mesh.lower, mesh.h, mesh.ncell, k, 0.0, "cpu").numpy()
        assert np.abs(got - beltrami.velocity(node_coordinates(mesh, k), 0.0).reshape(-1)).max() < 1e-14


def _run(args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.gpu
def test_bench_self_launches_two_ranks():
    r = _run(["--gpus", "2", "--cells", "32", "--steps", "3", "--warmup", "1"])
    assert r["n_gpus"] == 2 and r["config"]["partition"] == "2x1x1" and r["value"] > 0
    assert r["config"]["dofs"] == 3 * 129 * 65 * 65 + 65 * 33 * 33
    if torch.cuda.device_count() < 2:
        assert r.get("dry_run") is True
    r = _run(["--gpus", "2", "--cells", "32", "--steps", "3", "--warmup", "1", "--comm", "native"])
    assert r["n_gpus"] == 2 and r["config"]["comm"] == "native" and r["value"] > 0


@pytest.mark.gpu
def test_bench_cavity_configuration():
    r = _run(["--config", "cavity", "--cells", "16", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert "Q4/Q3" in r["metric"] and "incompressible stationary" in r["config"]["workload"]
    assert r["roofline"]["alg_bytes_per_dof"] == 70.8 and r["roofline"]["kernel"] == "ns_hox_kernel"
    r = _run(["--config", "cavity", "--cells", "16", "--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert r["scaling"] == "strong" and r["config"]["cells_per_gpu"] == 8 * 16 * 16
