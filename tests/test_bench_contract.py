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


@pytest.mark.gpu
def test_bench_falls_back_to_torch_transport_when_the_native_job_fails():
    """N > 1: every rank is a supervisor that runs the measurement in a child process; a failing job with the engine's own
    communicator (here: one rank raises during set-up, the other would wait in a collective) is stopped and repeated once
    with --comm torch, and the line says so"""
    env = dict(os.environ, ADAFLO_BENCH_INJECT_NATIVE_FAILURE="1", ADAFLO_BENCH_ATTEMPT_TIMEOUT="600")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cells", "16", "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 2 and r["config"]["comm"] == "torch" and r["value"] > 0
    assert "injected failure" in r["native_error"]
    assert len([l for l in out.stdout.splitlines() if l.startswith("{")]) == 1


_STUB_WORKER = r'''
import json, os, sys, time
rank, attempt = int(os.environ["RANK"]), int(os.environ["ADAFLO_BENCH_ATTEMPT"])
assert os.environ["ADAFLO_BENCH_WORKER"] == "1"
mode = os.environ["STUB_MODE"]
if attempt == 1 and mode in ("crash", "hang"):
    if rank == 1:
        if mode == "hang":
            time.sleep(3600)
        with open(os.path.join(os.environ["ADAFLO_BENCH_BOX"], "a1.err.1"), "w") as f:
            f.write("RuntimeError: ncclCommInitRank: unhandled system error (stub)")
        sys.exit(3)
    time.sleep(3600)                 # rank 0 waits in a collective that never completes
comm = sys.argv[sys.argv.index("--comm") + 1] if "--comm" in sys.argv else "native"
if rank == 0:
    print("some library chatter")
    print(json.dumps({"value": 1.0, "config": {"comm": comm}}), flush=True)
'''


@pytest.mark.parametrize("mode", ["ok", "crash", "hang"])
def test_supervisors_repeat_a_failed_native_job_with_the_torch_transport(tmp_path, mode):
    """bench.supervise on the CPU with a stub in place of the measuring process: a clean job is forwarded as it is; when
    one rank of the first attempt exits non-zero (or does not end within the limit) while the other waits for it, both
    children are stopped and a second job with --comm torch produces the line, which carries `native_error`"""
    stub = tmp_path / "stub_worker.py"
    stub.write_text(_STUB_WORKER)
    code = ("import sys, argparse; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.supervise(argparse.Namespace(comm='native'), script=%r, argv=['--gpus', '2']))" % (ROOT, str(stub)))
    port = 20000 + os.getpid() % 20000
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_PORT=str(port), STUB_MODE=mode,
                   ADAFLO_BENCH_ATTEMPT_TIMEOUT="3" if mode == "hang" else "60")
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and outs[1][0].strip() == ""
    r = json.loads(lines[0])
    if mode == "ok":
        assert r["config"]["comm"] == "native" and "native_error" not in r
    else:
        assert r["config"]["comm"] == "torch"
        assert ("ncclCommInitRank" in r["native_error"]) if mode == "crash" else ("-99" in r["native_error"])
    assert not os.path.exists("/tmp/adaflo_bench_%d_%d" % (port, os.getpid()))
