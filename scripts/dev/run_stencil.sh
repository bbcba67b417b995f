# development: parity tests that exercise q1_stencil_kernel, then the ops bench
mkdir -p gpurun_out/r03
timeout 1200 python -m pytest tests/test_ls_parity_gpu.py tests/test_ls_solver_steps_gpu.py tests/test_ns_parity_gpu.py tests/test_krylov_gpu.py tests/test_fdm_gpu.py tests/test_navier_stokes_gpu.py tests/test_two_phase_gpu.py -x -q -m gpu -k "not full_size" 2>&1 | tail -5
timeout 300 python scripts/bench_ops.py stencil 2>&1 | grep -v amdgpu.ids
