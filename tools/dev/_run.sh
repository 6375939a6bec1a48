timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lean or schedule_switches or folded or full_step or fused" 2>&1 | grep -v amdgpu | tee gpurun_out/lean_tests.log | tail -4
timeout 800 python tools/dev/gpu_fit_rate.py 2>&1 | grep -v amdgpu | tail -6
