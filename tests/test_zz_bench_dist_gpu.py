"""GPU, last in collection order (a `-x` run has then already reported everything else): bench.py launched the way the
driver launches it for N > 1."""
import pytest


@pytest.mark.gpu
def test_bench_rccl_path_with_one_rank():
    """bench.py's multi-GPU code path (RCCL init, barriers, all_gather of the outputs, max over ranks) under
    `python -m torch.distributed.run --nproc-per-node 1`: the driver launches exactly this with N ranks."""
    import json
    import os
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MREFSR_BENCH_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
                          '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '1', '--batch', '1', '--lr', '40', '--steps', '1',
                          '--warmup', '1', '--no-cpu-baseline'], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 1 and line['value'] > 0 and line['scaling'] == 'weak'
