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
                          '--warmup', '1', '--cpu-lr', '16', '--no-train-step'], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 1 and line['value'] > 0 and line['scaling'] == 'weak'
    # what the N > 1 line carries beside the contract fields (with and without the gather; a straggler must be visible;
    # the CPU baseline of rank 0 at any world size)
    assert line['step_ms']['min'] <= line['step_ms']['median'] <= line['step_ms']['max'] and line['value_median'] > 0
    assert line['rank_ms_per_step']['min'] <= line['rank_ms_per_step']['max']
    assert line['no_gather']['value'] > 0 and line['no_gather']['steps'] >= 2
    assert line['gather'].startswith('RCCL all_gather')
    assert line['cpu_baseline']['value'] > 0 and line['cpu_baseline']['cores'] >= 1
    assert 'clock_mhz' in line   # (null where sysfs does not expose the clock)


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_and_reports_a_ddp_training_step():
    """`python bench.py --gpus N` with no launcher around it (how the driver starts N = 1, and what a user types): the parent
    spawns torch.distributed.run before touching the GPU (scripts/dist_train.sh:14-16 does the same for the reference).
    MREFSR_BENCH_FORCE_DIST=1 takes that route with one rank; the line then also carries the configs[2]-shaped training step
    with net_g under DistributedDataParallel over RCCL."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(MREFSR_BENCH_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--batch', '1', '--lr', '40', '--steps', '1', '--warmup', '1',
                          '--no-cpu-baseline', '--train-steps', '3'], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 1 and line['value'] > 0
    ts = line['train_step']
    assert ts.get('error') is None, ts
    assert ts['parallelism'] == 'ddp1' and ts['ms_per_step'] > 0 and ts['loss'] > 0


@pytest.mark.gpu
def test_ddp_wrapped_training_engine_reproduces_the_reference_step(golden):
    """BASELINE configs[2] as the reference runs it: net_g inside DistributedDataParallel (base_model.py:98-101) over RCCL --
    here a one-rank process group, which still routes every gradient through DDP's reducer (bucket views, autograd hooks,
    the all-reduce) around the channels-last training engine's custom autograd nodes.  Loss, the 350 per-parameter gradient
    fingerprints and the post-Adam parameter sums must equal the reference's own optimize_parameters on the same inputs
    (e2e_c2.npz), exactly as the un-wrapped step does (tests/test_configs_gpu.py)."""
    import socket

    import torch
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel

    import test_configs_gpu as tc
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
    saved = tc._opt
    try:
        tc._opt = lambda is_train: dict(saved(is_train), dist=True)
        g = golden('e2e_c2')
        model, data, _ = tc._golden_model(g, True)
        assert isinstance(model.net_g, DistributedDataParallel)
        model.feed_data(data)
        tc._check_train_step_against_reference(g, model)
        # second step: the reducer's buckets are rebuilt after the first backward; gradients stay finite and the loss moves
        model.optimize_parameters(2)
        assert all(torch.isfinite(p.grad).all() for p in model.get_bare_model(model.net_g).parameters() if p.grad is not None)
    finally:
        tc._opt = saved
        dist.destroy_process_group()
