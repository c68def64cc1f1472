"""CPU, world_size 2 over gloo: the N > 1 plumbing of the sharded path (shard ranges, output
gather, max-over-ranks clock) and DistributedDataParallel gradient averaging with the model's
four-group Adam layout on a small stand-in network (the real nets need the GPU)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from mrefsr_amd import dist_util
    dist_util.init_dist('pytorch', backend='gloo')
    assert dist_util.get_dist_info() == (rank, world)
    # sharding: 13 samples -> [0,7) and [7,13)
    lo, hi = dist_util.shard_range(13)
    # gather of per-rank outputs
    out = torch.full((2, 3, 4, 4), float(rank))
    parts = dist_util.gather_outputs(out)
    gathered = torch.cat(parts)
    # the overlapped form bench.py uses: gathers in flight, waited for later (gloo runs them on a thread pool in any
    # order, so each gets its own buffers here; RCCL orders them on its stream and bench.py re-uses one set)
    b1, w1 = dist_util.gather_outputs(out + 10, None, async_op=True)
    b2, w2 = dist_util.gather_outputs(out + 20, None, async_op=True)
    w1.wait()
    w2.wait()
    for bufs, add in ((b1, 10), (b2, 20)):
        assert torch.equal(torch.cat(bufs), torch.cat([torch.full((2, 3, 4, 4), float(r) + add) for r in range(world)]))
    # clock
    t = dist_util.max_over_ranks(1.0 + rank)
    # DDP gradient averaging == single-process gradient over the concatenated batch
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.LeakyReLU(0.1), torch.nn.Conv2d(4, 3, 3, padding=1))
    ddp = torch.nn.parallel.DistributedDataParallel(net)
    g = torch.Generator().manual_seed(123)
    x_all, y_all = torch.randn(4, 3, 8, 8, generator=g), torch.randn(4, 3, 8, 8, generator=g)
    xs, ys = x_all[2 * rank:2 * rank + 2], y_all[2 * rank:2 * rank + 2]
    torch.nn.functional.l1_loss(ddp(xs), ys).backward()
    grads = [p.grad.clone() for p in net.parameters()]
    ref = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.LeakyReLU(0.1), torch.nn.Conv2d(4, 3, 3, padding=1))
    ref.load_state_dict(net.state_dict())
    torch.nn.functional.l1_loss(ref(x_all), y_all).backward()
    ok = all(torch.allclose(a, b.grad, atol=1e-6) for a, b in zip(grads, ref.parameters()))

    @dist_util.master_only
    def only0():
        return 'ran'
    q.put((rank, (lo, hi), gathered[:, 0, 0, 0].tolist(), t, ok, only0()))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == (0, 7) and res[1][1] == (7, 13)
    for r in res:
        assert r[2] == [0.0, 0.0, 1.0, 1.0]     # rank order preserved by the gather
        assert r[3] == 2.0                      # slowest rank defines the clock
        assert r[4]                             # DDP mean-of-shards gradient == full-batch gradient
    assert res[0][5] == 'ran' and res[1][5] is None


def test_slurm_launcher_environment():
    """dist_util.py:28-57: what a Slurm task exports before init_process_group (no Slurm needed: the host lookup is injected)"""
    import pytest
    from mrefsr_amd import dist_util
    env = dict(SLURM_PROCID='11', SLURM_NTASKS='16', SLURM_NODELIST='node[3-4]')
    assert dist_util.slurm_env(env, 8, first_host='node3') == dict(MASTER_PORT='29500', MASTER_ADDR='node3', WORLD_SIZE='16', LOCAL_RANK='3',
                                                                   RANK='11')
    assert dist_util.slurm_env(dict(env, MASTER_PORT='1234'), 8, first_host='node3')['MASTER_PORT'] == '1234'
    assert dist_util.slurm_env(dict(env, MASTER_PORT='1234'), 8, port=4321, first_host='node3')['MASTER_PORT'] == '4321'
    with pytest.raises(ValueError):
        dist_util.init_dist('mpi')


def test_bench_becomes_its_own_launcher(monkeypatch):
    """`python bench.py --gpus N` without RANK in the environment starts `python -m torch.distributed.run` with N ranks on
    127.0.0.1 and exits with its return code; under a launcher (RANK set) and at N = 1 it does nothing
    (scripts/dist_train.sh:14-16 is the reference's form of the same wrapper)."""
    import subprocess
    import types

    import torch

    import bench
    calls = []
    monkeypatch.setattr(subprocess, 'call', lambda cmd, env=None: calls.append((cmd, env)) or 7)
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 8)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3'])
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MREFSR_BENCH_FORCE_DIST'):
        monkeypatch.delenv(k, raising=False)
    assert bench.spawn_ranks_if_needed(types.SimpleNamespace(gpus=1)) is None and not calls      # N = 1: in-process
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks_if_needed(types.SimpleNamespace(gpus=4))
    assert e.value.code == 7                                                                       # the children's return code
    cmd, env = calls[0]
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1'] and cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[-4:] == ['--gpus', '4', '--steps', '3']
    assert env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and 'RANK' not in env
    monkeypatch.setenv('RANK', '0')
    assert bench.spawn_ranks_if_needed(types.SimpleNamespace(gpus=4)) is None and len(calls) == 1   # already a rank
    monkeypatch.delenv('RANK')
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 2)
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks_if_needed(types.SimpleNamespace(gpus=4))
    assert 'shows 2 GPU' in str(e.value.code)


def test_bench_line_of_an_n_gpu_run_is_assembled_as_the_contract_says():
    """rank 0's JSON line for N = 4 (bench.assemble_line: the function main() calls, fed with the measurements a 4-rank run produces):
    `value` is the whole job -- the pixels of all ranks' batches over the maximum-over-ranks time of the K timed steps --, the weak-
    scaling fields, the per-rank spread, the gather note, the companion figure without the exchange and what RCCL itself reported as
    its world size are on the line, and the line serialises"""
    import json
    import types

    import bench
    args = types.SimpleNamespace(gpus=4, steps=5, warmup=2, batch=8, refs=5, lr=160, mode='infer', cpu_lr=160, no_cpu_baseline=True,
                                 dtype='fp32', graph=False, miopen_find=False, no_train_step=True, train_steps=10, no_gather=False)
    world, elapsed = 4, 1.25                                    # seconds of the timed region, maximum over the ranks
    step_ms = [251.0, 249.0, 250.0, 252.0, 248.0]
    rank_elapsed = [1.21, 1.25, 1.22, 1.24]
    detail = {'conv_nhwc_k3': (60.0, 40, 2.0e13), 'conv_nhwc_k1': (5.0, 14, 1.0e12), 'conv_wino_k3': (90.0, 133, 3.4e13),
              'dcn_fwd': (29.0, 3, 3.6e12), 'mrattn_fwd': (4.7, 3, 2.6e10)}
    no_gather = dict(steps=2, ms_per_step=240.0, value=54.6, note='...')
    rccl = dict(all_reduce_of_ones=4, get_world_size=4, backend='nccl')
    res = bench.assemble_line(args, world, elapsed, step_ms, rank_elapsed, [31.0] * 5, detail, {'conv_wino_k3': 1.0e11},
                              dict(median=2100.0, min=1900.0, max=2400.0, samples=50), no_gather, rccl, None)
    json.loads(json.dumps(res))
    assert res['n_gpus'] == 4 and res['steps'] == 5 and res['warmup'] == 2 and res['scaling'] == 'weak' and res['higher_is_better'] is True
    assert res['unit'] == 'Mpix/s' and res['vs_baseline'] is None and res['data'] == 'synthetic' and res['dtype'] == 'f32'
    mpix_per_step = 4 * 8 * 640 * 640 / 1e6                    # all four ranks' batches
    assert abs(res['value'] - mpix_per_step * 5 / 1.25) < 1e-3 and abs(res['ms_per_step'] - 250.0) < 1e-6
    assert res['config']['parallelism'] == 'dp4' and res['config']['baseline_config'].startswith('configs[3]')
    assert res['rank_ms_per_step'] == dict(min=242.0, max=250.0, note=res['rank_ms_per_step']['note'])
    assert res['rccl'] == rccl and res['no_gather'] == no_gather and res['gather'].startswith('RCCL all_gather')
    assert res['step_ms']['median'] == 250.0 and abs(res['value_median'] - 4 * 8 * 640 * 640 / 1e6 / 0.25) < 1e-3
    assert res['roofline']['bound'] == 'mfma' and 0.0 < res['roofline']['frac'] < 1.0 and res['roofline']['launches'] == 5
    assert res['roofline_conv']['winograd']['launches'] == 133 and res['roofline_conv']['algorithmic_bytes_per_step'] == int(1.0e11)
    assert res['roofline_attn']['bound'] == 'hbm'
    # the same measurements at N = 1: configs[1], no per-rank fields
    args.gpus = 1
    one = bench.assemble_line(args, 1, elapsed, step_ms, None, [31.0] * 5, detail, None, None, None, None, None)
    assert one['n_gpus'] == 1 and abs(one['value'] - 8 * 640 * 640 / 1e6 * 5 / 1.25) < 1e-3 and one['config']['baseline_config'] == 'configs[1]'
    assert 'rank_ms_per_step' not in one and 'rccl' not in one and 'no_gather' not in one
