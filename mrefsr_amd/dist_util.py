"""One process per GPU over torch.distributed -- backend 'nccl' is RCCL (xGMI) on ROCm, 'gloo' on
CPU for tests.  Mirror of basicsr/utils/dist_util.py:10-82 (init_dist / get_dist_info /
master_only) plus the two data-path helpers the sharded path needs.  The hot path shards over
independent (LR, K-ref) samples: no collective inside the path; inference gathers the
outputs once, training all-reduces net_g's gradients through DistributedDataParallel."""
import functools
import os

import torch
import torch.distributed as dist


def init_dist(launcher='pytorch', backend='nccl', **kwargs):
    """launcher 'pytorch' = env:// rendezvous of torch.distributed.run (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR /
    MASTER_PORT), as dist_util.py:21-25; 'slurm' = one task per GPU under srun (dist_util.py:28-57); anything else raises
    ValueError like the reference (:18)."""
    if launcher == 'slurm':
        return _init_dist_slurm(backend, **kwargs)
    if launcher != 'pytorch':
        raise ValueError(f'Invalid launcher type: {launcher}')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    rank = int(os.environ['RANK'])
    if backend == 'nccl':
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', rank % max(torch.cuda.device_count(), 1))))
    dist.init_process_group(backend=backend, **kwargs)


def slurm_env(environ, n_gpus, port=None, first_host=None):
    """The rendezvous variables a Slurm task derives from its own environment (the arithmetic of dist_util.py:39-56, kept
    apart from the process-group call so it can be tested without Slurm): rank = SLURM_PROCID, world = SLURM_NTASKS,
    local rank = rank modulo the GPUs of the node, master = first host of SLURM_NODELIST, port = argument, else an existing
    MASTER_PORT, else 29500 (torch.distributed's default)."""
    proc_id, ntasks = int(environ['SLURM_PROCID']), int(environ['SLURM_NTASKS'])
    if first_host is None:
        import subprocess
        first_host = subprocess.getoutput(f"scontrol show hostname {environ['SLURM_NODELIST']} | head -n1")
    master_port = str(port) if port is not None else environ.get('MASTER_PORT', '29500')
    return dict(MASTER_PORT=master_port, MASTER_ADDR=first_host, WORLD_SIZE=str(ntasks), LOCAL_RANK=str(proc_id % max(n_gpus, 1)),
                RANK=str(proc_id))


def _init_dist_slurm(backend, port=None, **kwargs):
    env = slurm_env(os.environ, torch.cuda.device_count(), port)
    os.environ.update(env)
    if backend == 'nccl':
        torch.cuda.set_device(int(env['LOCAL_RANK']))
    dist.init_process_group(backend=backend, **kwargs)


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def master_only(func):
    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        if get_dist_info()[0] == 0:
            return func(*args, **kwargs)
    return wrapper


def shard_range(n_total, rank=None, world=None):
    """contiguous shard [lo, hi) of n_total samples for this rank (BASELINE config 4: batch 64 ->
    8 shards of 8); remainders go to the first ranks."""
    if rank is None:
        rank, world = get_dist_info()
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_outputs(out, buffers=None, async_op=False):
    """all_gather of per-rank output batches (equal shapes) -> list of world tensors (rank order).
    The one collective of sharded inference: (B,3,4h,4w) fp32 per rank.  async_op: returns (buffers, work) without making
    the compute stream wait for the collective (RCCL runs it on its own stream: the next batch's kernels overlap the
    ~3 ms ring of 8 x 39 MB); call work.wait() before reading the buffers."""
    rank, world = get_dist_info()
    if not (dist.is_available() and dist.is_initialized()):
        return ([out], None) if async_op else [out]
    if buffers is None:
        buffers = [torch.empty_like(out) for _ in range(world)]
    work = dist.all_gather(buffers, out.contiguous(), async_op=async_op)
    return (buffers, work) if async_op else buffers


def max_over_ranks(seconds, device=None):
    """the benchmark's clock: slowest rank defines the step time"""
    rank, world = get_dist_info()
    if world == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device or ('cuda' if dist.get_backend() == 'nccl' else 'cpu'))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
