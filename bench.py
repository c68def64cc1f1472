#!/usr/bin/env python3
"""Benchmark of the MRefSR hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole path over one synthetic batch per GPU, inputs resident in HBM:
VGG16 features of the up-sampled LR image and the K references -> 3x3-patch correlation + top-1
(HIP) -> offset planes (HIP) -> VGG19 taps -> MRAPARestorationNet (HIP convolutions -- direct and
Winograd split-operand implicit GEMM --, DynAgg glue, DCNv2, multi-reference attention) -> 4x output
on the device [+ all_gather of the outputs across ranks when N > 1: BASELINE configs[3]].

Workload at N=1 = BASELINE.json configs[1]: 5-ref 4x SR inference, LR 160x160 -> 640x640, batch 8,
fp32.  N > 1 keeps 8 samples per GPU (weak scaling; configs[3] = 64 samples over 8 GPUs).
Metric: output Mpix/s = N * B * 640 * 640 / 1e6 / t_step.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     for the correlation kernel, measured live with HIP events on its launch stream
  cpu_baseline the CPU port (oracle/pipeline.py) timed on this host's cores on a bounded sample (rank 0, any N)
  step_ms      median / min / max of the K timed steps (HIP events between steps) and value_median = the metric on the median
  clock_mhz    shader clock sampled from sysfs during the timed region (the convolution phase runs power-limited)
  at N > 1: rank_ms_per_step (min / max over ranks) and no_gather (the same loop without the RCCL all_gather of the outputs)
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MATRIX_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MATRIX_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense
HBM_PEAK_GBS = 8000.0


def synth_batch(b, k, lr, seed):
    """gt = 5x5-box-smoothed uniform noise; lq = bicubic down 4; up = bicubic
    up 4; ref_k = roll(gt, (17k, -23k)) + N(0, 0.02^2).  Generated on the CPU (same bits on every
    machine), then copied."""
    g = torch.Generator().manual_seed(seed)
    hr = 4 * lr
    gt = torch.rand(b, 3, hr, hr, generator=g)
    gt = F.avg_pool2d(F.pad(gt, (2, 2, 2, 2), mode='reflect'), 5, 1)
    lq = F.interpolate(gt, scale_factor=0.25, mode='bicubic', align_corners=False).clamp(0, 1)
    up = F.interpolate(lq, scale_factor=4, mode='bicubic', align_corners=False).clamp(0, 1)
    refs = []
    for kk in range(1, k + 1):
        r = torch.roll(gt, shifts=(17 * kk, -23 * kk), dims=(2, 3)) + 0.02 * torch.randn(gt.shape, generator=g)
        refs.append(r.clamp(0, 1))
    return dict(img_in_lq=lq, img_in_up=up, img_ref_list=torch.stack(refs, 1), img_in=gt)


def seeded_weights(model, seed=10):
    """random-init weights of the named architecture (no checkpoints offline): extractor / VGG19
    kaiming-normal; net_g keeps its own init (srntt normal 0.02, DynAgg uniform) with
    conv_offset_mask perturbed N(0, 1e-3^2) so the learned-offset path is exercised."""
    g = torch.Generator().manual_seed(seed)
    sds = {}
    for name in ('net_extractor', 'net_map', 'net_g'):
        net = model.get_bare_model(getattr(model, name))
        sd = net.state_dict()
        for key, v in sd.items():
            if name != 'net_g' and v.dim() == 4 and v.shape[-1] == 3:
                fan_in = v.shape[1] * 9
                sd[key] = torch.randn(v.shape, generator=g) * (2.0 / fan_in) ** 0.5
            elif name != 'net_g' and v.dim() == 1 and 'conv' in key:
                sd[key] = torch.zeros_like(v)
            elif name == 'net_g':
                if 'conv_offset_mask' in key:
                    sd[key] = torch.randn(v.shape, generator=g) * 1e-3
                elif v.dim() == 4:
                    sd[key] = torch.randn(v.shape, generator=g) * 0.02
                elif v.dim() == 1 and v.numel() > 1:
                    sd[key] = torch.zeros_like(v)
        # DynAgg main weights: uniform(+-1/sqrt(C*9)) like the module init
        if name == 'net_g':
            for key, v in sd.items():
                if key.endswith('_dyn_agg.weight'):
                    bound = 1.0 / (v.shape[1] * 9) ** 0.5
                    sd[key] = (torch.rand(v.shape, generator=g) * 2 - 1) * bound
        net.load_state_dict(sd)
        sds[name] = {k_: v_.cpu().clone() for k_, v_ in sd.items()}
    return sds


def build(args, dist_on):
    from mrefsr_amd.models import build_model
    opt = dict(
        name='bench', model_type='MultiRefRestorationModel', scale=4, crop_border=4, num_gpu=1, is_train=args.mode == 'train',
        dist=dist_on and args.mode == 'train',
        network_g=dict(type='MRAPARestorationNet', ngf=64, n_blocks=16, groups=8),
        network_map=dict(type='CorrespondenceGenerationArch', patch_size=3, stride=1,
                         vgg_layer_list=['relu1_1', 'relu2_1', 'relu3_1'], vgg_type='vgg19'),
        network_extractor=dict(type='ContrasMultiExtractorSep'), path={},
        train=dict(lr_g=1e-4, lr_offset=1e-4, lr_relu2_offset=1e-5, lr_relu3_offset=1e-6, beta_g=[0.9, 0.999],
                   scheduler=dict(type='MultiStepLR', milestones=[300000, 400000], gamma=0.5), net_g_pretrain_steps=0,
                   pixel_criterion='L1Loss', pixel_weight=1.0))
    import logging
    logging.getLogger('basicsr').setLevel(logging.ERROR)
    return build_model(opt)


def cpu_baseline(sds, args, model=None):
    """the CPU port (oracle/pipeline.py + oracle/mrefsr_oracle.c) on this host's cores, one sample
    of the same workload (B=1, same K, same LR size), timed once.  With ``model`` given the same sample also
    goes through the GPU path and the two results are compared (parity at the benchmark's full size)."""
    from oracle import c_api, pipeline
    # one thread per physical core up to 64: beyond that torch's intra-op pools and OpenMP teams
    # oversubscribe on these op sizes (256 threads measured 6x slower than 8 on the same pass)
    threads = max(1, min(64, (os.cpu_count() or 2) // 2))
    torch.set_num_threads(threads)
    c_api.set_num_threads(threads)
    data = synth_batch(1, args.refs, args.cpu_lr, seed=10)
    t0 = time.time()
    out, idx = pipeline.forward(sds['net_g'], sds['net_extractor'], sds['net_map'], data)
    dt = time.time() - t0
    mpix = out.shape[0] * out.shape[2] * out.shape[3] / 1e6
    res = dict(value=mpix / dt, unit='Mpix/s', cores=threads, kind='port',
               sample=f'B=1, K={args.refs}, LR {args.cpu_lr}x{args.cpu_lr} -> {4*args.cpu_lr}x{4*args.cpu_lr}, fp32, one pass '
                      f'({dt:.1f} s; torch-CPU convolutions + C/OpenMP matching, {c_api.num_threads()} OpenMP threads)')
    if model is not None and args.dtype == 'fp32':
        # the checker's result against the product's on the very same sample (north star: indices equal, pixels <= 1e-3)
        model.feed_data(data)
        model.test()
        model.check_numeric_range()
        gidx = model.max_idx.cpu().numpy().reshape(idx.shape)
        import torch.nn.functional as F
        resid = out - F.interpolate(data['img_in_lq'], None, 4, 'bilinear', False)   # what net_g adds to the bilinear base
        diff = float((model.output.cpu() - out).abs().max())
        res['parity_full_size'] = dict(max_abs_pixel_diff=diff, net_g_residual_max_abs=float(resid.abs().max()),
                                       diff_over_residual=diff / max(float(resid.abs().max()), 1e-30),
                                       match_index_mismatches=int((gidx != idx).sum()), match_indices=int(idx.size),
                                       note='GPU path vs oracle/pipeline.py on this sample; extractor features differ by fp32 '
                                            'rounding noise (HIP vs oneDNN convolutions), so a few near-tie matches may flip; with the '
                                            'random-init net_g of this benchmark the output is the bilinear base plus a small residual, '
                                            'hence diff_over_residual')
    return res


def train_step_figure(args, dist_on, rank):
    """BASELINE configs[2] per-GPU shape (B=4, K=5, LR 40x40 -> 160x160, fp32): `optimize_parameters` of
    multi_ref_restoration_model.py:197-279 -- frozen extractor / matching, net_g forward, L1, backward, four-group Adam --
    timed like the main loop (barrier + synchronize both sides, max over ranks).  With more than one rank (or
    MREFSR_BENCH_FORCE_DIST=1) net_g is wrapped in DistributedDataParallel over RCCL (base_model.py:98-101): the 4-GPU run of
    the shipped yml is this at N = 4.  launches / kernel_ms: kernel launches and summed kernel time of one step from the
    torch profiler's device trace (None when another profiler owns the device)."""
    import copy
    a = copy.copy(args)
    a.mode, a.batch, a.refs, a.lr = 'train', 4, 5, 40
    model = build(a, dist_on)
    seeded_weights(model)
    model.feed_data(synth_batch(a.batch, a.refs, a.lr, seed=100 + rank))
    for i in range(4):
        model.optimize_parameters(i + 1)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.train_steps):
        model.optimize_parameters(5 + i)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    from mrefsr_amd import dist_util
    dt = dist_util.max_over_ranks(time.perf_counter() - t0)
    loss = float(model.get_current_log()['l_g_pix'])
    launches, kernel_ms = None, None
    # (decided from the environment only, so that every rank takes the same branch: the extra step is a DDP step)
    do_prof = os.environ.get('MREFSR_BENCH_KINETO', '1') == '1' and not any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ)
    if do_prof and rank == 0:
        try:
            from torch.profiler import ProfilerActivity, profile
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                model.optimize_parameters(5 + args.train_steps)
                torch.cuda.synchronize()
            evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and 'emcpy' not in e.name and 'emset' not in e.name]
            if evs:
                launches, kernel_ms = len(evs), round(sum(e.device_time_total for e in evs) / 1e3, 3)
        except Exception as e:   # the figure is optional; the step itself has run (or raised inside optimize_parameters on every rank alike)
            launches, kernel_ms = None, None
    elif do_prof and dist_on:
        model.optimize_parameters(5 + args.train_steps)   # the other ranks' half of rank 0's profiled step (gradient all-reduce)
    world = dist.get_world_size() if dist_on else 1
    return dict(ms_per_step=round(dt / args.train_steps * 1e3, 2), steps=args.train_steps, launches=launches, kernel_ms=kernel_ms,
                samples_per_s=round(world * a.batch * args.train_steps / dt, 2), loss=loss,
                workload=f'configs[2] per-GPU shape: 5-ref training step (extractor + matching frozen, net_g fwd + L1 + bwd + 4-group Adam), '
                         f'B=4 per GPU, LR 40x40 -> 160x160, fp32{", DistributedDataParallel(net_g) over RCCL" if dist_on else ""}',
                parallelism=f'ddp{world}' if dist_on else 'dp1')


def spawn_ranks_if_needed(args):
    """`python bench.py --gpus N` started WITHOUT torch.distributed.run (no RANK in the environment): become the launcher --
    the reference wraps its launcher in the entry script the same way (scripts/dist_train.sh:14-16).  The N ranks are fresh
    child processes of `python -m torch.distributed.run` (rendezvous on 127.0.0.1, a free port); this parent has not made a
    single GPU call (importing torch and counting devices do not initialise HIP), only forwards the children's output --
    rank 0 prints the JSON line -- and exits with their return code.  N = 1 stays in-process unless
    MREFSR_BENCH_FORCE_DIST=1 asks for the RCCL code path with one rank."""
    if 'RANK' in os.environ:
        return
    if args.gpus <= 1 and os.environ.get('MREFSR_BENCH_FORCE_DIST') != '1':
        return
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but this node shows {have} GPU(s)')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or args.gpus) // max(args.gpus, 1))))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    sys.exit(subprocess.call(cmd, env=env))


class ClockSampler:
    """shader clock of this rank's GPU from sysfs while a timed region runs (no GPU API call in the reader thread:
    /sys/class/drm/card*/device/hwmon/hwmon*/freq1_input, or the starred level of pp_dpm_sclk).  The card is the one whose PCI
    address is the current device's; when that cannot be matched (a box exposes more cards in sysfs than it makes visible), every card
    is sampled and the busiest one is reported."""

    def __init__(self, index, period=0.02):
        import glob
        import threading
        self.stop_flag, self.period = False, period
        cards = sorted(glob.glob('/sys/class/drm/card[0-9]*/device/pp_dpm_sclk'))
        self.how = 'all cards, busiest'
        try:   # (the device is initialised by now: the model has been built)
            pr = torch.cuda.get_device_properties(torch.cuda.current_device())
            addr = f'{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.'
            mine = [c for c in cards if addr in os.path.realpath(os.path.dirname(c))]
            if mine:
                cards, self.how = mine[:1], 'PCI ' + addr + '0'
        except Exception:
            pass
        self.cards = []
        for c in cards:
            f = sorted(glob.glob(os.path.join(os.path.dirname(c), 'hwmon', 'hwmon*', 'freq1_input')))
            self.cards.append((c, f[0] if f else None))
        self.samples = [[] for _ in self.cards]
        self.thread = threading.Thread(target=self.run, daemon=True)

    @staticmethod
    def read(dpm, freq):
        try:
            if freq:
                return int(open(freq).read().strip()) / 1e6
            for ln in open(dpm).read().splitlines():
                if ln.rstrip().endswith('*'):
                    return float(''.join(ch for ch in ln.split(':')[1] if ch.isdigit() or ch == '.'))
        except Exception:
            pass
        return None

    def run(self):
        while not self.stop_flag:
            for i, (dpm, freq) in enumerate(self.cards):
                v = self.read(dpm, freq)
                if v:
                    self.samples[i].append(v)
            time.sleep(self.period)

    def __enter__(self):
        if self.cards:
            self.thread.start()
        return self

    def __exit__(self, *exc):
        self.stop_flag = True
        if self.cards:
            self.thread.join(timeout=1.0)

    def summary(self):
        best = None
        for i, smp in enumerate(self.samples):
            if smp:
                v = sorted(smp)
                if best is None or v[len(v) // 2] > best[0][len(best[0]) // 2]:
                    best = (v, i)
        if best is None:
            return None
        v, i = best
        return dict(median=round(v[len(v) // 2], 1), min=round(v[0], 1), max=round(v[-1], 1), samples=len(v),
                    source=('hwmon freq1_input' if self.cards[i][1] else 'pp_dpm_sclk level') + f' of {os.path.basename(os.path.dirname(os.path.dirname(self.cards[i][0])))} '
                           f'({self.how}) during the timed steps')


def assemble_line(args, world, elapsed, step_ms, rank_elapsed, corr_ms, detail, detail_bytes, clock_summary, no_gather_fig, rccl_world, train_fig):
    """rank 0's JSON line (everything but the CPU baseline) from the measurements of the timed region: a pure function of its
    arguments, the loaded library's self-description and the committed counter summaries under profiles/ -- no GPU call, so
    tests/test_dist_cpu.py assembles and checks the N > 1 line on a CPU host"""
    detail_bytes = detail_bytes or {}
    hr = 4 * args.lr
    mpix_step = world * args.batch * hr * hr / 1e6
    n_pair = args.batch * args.refs
    P = (args.lr - 2) ** 2
    alg_flops = 2.0 * P * P * 2304 * n_pair                         # 2 P^2 2304 per (sample, ref)
    alg_bytes = ((1 + args.refs) * 256 * args.lr ** 2 * 4 + 12 * args.refs * P) * args.batch
    exact_only = os.environ.get('MREFSR_CORR_EXACT', '0') == '1'
    from mrefsr_amd.archs import ref_map_util as _rmu
    fp16_pre = (not exact_only) and (not _rmu._BF16_PREFILTER)
    if exact_only:
        tiles = -(-(args.lr - 2) // 6) * -(-(args.lr - 2) // 14)
        exe_flops = 2.0 * 128 * 128 * 256 * tiles * tiles * n_pair
        exe_kernel, exe_dtype = 'corr_top1_kernel (exact fp32 MFMA)', 'f32'
    else:
        # which pre-filter kernel the loaded library launches and the matrix work it issues: asked of the library itself
        import ctypes as _C
        from mrefsr_amd import _lib as _l
        name, dt = _C.create_string_buffer(96), _C.c_int()
        per_pair = _l.load().mrefsr_corr_prefilter_info(1 if fp16_pre else 0, 256, args.lr, args.lr, name, 96, _C.byref(dt))
        exe_flops = float(per_pair) * n_pair
        exe_kernel = name.value.decode() + ' + corr_rescore_kernel (one mrefsr_corr_top1_prefilter_f32 call)'
        exe_dtype = ('fp16 (single plane, one v_mfma_f32_16x16x32_f16 per product, box-sum in registers, data-dependent window)'
                     if dt.value == 1 else 'bf16 (two-term split, 3 MFMAs per product)')
    # the least matrix work any pixel-Gram formulation needs: every (query pixel, reference pixel) product once
    gram_min_flops = 2.0 * (args.lr ** 2) ** 2 * 256 * n_pair
    exe_peak = FP32_MATRIX_PEAK_TFLOPS if exact_only else BF16_MATRIX_PEAK_TFLOPS
    avg_ms = sum(corr_ms) / max(len(corr_ms), 1)
    roof = None
    traffic, traffic_src, issue_cnt = None, None, None
    try:  # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)
        files = sorted(f for f in os.listdir(os.path.join(ROOT, 'profiles')) if f.endswith('corr_top1_pmc.json'))
        pmc = json.load(open(os.path.join(ROOT, 'profiles', files[-1])))
        if pmc.get('shape') == f'n_pair={n_pair} (B={args.batch},K={args.refs}), C=256, {args.lr}x{args.lr}' and \
                pmc.get('exact_only', True) == exact_only and pmc.get('pre_filter_kernel', '').split('<')[0] == exe_kernel.split(' ')[0].split('<')[0]:
            traffic, traffic_src = pmc['traffic_bytes'], 'profiles/' + files[-1]
            issue_cnt = pmc.get('issue')
    except Exception:
        pass
    if avg_ms > 0:
        exe_tf = exe_flops / (avg_ms * 1e-3) / 1e12
        alg_tf = alg_flops / (avg_ms * 1e-3) / 1e12
        roof = dict(bound='mfma', kernel=exe_kernel,
                    # the roofline number: MFMA FLOP actually issued per call / measured time of the whole call / dense peak of
                    # the issued dtype (MI355X_MICROARCH.md).  The reference formulation's work is reported beside it.
                    achieved=round(exe_tf, 2), peak=exe_peak, unit='TFLOP/s', frac=round(exe_tf / exe_peak, 4),
                    executed_mfma_dtype=exe_dtype, executed_mfma_flop_per_launch=exe_flops,
                    traffic=traffic, traffic_source=traffic_src, algorithmic_bytes=alg_bytes,
                    avg_launch_ms=round(avg_ms, 3), launches=len(corr_ms),
                    algorithmic_flop_per_launch=alg_flops, algorithmic_tflops=round(alg_tf, 2),
                    algorithmic_speedup_vs_fp32_matrix_peak=round(alg_tf / FP32_MATRIX_PEAK_TFLOPS, 3),
                    algorithmic_hbm_gbs=round(alg_bytes / (avg_ms * 1e-3) / 1e9, 2),
                    gram_minimum_flop_per_launch=gram_min_flops,
                    useful_frac=round(gram_min_flops / (avg_ms * 1e-3) / 1e12 / exe_peak, 4),
                    note='frac = executed MFMA FLOP / time of the whole correlation call (pre-filter + exact re-scoring + fallbacks) / '
                         'dense peak of the executed dtype; useful_frac = the same with only the pixel-Gram minimum (each query pixel x '
                         'reference pixel product once, 2 * (h*w)^2 * 256 FLOP per pair) counted as work: a kernel that stops issuing '
                         'redundant MFMAs lowers frac and raises useful_frac at the same time.  algorithmic_* = the fp32 work of the reference formulation '
                         '(2*P^2*2304 FLOP per (sample,ref), SURVEY 8d) over the same time: the kernels reach the same bits with '
                         'less matrix work (pixel-Gram restatement, 16-bit pre-filter + exact fp32 re-scoring of ~1.5 candidates '
                         'per query), so that figure exceeds the fp32 matrix peak; it is a speed-up, not a roofline fraction.')
    if roof is not None and issue_cnt and issue_cnt.get('SQ_INSTS_VALU'):
        # which issue port binds: wave instructions of one call (committed counter pass) against the SIMD cycles of the measured call
        # at the measured clock.  A wave64 VALU instruction holds its SIMD's 16 lanes for 4 cycles (SQ_ACTIVE_INST_VALU counts those
        # quad-cycles); a v_mfma_f32_16x16x32_f16 holds the matrix pipe for 16 cycles (4 passes) and its issue blocks the VALU port
        # for part of that (tools/ubench/mfma_valu_overlap.hip: ~4 VALU instructions hide under one 32-cycle MFMA).
        clk = (clock_summary or {}).get('median') or 2400.0
        clk = clk if clk >= 500.0 else 2400.0   # (a sensor that read an idle card: nominal)
        simd_cycles = 1024.0 * clk * 1e6 * avg_ms * 1e-3
        mfma_i = issue_cnt.get('SQ_INSTS_MFMA', 0.0)
        valu_i = issue_cnt['SQ_INSTS_VALU'] - mfma_i
        roof['issue'] = dict(
            valu_insts=valu_i, mfma_insts=mfma_i, lds_insts=issue_cnt.get('SQ_INSTS_LDS'), salu_insts=issue_cnt.get('SQ_INSTS_SALU'),
            vmem_insts=(issue_cnt.get('SQ_INSTS_VMEM_RD', 0.0) + issue_cnt.get('SQ_INSTS_VMEM_WR', 0.0)),
            clock_mhz=clk, simd_cycles=simd_cycles,
            valu_frac=round(4.0 * valu_i / simd_cycles, 4), mfma_frac=round(16.0 * mfma_i / simd_cycles, 4),
            valu_active_frac=round(4.0 * issue_cnt.get('SQ_ACTIVE_INST_VALU', 0.0) / simd_cycles, 4),
            source=traffic_src,
            note='wave instructions per call (SQ_INSTS_VALU counts MFMAs too: valu_insts = VALU - MFMA) x cycles each holds its port '
                 '(VALU 4, v_mfma_f32_16x16x32_f16 16) / (1024 SIMDs x measured clock x measured call time). valu_frac + mfma_frac '
                 'near 1 = the wave schedulers have no free slots: the call is issue-bound, not matrix-FLOP-bound')
    if args.mode == 'train':
        base_cfg = ('configs[2] (per-GPU shape of the 4-GPU DDP run)' if (args.batch, args.refs, args.lr, args.dtype) == (4, 5, 40, 'fp32')
                    else 'none (training step at a non-baseline shape)')
    elif (args.refs, args.lr, args.dtype) == (5, 160, 'fp32') and args.batch == 8:
        base_cfg = 'configs[1]' if world == 1 else 'configs[3] (per-GPU batch 8 + RCCL all_gather of outputs)'
    elif (args.batch, args.refs, args.lr, args.dtype) == (1, 10, 320, 'bf16'):
        base_cfg = 'configs[4]'
    elif (args.batch, args.refs, args.lr, args.dtype) == (1, 1, 40, 'fp32'):
        base_cfg = 'configs[0] shape (on the GPU; the reference runs it on the CPU)'
    else:
        base_cfg = 'none (not a BASELINE.json configuration)'
    res = dict(metric='4x SR Mpix/sec, 5-ref 160x160->640x640; PSNR within 0.01 dB of ref', value=round(mpix_step * args.steps / elapsed, 4),
               unit='Mpix/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=round(elapsed / args.steps * 1e3, 2), higher_is_better=True, scaling='weak', vs_baseline=None,
               dtype='f32' if args.dtype == 'fp32' else 'bf16', data='synthetic',
               config=dict(workload=f'{args.refs}-ref 4x SR {"inference" if args.mode == "infer" else "training step"}, '
                                    f'LR {args.lr}x{args.lr} -> {hr}x{hr}, batch {args.batch} per GPU, {args.dtype}, random-init weights',
                           baseline_config=base_cfg,
                           per_gpu_batch=args.batch, refs=args.refs, lr=args.lr, mode=args.mode,
                           parallelism=f'dp{world}', miopen_find=bool(args.miopen_find), hip_graph=bool(args.graph)),
               roofline=roof)
    sm = sorted(step_ms)
    med = sm[len(sm) // 2] if len(sm) % 2 else 0.5 * (sm[len(sm) // 2 - 1] + sm[len(sm) // 2])
    res['step_ms'] = dict(median=round(med, 2), min=round(sm[0], 2), max=round(sm[-1], 2),
                          note='rank 0, HIP events between consecutive steps of the timed region; value / ms_per_step are the '
                               'whole region over K (the contract), value_median = the metric on the median step (SURVEY 8d)')
    res['value_median'] = round(mpix_step / (med * 1e-3), 4)
    res['clock_mhz'] = clock_summary
    if rank_elapsed is not None:
        res['rank_ms_per_step'] = dict(min=round(min(rank_elapsed) / args.steps * 1e3, 2), max=round(max(rank_elapsed) / args.steps * 1e3, 2),
                                       note='each rank\'s own clock around the timed region (a straggler shows as max >> min)')
        res['gather'] = 'off (--no-gather)' if args.no_gather else 'RCCL all_gather of the outputs inside the step (async, overlapped with the next batch)'
    if no_gather_fig is not None:
        res['no_gather'] = no_gather_fig
    if rccl_world is not None:
        res['rccl'] = rccl_world
    if detail and detail.get('conv_nhwc_k3'):
        # the kernel that now takes most of the step: the bf16-split implicit-GEMM convolution
        ms3, n3, fl3 = detail['conv_nhwc_k3']
        ms1, n1, fl1 = detail.get('conv_nhwc_k1', (0.0, 0, 0.0))
        msw, nw, flw = detail.get('conv_wino_k3', (0.0, 0, 0.0))   # launches on the Winograd F(2x2, 3x3) kernel (terms 17): 16 MFMA products per 36 direct ones
        conv_ms = ms3 + ms1 + msw
        ach = (fl3 + fl1 + flw) / (conv_ms * 1e-3) / 1e12
        from mrefsr_amd.archs import nhwc as _nhwc
        nprod = {16: 3, 6: 6, 3: 3, 1: 1}[_nhwc.TERMS]
        exe_dtype = {16: 'fp16 (exact two-term split of both operands, 3 MFMAs per fp32-equivalent product)',
                     6: 'bf16 (exact three-term split, 6 MFMAs per fp32-equivalent product)',
                     3: 'bf16 (two-term split, 3 MFMAs per product; reduced accuracy, experiments only)',
                     1: 'bf16 arithmetic (one MFMA per product; --dtype bf16)'}[_nhwc.TERMS]
        conv_traffic, conv_traffic_src = None, None
        try:  # HBM-side bytes per step of the two conv kernels from the committed per-step PMC summary (same workload only)
            if (args.batch, args.refs, args.lr, args.dtype) == (8, 5, 160, 'fp32'):
                pfile = sorted(f for f in os.listdir(os.path.join(ROOT, 'profiles')) if f.endswith('_bench_pmc_per_step.json'))[-1]
                pm = json.load(open(os.path.join(ROOT, 'profiles', pfile)))['kernels']
                conv_traffic = sum(v.get('hbm_read_bytes(FETCH_SIZE*1024*2)', 0) + v.get('hbm_write_bytes(WRITE_SIZE*1024)', 0)
                                   for k, v in pm.items() if k.startswith(('conv_nhwc', 'conv_wino', 'conv1x1')))   # (both generations of the Winograd kernel)
                conv_traffic_src = 'profiles/' + pfile
        except Exception:
            pass
        res['roofline_conv'] = dict(
            bound='mfma', kernel=f'conv_nhwc_kernel<MODE {dict([(16, 2), (6, 0), (3, 1), (1, 3)])[_nhwc.TERMS]}, 3> + conv1x1_kernel + conv_nhwc8_kernel + conv_wino4_kernel + conv_wino_kernel (mrefsr_conv_nhwc_f32 / mrefsr_conv_dynagg_f32: every 3x3 / 1x1 convolution of the path)',
            achieved=round(nprod * (fl3 + fl1 + flw / 2.25) / (conv_ms * 1e-3) / 1e12, 1), peak=BF16_MATRIX_PEAK_TFLOPS, unit='TFLOP/s',
            frac=round(nprod * (fl3 + fl1 + flw / 2.25) / (conv_ms * 1e-3) / 1e12 / BF16_MATRIX_PEAK_TFLOPS, 4),
            executed_mfma_tflop_per_step=round(nprod * (fl3 + fl1 + flw / 2.25) / 1e12, 2), direct_equivalent_mfma_tflop_per_step=round(nprod * (fl3 + fl1 + flw) / 1e12, 2),
            direct_equivalent_frac=round(nprod * ach / BF16_MATRIX_PEAK_TFLOPS, 4),
            winograd=dict(launches=nw, ms_per_step=round(msw, 2), direct_tflop_per_step=round(flw / 1e12, 2),
                          note='conv_wino4_kernel (whole 16 x 16 tiles and cout blocks) / conv_wino_kernel (the rest): F(2x2, 3x3), 2.25x fewer MFMAs per output; the layer shapes they take: archs/nhwc.wino_applies'),
            fp32_equivalent_tflops=round(ach, 2), fp32_equivalent_speedup_vs_fp32_matrix_peak=round(ach / FP32_MATRIX_PEAK_TFLOPS, 3),
            traffic=conv_traffic, traffic_source=conv_traffic_src,
            algorithmic_bytes_per_step=int(sum(detail_bytes.get(k, 0.0) for k in ('conv_nhwc_k3', 'conv_nhwc_k1', 'conv_wino_k3'))),
            launches_per_step=n3 + n1 + nw, ms_per_step=round(conv_ms, 2), algorithmic_tflop_per_step=round((fl3 + fl1 + flw) / 1e12, 2),
            executed_mfma_dtype=exe_dtype, conv_terms=_nhwc.TERMS,
            note='achieved / frac = 16-bit MFMA FLOP EXECUTED (products per fp32-equivalent multiply x direct-convolution FLOPs '
                 '2*N*H*W*Cin*Cout*k*k, real channel counts) of all convolution launches of one step / their summed HIP-event time '
                 '(extra untimed step), against the 2.5 PF dense 16-bit matrix peak; fp32_equivalent_* = the same time priced as '
                 'fp32 convolution work (results are fp32-equivalent, DESIGN 3.3); zero-padded channels of Cin=3 / Cout=216,32,3 '
                 'layers are not counted as work.  Launches on the Winograd kernel execute 1 / 2.25 of their direct-convolution FLOP: '
                 'direct_equivalent_* prices them as direct convolutions (the figure comparable with earlier rounds).')
        if detail.get('dcn_fwd'):
            msd, nd, fld = detail['dcn_fwd']
            dcn_traffic = None
            try:
                if (args.batch, args.refs, args.lr, args.dtype) == (8, 5, 160, 'fp32'):
                    dfile = sorted(f for f in os.listdir(os.path.join(ROOT, 'profiles')) if f.endswith('_dcn_fwd_pmc.json'))[-1]
                    dcn_traffic = json.load(open(os.path.join(ROOT, 'profiles', dfile)))['traffic_bytes_per_step']
            except Exception:
                pass
            res['roofline_conv']['dcn_fwd'] = dict(ms_per_step=round(msd, 2), launches=nd, tflops=round(fld / (msd * 1e-3) / 1e12, 1),
                                                   traffic=dcn_traffic, algorithmic_bytes=sum(((2 * c + 216) * (640 * 640 // s_ ** 2) * 4 + 36 * c * c) * n_pair
                                                                                              for c, s_ in ((256, 4), (128, 2), (64, 1))) if args.lr == 160 else None,
                                                   peak=FP32_MATRIX_PEAK_TFLOPS, note=('fused gather + fp32 MFMA + bias + LeakyReLU (MREFSR_DCN_BF16=0)'
                                                         if os.environ.get('MREFSR_DCN_BF16') == '0' and args.dtype != 'bf16' else
                                                         ('fused gather + bf16 arithmetic MFMA + bias + LeakyReLU' if args.dtype == 'bf16' else
                                                          'fused gather + bf16 three-term split MFMA (fp32-equivalent, 6 products) + bias + LeakyReLU'
                                                          if os.environ.get('MREFSR_DCN_TERMS') == '6' else
                                                          'fused gather + fp16 two-term split MFMA (fp32-equivalent, 3 products) + bias + LeakyReLU')))
    if detail and detail.get('mrattn_fwd'):
        msa, na, bya = detail['mrattn_fwd']
        res['roofline_attn'] = dict(bound='hbm', kernel='mrattn_fwd_nhwc_kernel<C> (3 launches per step: C = 256 / 128 / 64)',
                                    achieved=round(bya / (msa * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                                    frac=round(bya / (msa * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), ms_per_step=round(msa, 3), launches=na,
                                    algorithmic_bytes_per_step=bya,
                                    note='algorithmic bytes (3K+3)*c*H*W*4 per sample and scale (SURVEY 8d) / summed HIP-event time')
    if train_fig is not None:
        res['train_step'] = train_fig
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=8, help='samples per GPU')
    ap.add_argument('--refs', type=int, default=5)
    ap.add_argument('--lr', type=int, default=160, help='LR side (output is 4x)')
    ap.add_argument('--mode', choices=('infer', 'train'), default='infer')
    ap.add_argument('--cpu-lr', type=int, default=160)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dtype', choices=('fp32', 'bf16'), default='fp32',
                    help='bf16: BASELINE configs[4] arithmetic (weights / activations rounded to bf16, fp32 accumulate)')
    ap.add_argument('--graph', action='store_true', help='replay the inference pass (MREFSR_GRAPH=1) / the training step (MREFSR_TRAIN_GRAPH=1) as hipGraphs: small shapes')
    ap.add_argument('--miopen-find', action='store_true', help='torch.backends.cudnn.benchmark = True')
    ap.add_argument('--no-train-step', action='store_true', help='skip the training-step figure (train_step on the JSON line) after the inference loop')
    ap.add_argument('--train-steps', type=int, default=10, help='timed training steps of the train_step figure')
    ap.add_argument('--no-gather', action='store_true', help='N > 1: leave the RCCL all_gather of the outputs out of the step (the line then carries only that figure)')
    args = ap.parse_args()
    spawn_ranks_if_needed(args)   # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (before any GPU call)
    if args.graph:
        os.environ['MREFSR_TRAIN_GRAPH' if args.mode == 'train' else 'MREFSR_GRAPH'] = '1'
    if args.dtype == 'bf16':
        from mrefsr_amd.archs import nhwc as _nh
        _nh.set_arithmetic('bf16')

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # MREFSR_BENCH_FORCE_DIST=1: take the multi-GPU code path (RCCL init, all_gather, barriers) with one rank too -- a way to
    # exercise it on a 1-GPU box under `python -m torch.distributed.run --nproc-per-node 1`
    dist_on = world > 1 or (os.environ.get('MREFSR_BENCH_FORCE_DIST') == '1' and 'RANK' in os.environ)
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run'
    torch.cuda.set_device(local_rank)
    from mrefsr_amd import dist_util
    rccl_world = None
    if dist_on:
        dist_util.init_dist('pytorch', backend='nccl')  # RCCL over xGMI, one process per GPU
        ones = torch.ones(1, device='cuda')
        dist.all_reduce(ones)                            # the world size as the collective itself sees it (every rank contributes 1)
        rccl_world = dict(all_reduce_of_ones=int(ones.item()), get_world_size=dist.get_world_size(), backend=dist.get_backend())
    torch.backends.cudnn.benchmark = bool(args.miopen_find)

    from mrefsr_amd import hip
    model = build(args, dist_on)
    sds = seeded_weights(model)
    data = synth_batch(args.batch, args.refs, args.lr, seed=10 + rank)
    model.feed_data(data)  # inputs now resident in HBM
    torch.cuda.synchronize()
    gather = [torch.empty(args.batch, 3, 4 * args.lr, 4 * args.lr, device='cuda') for _ in range(world)] if dist_on else None

    pending = []   # (work handle, the tensor being gathered) of the collective in flight

    dbg_sync = os.environ.get('MREFSR_BENCH_SYNC_STEPS') == '1'   # debugging: fence and name every step on stderr
    with_gather = [not args.no_gather]

    def step(i):
        if args.mode == 'train':
            model.optimize_parameters(i + 1)
            if dbg_sync:
                torch.cuda.synchronize()
                st = getattr(model, '_tgraph', None) or {}
                print(f'[step {i + 1}] graph={"replay" if st.get("fb") is not None else "eager"} eager={st.get("eager")} changes={st.get("changes")}', file=sys.stderr, flush=True)
        else:
            model.test()
            model.check_numeric_range()   # fp16-split convolutions: 4-byte flag readback, part of the step
            if dist_on and with_gather[0]:  # BASELINE configs[3]: RCCL gather of the outputs, overlapped with the next batch's kernels
                pending[:] = [dist_util.gather_outputs(model.output, gather, async_op=True)[1], model.output]

    def timed_region(n_steps, first):
        """exactly n_steps steps between barrier + synchronize on both sides; -> (seconds (max over ranks), this rank's seconds,
        per-step milliseconds of this rank from HIP events recorded between the steps)"""
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(n_steps):
            step(first + i)
            marks[i + 1].record()
        if pending and pending[0] is not None:
            pending[0].wait()   # the last batch's gather belongs to the timed region
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0
        pending[:] = []
        return dist_util.max_over_ranks(mine), mine, [marks[i].elapsed_time(marks[i + 1]) for i in range(n_steps)]

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    hip.set_kernel_timing(True)
    with ClockSampler(local_rank) as clock:
        elapsed, elapsed_mine, step_ms = timed_region(args.steps, args.warmup)
    rank_elapsed = None
    if dist_on:
        t = torch.tensor([elapsed_mine], device='cuda', dtype=torch.float64)
        allt = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        rank_elapsed = [float(x.item()) for x in allt]
    corr_ms = hip.kernel_timings().get('corr_top1', [])
    hip.set_kernel_timing(False)
    detail, detail_bytes = None, {}
    if rank == 0 and args.mode == 'infer':
        # one extra, untimed pass with an event pair around every convolution / DCN launch (rank 0 only, so
        # WITHOUT the collective of step(): the other ranks are already past the timed region)
        hip.set_kernel_timing(True, detail=True)
        model.test()
        torch.cuda.synchronize()
        detail = {k: (sum(v), len(v), hip.kernel_work().get(k, 0.0)) for k, v in hip.kernel_timings().items()}
        detail_bytes = hip.kernel_bytes()
        hip.set_kernel_timing(False)

    no_gather_fig = None
    if dist_on and args.mode == 'infer' and with_gather[0]:
        # the same loop without the exchange (half the steps, at least 2: a reported companion figure, every rank takes part)
        with_gather[0] = False
        k2 = max(2, args.steps // 2)
        e2, _, s2 = timed_region(k2, args.warmup + args.steps)
        with_gather[0] = True
        no_gather_fig = dict(steps=k2, ms_per_step=round(e2 / k2 * 1e3, 2), value=round(world * args.batch * (4 * args.lr) ** 2 / 1e6 * k2 / e2, 4),
                             note='the timed loop again without the RCCL all_gather of the outputs (per-rank inference only)')
    train_fig = None
    if args.mode == 'infer' and not args.no_train_step and args.dtype == 'fp32':
        try:
            train_fig = train_step_figure(args, dist_on, rank)
        except Exception as e:   # a reported extra, never a reason to lose the headline line (a rank-local failure under DDP would
            train_fig = dict(ms_per_step=None, error=f'{type(e).__name__}: {e}')   # still stall its peers: RCCL's watchdog ends them)
    if rank == 0:
        res = assemble_line(args, world, elapsed, step_ms, rank_elapsed, corr_ms, detail, detail_bytes, clock.summary(), no_gather_fig, rccl_world, train_fig)
        if not args.no_cpu_baseline:   # (rank 0 at any N: the other ranks wait at the closing barrier)
            try:
                res['cpu_baseline'] = cpu_baseline(sds, args, model)
            except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU line
                res['cpu_baseline'] = dict(value=None, unit='Mpix/s', cores=os.cpu_count(), kind='port', sample=f'failed: {e}')
        print(json.dumps(res), flush=True)
    if dist_on:
        dist.barrier()   # (the peers of rank 0 wait here while it times the CPU baseline)
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
