#!/usr/bin/env python
"""bench.py — images/sec of the Faster R-CNN ResNet-50 train step (BASELINE.json metric) on N MI355X GPUs of one
node.

    python bench.py --gpus N --steps K --warmup W          # N > 1: re-executes itself under torch.distributed.run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = forward + loss + backward + (gradient all-reduce) + optimizer update of one batch of synthetic images
that are resident in HBM before the timed region.  Default workload = BASELINE configs[1] (`--workload frcnn_r50`):
Faster R-CNN ResNet-50 (no FPN), batch 2 per GPU, 1024x1024, 8 gt boxes/image, 80 classes, 49 152 anchors/image,
RPN minibatch 256, RCNN minibatch 256, fp32, random-init weights.  The other BASELINE configs are selectable with
`--workload {frcnn_vgg16, ssd300_b32, frcnn_r101, frcnn_r50_coco}` and print the same contract line.
Weak scaling: per-GPU batch fixed, gradients all-reduced over RCCL (bucketed, overlapped with the backward).

Prints ONE JSON line (rank 0) with the driver's contract plus
  roofline     — the convolution MFMA kernel class with the LARGEST time per step (over forward, backward-data and
                 backward-weight classes alike): FLOPs the launch executes / mean launch duration vs the dense MFMA
                 peak of the compute dtype.  Durations are HIP events recorded by the C library on the launch stream
                 directly around that kernel, in profiling steps run right after the timed region with the
                 weight-gradient / proposal side streams SERIALISED onto one stream (un-overlapped timing; the timed
                 region itself uses the production multi-stream schedule).  `whole_step` = algorithmic conv FLOPs of
                 the step / timed step time.
  cpu_baseline — the CPU oracle (oracle/, kind "port": the TF reference cannot run here) timed on this host's
                 cores on a bounded sample: 1 warm-up + median of 5 steps, and a 1-thread step beside it.
"""
import argparse
import json
import os
import socket
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md: dense matrix peaks
# bf16x3: fp32 arithmetic as six bf16 MFMA products per fp32 product (conv_half.h): 2500 / 6 fp32-equivalent TFLOP/s
PEAK_TFLOPS = {'f32': 157.3, 'f16': 2500.0, 'bf16': 2500.0, 'bf16x3': 2500.0 / 6}
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # name: model type, architecture, per-GPU batch, H, W, classes, gt boxes/image, BASELINE.json configs[] index
    'frcnn_r50': dict(model='fasterrcnn', arch='resnet_v1_50', batch=2, H=1024, W=1024, classes=80, G=8, cfg=1),
    'frcnn_vgg16': dict(model='fasterrcnn', arch='vgg_16', batch=1, H=600, W=800, classes=20, G=3, cfg=0),
    'ssd300_b32': dict(model='ssd', arch='truncated_vgg_16', batch=32, H=300, W=300, classes=20, G=4, cfg=2),
    'frcnn_r101': dict(model='fasterrcnn', arch='resnet_v1_101', batch=2, H=1024, W=1024, classes=80, G=8, cfg=3),
    'frcnn_r50_coco': dict(model='fasterrcnn', arch='resnet_v1_50', batch=2, H=800, W=1333, classes=80, G=8, cfg=4),
}


def synth_batch(B, H, W, G, num_classes, seed, device, lo=32, hi=512):
    g = torch.Generator().manual_seed(seed)
    images = (torch.rand((B, H, W, 3), generator=g) * 255.0).to(device)
    gt = torch.zeros((B, G, 5), dtype=torch.float32)
    hi = min(hi, min(H, W) // 2)
    for b in range(B):
        wh = torch.randint(lo, hi + 1, (G, 2), generator=g)
        x = (torch.rand((G,), generator=g) * (W - wh[:, 0]).float()).floor()
        y = (torch.rand((G,), generator=g) * (H - wh[:, 1]).float()).floor()
        gt[b, :, 0], gt[b, :, 1] = x, y
        gt[b, :, 2], gt[b, :, 3] = x + wh[:, 0] - 1, y + wh[:, 1] - 1
        gt[b, :, 4] = torch.randint(0, num_classes, (G,), generator=g).float()
    cnt = torch.full((B,), G, dtype=torch.int32)
    return images, (gt.to(device), cnt.to(device))


def condition_weights(model, arch):
    """Random-init stand-in for pretrained statistics (no checkpoint can be downloaded): with identity BatchNorm
    statistics raw 0..255 pixels and 16+ stacked residual adds drive the activations to O(1e3) and momentum-SGD
    diverges to NaN within 3 steps.  Only FROZEN statistics / the first layer are touched (ResNet conv1: pixel variance
    x fan-in gain, last BN of every bottleneck: 16 => residual branch x 1/4; VGG: conv1_1 weights / pixel std);
    architecture, shapes and work per step are unchanged, and the loss stays finite and decreases."""
    sd = model.state_dict()
    if arch.startswith('resnet'):
        sd['truncated_base_network/%s/conv1/BatchNorm/moving_variance' % arch].fill_(73.6 ** 2 * 2)
        for k in sd:
            if k.endswith('conv3/BatchNorm/moving_variance'):
                sd[k].fill_(16.0)
    elif arch == 'vgg_16':
        sd['truncated_base_network/vgg_16/conv1/conv1_1/weights'].mul_(1.0 / 73.6)
    model.load_state_dict(sd)
    return model


def build(wl, device, dtype='f32', half_storage=True):
    from luminoth_amd.models import get_model
    from luminoth_amd.utils.config import get_config
    if wl['model'] == 'ssd':
        cfg = get_config({'model': {'type': 'ssd', 'network': {'num_classes': wl['classes']}},
                          'train': {'seed': 0, 'debug': False}})
    else:
        bn = {'architecture': wl['arch']}
        if wl['arch'] == 'vgg_16':
            bn['fine_tune_from'] = 'conv3'     # the reference default "block2" is ResNet-only (raises for VGG there too)
        if dtype != 'f32':
            bn['compute_dtype'] = dtype
            # f16 / bf16: 16-bit activations, activation gradients and working weights in HBM for the ResNet trunk
            # (csrc/conv_hs.h); --fp32-storage keeps the round-2 path (fp32 tensors, operands rounded
            # on their way into LDS)
            if half_storage and dtype in ('f16', 'bf16') and wl['arch'].startswith('resnet_v1'):
                bn['storage_dtype'] = dtype
        cfg = get_config({'model': {'type': 'fasterrcnn', 'network': {'num_classes': wl['classes']},
                                    'base_network': bn}, 'train': {'seed': 0, 'debug': False}})
    model = get_model(wl['model'])(cfg, device=device)
    if wl['model'] != 'ssd':
        condition_weights(model, wl['arch'])
    return cfg, model


def inputs(wl, seed, device):
    if wl['model'] == 'ssd':      # O(1) inputs keep the random-init SSD finite (no BN, no mean subtraction)
        images, gts = synth_batch(wl['batch'], wl['H'], wl['W'], wl['G'], wl['classes'], seed, device, lo=30, hi=200)
        return images / 127.5 - 1.0, gts
    return synth_batch(wl['batch'], wl['H'], wl['W'], wl['G'], wl['classes'], seed, device)


def cpu_baseline(wl, sd, steps=5):
    """The CPU oracle (oracle/model.py: torch-CPU fp32 convs + numpy box / NMS stages — a port, the TF reference cannot
    run here) timed on this host for full train steps over one batch of the benchmark shape: 1 warm-up + median of
    `steps` steps on min(cores, 32) threads (the oracle's scaling peaks there on the 256-thread GPU hosts), then ONE
    step of one image on a single thread for a per-core figure."""
    if wl['model'] == 'ssd':
        return None
    from oracle.model import OracleFasterRCNN
    cores = min(os.cpu_count() or 1, 32)
    kw = {'fine_tune_from': 'conv3'} if wl['arch'] == 'vgg_16' else {}
    images, (gt, _) = synth_batch(wl['batch'], wl['H'], wl['W'], wl['G'], wl['classes'], 1234, 'cpu')
    ims = [images[i] for i in range(wl['batch'])]
    gts = [gt[i].numpy() for i in range(wl['batch'])]
    torch.set_num_threads(cores)
    oracle = OracleFasterRCNN(sd, arch=wl['arch'], num_classes=wl['classes'], seed=0, **kw)
    mom, times = None, []
    t_all = time.time()
    for i in range(steps + 1):
        t0 = time.time()
        _, _, mom = oracle.train_step(ims, gts, mom_state=mom)
        if i > 0:
            times.append(time.time() - t0)
    med = float(np.median(times))
    torch.set_num_threads(1)
    oracle1 = OracleFasterRCNN(sd, arch=wl['arch'], num_classes=wl['classes'], seed=0, **kw)
    t0 = time.time()
    oracle1.train_step(ims[:1], gts[:1])
    t1 = time.time() - t0
    torch.set_num_threads(cores)
    return {'value': wl['batch'] / med, 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'one_thread_value': 1.0 / t1,
            'sample': '%d-image batch %dx%d, full oracle train step (torch-CPU fp32 + numpy): 1 warm-up + median of %d '
                      'steps on %d threads (%.2fs/step), plus one 1-image step on 1 thread (%.1fs); %.0fs in all'
                      % (wl['batch'], wl['H'], wl['W'], steps, cores, med, t1, time.time() - t_all)}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/*_pmc_traffic.json, produced by
    scripts/gpu_evidence_profiles.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `bench.py --serial`; the
    counters cannot be read from inside the process).  (None, None) if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json')))
    if not files:
        return None, None
    for f in files[::-1]:           # newest first; a summary covers the kernels of the workload it was taken on
        try:
            k = json.load(open(f))['kernels'].get(kernel.replace(' ', ''))
        except Exception:
            continue
        if k:
            return k['hbm_bytes_per_launch'], os.path.relpath(f, ROOT)
    return None, None


def rocprof_avg_ms(kernel):
    """Average duration (ms) of `kernel` in the newest committed rocprofv3 kernel trace of the serialised profiling steps
    (profiles/*_bench_roofline_steps_kernel_stats.csv: `rocprofv3 --kernel-trace --stats -- python bench.py`, reduced by
    scripts/make_profile_summary.py) — printed beside the live HIP-event figure so that the two can be compared on the
    line itself.  (None, None) if absent."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_bench_roofline_steps_kernel_stats.csv')))
    want = kernel.replace(' ', '')
    for f in files[::-1]:
        try:
            for row in csv.DictReader(open(f)):
                name = (row.get('Name') or row.get('name') or row.get('kernel') or '').replace(' ', '')
                if name.startswith('void'):
                    name = name[4:]
                if name.split('(')[0] == want:
                    avg = row.get('AverageNs') or row.get('avg_ns') or row.get('Average')
                    if avg:
                        return float(avg) * 1e-6, os.path.relpath(f, ROOT)
        except Exception:
            continue
    return None, None


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # defaults: 60 timed steps after 15 warm-up steps (< 1 s of GPU time).  Since round 4 a step is replayed from a recorded
    # launch plan with one host call (the host is done enqueueing after ~1 ms of a 6.6 ms step), so the driver's cold
    # `--steps 20 --warmup 5` lands within 1 % of this warm run (profiles/r04_bench_line_cold_20_5.json); in round 3, with
    # ~280 Python -> ctypes launches per step, three consecutive processes at 5 + 20 steps gave 7.72, 7.53, 7.20 ms/step.
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--warmup', type=int, default=15)
    ap.add_argument('--workload', default='frcnn_r50', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=None, help='images per GPU (default: the workload\'s)')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'f16', 'bf16', 'bf16x3'],
                    help='convolution compute dtype (f32 = the parity dtype of north_star)')
    ap.add_argument('--fp32-storage', action='store_true',
                    help='with --dtype f16 / bf16: keep fp32 tensors in HBM (round-2 path) instead of the half-storage trunk')
    ap.add_argument('--serial', action='store_true',
                    help='run EVERY step on one stream (the schedule of the roofline profiling steps): the command '
                         'the rocprofv3 summaries under profiles/*_serial_* are taken from')
    ap.add_argument('--phases', type=int, default=0, metavar='N',
                    help='after the timed steps, run N more with HIP-event marks and report the mean un-profiled timeline '
                         'of the three streams ("phases" in the JSON line; Faster R-CNN workloads)')
    ap.add_argument('--no-lookahead', action='store_true',
                    help='do not tell the step which batch comes next (no cross-step prefetch of the frozen trunk prefix)')
    ap.add_argument('--alt', action='store_true',
                    help='also re-run the step with bf16x3 convolutions and report it as `alt_arithmetic` (opt-in since '
                         'round 3: its gain did not reproduce on the driver\'s box)')
    ap.add_argument('--no-alt', action='store_true', help='(accepted for compatibility; the alt run is opt-in now)')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip the BASELINE configs[4] leg (f16 half-storage step at 800x1333) the default run adds as '
                         '`other_configs.frcnn_r50_coco_f16`')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=5)
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # launched plainly with --gpus N: become N ranks (one process per GPU over RCCL)
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.execvp(sys.executable, [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
                                   '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
                                   '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:])
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU, or run `python bench.py '
                         '--gpus N` and let it spawn them)' % (args.gpus, world))
    if world > 1:
        # main, aux, weight-gradient side stream, the bucket stream and RCCL's own: more streams than the 4
        # hardware queues HIP creates by default (read at runtime initialisation, i.e. before the first cuda call)
        os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm GPU (the product path has no CPU fallback)')
    backend = os.environ.get('LUMINOTH_AMD_DIST_BACKEND', 'nccl')   # 'gloo': ranks may share a GPU (control-flow test)
    if backend != 'nccl':
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)

    from luminoth_amd import kernels as K
    from luminoth_amd import plan as P
    from luminoth_amd.models.base import layers as L
    from luminoth_amd.utils import training as T

    T.issue_from_high_priority_stream(device)      # what luminoth_amd.train.run does: critical path first at the dispatcher

    side_default = L.SideStream.enabled        # LUMINOTH_AMD_SIDE_STREAM=0: weight gradients on the issuing stream

    def serialise(on):
        T.FUSED_STEP = not on
        L.SideStream.enabled = side_default and not on

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_workload(name, dtype, steps, warmup, batch=None, want_roofline=True, phases_n=0, keep_sd=False):
        """Build the workload's model, run `warmup` untimed + `steps` timed train steps (barrier + synchronize on both
        sides of the timed block), then the optional diagnostic legs.  -> dict of measurements."""
        wl = dict(WORKLOADS[name])
        if batch:
            wl['batch'] = batch
        cfg, model = build(wl, device, dtype, half_storage=not args.fp32_storage)
        T.broadcast_parameters(model)
        sd0 = model.state_dict() if keep_sd else None
        opt = T.get_optimizer(cfg.train, model)
        images, gts = inputs(wl, 100 + rank, device)
        # a second synthetic batch: steps alternate between the two like a data loader handing over batch after batch,
        # and each step is told which batch comes next (the look-ahead luminoth_amd.train.run gives the model)
        batches = [(images, gts), inputs(wl, 1000 + rank, device)]
        counter = [0]

        def step_fn():
            i = counter[0]
            counter[0] += 1
            cur, nxt = batches[i % 2], batches[(i + 1) % 2]
            if args.no_lookahead:
                return T.train_step(model, opt, cur[0], cur[1])
            return T.train_step(model, opt, cur[0], cur[1], next_image=nxt[0], next_gt=nxt[1])

        serialise(args.serial)
        for _ in range(warmup):
            step_fn()
        sync()
        # per-step times: one HIP event per step boundary on the issuing stream (the MEDIAN step time is
        # reported beside the block mean; the events cost ~2 us of host time each and no GPU time)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(steps):
            total, _ = step_fn()
            marks[i + 1].record()
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0])
        per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
        loss_val = float(total.detach())
        assert np.isfinite(loss_val), 'train step diverged (loss %r)' % loss_val
        res = {'wl': wl, 'dt': dt, 'steps': steps, 'warmup': warmup, 'loss': loss_val, 'model': model, 'sd0': sd0,
               'ms_per_step': dt / steps * 1e3, 'ms_per_step_median': float(np.median(per_step)),
               'ms_per_step_min': float(np.min(per_step)), 'ms_per_step_max': float(np.max(per_step))}
        plans = [pl for S in getattr(model, '_step_state', {}).values() for pl in S['plans'].values()]
        res['launch_plan'] = {'enabled': bool(P.ENABLED and plans), 'plans': len(plans),
                              'kernel_launches_per_step': max([pl.n_kernels for pl in plans] or [0]),
                              'plan_nodes_per_step': max([pl.n_nodes for pl in plans] or [0])}

        if phases_n > 0 and hasattr(model, 'record_phases') and not args.serial:
            # timeline marks (library events inside the step, part of the launch plan): the first steps after arming
            # re-record the plans with the marks in them and are not counted
            model.record_phases(phases_n + 6)
            for _ in range(6):
                step_fn()
            torch.cuda.synchronize()
            model._phase_sum, model._phase_n, model._phase_next_n = {}, 0, 0
            for S in model._step_state.values():
                S.get('pending', {}).clear()
            for _ in range(phases_n):
                step_fn()
            res['phases'] = {k: round(v, 3) for k, v in sorted(model.phase_times().items(), key=lambda kv: kv[1])}

        nprof = min(steps, 3)
        if want_roofline:
            # every rank takes the profiling steps (they contain the gradient all-reduce); rank 0 records
            serialise(True)
            T.train_step(model, opt, images, gts)          # settle allocations of the serial schedule
            torch.cuda.synchronize()
            if rank == 0:
                K._Profile.start()
            for _ in range(nprof):
                T.train_step(model, opt, images, gts)
            prof = K._Profile.stop() if rank == 0 else None
            serialise(args.serial)
            if rank == 0 and prof:
                res['roofline'] = roofline_of(prof, nprof, dtype, dt, steps,
                                              rocprof_ok=(name == 'frcnn_r50' and dtype == 'f32' and not batch))
        if world > 1:
            # data-parallel sanity: after the same number of identical updates every replica must hold the SAME bits
            # (seeded init + broadcast, ring all-reduce hands every rank the same sums, one update kernel)
            torch.cuda.synchronize()
            flat = model.store.flat
            chk = torch.stack([flat.view(torch.int32).to(torch.int64).sum(), flat.double().abs().sum().view(torch.int64)])
            if dist.get_backend() != 'nccl':
                chk = chk.cpu()          # gloo gathers host tensors
            allc = [torch.zeros_like(chk) for _ in range(world)]
            dist.all_gather(allc, chk)
            same = all(bool(torch.equal(c, allc[0])) for c in allc)
            res['replicas_identical'] = same
            if not same and rank == 0:
                # reported on the line (`dist.replicas_identical_after_timed_steps`: false) instead of raised: a diverged
                # replica set makes the number suspect, but a traceback here would leave the scaling record empty
                sys.stderr.write('bench.py: WARNING data-parallel replicas diverged: parameter checksums %r\n'
                                 % [c.tolist() for c in allc])
            dist.barrier()
        return res

    def roofline_of(prof, nprof, dtype, dt, steps, rocprof_ok=False):
        peak = PEAK_TFLOPS[dtype]
        name = max(prof, key=lambda k: prof[k]['ms'])          # the dominant kernel class, whichever pass it is in
        step_flops = sum(v['direct_flops'] for v in prof.values()) / nprof
        exec_flops = sum(v['flops'] for v in prof.values()) / nprof
        r = prof[name]
        fl = r['flops'] / r['launches']
        by = r['bytes'] / r['launches']
        ms = r['ms'] / r['launches']
        ms_raw = r['ms_raw'] / r['launches']
        achieved = fl / (ms * 1e-3) / 1e12
        traffic, traffic_src = pmc_traffic(name)
        # the committed kernel trace is the one of the DEFAULT line (frcnn_r50, fp32, default batch): the same kernel name on
        # another workload runs other layer shapes, so no rocprofv3 figure is quoted there
        rp_ms, rp_src = rocprof_avg_ms(name) if rocprof_ok else (None, None)
        # which roofline bounds this kernel: the larger of its two ideal times (fp32 convolutions are always
        # matrix-bound; with f16 / bf16 operands the tensors in HBM become the limit on the thin layers)
        t_mfma, t_hbm = fl / (peak * 1e12), by / (PEAK_HBM_GBS * 1e9)
        if t_hbm > t_mfma:
            unit, pk, per_launch = 'GB/s', PEAK_HBM_GBS, by / 1e9
            bound = {'bound': 'hbm', 'mfma_tflops': achieved}
        else:
            unit, pk, per_launch = 'TFLOP/s', peak, fl / 1e12
            bound = {'bound': 'mfma'}
        bound.update({'achieved': per_launch / (ms * 1e-3), 'peak': pk, 'unit': unit, 'frac': per_launch / (ms * 1e-3) / pk,
                      # the same figure from the event interval as measured (no calibration: a lower bound) and from the
                      # committed rocprofv3 summary of this command (kernel begin / end timestamps)
                      'frac_raw_event_interval': per_launch / (ms_raw * 1e-3) / pk,
                      'frac_rocprofv3': (per_launch / (rp_ms * 1e-3) / pk) if rp_ms else None,
                      'rocprofv3_avg_ms': rp_ms, 'rocprofv3_source': rp_src})
        return dict(bound, **{
            'kernel': name, 'traffic': traffic, 'algorithmic_bytes_per_launch': by,
            'traffic_unit': 'HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE PMC passes)',
            'traffic_source': traffic_src,
            'launches_per_step': r['launches'] / nprof, 'flops_per_launch': fl, 'ms_per_launch': ms,
            'ms_per_launch_raw_event_interval': ms_raw,
            'event_pair_overhead_ms': K._Profile.overhead_ms,
            'ms_per_step': r['ms'] / nprof,
            'timing': 'HIP events around the kernel on its launch stream, %d serialised profiling steps; `frac` = minus the '
                      'interval of an event pair around an EMPTY kernel (dispatch latency, calibrated in this process), '
                      '`frac_raw_event_interval` = as measured, `frac_rocprofv3` = from the committed kernel trace' % nprof,
            # conv_flops = the ALGORITHMIC count of SURVEY.md 8(d) (direct convolution); executed_flops = what the
            # launched kernels actually multiply (Winograd F(4x4,3x3) layers do 4x fewer): the second is the honest
            # measure of how busy the matrix pipe is, the first of how fast the step's defined work gets done
            'whole_step': {'conv_flops': step_flops, 'tflops': step_flops / dt * steps / 1e12,
                           'frac': step_flops / dt * steps / 1e12 / peak,
                           'executed_flops': exec_flops,
                           'executed_tflops': exec_flops / dt * steps / 1e12,
                           'executed_frac': exec_flops / dt * steps / 1e12 / peak,
                           'conv_kernel_ms_per_step': sum(v['ms'] for v in prof.values()) / nprof,
                           'executed_frac_of_conv_kernel_time':
                               exec_flops / (sum(v['ms'] for v in prof.values()) / nprof * 1e-3) / 1e12 / peak},
            'all_conv_kernels': {k: {'launches_per_step': v['launches'] / nprof,
                                     'tflops': v['flops'] / (v['ms'] * 1e-3) / 1e12,
                                     'gbs': v['bytes'] / (v['ms'] * 1e-3) / 1e9,
                                     'ms_per_step': v['ms'] / nprof}
                                 for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])}})

    head = run_workload(args.workload, args.dtype, args.steps, args.warmup, batch=args.batch,
                        want_roofline=not args.no_roofline, phases_n=args.phases,
                        keep_sd=(rank == 0 and world == 1 and not args.no_cpu_baseline))
    wl, dt, model = head['wl'], head['dt'], head['model']
    plan_on = head['launch_plan']['enabled']
    schedule = ('three streams; ' + ('recorded launch plan replayed with one host call per step (%d kernel launches per step, '
                                      'identical to the eager step)' % head['launch_plan']['kernel_launches_per_step']
                                      if plan_on else 'eager launches') +
                ('' if args.no_lookahead or args.serial else
                 '; the frozen trunk prefix (conv1 + block1) and the anchor targets of the NEXT batch are computed in idle '
                 'slots of the step (main stream waiting for the RCNN branch / idle proposal stream): one prefix and one '
                 'target pass per step, every step'))

    if rank == 0:
        gb = wl['batch'] * world
        metric = 'images/sec (1024x1024) Faster R-CNN ResNet-50 train step'
        if args.workload != 'frcnn_r50':
            metric = 'images/sec (%dx%d) %s %s train step' % (wl['W'], wl['H'], wl['model'], wl['arch'])
        out = {
            'metric': metric,
            'value': gb * args.steps / dt, 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'ms_per_step_median': head['ms_per_step_median'], 'ms_per_step_min': head['ms_per_step_min'],
            'ms_per_step_max': head['ms_per_step_max'], 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[%d] (%s): %s %s, %dx%d (HxW) synthetic, batch %d/GPU, %d classes, '
                                   '%d gt/image, fwd+loss+bwd+optimizer update%s'
                                   % (wl['cfg'], args.workload, wl['model'], wl['arch'], wl['H'], wl['W'], wl['batch'],
                                      wl['classes'], wl['G'], ' [single-stream schedule]' if args.serial else ''),
                       'global_batch': gb, 'parallelism': 'dp%d' % world, 'final_total_loss': head['loss'],
                       'storage': (getattr(getattr(model, 'base_network', None), 'storage_dtype', None) or 'f32') +
                                  (' trunk activations / gradients / working weights in HBM, fp32 master weights'
                                   if getattr(getattr(model, 'base_network', None), 'storage_dtype', None) else ''),
                       'schedule': schedule, 'launch_plan': head['launch_plan']},
            'roofline': head.get('roofline'),
            # what the collective layer saw (SCALE_rNN.json can show that RCCL ran with N ranks)
            'dist': {'world_size': dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1,
                     'backend': dist.get_backend() if (dist.is_available() and dist.is_initialized()) else None,
                     'rccl_version': '.'.join(str(v) for v in torch.cuda.nccl.version())
                     if hasattr(torch.cuda, 'nccl') else None,
                     'GPU_MAX_HW_QUEUES': os.environ.get('GPU_MAX_HW_QUEUES'),
                     'streams': 'issue (high priority), proposal/RCNN (high priority), weight-gradient x2' +
                                (', gradient-bucket, RCCL internal' if world > 1 else ''),
                     'buckets': getattr(T.ACTIVE_BUCKETS, 'describe', lambda: None)(),
                     'replicas_identical_after_timed_steps': head.get('replicas_identical')},
        }
        if head.get('phases'):
            out['phases_ms'] = head['phases']
    sd0 = head['sd0']
    del head, model
    other = {}
    if args.workload == 'frcnn_r50' and args.dtype == 'f32' and not args.serial and not args.no_other_configs:
        # BASELINE configs[4] ("fp16 MFMA path, 1333x800 COCO shapes") on the same record: the half-storage trunk at its own
        # geometry, 15 + 60 steps (~0.4 s of GPU time), with its own roofline leg.  Every rank runs it (it contains the
        # gradient exchange); reported BESIDE `value`, never as it.
        torch.cuda.empty_cache()
        o = run_workload('frcnn_r50_coco', 'f16', 60, 15, want_roofline=not args.no_roofline)
        if rank == 0:
            owl = o['wl']
            other['frcnn_r50_coco_f16'] = {
                'config': 'BASELINE configs[4] (frcnn_r50_coco): fasterrcnn resnet_v1_50, %dx%d (HxW) synthetic, batch %d/GPU, '
                          'f16 MFMA operands, 16-bit trunk activations / gradients / working weights in HBM, fp32 master '
                          'weights and accumulation' % (owl['H'], owl['W'], owl['batch']),
                'value': owl['batch'] * world * o['steps'] / o['dt'], 'unit': 'images/sec', 'dtype': 'f16',
                'ms_per_step': o['ms_per_step'], 'ms_per_step_median': o['ms_per_step_median'], 'steps': o['steps'],
                'warmup': o['warmup'], 'final_total_loss': o['loss'], 'launch_plan': o['launch_plan'],
                'roofline': {k: v for k, v in (o.get('roofline') or {}).items() if k != 'all_conv_kernels'} or None}
        del o
    if rank == 0:
        if other:
            out['other_configs'] = other
        if args.dtype == 'f32' and world == 1 and args.alt and wl['model'] != 'ssd' and not args.serial:
            # the same step with the convolutions in bf16x3 (fp32 arithmetic on the bf16 matrix pipe),
            # measured in this process right after the headline run: reported BESIDE `value`, never as it
            a = run_workload(args.workload, 'bf16x3', args.steps, args.warmup, batch=args.batch, want_roofline=False)
            out['alt_arithmetic'] = {
                'dtype': 'bf16x3', 'value': gb * args.steps / a['dt'], 'unit': 'images/sec',
                'ms_per_step': a['ms_per_step'], 'steps': args.steps, 'warmup': args.warmup,
                'final_total_loss': a['loss'],
                'note': 'same workload, schedule, tensors and tolerances; every fp32 convolution operand split exactly into '
                        'three bf16 pieces, six v_mfma_f32_32x32x16_bf16 per fp32 product, fp32 accumulate (bit-exact with the '
                        'native kernels on integer data, same error against float64: tests/test_gpu_x3.py).  Not the headline: '
                        '`value` above is the native fp32-MFMA path.'}
            del a
        if not args.no_cpu_baseline and world == 1:          # reported on rank 0 at N = 1 only
            cb = cpu_baseline(wl, sd0, args.cpu_steps)
            if cb is not None:
                out['cpu_baseline'] = cb
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
