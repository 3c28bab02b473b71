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

Arithmetic of the headline (round 6).  `value` is measured with the trunk / RPN convolutions in bf16x3:
every fp32 operand split EXACTLY into three bf16 pieces, six v_mfma_f32_32x32x16_bf16 products per fp32 product, fp32
accumulation (csrc/conv_x3.h) — fp32 arithmetic on the bf16 matrix pipe: bit-identical to the native fp32-MFMA kernels on
representable data, the same error against float64, and every parity test of the fp32 contract passes under the SAME bounds
(tests/test_gpu_model.py[bf16x3], tests/test_gpu_ref_tf_golden.py[bf16x3], tests/test_gpu_x3.py).  The line says so in
`dtype` ("f32 (bf16x3 exact split)"), prices `roofline.frac` against 2500 / 6 = 417 TFLOP/s of fp32-equivalent work and
carries the SAME step on the native fp32 MFMA, measured in the same process, beside it (`native_fp32_mfma`).
`--dtype f32` makes the native path the headline again.

Prints ONE JSON line (rank 0) with the driver's contract plus
  roofline     — the convolution MFMA kernel class with the LARGEST time per step (over forward, backward-data and
                 backward-weight classes alike): FLOPs (or, for a kernel whose HBM time exceeds its MFMA time, algorithmic
                 bytes) of one launch / its mean launch duration vs the peak.  The launch duration comes from profiling
                 steps run right after the timed region with the weight-gradient / proposal side streams SERIALISED onto
                 one stream (un-overlapped timing; the timed region itself uses the production multi-stream schedule),
                 measured twice: (1) HIP events recorded by the C library on the launch stream directly around the kernel
                 (`frac_raw_event_interval`), (2) at N = 1, the same steps in a child process under
                 `rocprofv3 --kernel-trace --stats` ON THIS BOX IN THIS RUN (`frac_rocprofv3`, kernel begin / end
                 timestamps).  `frac` = (2) when the profiler ran, else (1); `frac_source` says which.  Nothing on the line
                 is read from a committed profile except `traffic` (PMC bytes need separate `--pmc` passes:
                 scripts/r5_evidence.sh -> profiles/r05_pmc.json, `traffic_source`).
                 `whole_step` = algorithmic conv FLOPs of the step / timed step time.
  roofline.step — the whole step as tracked numbers: kernel launches per step (the replayed launch plan), HBM bytes per
                 step by the memory-side counters (the newest committed PMC summary of this workload / dtype:
                 profiles/*_pmc_traffic.json `step`, scripts/pmc_reduce.py) against the algorithmic bytes of SURVEY.md 8(d)
                 (tools/flops.py: resnet_frcnn_step_bytes), and their ratio.
  cpu_baseline — the CPU oracle (oracle/, kind "port": the TF reference cannot run here) timed on this host's
                 cores on a bounded sample: 1 warm-up + median of 5 steps, and a 1-thread step beside it.
  dist         — N > 1: which exchange mode produced `value` (`mode`), the modes that failed before it (`fallbacks`: the
                 ladder bucketed all-reduce + launch plan -> one all-reduce + launch plan -> one all-reduce, eager
                 launches), replica bit-identity after the timed steps, the collective timeout.
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
DTYPE_LABEL = {'f32': 'f32', 'f16': 'f16', 'bf16': 'bf16', 'bf16x3': 'f32 (bf16x3 exact split)'}

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


def pmc_step_traffic(workload, dtype):
    """HBM bytes per STEP by the memory-side counters, from the newest committed PMC summary taken on this workload and
    dtype (profiles/*_pmc_traffic.json: `workload`, `dtype`, `step.hbm_bytes`; scripts/pmc_reduce.py).  (None, None) if
    there is none."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json')))[::-1]:
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get('workload') == workload and d.get('dtype') == dtype and (d.get('step') or {}).get('hbm_bytes'):
            return d['step'], os.path.relpath(f, ROOT)
    return None, None


def algorithmic_step_bytes(wl, dtype, model):
    """SURVEY.md 8(d)'s compulsory HBM bytes of one train step (tools/flops.py); ResNet Faster R-CNN workloads only."""
    if wl['model'] != 'fasterrcnn' or not wl['arch'].startswith('resnet_v1'):
        return None
    from tools import flops
    half = getattr(getattr(model, 'base_network', None), 'storage_dtype', None) in ('f16', 'bf16')
    return flops.resnet_frcnn_step_bytes(wl['arch'], wl['H'], wl['W'], wl['batch'], int(model.store.flat.numel()),
                                         2 if half else 4)['step']


def _short_kernel_name(name):
    name = name.replace('void ', '')
    i = name.find('(')
    return (name[:i] if i > 0 else name).replace(' ', '')


def reduce_kernel_trace(path, nprof):
    """rocprofv3 kernel_trace.csv -> {kernel: {'avg_ms', 'calls_per_step'}, '_step_ms': ...} over the dispatches of the LAST
    `nprof` steps.  A step ends with its optimizer launch (k_sgd_momentum / k_optimizer): model construction, weight
    upload and the settling steps in front are excluded.  None if the trace holds fewer than nprof + 1 optimizer launches."""
    import csv
    rows = []
    with open(path) as f:
        for row in csv.DictReader(f):
            rows.append((int(row['Start_Timestamp']), int(row['End_Timestamp']), _short_kernel_name(row['Kernel_Name'])))
    rows.sort()
    is_opt = lambda n: 'k_sgd_momentum' in n or 'k_optimizer' in n or 'k_sgd_early' in n      # noqa: E731
    # a step ends with its LAST optimizer launch: with per-range updates under the backward (LUMINOTH_AMD_EARLY_UPDATE=1)
    # a step issues several; an optimizer launch followed (before the next step's kernels) by another one is not an end
    ends = []
    for i, row in enumerate(rows):
        if is_opt(row[2]) and 'early' not in row[2]:
            if ends and all(is_opt(r[2]) for r in rows[ends[-1] + 1:i]):
                ends[-1] = i
            else:
                ends.append(i)
    if len(ends) < nprof + 1:
        return None
    win = rows[ends[-nprof - 1] + 1:ends[-1] + 1]
    agg = {}
    for s0, e0, n in win:
        a = agg.setdefault(n, [0, 0])
        a[0] += e0 - s0
        a[1] += 1
    out = {n: {'avg_ms': t / float(c) * 1e-6, 'calls_per_step': c / float(nprof)} for n, (t, c) in agg.items()}
    out['_step_ms'] = (win[-1][1] - rows[ends[-nprof - 1]][1]) / float(nprof) * 1e-6
    return out


def rocprof_child(workload, dtype, batch=None, fp32_storage=False, nprof=3, keep_dir=None, timeout=150):
    """The rocprofv3 figure of the roofline, measured LIVE on this box: runs
        rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --roofline-child --workload W --dtype D
    as a child process (the same serialised profiling steps the HIP-event leg times: 2 settling + `nprof` recorded steps on
    one stream), reads its kernel trace and returns {kernel: {'avg_ms', 'calls_per_step'}} over the dispatches of the LAST
    `nprof` steps (optimizer launch to optimizer launch), plus '_step_ms' and '_cmd'.  None (with the reason on stderr) if
    rocprofv3 is missing, fails or times out: the line then says `frac_source: hip_events`.
    `keep_dir`: copy the raw kernel trace / stats CSVs there (scripts/r5_evidence.sh turns them into profiles/r05_*)."""
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        sys.stderr.write('bench.py: rocprofv3 not found; roofline from HIP events only\n')
        return None
    try:
        tmp = tempfile.mkdtemp(prefix='lmh_rocprof_', dir='/tmp' if os.access('/tmp', os.W_OK) else None)
    except Exception as e:
        sys.stderr.write('bench.py: no scratch directory for the rocprofv3 child (%r); roofline from HIP events only\n' % (e,))
        return None
    child = [sys.executable, os.path.abspath(__file__), '--roofline-child', '--workload', workload, '--dtype', dtype,
             '--roofline-steps', str(nprof)]
    if batch:
        child += ['--batch', str(batch)]
    if fp32_storage:
        child += ['--fp32-storage']
    cmd = [exe, '--kernel-trace', '--stats', '--output-format', 'csv', '-d', tmp, '-o', 'rp', '--'] + child
    env = dict(os.environ, TMPDIR='/tmp')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, cwd=tmp, env=dict(env, TMPDIR='/tmp' if os.access('/tmp', os.W_OK) else tmp), capture_output=True, text=True, timeout=timeout)
        if r.returncode != 0:
            sys.stderr.write('bench.py: rocprofv3 child rc %d: %s\n' % (r.returncode, r.stderr[-600:]))
            return None
        traces = glob.glob(os.path.join(tmp, '**', '*kernel_trace.csv'), recursive=True)
        if not traces:
            sys.stderr.write('bench.py: rocprofv3 child wrote no kernel trace\n')
            return None
        out = reduce_kernel_trace(traces[0], nprof)
        if out is None:
            sys.stderr.write('bench.py: rocprofv3 child trace holds fewer than %d optimizer launches\n' % (nprof + 1))
            return None
        out['_cmd'] = 'rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py ' + ' '.join(child[2:])
        if keep_dir:
            os.makedirs(keep_dir, exist_ok=True)
            for f in glob.glob(os.path.join(tmp, '**', '*.csv'), recursive=True):
                if f.endswith('kernel_trace.csv') or f.endswith('kernel_stats.csv'):
                    shutil.copy(f, os.path.join(keep_dir, os.path.basename(f)))
        return out
    except Exception as e:          # a profiler problem must never cost the bench line
        sys.stderr.write('bench.py: rocprofv3 child failed: %r\n' % (e,))
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


class StepFailure(RuntimeError):
    """The timed block of one exchange mode failed on some rank (agreed on by all ranks)."""


# N > 1: the exchange modes, most overlapped first (environment each one adds on top of the previous)
LADDER = [{}, {'LUMINOTH_AMD_BUCKETED_ALLREDUCE': '0'}, {'LUMINOTH_AMD_BUCKETED_ALLREDUCE': '0', 'LUMINOTH_AMD_PLAN': '0'}]


def run_ladder(world, attempt, environ=None):
    """First-contact safety net of the multi-GPU run.  `attempt(mode)` runs the timed block in exchange mode `mode`
    ({'bucketed_allreduce_under_backward', 'launch_plan'}) and returns its result dict, or raises StepFailure.  A mode that
    raises, or that leaves the replicas with different bits (`replicas_identical` False) while a simpler mode is left, is
    recorded and the next one runs.  -> (result, mode, fallbacks); result None when every mode failed.  N = 1: one mode."""
    environ = os.environ if environ is None else environ
    ladder = LADDER if world > 1 else LADDER[:1]
    fallbacks = []
    for li, env in enumerate(ladder):
        environ.update(env)
        mode = {'bucketed_allreduce_under_backward': world > 1 and environ.get('LUMINOTH_AMD_BUCKETED_ALLREDUCE', '1') != '0',
                'launch_plan': environ.get('LUMINOTH_AMD_PLAN', '1') != '0'}
        try:
            res = attempt(mode)
        except StepFailure as e:
            fallbacks.append(dict(mode, error=str(e)))
            continue
        if world > 1 and res.get('replicas_identical') is False and li + 1 < len(ladder):
            fallbacks.append(dict(mode, error='replicas diverged after the timed steps'))
            continue
        return res, mode, fallbacks
    return None, None, fallbacks


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
    ap.add_argument('--dtype', default='auto', choices=['auto', 'f32', 'f16', 'bf16', 'bf16x3'],
                    help='convolution arithmetic.  auto (default) = bf16x3 for the Faster R-CNN workloads (fp32 arithmetic as an '
                         'exact three-way bf16 split, the headline since round 6) and f32 for SSD; f32 = the native fp32 MFMA')
    ap.add_argument('--no-native', action='store_true',
                    help='with bf16x3: do not re-run the step on the native fp32 MFMA for `native_fp32_mfma`')
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
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip the BASELINE configs[4] leg (f16 half-storage step at 800x1333) the default run adds as '
                         '`other_configs.frcnn_r50_coco_f16`')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=5)
    ap.add_argument('--no-rocprof', action='store_true',
                    help='do not run the rocprofv3 child processes; `roofline.frac` then comes from the HIP-event intervals')
    ap.add_argument('--rocprof-keep', default=None, metavar='DIR',
                    help='keep the raw kernel trace / stats CSVs of the rocprofv3 children under DIR/<workload>_<dtype>/')
    ap.add_argument('--roofline-child', action='store_true',
                    help='(internal) only the serialised roofline profiling steps of --workload / --dtype: the command the '
                         'parent process runs under rocprofv3')
    ap.add_argument('--roofline-steps', type=int, default=3, help='serialised profiling steps of the roofline leg')
    ap.add_argument('--collective-timeout', type=int, default=int(os.environ.get('LUMINOTH_AMD_COLLECTIVE_TIMEOUT', '180')),
                    help='seconds after which a collective that does not complete aborts the rank (N > 1): a hang becomes '
                         'a non-zero exit code instead of an empty record')
    ap.add_argument('--inject-bucket-failure', action='store_true',
                    help='(test hook) the bucketed gradient exchange raises on its first early bucket: exercises the '
                         'fall-back ladder bucketed+plan -> one all-reduce+plan -> one all-reduce, eager launches')
    args = ap.parse_args()
    if args.dtype == 'auto':
        args.dtype = 'f32' if WORKLOADS[args.workload]['model'] == 'ssd' else 'bf16x3'

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
    if world > 1 or 'WORLD_SIZE' in os.environ:
        # main, aux, weight-gradient side stream, the bucket stream and RCCL's own: more streams than the 4
        # hardware queues HIP creates by default (read at runtime initialisation, i.e. before the first cuda call).
        # Every process started by a launcher (torch.distributed.run sets WORLD_SIZE, also for ONE rank) gets the same
        # mapping, so the N = 1 point of a scaling run and its N > 1 points differ in the exchange only; the plain
        # `python bench.py` headline keeps HIP's default (A/B of the two at N = 1: DESIGN.md 8).
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
        import datetime
        tmo = datetime.timedelta(seconds=max(30, args.collective_timeout))
        # the watchdog of the NCCL (= RCCL) backend tears the process down when a collective exceeds `tmo`
        os.environ.setdefault('TORCH_NCCL_ASYNC_ERROR_HANDLING', '1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)

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

    if world > 1:
        # last resort: a hang that no collective timeout sees (a stream waiting for an event that never fires) still ends
        # the process with a traceback and a non-zero exit code instead of leaving the scaling record empty
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get('LUMINOTH_AMD_BENCH_WATCHDOG', '1500')), exit=True)

    if args.inject_bucket_failure:
        _orig_launch = T.GradientBuckets._launch

        def _failing_launch(self, lo, hi):
            raise RuntimeError('injected failure of the bucketed gradient exchange (--inject-bucket-failure)')
        T.GradientBuckets._launch = _failing_launch

    if args.roofline_child:
        # what the parent runs under rocprofv3: the serialised profiling steps only (same calls as the HIP-event leg)
        wl = dict(WORKLOADS[args.workload])
        if args.batch:
            wl['batch'] = args.batch
        cfg, model = build(wl, device, args.dtype, half_storage=not args.fp32_storage)
        opt = T.get_optimizer(cfg.train, model)
        images, gts = inputs(wl, 100 + rank, device)
        serialise(True)
        for _ in range(2 + args.roofline_steps):
            T.train_step(model, opt, images, gts)
        torch.cuda.synchronize()
        print(json.dumps({'roofline_child': 'ok', 'workload': args.workload, 'dtype': args.dtype,
                          'steps': args.roofline_steps}))
        return

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
        failure = None
        try:
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
        except Exception as e:      # N > 1: the ladder in main() retries with a simpler exchange; N = 1: re-raised below
            if world == 1:
                raise
            import traceback
            failure = '%s: %s' % (type(e).__name__, e)
            sys.stderr.write('bench.py: rank %d: timed block failed:\n%s' % (rank, traceback.format_exc()))
        if world > 1:
            # every rank learns whether ANY rank failed (a deterministic failure hits all ranks at the same call, so this
            # collective pairs up; a one-sided failure ends in the collective timeout and a non-zero exit code)
            flag = torch.tensor([1.0 if failure else 0.0, dt if not failure else 0.0], dtype=torch.float64)
            flag = flag.to(device) if dist.get_backend() == 'nccl' else flag
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if float(flag[0]) > 0:
                K.TAILS.abort()
                T.install_buckets(None)
                del model, opt
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
                raise StepFailure(failure or 'another rank failed')
            dt = float(flag[1])
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

        nprof = max(1, min(steps, args.roofline_steps))
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
                # ... and the SAME steps once more in a child process under rocprofv3 (kernel begin / end timestamps of this
                # very box and run: what `frac` is computed from; N = 1 only, like the CPU baseline)
                rp = None
                if world == 1 and not args.no_rocprof:
                    rp = rocprof_child(name, dtype, batch=batch, fp32_storage=args.fp32_storage, nprof=nprof,
                                       keep_dir=os.path.join(args.rocprof_keep, '%s_%s' % (name, dtype))
                                       if args.rocprof_keep else None)
                # the step as tracked numbers: launches from the replayed plan, counter bytes from the
                # newest committed PMC summary of this workload / dtype, algorithmic bytes from tools/flops.py
                pst, psrc = pmc_step_traffic(name, dtype)
                alg = algorithmic_step_bytes(wl, dtype, model) if (batch is None or batch == WORKLOADS[name]['batch']) else None
                step_block = {
                    'launches': res['launch_plan']['kernel_launches_per_step'] or None,
                    'hbm_bytes_counter': pst['hbm_bytes'] if pst else None,
                    'hbm_bytes_counter_source': (psrc + ': sum over the kernels of one serialised step of 2 x FETCH_SIZE + '
                                                 'WRITE_SIZE (separate --pmc passes, gfx950 correction)') if pst else None,
                    'launches_in_counter_profile': pst.get('launches') if pst else None,
                    'hbm_bytes_algorithmic': alg,
                    'ratio': (pst['hbm_bytes'] / alg) if (pst and alg) else None}
                res['roofline'] = roofline_of(prof, nprof, dtype, dt, steps, rp, step_block)
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

    def roofline_of(prof, nprof, dtype, dt, steps, rp=None, step_block=None):
        peak = PEAK_TFLOPS[dtype]
        name = max(prof, key=lambda k: prof[k]['ms'])          # the dominant kernel class, whichever pass it is in
        step_flops = sum(v['direct_flops'] for v in prof.values()) / nprof
        exec_flops = sum(v['flops'] for v in prof.values()) / nprof
        r = prof[name]
        fl = r['flops'] / r['launches']
        by = r['bytes'] / r['launches']
        ms_ev = r['ms'] / r['launches']
        ms_raw = r['ms_raw'] / r['launches']
        traffic, traffic_src = pmc_traffic(name)
        rk = (rp or {}).get(name.replace(' ', ''))
        rp_ms = rk['avg_ms'] if rk else None
        # `frac` is computed from the rocprofv3 kernel duration of THIS run (child process above) whenever the profiler ran;
        # the HIP-event figures stay beside it, labelled.  Without the profiler (N > 1, --no-rocprof, profiler failure) it is
        # the RAW event interval — dispatch latency included, a lower bound — never the calibrated one.
        ms = rp_ms if rp_ms else ms_raw
        src = ('rocprofv3 --kernel-trace, child process of this run: average duration of the kernel over the dispatches of '
               '%d serialised steps' % nprof) if rp_ms else 'hip_events: raw interval of an event pair around the kernel (lower bound)'
        achieved = fl / (ms * 1e-3) / 1e12
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
                      'frac_source': src,
                      'frac_rocprofv3': (per_launch / (rp_ms * 1e-3) / pk) if rp_ms else None,
                      'frac_raw_event_interval': per_launch / (ms_raw * 1e-3) / pk,
                      'frac_event_interval_minus_empty_pair': per_launch / (ms_ev * 1e-3) / pk,
                      'rocprofv3': None if not rp else {
                          'cmd': rp['_cmd'], 'avg_ms': rp_ms, 'calls_per_step': rk['calls_per_step'] if rk else None,
                          'step_ms_under_profiler': rp['_step_ms'],
                          'top_kernels': {k: {'avg_us': round(v['avg_ms'] * 1e3, 2), 'calls_per_step': v['calls_per_step']}
                                          for k, v in sorted(((k, v) for k, v in rp.items() if not k.startswith('_')),
                                                             key=lambda kv: -kv[1]['avg_ms'] * kv[1]['calls_per_step'])[:8]}}})
        return dict(bound, **{
            'kernel': name, 'traffic': traffic, 'algorithmic_bytes_per_launch': by,
            'step': step_block,
            'traffic_unit': 'HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE PMC passes)',
            'traffic_source': traffic_src,
            'launches_per_step': r['launches'] / nprof, 'flops_per_launch': fl, 'ms_per_launch': ms,
            'ms_per_launch_raw_event_interval': ms_raw,
            'event_pair_overhead_ms': K._Profile.overhead_ms,
            'ms_per_step': ms * r['launches'] / nprof,
            'timing': '%d serialised profiling steps (every stream of the step on one); `frac` / `ms_per_launch` from '
                      '`frac_source`; `frac_raw_event_interval` = HIP events around the kernel on its launch stream, as '
                      'measured; `frac_event_interval_minus_empty_pair` = the same minus the interval of an event pair around '
                      'an EMPTY kernel (dispatch latency, calibrated in this process)' % nprof,
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

    # ---- N > 1: first contact with RCCL has a safety net (run_ladder): a mode whose timed block raises on any rank, or leaves
    # the replicas with different bits, is reported under `dist.fallbacks` and the next one produces the number (`dist.mode`
    # says which did).  N = 1 has one mode and no collective.
    def attempt(mode):
        P.ENABLED = mode['launch_plan']
        try:
            res = run_workload(args.workload, args.dtype, args.steps, args.warmup, batch=args.batch,
                               want_roofline=not args.no_roofline, phases_n=args.phases,
                               keep_sd=(rank == 0 and world == 1 and not args.no_cpu_baseline))
        except StepFailure:
            raise
        if world > 1 and res.get('replicas_identical') is False:
            T.install_buckets(None)
            torch.cuda.empty_cache()
        return res

    # ---- N > 1: what the collective costs on THIS node, before the first step: the all-reduce of the
    # real flat gradient and of one 12 MB bucket, 5 repetitions each; the bucket size of the exchange follows from the
    # measured latency / bandwidth (LUMINOTH_AMD_BUCKET_MB still overrides)
    probe = None
    if world > 1:
        try:
            _cfg, _m = build(dict(WORKLOADS[args.workload]), device, args.dtype, half_storage=not args.fp32_storage)
            numel = int(_m.store.flat.numel())
            del _cfg, _m
            torch.cuda.empty_cache()
            probe = T.allreduce_probe(numel, device)
            if probe and not os.environ.get('LUMINOTH_AMD_BUCKET_MB'):
                T.PROBED_BUCKET_BYTES = probe['bucket_bytes_from_probe']
        except Exception as e:      # the probe must never cost the run
            probe = {'error': '%s: %s' % (type(e).__name__, e)}
            sys.stderr.write('bench.py: all-reduce probe failed: %r\n' % (e,))

    head, mode, fallbacks = run_ladder(world, attempt)
    if head is None:
        if rank == 0:
            sys.stderr.write('bench.py: every exchange mode failed: %r\n' % (fallbacks,))
        raise SystemExit(3)
    head['mode'] = mode
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
            'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPE_LABEL[args.dtype], 'data': 'synthetic',
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
                     # the collective as measured on this node before the first step: bytes, ms, GB/s per rank, the ranks the
                     # collective itself saw (all-reduce of ones), and the bucket size derived from it
                     'allreduce_probe': probe,
                     'replicas_identical_after_timed_steps': head.get('replicas_identical'),
                     # which exchange mode produced `value`, and the modes that failed before it (first-contact ladder)
                     'mode': head['mode'], 'fallbacks': fallbacks,
                     'collective_timeout_s': args.collective_timeout if world > 1 else None},
        }
        if head.get('phases'):
            out['phases_ms'] = head['phases']
    sd0 = head['sd0']
    del head, model
    other = {}
    native = None
    if args.dtype == 'bf16x3' and not args.no_native and not args.serial:
        # the SAME step on the native fp32 MFMA (v_mfma_f32_32x32x2_f32), same process, same inputs: reported beside `value`
        # (every rank runs it: it contains the gradient exchange)
        torch.cuda.empty_cache()
        try:
            native = run_workload(args.workload, 'f32', args.steps, args.warmup, batch=args.batch, want_roofline=False)
        except StepFailure as e:
            native = {'error': str(e)}
        if rank == 0:
            if 'error' in native:
                out['native_fp32_mfma'] = native
            else:
                out['native_fp32_mfma'] = {
                    'dtype': 'f32', 'value': gb * args.steps / native['dt'], 'unit': 'images/sec',
                    'ms_per_step': native['ms_per_step'], 'ms_per_step_median': native['ms_per_step_median'],
                    'steps': args.steps, 'warmup': args.warmup, 'final_total_loss': native['loss'],
                    'launch_plan': native['launch_plan'],
                    'note': 'same workload, schedule, tensors, inputs and tolerances with every convolution on '
                            'v_mfma_f32_32x32x2_f32 (157.3 TFLOP/s peak): the arithmetic of rounds 1-5\'s headline'}
        native = None
    if args.workload == 'frcnn_r50' and args.dtype in ('f32', 'bf16x3') and not args.serial and not args.no_other_configs:
        # BASELINE configs[4] ("fp16 MFMA path, 1333x800 COCO shapes") on the same record: the half-storage trunk at its own
        # geometry, 15 + 60 steps (~0.4 s of GPU time), with its own roofline leg.  Every rank runs it (it contains the
        # gradient exchange); reported BESIDE `value`, never as it.
        torch.cuda.empty_cache()
        try:
            o = run_workload('frcnn_r50_coco', 'f16', 60, 15, want_roofline=not args.no_roofline)
        except StepFailure as e:
            o = None
            other['frcnn_r50_coco_f16'] = {'error': str(e)}
        if rank == 0 and o is not None:
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
        if not args.no_cpu_baseline and world == 1:          # reported on rank 0 at N = 1 only
            cb = cpu_baseline(wl, sd0, args.cpu_steps)
            if cb is not None:
                out['cpu_baseline'] = cb
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()
        # Two models, their launch plans (HIP events, recorded kernel arguments), side streams and the process group are
        # torn down by the interpreter in no particular order at exit; with two ranks on one GPU that ended in a SIGABRT
        # of rank 0 AFTER the line was printed about one run in three (round 6) — and a non-zero exit code makes the
        # launcher report the whole run as failed.  The record is complete at this point: leave without the teardown.
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == '__main__':
    main()
