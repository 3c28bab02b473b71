"""`lumi train` re-hosted on the MI355X hot path (reference: luminoth/train.py:19-326).

    python -m luminoth_amd.train -c config.yml -o train.num_epochs=2 -o model.network.num_classes=80
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m luminoth_amd.train -c config.yml

Same flow as the reference's `run()`: merged config (`get_config`: base_config + custom files + `-o`
overrides) -> `get_model(config.model.type)(config)` -> dataset -> loop {train step, the reference's log line}
-> periodic checkpoints under `job_dir/run_name` (model variables only: optimizer slots are re-initialised on
restart exactly like train.py:93-112) with `checkpoints_max_keep`, resume from the latest one.  What replaces
the TF machinery: eager step on HIP kernels instead of `MonitoredTrainingSession`, one process per GPU with an
RCCL all-reduce instead of `TF_CONFIG` parameter servers (synchronous, averaged gradients: SURVEY.md §8e).
"""
import argparse
import glob
import json
import logging
import os
import re
import sys
import time

import numpy as np

log = logging.getLogger('luminoth_amd')
CKPT_RE = re.compile(r'model\.ckpt-(\d+)\.npz$')


# ------------------------------------------------------------ checkpoints ----
def checkpoint_dir(config):
    """train.py:171-181: job_dir/run_name, job_dir, or None (nothing is saved)."""
    job_dir = config.train.get('job_dir')
    if not job_dir:
        return None
    run_name = config.train.get('run_name')
    return os.path.join(job_dir, run_name) if run_name else job_dir


def list_checkpoints(ckpt_dir):
    out = []
    for f in glob.glob(os.path.join(ckpt_dir, 'model.ckpt-*.npz')):
        m = CKPT_RE.search(f)
        if m:
            out.append((int(m.group(1)), f))
    return sorted(out)


def save_checkpoint(model, global_step, ckpt_dir, max_to_keep=1):
    """Model variables + global_step (no optimizer slots), `max_to_keep` newest files kept."""
    os.makedirs(ckpt_dir, exist_ok=True)
    path = os.path.join(ckpt_dir, 'model.ckpt-%d.npz' % global_step)
    sd = {k: np.asarray(v.detach().cpu().numpy() if hasattr(v, 'detach') else v) for k, v in model.state_dict().items()}
    tmp = path + '.tmp.npz'
    np.savez(tmp, global_step=np.int64(global_step), **sd)
    os.replace(tmp, path)
    with open(os.path.join(ckpt_dir, 'checkpoint'), 'w') as f:       # the TF-style pointer file
        f.write('model_checkpoint_path: "%s"\n' % os.path.basename(path))
    for _, old in list_checkpoints(ckpt_dir)[:-max(1, int(max_to_keep))]:
        os.remove(old)
    return path


def restore_latest(model, ckpt_dir):
    """Returns the restored global_step, or None when there is nothing to resume from."""
    if not ckpt_dir or not os.path.isdir(ckpt_dir):
        return None
    ckpts = list_checkpoints(ckpt_dir)
    if not ckpts:
        return None
    step, path = ckpts[-1]
    data = np.load(path)
    model.load_state_dict({k: data[k] for k in data.files if k != 'global_step'})
    return int(data['global_step'])


# ------------------------------------------------------------------- run ----
def run(config, get_model_fn=None, get_dataset_fn=None, train_step_fn=None, max_steps=None):
    """The reference's run(config, ...) (train.py:19-269).  Returns the last global step."""
    from luminoth_amd.models import get_model
    from luminoth_amd.datasets import get_dataset
    from luminoth_amd.utils import training
    get_model_fn = get_model_fn or get_model
    get_dataset_fn = get_dataset_fn or get_dataset
    train_step_fn = train_step_fn or training.train_step

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    is_chief = rank == 0
    log_prefix = '[worker-{}] - '.format(rank) if world > 1 else ''
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        if not dist.is_initialized():
            dist.init_process_group('nccl')

    model = get_model_fn(config.model.type)(config)
    try:
        config['dataset']['type']
    except KeyError:
        raise KeyError('dataset.type should be set on the custom config.')
    dataset = get_dataset_fn(config.dataset.type)(config)
    optimizer = training.get_optimizer(config.train, model)

    ckpt_dir = checkpoint_dir(config)
    if ckpt_dir is None:
        log.warning('`job_dir` is not defined. Checkpoints and logs will not be saved.')
    global_step = restore_latest(model, ckpt_dir)
    if global_step is None:
        global_step = 0
    else:
        log.info('%sRestored checkpoint at step %d from %s', log_prefix, global_step, ckpt_dir)
    pretrained = None
    if global_step == 0 and hasattr(model, 'get_checkpoint_file') and hasattr(model, 'get_base_network_checkpoint_vars'):
        pretrained = model.get_checkpoint_file()                  # train.py:114-127
        if pretrained:
            from luminoth_amd.utils.tf_checkpoint import restore_base_network
            names = restore_base_network(model, pretrained)
            log.info('%sLoaded %d base-network variables from %s', log_prefix, len(names), pretrained)
    if global_step == 0 and not pretrained:
        log.warning('%sno pretrained base-network weights (model.base_network.weights) and no checkpoint to resume: '
                    'training starts from random initialisation (BatchNorm statistics are identity)', log_prefix)
    optimizer.global_step = global_step        # slots (momentum) start from zero, like train.py:93-112
    training.broadcast_parameters(model)

    save_secs = config.train.get('save_checkpoint_secs')
    max_keep = config.train.get('checkpoints_max_keep', 1)
    last_save = time.time()
    log.info('%sStarting training for %s', log_prefix, type(model).__name__)
    step = global_step
    for batch in dataset:
        before = time.time()
        total_loss, _ = train_step_fn(model, optimizer, batch['image'], batch['bboxes'])
        train_loss = float(total_loss)          # the per-step fetch of train.py:237-239 (host sync)
        step += 1
        log.info('%sstep: %d, file: %s, train_loss: %s, in %.2fs', log_prefix, step, batch.get('filename'),
                 train_loss, time.time() - before)
        if is_chief and ckpt_dir and save_secs and time.time() - last_save >= save_secs:
            save_checkpoint(model, step, ckpt_dir, max_keep)
            last_save = time.time()
        if max_steps is not None and step - global_step >= max_steps:
            break
    else:
        log.info('%sfinished training after %s epoch limit', log_prefix, config.train.get('num_epochs'))
    if is_chief and ckpt_dir:
        save_checkpoint(model, step, ckpt_dir, max_keep)
    return step


def main(argv=None):
    ap = argparse.ArgumentParser(description='Train models (luminoth/train.py:271-326)')
    ap.add_argument('--config', '-c', dest='config_files', action='append', required=True, help='Config to use.')
    ap.add_argument('--job-dir', help='Job directory.')
    ap.add_argument('--override', '-o', dest='override_params', action='append', default=[],
                    help='Override model config params.')
    ap.add_argument('--max-steps', type=int, default=None)
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format='%(levelname)s:%(name)s:%(message)s')
    from luminoth_amd.utils.config import get_config
    overrides = list(args.override_params)
    if args.job_dir:
        overrides.append('train.job_dir={}'.format(args.job_dir))
    try:
        config = get_config(args.config_files, override_params=overrides)
    except KeyError:
        raise KeyError('model.type should be set on the custom config.')
    return run(config, max_steps=args.max_steps)


if __name__ == '__main__':
    main()
