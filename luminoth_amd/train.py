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
CKPT_RE = re.compile(r'model\.ckpt-(\d+)\.(?:index|npz)$')


# ------------------------------------------------------------ checkpoints ----
def checkpoint_dir(config):
    """train.py:171-181: job_dir/run_name, job_dir, or None (nothing is saved)."""
    job_dir = config.train.get('job_dir')
    if not job_dir:
        return None
    run_name = config.train.get('run_name')
    return os.path.join(job_dir, run_name) if run_name else job_dir


def list_checkpoints(ckpt_dir):
    """[(global_step, path)] sorted by step.  `path` is the TensorFlow V2 bundle PREFIX `model.ckpt-N` (files
    `.index` + `.data-00000-of-00001`, what `tf.train.Saver` writes and the reference's tooling reads); `.npz`
    files written by round-1 builds of this repo are still listed (path ends in .npz)."""
    found = {}
    for f in glob.glob(os.path.join(ckpt_dir, 'model.ckpt-*')):
        m = CKPT_RE.search(f)
        if m:
            step = int(m.group(1))
            path = f[:-len('.index')] if f.endswith('.index') else f
            if step not in found or not path.endswith('.npz'):
                found[step] = path
    return sorted(found.items())


def _remove_checkpoint(path):
    for f in ([path] if path.endswith('.npz') else glob.glob(path + '.index') + glob.glob(path + '.data-*')):
        os.remove(f)


def save_checkpoint(model, global_step, ckpt_dir, max_to_keep=1):
    """Model variables + `global_step` (no optimizer slots, like train.py:93-112) as a TensorFlow V2 tensor
    bundle under the variables' TF names, so checkpoints travel both ways between this path and the reference's
    tooling (`tf.train.Saver`, `lumi predict --checkpoint`); the `checkpoint` pointer file names the prefix.
    The `max_to_keep` newest checkpoints are kept."""
    from luminoth_amd.utils import tf_checkpoint
    os.makedirs(ckpt_dir, exist_ok=True)
    prefix = os.path.join(ckpt_dir, 'model.ckpt-%d' % global_step)
    sd = {k: np.asarray(v.detach().cpu().numpy() if hasattr(v, 'detach') else v)
          for k, v in model.state_dict().items()}
    sd['global_step'] = np.asarray(global_step, dtype=np.int64)
    tmp = prefix + '.tmp'
    tf_checkpoint.save_v2(tmp, sd)
    os.replace(tf_checkpoint._shard_name(tmp, 0, 1), tf_checkpoint._shard_name(prefix, 0, 1))
    os.replace(tmp + '.index', prefix + '.index')       # the index goes last: a visible index has its data
    with open(os.path.join(ckpt_dir, 'checkpoint'), 'w') as f:       # the TF pointer file
        f.write('model_checkpoint_path: "%s"\n' % os.path.basename(prefix))
    for _, old in list_checkpoints(ckpt_dir)[:-max(1, int(max_to_keep))]:
        _remove_checkpoint(old)
    return prefix


def load_checkpoint_variables(path):
    """{name: ndarray} (incl. `global_step` when present) of a checkpoint written by this repo or by TensorFlow
    (V2 bundle prefix / V1 file), or of a legacy `.npz`."""
    if path.endswith('.npz'):
        data = np.load(path)
        return {k: data[k] for k in data.files}
    from luminoth_amd.utils import tf_checkpoint
    return tf_checkpoint.load_checkpoint(path)


def restore_checkpoint(model, path):
    """Loads the model variables of `path`; returns its global step (parsed from the file name when the
    checkpoint holds no `global_step` tensor)."""
    values = load_checkpoint_variables(path)
    step = values.pop('global_step', None)
    model.load_state_dict(values)        # extra tensors (optimizer slots of a TF-written file) are ignored, missing ones raise
    if step is None:
        m = re.search(r'-(\d+)(?:\.npz)?$', path)
        step = int(m.group(1)) if m else 0
    return int(step)


def restore_latest(model, ckpt_dir):
    """Returns the restored global_step, or None when there is nothing to resume from."""
    if not ckpt_dir or not os.path.isdir(ckpt_dir):
        return None
    ckpts = list_checkpoints(ckpt_dir)
    if not ckpts:
        return None
    return restore_checkpoint(model, ckpts[-1][1])


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
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count())
        if not dist.is_initialized():
            dist.init_process_group(os.environ.get('LUMINOTH_AMD_DIST_BACKEND', 'nccl'))   # 'gloo': CPU control-flow tests

    prev_stream = hp_stream = None
    try:
        import torch
        if torch.cuda.is_available():
            prev_stream = torch.cuda.current_stream()
            hp_stream = training.issue_from_high_priority_stream(prev_stream.device)
    except ImportError:
        pass
    try:
        return _run(config, get_model_fn, get_dataset_fn, train_step_fn, max_steps, rank, world, is_chief, log_prefix)
    finally:
        if hp_stream is not None:
            prev_stream.wait_stream(hp_stream)
            torch.cuda.set_stream(prev_stream)


def _run(config, get_model_fn, get_dataset_fn, train_step_fn, max_steps, rank, world, is_chief, log_prefix):
    from luminoth_amd.utils import training
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
    # one batch of look-ahead: the step is handed the NEXT image as well, so the model may run that image's frozen
    # trunk prefix in the slot where its main stream waits for the proposal / RCNN branch (FasterRCNN.train_step)
    lookahead = train_step_fn is training.train_step
    it = iter(dataset)
    nxt = next(it, None)
    while nxt is not None:
        # no look-ahead behind the last step of a bounded run: the extra batch would be decoded, uploaded, its prefix
        # and anchor targets computed — and then dropped, one record consumed for nothing
        is_last = max_steps is not None and step + 1 - global_step >= max_steps
        batch, nxt = nxt, (None if is_last else next(it, None))
        before = time.time()
        if lookahead and nxt is not None:
            total_loss, _ = train_step_fn(model, optimizer, batch['image'], batch['bboxes'], next_image=nxt['image'],
                                          next_gt=nxt['bboxes'])
        else:
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
