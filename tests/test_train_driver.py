"""CPU: the `lumi train` re-host (luminoth_amd/train.py) with a mock model — the port of the reference's
luminoth/train_test.py:19-157 (MockFasterRCNN with one weight, loop runs, checkpoint is written and holds the
trained value, resume continues from the saved global step)."""
import os

import numpy as np
import torch

from luminoth_amd import train as T
from luminoth_amd.utils.config import get_config


class MockModel(object):
    """train_test.py:19-52: loss = reduce_sum(w * 0) ... here a weight that every step moves by +0.5."""

    def __init__(self, config):
        self.w = torch.tensor([2.0, 2.5])
        self.steps = 0

    def state_dict(self):
        return {'mockfasterrcnn/w': self.w.clone()}

    def load_state_dict(self, sd, strict=True):
        self.w = torch.as_tensor(np.asarray(sd['mockfasterrcnn/w'])).clone()


class MockOptimizer(object):
    global_step = 0


def mock_train_step(model, optimizer, image, gt_boxes):
    assert image.shape[-1] == 3 and len(gt_boxes) == image.shape[0]
    model.w = model.w + 0.5
    model.steps += 1
    optimizer.global_step += 1
    return torch.tensor(float(model.w.sum())), {}


def make_config(tmpdir, **over):
    ov = ['train.num_epochs=1', 'dataset.type=synthetic', 'dataset.num_images=3', 'dataset.height=64',
          'dataset.width=96', 'dataset.image_preprocessing.max_size=96', 'train.save_checkpoint_secs=0']
    ov += ['%s=%s' % kv for kv in over.items()]
    if tmpdir is not None:
        ov += ['train.job_dir=%s' % tmpdir, 'train.run_name=test_runname']
    else:
        ov += ['train.job_dir=']
    return get_config({'model': {'type': 'fasterrcnn'}}, ov)


def patched_run(config, monkeypatch, **kw):
    from luminoth_amd.utils import training
    monkeypatch.setattr(training, 'get_optimizer', lambda cfg, model: MockOptimizer())
    monkeypatch.setattr(training, 'broadcast_parameters', lambda model: None)
    return T.run(config, get_model_fn=lambda t: MockModel, train_step_fn=mock_train_step, **kw)


def test_train_runs_without_job_dir(monkeypatch):
    assert patched_run(make_config(None), monkeypatch) == 3          # train_test.py:108-122: "This should not fail"


def test_train_saves_checkpoint_and_resumes(tmp_path, monkeypatch):
    cfg = make_config(str(tmp_path))
    assert patched_run(cfg, monkeypatch) == 3
    # checkpoints are TensorFlow V2 tensor bundles (what tf.train.Saver writes: train.py:93-112, 153)
    run_dir = tmp_path / 'test_runname'
    assert (run_dir / 'model.ckpt-3.index').exists() and (run_dir / 'model.ckpt-3.data-00000-of-00001').exists()
    assert (run_dir / 'checkpoint').read_text().strip() == 'model_checkpoint_path: "model.ckpt-3"'
    from luminoth_amd.utils import tf_checkpoint
    data = tf_checkpoint.load_v2(str(run_dir / 'model.ckpt-3'))
    np.testing.assert_allclose(data['mockfasterrcnn/w'], [3.5, 4.0])  # 3 steps of +0.5 (train_test.py:155-157 analogue)
    assert int(data['global_step']) == 3 and data['global_step'].dtype == np.int64
    # resume: continues from step 3 with the saved weights; old checkpoints beyond max_to_keep are dropped
    assert patched_run(cfg, monkeypatch) == 6
    files = sorted(os.listdir(str(run_dir)))
    assert 'model.ckpt-6.index' in files and not [f for f in files if f.startswith('model.ckpt-3')]
    np.testing.assert_allclose(tf_checkpoint.load_v2(str(run_dir / 'model.ckpt-6'))['mockfasterrcnn/w'], [5.0, 5.5])


def test_restore_reads_legacy_npz_and_foreign_tf_checkpoints(tmp_path):
    """A round-1 `.npz` checkpoint and a TF-written bundle with optimizer slots / no global_step both restore."""
    from luminoth_amd.utils import tf_checkpoint
    m = MockModel(None)
    np.savez(str(tmp_path / 'model.ckpt-5.npz'), global_step=np.int64(5), **{'mockfasterrcnn/w': np.array([7.0, 8.0], np.float32)})
    assert T.restore_latest(m, str(tmp_path)) == 5
    np.testing.assert_allclose(m.w.numpy(), [7.0, 8.0])
    tf_checkpoint.save_v2(str(tmp_path / 'model.ckpt-12'), {'mockfasterrcnn/w': np.array([1.0, 2.0], np.float32),
                                                           'mockfasterrcnn/w/Momentum': np.zeros(2, np.float32)})
    assert [s for s, _ in T.list_checkpoints(str(tmp_path))] == [5, 12]
    assert T.restore_latest(m, str(tmp_path)) == 12          # step parsed from the file name
    np.testing.assert_allclose(m.w.numpy(), [1.0, 2.0])


def test_checkpoint_rotation_and_dataset_registry(tmp_path):
    m = MockModel(None)
    for s in (1, 2, 3, 4):
        T.save_checkpoint(m, s, str(tmp_path), max_to_keep=2)
    assert [s for s, _ in T.list_checkpoints(str(tmp_path))] == [3, 4]
    from luminoth_amd.datasets import get_dataset
    import pytest
    with pytest.raises(ValueError):
        get_dataset('nope')
    ds = get_dataset('synthetic')(make_config(None, **{'train.batch_size': 2, 'dataset.num_images': 4}))
    batches = list(ds)
    assert len(batches) == 2 and batches[0]['image'].shape == (2, 64, 96, 3) and batches[0]['bboxes'][0].shape == (8, 5)
    b = batches[0]['bboxes'][0]
    assert (b[:, 2] < 96).all() and (b[:, 3] < 64).all() and (b[:, 0] <= b[:, 2]).all()


def test_cli_dispatch(monkeypatch, capsys):
    """luminoth/cli.py: train / predict / eval are hosted behind one entry point; the other groups are refused."""
    from luminoth_amd import __main__ as cli
    assert cli.main(['--help']) == 0 and 'train' in capsys.readouterr().out
    assert cli.main([]) == 2
    assert cli.main(['cloud', 'gc', 'train']) == 2 and 'not hosted' in capsys.readouterr().err
    assert cli.main(['frobnicate']) == 2
    seen = {}
    import luminoth_amd.train as T
    monkeypatch.setattr(T, 'main', lambda argv: seen.setdefault('argv', argv) and 17)
    assert cli.main(['train', '-c', 'x.yml', '-o', 'a=b']) == 0 and seen['argv'] == ['-c', 'x.yml', '-o', 'a=b']
    import luminoth_amd.predict as P
    monkeypatch.setattr(P, 'main', lambda argv: 2)
    assert cli.main(['predict']) == 2


def test_bounded_run_does_not_pull_a_batch_it_will_not_train_on(monkeypatch):
    """with max_steps the look-ahead used to consume (decode, upload, prefix) one extra record."""
    from luminoth_amd.datasets import synthetic
    pulled = []
    orig_iter = synthetic.SyntheticObjectDetectionDataset.__iter__

    def counting_iter(self):
        for b in orig_iter(self):
            pulled.append(1)
            yield b
    monkeypatch.setattr(synthetic.SyntheticObjectDetectionDataset, '__iter__', counting_iter)
    assert patched_run(make_config(None), monkeypatch, max_steps=2) == 2
    assert len(pulled) == 2


def test_optimizer_arguments_are_not_swallowed():
    """the reference forwards every remaining optimizer key to the TF constructor (training.py:64-81), which
    raises on unknown keywords; Adam's bias-correction step follows the restored global_step."""
    import pytest
    from luminoth_amd.params import ParamStore
    from luminoth_amd.utils import training

    class M(object):
        pass
    m = M()
    m.store = ParamStore()
    m.store.add('w', (4,), lambda s, g: torch.ones(s), trainable=True)
    m.store.build(torch.device('cpu'), seed=0)

    def cfg(**opt):
        return get_config({'model': {'type': 'fasterrcnn'}, 'train': {'optimizer': dict(_replace=True, **opt)}}).train
    assert training.get_optimizer(cfg(type='momentum', momentum=0.7), m).momentum == 0.7
    with pytest.raises(TypeError):
        training.get_optimizer(cfg(type='momentum', momentun=0.7), m)              # typo
    with pytest.raises(TypeError):
        training.get_optimizer(cfg(type='adam', learning_rate=0.1), m)             # TF: multiple values for learning_rate
    # round 4: the two keyword arguments that used to raise NotImplementedError have kernels now (optim.hip kinds 3, 4)
    rms = training.get_optimizer(cfg(type='rmsprop', centered=True), m)
    assert rms.centered and rms._slot3 is not None and float(rms._slot3.abs().max()) == 0.0
    assert training.get_optimizer(cfg(type='momentum', use_nesterov=True), m).use_nesterov
    adam = training.get_optimizer(cfg(type='adam', beta1=0.8), m)
    assert adam.beta1 == 0.8 and adam._t == 1
    adam.global_step = 500                                                         # what train.run does on resume
    assert adam._t == 501
