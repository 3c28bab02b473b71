"""CPU-only tests of the host side of the boundary: config merge semantics, registry, anchors,
initialisers, variable naming / trainable-set logic (no kernels launched)."""
import numpy as np
import pytest
import torch

from luminoth_amd.utils import config as C
from luminoth_amd.utils.anchors import generate_anchors_reference, truncate_reference, all_anchors_numpy
from oracle import boxes as obx


def test_get_model_registry():
    # luminoth/models/models.py:11-17
    from luminoth_amd.models import get_model
    from luminoth_amd.models.fasterrcnn.fasterrcnn import FasterRCNN
    assert get_model('FasterRCNN') is FasterRCNN
    with pytest.raises(ValueError):
        get_model('yolo')


def test_config_merge_replace_override_and_types():
    # luminoth/utils/config.py:73-196
    cfg = C.get_config({'model': {'type': 'fasterrcnn', 'network': {'num_classes': 80}},
                        'train': {'learning_rate': {'decay_method': 'piecewise_constant', 'boundaries': [1],
                                                    'values': [1., .1]}}},
                       ['model.rpn.proposals.post_nms_top_n=300', 'train.seed=7', 'train.debug=True',
                        'model.base_network.fine_tune_from=None'])
    assert cfg.model.network.num_classes == 80 and cfg.model.network.with_rcnn is True
    assert cfg.model.rpn.proposals.post_nms_top_n == 300 and cfg.model.rpn.proposals.pre_nms_top_n == 12000
    assert cfg.train.seed == 7 and cfg.train.debug is True and cfg.model.base_network.fine_tune_from is None
    # `_replace: True` on learning_rate: the default's `learning_rate: 0.0003` key is gone
    assert 'learning_rate' not in cfg.train.learning_rate and '_replace' not in cfg.train.learning_rate
    with pytest.raises(ValueError):
        C.get_config({'model': {'type': 'fasterrcnn', 'network': {'num_classes': 'many'}}})
    with pytest.raises(ValueError):
        C.parse_override(['a=b=c'])
    assert C.parse_override(['a.b=1', 'a.c=0.5', 'd=None', 'e=true', 'f=x']) == \
        {'a': {'b': 1, 'c': .5}, 'd': None, 'e': True, 'f': 'x'}


def test_anchor_reference_matches_oracle_and_quirk():
    ref = generate_anchors_reference(256, [.5, 1, 2], [.25, .5, 1, 2])
    np.testing.assert_allclose(ref, obx.generate_anchors_reference(256, np.array([.5, 1, 2]),
                                                                   np.array([.25, .5, 1, 2])), rtol=1e-12)
    with pytest.raises(ValueError):
        generate_anchors_reference(1, [.5], [.5])
    np.testing.assert_array_equal(all_anchors_numpy(ref, 5, 7, 16), obx.generate_anchors(ref, 5, 7, 16))
    assert truncate_reference(np.array([[-22.13, 29.9, -0.5, 0.5]])).tolist() == [[-22, 29, 0, 0]]


def test_initializers_and_activations():
    from luminoth_amd.utils.vars import get_initializer, get_activation_function
    g = torch.Generator().manual_seed(0)
    w = get_initializer({'type': 'random_normal_initializer', 'mean': 0., 'stddev': 0.01})((3, 3, 64, 128), g)
    assert abs(float(w.std()) - 0.01) < 5e-4
    u = get_initializer({'type': 'variance_scaling_initializer', 'factor': 1.0, 'uniform': True,
                         'mode': 'FAN_AVG'})((1024, 81), g)
    assert float(u.abs().max()) <= np.sqrt(3.0 / ((1024 + 81) / 2)) + 1e-6
    with pytest.raises(ValueError):
        get_initializer({'type': 'nope'})
    # tf.truncated_normal re-draws what falls outside two standard deviations (vars.py:4-5): no mass ON the bounds, and the
    # standard deviation of the truncated law (0.8796 sigma), not that of a clamped one (0.9543 sigma)
    t = get_initializer({'type': 'truncated_normal_initializer', 'mean': 0., 'stddev': 0.5})((512, 512), g)
    assert float(t.abs().max()) < 1.0 and int((t.abs() == 1.0).sum()) == 0
    assert abs(float(t.std()) / 0.5 - 0.8796) < 5e-3
    assert get_activation_function('relu6') == 'relu6' and get_activation_function(None) is None
    assert get_activation_function('elu') == 'elu' and get_activation_function('leaky_relu') == 'leaky_relu'      # any tf.nn activation
    with pytest.raises(ValueError):
        get_activation_function('swish')


def test_trainable_variable_selection_resnet():
    # base_network.py:211-241 + truncated_base_network.py:97-144
    from luminoth_amd.models.base.truncated_base_network import TruncatedBaseNetwork
    cfg = C.get_config({'model': {'type': 'fasterrcnn', 'base_network': {'architecture': 'resnet_v1_50'}}})
    net = TruncatedBaseNetwork(cfg.model.base_network)
    names = net.get_trainable_var_names()
    assert names[0].endswith('resnet_v1_50/block2/unit_1/bottleneck_v1/shortcut/weights')
    assert names[-1].endswith('resnet_v1_50/block3/unit_6/bottleneck_v1/conv3/BatchNorm/gamma')
    assert not any('/block1/' in n or '/block4/' in n or n.endswith('conv1/weights') and '/block' not in n
                   for n in names)
    # R50 to block4/unit_3: 156 trainable vars in total (truncated_base_network_test.py:61-133);
    # block2+block3 = (4+6) units * 9 vars + 2 shortcuts * 3 = 96
    assert len(names) == 96
    cfg101 = C.get_config({'model': {'type': 'fasterrcnn'}})
    net101 = TruncatedBaseNetwork(cfg101.model.base_network)
    n101 = net101.get_trainable_var_names()
    assert any('/block4/' in n for n in n101) and net101.tail is not None     # tail only for R101
    all_vars = net._ordered_var_names()
    assert len(all_vars) == 3 + 16 * 9 + 4 * 3                                 # conv1 + 16 units + 4 shortcuts
    bad = C.get_config({'model': {'type': 'fasterrcnn', 'base_network': {'architecture': 'resnet_v1_50',
                                                                          'fine_tune_from': 'nope'}}})
    with pytest.raises(ValueError):
        TruncatedBaseNetwork(bad.model.base_network).get_trainable_var_names()
    with pytest.raises(ValueError):
        TruncatedBaseNetwork(C.Config({'architecture': 'lenet'}))


def test_storage_dtype_key_selects_the_half_storage_layers():
    """Extension key model.base_network.storage_dtype (BASELINE configs[4]; SURVEY.md 8(d): fp16 activations / weights, fp32
    master weights): which layers become half-storage layers is host logic."""
    from luminoth_amd.models.base import layers as L
    from luminoth_amd.models.base.truncated_base_network import TruncatedBaseNetwork
    base = {'architecture': 'resnet_v1_50', 'storage_dtype': 'bf16'}
    net = TruncatedBaseNetwork(C.get_config({'model': {'type': 'fasterrcnn', 'base_network': base}}).model.base_network)
    assert net.storage_dtype == 'bf16' and net.compute_dtype == 'bf16'          # half storage implies half MFMA operands
    nodes = net.trunk.nodes
    assert nodes[0].layer.storage is None and nodes[0].layer.cin == 3            # the fp32 stem convolution
    assert isinstance(nodes[1], L.MaxPoolNode) and nodes[1].storage == 'bf16'    # the pool writes the first 16-bit tensor
    hs = [l for n in nodes[2:] for l in n.layers]
    assert hs == net._hs_layers and len(hs) == 42 and all(l.storage == 'bf16' and l.compute == 'bf16' for l in hs)
    assert [l.hs_out_f32 for l in hs].count(True) == 1 and nodes[-1].conv3.hs_out_f32      # fp32 feature map handed on
    assert L.HS_LOSS_SCALE == {'f16': 1024.0, 'bf16': 1.0}
    plain = TruncatedBaseNetwork(C.get_config({'model': {'type': 'fasterrcnn', 'base_network': {
        'architecture': 'resnet_v1_50'}}}).model.base_network)
    assert plain.storage_dtype is None and not plain._hs_layers and all(l.storage is None for l in plain.trunk.all_layers())
    with pytest.raises(ValueError):
        TruncatedBaseNetwork(C.Config({'architecture': 'resnet_v1_50', 'storage_dtype': 'fp8'}))
    with pytest.raises(ValueError):               # storage and compute types must agree
        TruncatedBaseNetwork(C.Config({'architecture': 'resnet_v1_50', 'storage_dtype': 'f16', 'compute_dtype': 'bf16'}))
    with pytest.raises(NotImplementedError):      # VGG trunks / the R101 tail keep fp32 tensors
        TruncatedBaseNetwork(C.Config({'architecture': 'vgg_16', 'storage_dtype': 'f16'}))
    with pytest.raises(NotImplementedError):      # resnet_v2's stand-alone BatchNorm layers are fp32 only
        TruncatedBaseNetwork(C.Config({'architecture': 'resnet_v2_50', 'storage_dtype': 'f16'}))
    # round 4: ResNet-101 — the 33-unit trunk keeps 16-bit tensors, the block4 tail on the fp32 ROI crops keeps fp32 tensors
    r101 = TruncatedBaseNetwork(C.get_config({'model': {'type': 'fasterrcnn', 'base_network': {
        'architecture': 'resnet_v1_101', 'storage_dtype': 'f16'}}}).model.base_network)
    assert len(r101._hs_layers) == 3 * 3 + 4 * 3 + 23 * 3 + 3 and r101.tail is not None
    assert all(l.storage is None and l.compute == 'f16' for l in r101.tail.all_layers())


def test_oracle_half_storage_layer_rounds_where_the_kernels_round():
    """oracle/torch_ops.py HalfStorageConvFn (the restatement csrc/conv_hs.h is compared with): stored tensors are 16-bit
    values, the loss scale is exact, and with rounding switched off it is the plain fp32 layer."""
    import oracle.torch_ops as ot
    torch.manual_seed(0)
    x = torch.randn(1, 6, 7, 64).to(torch.float16).float().requires_grad_(True)
    w = (torch.randn(3, 3, 64, 64) * 0.05).requires_grad_(True)
    scale = (1 + 0.1 * torch.randn(64)).requires_grad_(True)
    shift = (0.1 * torch.randn(64)).requires_grad_(True)
    cfg = dict(quant='f16', stride=1, dilation=1, padding='SAME', act='relu', out_f32=False, round_dx=True, loss_scale=1024.0)
    y = ot.HalfStorageConvFn.apply(x, w, scale, shift, None, None, cfg)
    assert torch.equal(y, y.to(torch.float16).float())                      # the stored activation is an f16 value
    gy = torch.randn_like(y) * 1e-6                                          # gradient-sized: below f16's normal range unscaled
    dx, dw = torch.autograd.grad(y, [x, w], gy)
    assert torch.equal(dx * 1024.0, (dx * 1024.0).to(torch.float16).float())   # round_dx: a 16-bit tensor times 2^-10
    y32 = torch.relu(ot.conv2d_nhwc(x, w.to(torch.float16).float(), 1, 1, 'SAME') * scale + shift)
    assert float((y - y32).abs().max()) <= 2.0 ** -11 * float(y32.abs().max()) * 1.001
    dx32, dw32 = torch.autograd.grad(y32, [x, w], gy)
    assert float((dw - dw32).abs().max()) <= 2e-3 * float(dw32.abs().max())    # operand rounding only: the scale kept 1e-6 alive
    assert float(dw.abs().max()) > 0


def test_launch_stream_override_nests_and_restores():
    """kernels.launch_on: the launch-stream override used for the weight-gradient stream (no torch stream context) nests,
    restores on exceptions, and is what every launch of the module resolves its stream from."""
    from luminoth_amd import kernels as K

    class FakeStream(object):
        def __init__(self, h):
            self.cuda_stream = h

    a, b = FakeStream(0x1000), FakeStream(0x2000)
    assert K._stream_override is None
    with K.launch_on(a):
        assert K._stream_id() == 0x1000 and K._stream().value == 0x1000 and K._stream_override_obj is a
        with K.launch_on(b):
            assert K._stream_id() == 0x2000 and K._stream_override_obj is b
        assert K._stream_id() == 0x1000 and K._stream_override_obj is a
        with pytest.raises(RuntimeError):
            with K.launch_on(b):
                raise RuntimeError('inside')
        assert K._stream_id() == 0x1000
    assert K._stream_override is None and K._stream_override_obj is None


def test_param_store_layout_cpu():
    from luminoth_amd.params import ParamStore
    st = ParamStore()
    ones = lambda s, g: torch.ones(s)
    st.add('a/weights', (3, 5), ones, trainable=True, wd=1e-3)
    st.add('f/weights', (7,), ones, trainable=False, wd=5e-4)
    st.add('a/biases', (5,), ones, trainable=True)
    st.build(torch.device('cpu'), seed=0)
    assert st.flat.numel() == 16 + 8 and st.frozen.numel() == 8           # 4-float segment padding
    assert st.seg_offset.tolist() == [0, 16, 24] and st.seg_wd.tolist() == pytest.approx([1e-3, 0.0])
    st.grads['a/biases'].fill_(2.)
    assert float(st.grad[16:21].sum()) == 10.
    sd = st.state_dict()
    sd['a/weights'] = torch.zeros(3, 5)
    st.load_state_dict(sd)
    assert float(st.flat[:15].abs().sum()) == 0.


def test_gradient_bucket_plan_on_resnet50_layout():
    """utils/training.py GradientBuckets._plan (pure host logic): for the ResNet-50 Faster R-CNN parameter layout the
    early buckets are contiguous, disjoint, end-aligned with whole trunk nodes, cover the heads, and leave only the
    first trunk nodes + the BatchNorm block to finish()."""
    from luminoth_amd.models import get_model
    from luminoth_amd.utils.config import get_config
    from luminoth_amd.utils.training import GradientBuckets
    cfg = get_config({'model': {'type': 'fasterrcnn', 'network': {'num_classes': 80},
                                'base_network': {'architecture': 'resnet_v1_50'}}, 'train': {'seed': 0}})
    model = get_model('fasterrcnn')(cfg, device='cpu')
    st = model.store
    numel = int(st.grad.numel())
    trunk = model.base_network.trunk
    nodes = trunk.nodes[trunk.first_trainable():]
    b = GradientBuckets(st, reduce_fn=lambda t: None, bucket_bytes=12 << 20)
    plan = b._plan(nodes)
    assert b._plan(nodes) is plan                                         # cached per trunk
    assert len(nodes) in plan                                              # heads bucket launched before the loop
    heads_lo, heads_hi = plan[len(nodes)]
    assert heads_hi == numel and heads_lo % 4 == 0
    rpn_lo = min(o for n, (t, o, c) in st.offsets.items() if t and '/rpn/' in n)
    assert heads_lo <= rpn_lo                                              # RPN + RCNN are inside the heads bucket
    trunk_names = set(nm for n in nodes for l in n.layers for nm in l.var_names())
    trunk_hi = max(o + c for n, (t, o, c) in st.offsets.items() if t and n in trunk_names)
    assert heads_lo >= trunk_hi                                            # no trunk parameter inside it
    ranges = sorted(v for k, v in plan.items())
    for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
        assert a1 <= b0                                                    # disjoint
    node_buckets = {k: v for k, v in plan.items() if k < len(nodes)}
    assert node_buckets and all((hi - lo) * 4 >= 12 << 20 for lo, hi in node_buckets.values())
    for j, (lo, hi) in node_buckets.items():
        w0 = min(st.offsets[l.w_name][1] for l in nodes[j].layers)
        assert lo == w0                                                    # starts at the first weight of node j
    covered = sum(hi - lo for lo, hi in plan.values())
    assert 0.85 * numel < covered < numel                                  # >85 % of the buffer goes out early
    # a trunk whose weights are not laid out node after node gets no early buckets
    b2 = GradientBuckets(st, reduce_fn=lambda t: None, bucket_bytes=1 << 20)
    assert b2._plan(list(reversed(nodes))) == {}


def test_data_parallel_sharding_is_disjoint_and_equal_length(monkeypatch):
    """every rank must see different records and run the same number of steps."""
    from luminoth_amd.utils import sharding
    order = list(np.random.RandomState(0).permutation(103))
    parts = [sharding.shard_order(order, r, 4) for r in range(4)]
    assert [len(p) for p in parts] == [25] * 4
    flat = [i for p in parts for i in p]
    assert len(set(flat)) == 100 and set(flat) <= set(order)
    assert sharding.shard_order(order, 0, 1) == order
    # synthetic dataset: batches dealt rank::world, same count on every rank
    from luminoth_amd.datasets.synthetic import SyntheticObjectDetectionDataset
    from luminoth_amd.utils.config import get_config
    cfg = get_config({'model': {'type': 'fasterrcnn'}, 'dataset': {'type': 'synthetic', 'num_images': 7, 'height': 96,
                                                                  'width': 96, 'boxes_per_image': 1},
                      'train': {'batch_size': 1, 'num_epochs': 1, 'seed': 3}})
    seen = []
    for r in range(2):
        monkeypatch.setattr(sharding, 'rank_world', lambda r=r: (r, 2))
        ds = SyntheticObjectDetectionDataset(cfg)
        names = [b['filename'][0] for b in ds]
        assert len(names) == len(ds) == 3
        seen.append(names)
    assert not set(seen[0]) & set(seen[1])


def test_get_config_requires_model_type():
    """luminoth/utils/config.py:16: `custom_config['model']['type']` — a config without it is a KeyError, not a
    silently assumed Faster R-CNN (the CLIs turn it into 'model.type should be set on the custom config.')."""
    import pytest
    from luminoth_amd.utils.config import get_config
    with pytest.raises(KeyError):
        get_config({'train': {'seed': 0}})
    assert get_config({'model': {'type': 'ssd'}}).model.type == 'ssd'


def test_lds_sort_pass_grouping_sorts():
    """Host twin of the LDS phases of the u64 sort (csrc/proposals.hip: sort_lds_passes / sort_lds_step, round 4): up to
    three compare-exchange passes of a bitonic merge step run on the 2^S keys a thread owns (the keys that differ in exactly
    the S stride bits), the remainder of a step as a group of two or one.  The index arithmetic — group g -> base with the
    stride bits cleared, direction from bit k of the GLOBAL index, stride-j0 >= chunk passes done globally — is restated
    here and must sort; the device kernels are pinned against np.sort in tests/test_gpu_kernels.py::test_sort_u64."""
    import numpy as np

    def passes(s, chunk, gbase, k, j, S):
        E, jl = 1 << S, j >> (S - 1)
        for g in range(chunk // E):
            low = g & (jl - 1)
            base = ((g - low) << S) | low
            asc = ((gbase + base) & k) == 0
            idx = [base + e * jl for e in range(E)]
            v = [s[i] for i in idx]
            for q in range(S):
                bit = 1 << (S - 1 - q)
                for e in range(E):
                    if not (e & bit) and (v[e] > v[e | bit]) == asc:
                        v[e], v[e | bit] = v[e | bit], v[e]
            for i, x in zip(idx, v):
                s[i] = x

    def step(s, chunk, gbase, k, j0):
        j = j0
        while j > 0:
            if j >= 4:
                passes(s, chunk, gbase, k, j, 3)
                j >>= 3
            elif j == 2:
                passes(s, chunk, gbase, k, j, 2)
                j = 0
            else:
                passes(s, chunk, gbase, k, j, 1)
                j = 0

    rs = np.random.RandomState(5)
    for n, chunk in ((2, 2), (8, 8), (64, 64), (256, 64), (1024, 128), (2048, 2048), (4096, 512)):
        keys = [int(x) for x in rs.randint(0, 1000, size=n)]
        ref = sorted(keys)
        for c0 in range(0, n, chunk):                       # k_sort_local
            s = keys[c0:c0 + chunk]
            k = 2
            while k <= chunk:
                step(s, chunk, c0, k, k >> 1)
                k <<= 1
            keys[c0:c0 + chunk] = s
        k = chunk * 2
        while k <= n:                                       # global passes + k_sort_merge_local
            j = k >> 1
            while j >= chunk:
                for t in range(n // 2):
                    i = 2 * t - (t & (j - 1))
                    if (keys[i] > keys[i + j]) == ((i & k) == 0):
                        keys[i], keys[i + j] = keys[i + j], keys[i]
                j >>= 1
            for c0 in range(0, n, chunk):
                s = keys[c0:c0 + chunk]
                step(s, chunk, c0, k, chunk >> 1)
                keys[c0:c0 + chunk] = s
            k <<= 1
        assert keys == ref, (n, chunk)


def test_bench_kernel_trace_reduction(tmp_path):
    """bench.py's rocprofv3 leg: `roofline.frac` comes from the kernel trace of a child run; the reduction keeps the
    dispatches of the LAST n steps only (optimizer launch to optimizer launch) and averages per kernel name."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rows = ['"Kind","Agent_Id","Queue_Id","Kernel_Name","Start_Timestamp","End_Timestamp"']
    t = 1000
    # model build noise, then 5 steps: conv of 50 us in the first two (settling), 40 / 42 / 44 us in the last three
    rows.append('"KERNEL_DISPATCH",1,1,"__amd_rocclr_copyBuffer",%d,%d' % (t, t + 500)); t += 1000
    for dur in (50000, 50000, 40000, 42000, 44000):
        for _ in range(2):
            rows.append('"KERNEL_DISPATCH",1,1,"void k_conv_fwd<128, 128, false>(lmh_conv_desc, float const*)",%d,%d' % (t, t + dur))
            t += dur + 1000
        rows.append('"KERNEL_DISPATCH",1,1,"k_sgd_momentum(float*, float const*)",%d,%d' % (t, t + 10000)); t += 11000
    f = tmp_path / 'rp_kernel_trace.csv'
    f.write_text('\n'.join(rows) + '\n')
    out = bench.reduce_kernel_trace(str(f), 3)
    k = out['k_conv_fwd<128,128,false>']
    assert abs(k['avg_ms'] - 0.042) < 1e-9 and k['calls_per_step'] == 2.0
    assert out['k_sgd_momentum']['calls_per_step'] == 1.0
    assert '__amd_rocclr_copyBuffer' not in out
    assert abs(out['_step_ms'] - (2 * 42000 + 2 * 1000 + 11000) * 1e-6) < 1e-9
    assert bench.reduce_kernel_trace(str(f), 5) is None          # needs n + 1 optimizer launches
    # per-range updates (LUMINOTH_AMD_EARLY_UPDATE=1): early range updates under the backward and SEVERAL
    # consecutive optimizer launches at the end of a step are ONE step end
    rows, t = [rows[0]], 1000
    for dur in (50000, 50000, 40000, 42000, 44000):
        rows.append('"KERNEL_DISPATCH",1,1,"void k_conv_fwd<128, 128, false>(lmh_conv_desc, float const*)",%d,%d' % (t, t + dur)); t += dur + 1000
        rows.append('"KERNEL_DISPATCH",1,1,"k_sgd_early_range(float*)",%d,%d' % (t, t + 3000)); t += 4000
        rows.append('"KERNEL_DISPATCH",1,1,"void k_conv_fwd<128, 128, false>(lmh_conv_desc, float const*)",%d,%d' % (t, t + dur)); t += dur + 1000
        for _ in range(3):
            rows.append('"KERNEL_DISPATCH",1,1,"k_sgd_momentum_range(float*)",%d,%d' % (t, t + 3000)); t += 4000
    f.write_text('\n'.join(rows) + '\n')
    out = bench.reduce_kernel_trace(str(f), 3)
    assert out['k_conv_fwd<128,128,false>']['calls_per_step'] == 2.0 and abs(out['k_conv_fwd<128,128,false>']['avg_ms'] - 0.042) < 1e-9
    assert out['k_sgd_momentum_range']['calls_per_step'] == 3.0 and out['k_sgd_early_range']['calls_per_step'] == 1.0


def test_bench_step_block_sources(tmp_path, monkeypatch):
    """`roofline.step`: counter bytes per step come from the newest committed PMC summary of the SAME
    workload and dtype, algorithmic bytes from tools/flops.py (~4.7 GB for ResNet-50 at 2 x 1024^2)."""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_mod2', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    (tmp_path / 'profiles').mkdir()
    json.dump({'workload': 'frcnn_r50', 'dtype': 'f32', 'step': {'hbm_bytes': 1.0e10, 'launches': 250}, 'kernels': {}},
              open(tmp_path / 'profiles' / 'r01_a_pmc_traffic.json', 'w'))
    json.dump({'workload': 'frcnn_r50', 'dtype': 'bf16x3', 'step': {'hbm_bytes': 9.0e9, 'launches': 226}, 'kernels': {}},
              open(tmp_path / 'profiles' / 'r02_b_pmc_traffic.json', 'w'))
    json.dump({'kernels': {}}, open(tmp_path / 'profiles' / 'r03_c_pmc_traffic.json', 'w'))       # an old-layout file: skipped
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    st, src = bench.pmc_step_traffic('frcnn_r50', 'bf16x3')
    assert st['hbm_bytes'] == 9.0e9 and src.endswith('r02_b_pmc_traffic.json')
    assert bench.pmc_step_traffic('frcnn_r50', 'f32')[0]['launches'] == 250
    assert bench.pmc_step_traffic('frcnn_r50_coco', 'f16') == (None, None)

    class _M(object):
        class base_network(object):
            storage_dtype = None

        class store(object):
            class flat(object):
                @staticmethod
                def numel():
                    return 13500000
    b = bench.algorithmic_step_bytes(bench.WORKLOADS['frcnn_r50'], 'f32', _M)
    assert 4.3e9 < b < 5.1e9, b
    _M.base_network.storage_dtype = 'f16'
    assert bench.algorithmic_step_bytes(bench.WORKLOADS['frcnn_r50_coco'], 'f16', _M) < 0.6 * b
    assert bench.algorithmic_step_bytes(bench.WORKLOADS['ssd300_b32'], 'f32', _M) is None


def test_bench_exchange_ladder():
    """bench.py --gpus N: the fall-back ladder bucketed + plan -> one all-reduce + plan -> one all-reduce, eager (first contact
    with an 8-GPU node must not end in an empty record).  Pure host logic: exercised here with stand-in attempts."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_mod2', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    B, NB = {'bucketed_allreduce_under_backward': True, 'launch_plan': True}, {'bucketed_allreduce_under_backward': False, 'launch_plan': True}
    E_ = {'bucketed_allreduce_under_backward': False, 'launch_plan': False}

    def scripted(outcomes):
        seen = []

        def attempt(mode):
            seen.append(dict(mode))
            o = outcomes[len(seen) - 1]
            if isinstance(o, Exception):
                raise o
            return dict(o)
        return attempt, seen

    # the bucketed exchange raises: ONE fall-back, the second mode produces the result
    a, seen = scripted([bench.StepFailure('boom'), {'dt': 1.0, 'replicas_identical': True}])
    res, mode, fb = bench.run_ladder(8, a, environ={})
    assert res['dt'] == 1.0 and mode == NB and seen == [B, NB]
    assert fb == [dict(B, error='boom')]
    # diverged replicas in the first two modes: eager launches with one all-reduce produce the number
    a, seen = scripted([{'replicas_identical': False}, {'replicas_identical': False}, {'dt': 2.0, 'replicas_identical': True}])
    res, mode, fb = bench.run_ladder(2, a, environ={})
    assert res['dt'] == 2.0 and mode == E_ and [f['error'] for f in fb] == ['replicas diverged after the timed steps'] * 2
    # the LAST mode is reported even when it diverged (the line then says so), and a ladder that only raises returns None
    a, _ = scripted([bench.StepFailure('a'), bench.StepFailure('b'), {'dt': 3.0, 'replicas_identical': False}])
    res, mode, fb = bench.run_ladder(4, a, environ={})
    assert res['replicas_identical'] is False and mode == E_ and len(fb) == 2
    a, _ = scripted([bench.StepFailure('a'), bench.StepFailure('b'), bench.StepFailure('c')])
    assert bench.run_ladder(4, a, environ={})[0] is None
    # one GPU: one mode, no bucketed exchange, the user's environment decides about the plan
    a, seen = scripted([{'dt': 4.0}])
    res, mode, fb = bench.run_ladder(1, a, environ={'LUMINOTH_AMD_PLAN': '0'})
    assert mode == {'bucketed_allreduce_under_backward': False, 'launch_plan': False} and fb == [] and len(seen) == 1
    # a pre-set environment is respected by the first mode
    a, seen = scripted([{'dt': 5.0, 'replicas_identical': True}])
    res, mode, fb = bench.run_ladder(8, a, environ={'LUMINOTH_AMD_BUCKETED_ALLREDUCE': '0'})
    assert mode == NB


def test_step_state_helpers():
    """Host pieces of the per-shape step state (round 5): the gt CAPACITY of a state's fixed buffers (the gt count of a
    batch no longer creates states), the in-place version fingerprint of a gt argument (a loader that refills its buffers is not
    served a stale copy), and the bytes a recorded plan pins (distinct storages, nested containers)."""
    import numpy as np
    import torch
    from luminoth_amd import plan as P
    from luminoth_amd.models.fasterrcnn.fasterrcnn import FasterRCNN
    assert [FasterRCNN._gt_bucket(g) for g in (0, 1, 8, 9, 16, 17, 100)] == [8, 8, 8, 16, 16, 32, 128]
    t = torch.zeros(3, 5)
    v0 = FasterRCNN._gt_versions(t)
    t.add_(1.0)
    assert FasterRCNN._gt_versions(t) != v0
    pair = (torch.zeros(2, 4, 5), torch.zeros(2, dtype=torch.int32))
    f0 = FasterRCNN._gt_versions(pair)
    pair[1].fill_(3)
    f1 = FasterRCNN._gt_versions(pair)
    assert f0 != f1 and f0[0] == f1[0]
    assert FasterRCNN._gt_versions([np.zeros((2, 5), np.float32)]) == (None,) and FasterRCNN._gt_versions(None) is None

    class FakePlan(object):
        pass
    base = torch.zeros(1000)
    pl = FakePlan()
    pl.keep = [base, base[10:20], {'a': torch.zeros(8, dtype=torch.float64), 'b': (base[:5], [torch.zeros(3)])}, 7, None]
    assert P.pinned_bytes(pl) == 4000 + 64 + 12          # views share their storage; counted once
    assert P.pinned_bytes(pl) == pl._pinned
