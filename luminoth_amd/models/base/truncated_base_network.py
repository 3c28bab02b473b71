"""TruncatedBaseNetwork — feature extractor truncated at an endpoint, with the
ResNet-101 `block4` tail for pooled ROIs.  Mirrors the public surface of
luminoth/models/base/truncated_base_network.py:19-169:
`__call__(inputs, is_training)`, `_build_tail(inputs, is_training)`,
`get_trainable_vars()`; default endpoints from :8-16.
"""
import torch

from luminoth_amd.models.base import layers as L
from luminoth_amd.models.base import networks
from luminoth_amd.kernels import COMPUTE as K_COMPUTE
from luminoth_amd.models.base.base_network import BaseNetwork, he_normal, ones, zeros

DEFAULT_ENDPOINTS = {
    'resnet_v1_50': 'block3', 'resnet_v1_101': 'block3', 'resnet_v1_152': 'block3',
    'resnet_v2_50': 'block3', 'resnet_v2_101': 'block3', 'resnet_v2_152': 'block3',
    'vgg_16': 'conv5/conv5_3',
}


class _TrunkFn(torch.autograd.Function):
    """Autograd boundary around the trainable part of a Trunk.  Parameter
    gradients are written straight into the flat gradient buffer by the node
    backward passes; autograd only routes the activation gradient."""

    @staticmethod
    def forward(ctx, x, anchor, trunk, start, need_dx):
        y, saved = trunk.forward(x, save_from=0)
        ctx.trunk, ctx.saved, ctx.need_dx = trunk, saved, need_dx
        return y

    @staticmethod
    def backward(ctx, dy):
        dx = ctx.trunk.backward(ctx.saved, dy.contiguous(), 0, need_dx_first=ctx.need_dx)
        ctx.saved = None
        return dx, None, None, None, None


class TruncatedBaseNetwork(BaseNetwork):
    def __init__(self, config, name='truncated_base_network', build_unused_tail_vars=True):
        super(TruncatedBaseNetwork, self).__init__(config, name=name)
        self._endpoint = config.get('endpoint') or DEFAULT_ENDPOINTS[config.get('architecture')]
        self._freeze_tail = config.get('freeze_tail')
        self._use_tail = config.get('use_tail')
        wd = self._weight_decay()
        arch = self._architecture
        in_sub = self._in_sub()
        self._in_sub_vals = in_sub
        if self.resnet_v1_type:
            # slim builds the whole net (block4 included) even though only the
            # endpoint is used: its variables exist and are regularised.
            nodes, endpoints = networks.resnet_v1_nodes(arch, name, wd, he_normal,
                                                        output_stride=config.get('output_stride'))
            if self._endpoint not in endpoints:
                raise ValueError('"{}" is an invalid value of endpoint for this architecture.'.format(
                    '%s/%s/%s' % (name, arch, self._endpoint)))
            cut = endpoints[self._endpoint] + 1
            self._all_nodes = nodes
            self.trunk = L.Trunk(nodes[:cut])
            self._unused_nodes = nodes[cut:]
            self.tail = None
            if arch == 'resnet_v1_101':
                # same variables as the (unused) trunk block4, applied at stride 1 / rate 1
                self.tail = L.Trunk(networks.resnet_v1_tail_nodes(arch, name, wd, he_normal))
            self.feat_channels = self.trunk.nodes[-1].conv3.cout
            self.tail_channels = 2048 if (self.tail is not None and self._use_tail) else self.feat_channels
        elif self.resnet_v2_type:
            # base_network.py:94-101: slim resnet_v2 (pre-activation bottlenecks); no pooled-ROI tail (the reference only
            # builds one for resnet_v1_101, truncated_base_network.py:58)
            nodes, endpoints, unused = networks.resnet_v2_nodes(arch, name, wd, he_normal, zeros,
                                                                output_stride=config.get('output_stride'))
            if self._endpoint not in endpoints:
                raise ValueError('"{}" is an invalid value of endpoint for this architecture.'.format(
                    '%s/%s/%s' % (name, arch, self._endpoint)))
            cut = endpoints[self._endpoint] + 1
            self._all_nodes = nodes
            self.trunk = L.Trunk(nodes[:cut])
            self._unused_nodes = nodes[cut:]
            self._unused_layers = unused
            self.tail = None
            self.feat_channels = self.trunk.nodes[-1].conv3.cout
            self.tail_channels = self.feat_channels
        elif self.vgg_type or self.truncated_vgg_type:
            nodes, endpoints = networks.vgg16_nodes(name, arch, wd, he_normal, zeros, in_sub=None)
            if self._endpoint not in endpoints:
                raise ValueError('"{}" is an invalid value of endpoint for this architecture.'.format(
                    '%s/%s/%s' % (name, arch, self._endpoint)))
            cut = endpoints[self._endpoint] + 1
            self._all_nodes = nodes[:cut]
            self.trunk = L.Trunk(nodes[:cut])
            self._unused_nodes = []
            # full slim vgg_16 (not the SSD 'truncated_vgg_16'): the classifier's variables exist and are regularised
            self._unused_layers = networks.vgg16_unused_fc_layers(name, arch, wd, he_normal) if self.vgg_type else []
            self.tail = None
            self.feat_channels = self.trunk.nodes[-1].layer.cout
            self.tail_channels = self.feat_channels
        self.bn_table = L.BNTable()
        self._in_sub = None
        # extension key: model.base_network.compute_dtype in {None, 'f32', 'f16', 'bf16'} — half-precision MFMA
        # operands with fp32 accumulation for every backbone / tail convolution (BASELINE configs[4])
        self.compute_dtype = config.get('compute_dtype')
        if config.get('storage_dtype') in ('f16', 'bf16') and self.compute_dtype in (None, 'f32', 'fp32', 'float32'):
            self.compute_dtype = config.get('storage_dtype')         # half storage implies half MFMA operands
        if self.compute_dtype not in (None, 'f32', 'fp32', 'float32'):
            if self.compute_dtype not in K_COMPUTE:
                raise ValueError('Invalid compute_dtype: "{}"'.format(self.compute_dtype))
            for layer in self._creation_order_layers() + (self.tail.all_layers() if self.tail else []):
                layer.compute = self.compute_dtype
        # extension key: model.base_network.storage_dtype in {None, 'f16', 'bf16'} — the trunk keeps its activations,
        # activation gradients and working weight copies as 16-bit tensors in HBM (csrc/conv_hs.h; SURVEY.md 8(d) config 5:
        # "fp16 activations/weights with fp32 accumulate + fp32 master weights").  The fp32 stem convolution feeds a
        # max-pool that writes the half tensor; the last trunk layer hands the feature map on as fp32.
        self.storage_dtype = config.get('storage_dtype')
        self._hs_layers = []
        self.extra_hs_layers = []     # half-storage layers of the heads (the RPN 3x3 convolution): their working weight
        #                               copies are refreshed in the same launch as the trunk's
        if self.storage_dtype not in (None, 'f32', 'fp32', 'float32'):
            if self.storage_dtype not in ('f16', 'bf16'):
                raise ValueError('Invalid storage_dtype: "{}"'.format(self.storage_dtype))
            if K_COMPUTE[self.compute_dtype] != K_COMPUTE[self.storage_dtype]:
                raise ValueError('storage_dtype "{}" needs the same compute_dtype (got "{}")'.format(
                    self.storage_dtype, self.compute_dtype))
            if not self.resnet_v1_type:
                raise NotImplementedError('storage_dtype: implemented for the resnet_v1 trunks (BASELINE configs[4] is '
                                          'ResNet-50; VGG / SSD / resnet_v2 keep fp32 tensors)')
            # ResNet-101 (round 4): the TRUNK keeps 16-bit tensors; the block4 tail runs on the fp32 ROI crops of the fp32
            # feature map with the same 16-bit MFMA operands but fp32 tensors (conv_half.h) — its layers keep storage None
            nodes = self.trunk.nodes
            if not (isinstance(nodes[1], L.MaxPoolNode) and all(isinstance(n, L.BottleneckNode) for n in nodes[2:])):
                raise NotImplementedError('storage_dtype: unexpected trunk structure')
            nodes[1].storage = self.storage_dtype
            for n in nodes[2:]:
                for layer in n.layers:
                    if layer.cin % 64 or layer.cout % 64:
                        raise NotImplementedError('storage_dtype: channel counts must be multiples of 64 (%s)' % layer.scope)
                    layer.compute = layer.storage = self.storage_dtype
                    self._hs_layers.append(layer)
            nodes[-1].conv3.hs_out_f32 = True

    # ---- variables --------------------------------------------------------------
    def _creation_order_layers(self):
        return [l for n in self._all_nodes for l in n.layers] + list(getattr(self, '_unused_layers', []))

    def get_trainable_var_names(self):
        """truncated_base_network.py:97-144: fine-tune range cut at the last
        variable containing the endpoint, plus block4 for the R101 tail."""
        all_tr = super(TruncatedBaseNetwork, self).get_trainable_var_names()
        idx = None
        for i, n in enumerate(all_tr):
            if self._endpoint in n:
                idx = i
        names = [] if idx is None else all_tr[:idx + 1]
        if self._use_tail and not self._freeze_tail and self._architecture == 'resnet_v1_101':
            for i, n in enumerate(all_tr):
                if 'block4' in n:
                    names += all_tr[i:]
                    break
            else:
                raise ValueError('"block4" not present in the trainable vars retrieved from base network.')
        return names

    def register(self, store, base_trainable=True):
        """Declare every variable; decides trainability like the reference."""
        tr = set(self.get_trainable_var_names()) if base_trainable else set()
        self._trainable_names = tr
        tail_layers = {l.scope: l for l in (self.tail.all_layers() if self.tail else [])}
        for layer in self._creation_order_layers():
            if isinstance(layer, L.PreactLayer):         # resnet_v2 `preact` / `postnorm`: a BatchNorm without a convolution
                layer.trainable = ('%s/gamma' % layer.scope) in tr
                self.bn_table.add(layer.scope, layer.cout, layer.trainable, prefix=layer.scope)
                continue
            layer.trainable = layer.w_name in tr
            store.add(layer.w_name, (layer.k, layer.k, layer.cin, layer.cout), layer.init or he_normal,
                      trainable=layer.trainable, wd=layer.wd)
            if layer.norm == 'bn':
                self.bn_table.add(layer.scope, layer.cout, ('%s/BatchNorm/gamma' % layer.scope) in tr)
            elif layer.norm == 'bias':
                store.add(layer.b_name, (layer.cout,), zeros, trainable=layer.b_name in tr)
            if layer.scope in tail_layers:
                tail_layers[layer.scope].trainable = layer.trainable
        self.bn_table.register(store, ones, zeros)

    def bind(self, store):
        self.store = store
        self.bn_table.bind(store)
        for layer in self._creation_order_layers():
            layer.bind(store, self.bn_table)
        if self.tail is not None:
            for layer in self.tail.all_layers():
                layer.bind(store, self.bn_table)
        if self._in_sub_vals is not None:
            self._in_sub = torch.tensor(self._in_sub_vals, dtype=torch.float32, device=store.flat.device)
            first = self.trunk.nodes[0]
            if hasattr(first, 'in_sub'):
                first.in_sub = self._in_sub
        self._anchor = torch.zeros(1, device=store.flat.device, requires_grad=True)

    # ---- forward ------------------------------------------------------------------
    def set_bn_mode(self, is_training):
        """base_network.py:82-93 / truncated_base_network.py:61-76: with `train_batch_norm: True` every BatchNorm of the
        network (frozen blocks included — slim hands `is_training` to all of them) uses the statistics of the batch while
        training and advances its moving averages; otherwise the frozen-statistics path.  Statistics are per process:
        under data parallelism every replica normalises with its own batch, like the reference's per-worker graphs.
        -> whether the training-mode path is on."""
        # resnet_v2: base_network.py:94-101 hands `is_training` itself to slim (not gated by train_batch_norm)
        on = bool(is_training and (self._config.get('train_batch_norm') or self.resnet_v2_type))
        if on and not self.resnet_type:
            raise NotImplementedError('train_batch_norm: only the resnet networks have BatchNorm layers')
        for layer in self._creation_order_layers() + (self.tail.all_layers() if self.tail else []):
            layer.bn_train = on and layer.norm == 'bn'
        return on

    def _run(self, trunk, x, is_training):
        start = trunk.first_trainable()
        needs_grad = torch.is_grad_enabled() and (start < len(trunk.nodes) or x.requires_grad)
        if not needs_grad:
            y, _ = trunk.forward(x, save_from=None)
            return y
        if not x.requires_grad:
            # frozen prefix: forward only, nothing saved
            pre = L.Trunk(trunk.nodes[:start])
            x, _ = pre.forward(x, save_from=None)
            sub = L.Trunk(trunk.nodes[start:])
            return _TrunkFn.apply(x, self._anchor, sub, 0, False)
        return _TrunkFn.apply(x, self._anchor, trunk, 0, True)

    def feature_hw(self, H, W):
        """Spatial size of the feature map for an (H, W) input (no launch)."""
        return self.trunk.out_hw(H, W)

    def __call__(self, inputs, is_training=False):
        """inputs (B,H,W,3) fp32 RGB 0..255 -> feature map (B,fh,fw,C)."""
        self.set_bn_mode(is_training)
        self.bn_table.refresh()
        if self._hs_layers:
            L.prepare_half_weights(self._hs_layers + self.extra_hs_layers, self.storage_dtype)   # weights may have changed
        return self._run(self.trunk, inputs.contiguous(), is_training)

    @property
    def has_tail(self):
        return bool(self._use_tail and self.tail is not None)

    def _build_tail(self, inputs, is_training=False):
        if not self._use_tail or self.tail is None:
            return inputs
        self.set_bn_mode(is_training)
        return self._run(self.tail, inputs, is_training)
