"""Eager numpy stand-in for the slice of TensorFlow 1.x + Sonnet that the reference's hot-path GRAPH code uses.

Test infrastructure only (used by tests/golden/make_golden_ref_tf.py to RUN the reference's own
`luminoth/models/{fasterrcnn,ssd}/*.py` and `luminoth/utils/*_tf.py` in the build container, where TensorFlow is
not installed).  A "tensor" is a plain numpy array; every op executes immediately.

What is restated here and what is not:

  * structural ops (`where`, `boolean_mask`, `gather`, `gather_nd`, `scatter_nd`, `sparse_to_dense`, `unique`,
    `nn.top_k`, `cond`, `one_hot`, reductions, comparisons, ...) follow the TF 1.x documented semantics — the ones
    whose corner cases matter are unit-tested in tests/test_tf_shim.py (`sparse_to_dense` duplicate handling and
    `validate_indices`, `where` with a vector condition over matrix operands, `top_k` tie order, `unique` order,
    `scatter_nd` accumulation);
  * arithmetic stays in the operand dtype the way TF's strict typing forces it: python scalars adopt the tensor's
    dtype (numpy >= 2 weak-scalar promotion does exactly that); mixing two different array dtypes in a binary op
    RAISES like TF does (`_same`), so a silent float64 promotion cannot creep into a fixture;
  * `image.non_max_suppression` and `image.crop_and_resize` are TensorFlow C++ kernels (third party, not under
    /root/reference): they are taken from `oracle/tfops.py`, the restatement SURVEY.md §8(c) anchors on the
    reference's call sites — the fixtures pin the reference's COMPOSITION around them, not those kernels;
  * `random_shuffle` is a hook (`set_random_shuffle`): TF's Philox stream is not reproducible, any permutation is a
    valid outcome, so the fixture generator installs the permutation that corresponds to the shared counter RNG
    (`oracle/rng.py` == `lmh_hash_u32`) and the reference's own subsample code runs on it unmodified;
  * `summary`, `name_scope`, `variable_scope`, `control_dependencies`, `logging` are inert;
  * `snt.Conv2D` / `snt.Linear` (round 5: the reference's head `_build`s — rpn.py:148-172, rcnn.py:185-239,
    ssd/ssd.py:73-109 — are executed too) are direct numpy convolutions / matrix products in float32 with Sonnet's
    variable layout (`w`: HWIO / (in, out), `b`: (out,)), SAME = TF's padding rule; their variables come from
    `seeded_variable(module name, 'w' | 'b', shape)` — a fixed function of the NAME, so the generator and the tests
    that replay a fixture rebuild the same weights without storing them (initializers / regularizers are accepted
    and ignored: the layouts are what is pinned, not the init distribution).

  * round 6 (the reference's TOP-LEVEL composition is executed too: fasterrcnn.py:22-259,337-358 and
    models/base/{base_network,truncated_base_network}.py): variable scopes are real (`variable_scope` pushes names; a Sonnet
    module takes its scope where it is CONSTRUCTED and enters it when called), variables are objects with TensorFlow names
    (`Variable`: `.name` = '<scope>/<var>:0', `.op.name`, collections) kept in creation order, so
    `snt.get_variables_in_module`, `tf.get_collection(MODEL_VARIABLES, scope)`, `module.variable_scope.name` and the
    reference's own `get_trainable_vars` / `get_base_network_checkpoint_vars` run unmodified; `l2_regularizer` is real and,
    while `track_regularizers(True)` is in effect, a layer built with `regularizers={'w': fn}` adds fn(w) to the
    regularization losses and its variable name to `regularized_names()` (off by default: the round-5 fixtures were made
    with inert regularizers and stay byte-identical);
  * `expand_dims` returns a `Tensor` view that applies TensorFlow's rule for `ndarray + tensor`: the ndarray is converted
    to the TENSOR's dtype (float -> int truncates toward zero) — fasterrcnn.py:299-302 adds the float64 anchor reference to
    an int32 grid that way, which is what makes `all_anchors` int32.

`install()` puts `tensorflow`, `sonnet` and `easydict` stand-ins into `sys.modules`; the slim stand-in of the top-level
fixtures lives in tests/golden/slim_standin.py.
"""
import builtins
import collections
import contextlib
import sys
import types

import numpy as np

float32, float64, int32, int64, bool = np.float32, np.float64, np.int32, np.int64, np.bool_   # noqa: A001
uint8 = np.uint8

_shuffle_hook = None
_losses = []
_reg_losses = []


def set_random_shuffle(fn):
    """fn(value, seed, caller_name) -> permuted value (along axis 0)."""
    global _shuffle_hook
    _shuffle_hook = fn


def reset_losses():
    del _losses[:]
    del _reg_losses[:]


# ------------------------------------------------------------------------------------------------ helpers ----
def _t(x, dtype=None):
    if isinstance(x, np.ndarray) and dtype is None:
        return x
    if isinstance(x, (list, tuple)) and dtype is None and len(x) and all(isinstance(v, np.ndarray) for v in x):
        return np.stack(x)
    a = np.asarray(x, dtype=dtype)
    if dtype is None and not isinstance(x, (np.ndarray, np.generic)):
        # python literals: TF defaults (float -> float32, int -> int32)
        if a.dtype == np.float64:
            a = a.astype(np.float32)
        elif a.dtype == np.int64:
            a = a.astype(np.int32)
    return a


def _same(x, y):
    """Binary-op operand coercion with TF's rule: a python scalar / list adopts the tensor's dtype, two tensors must
    already agree."""
    xa, ya = isinstance(x, (np.ndarray, np.generic)), isinstance(y, (np.ndarray, np.generic))
    if xa and ya:
        if x.dtype != y.dtype:
            raise TypeError('tf shim: dtype mismatch %s vs %s (TensorFlow would raise)' % (x.dtype, y.dtype))
        return x, y
    if xa:
        return x, np.asarray(y, dtype=x.dtype)
    if ya:
        return np.asarray(x, dtype=y.dtype), y
    return _t(x), _t(y)


def convert_to_tensor(value, dtype=None, name=None):
    return _t(value, dtype)


def constant(value, dtype=None, shape=None, name=None):
    return _t(value, dtype)


identity = stop_gradient = lambda x, name=None: _t(x)   # noqa: E731


def cast(x, dtype, name=None):
    x = _t(x)
    if np.issubdtype(x.dtype, np.floating) and np.issubdtype(np.dtype(dtype), np.integer):
        return np.trunc(x).astype(dtype)
    return x.astype(dtype)


def to_float(x, name=None):
    return cast(x, np.float32)


def to_int32(x, name=None):
    return cast(x, np.int32)


def shape(x, name=None, out_type=np.int32):
    return np.array(np.shape(x), dtype=out_type)


def size(x, name=None, out_type=np.int32):
    return np.array(np.size(x), dtype=out_type)


def reshape(x, shape, name=None):   # noqa: A002
    shape = [int(s) for s in np.asarray(shape).reshape(-1)] if not isinstance(shape, (list, tuple)) else \
        [int(s) for s in shape]
    return np.reshape(_t(x), shape)


def squeeze(x, axis=None, name=None):
    return np.squeeze(_t(x), axis=None if axis is None else tuple(np.atleast_1d(axis)))


def expand_dims(x, axis, name=None):
    return Tensor(np.expand_dims(_t(x), axis))      # (a graph tensor: `ndarray + tensor` converts the ndarray, see Tensor)


def transpose(x, perm=None, name=None):
    return np.transpose(_t(x), perm)


def concat(values, axis, name=None):
    values = [_t(v) for v in values]
    dt = values[0].dtype
    for v in values:
        if v.dtype != dt:
            raise TypeError('tf shim: concat dtype mismatch')
    return np.concatenate(values, axis=axis)


def stack(values, axis=0, name=None):
    return np.stack([_t(v) for v in values], axis=axis)


def unstack(x, num=None, axis=0, name=None):
    x = _t(x)
    return [np.take(x, i, axis=axis) for i in builtins.range(x.shape[axis])]


def split(x, num_or_size_splits, axis=0, name=None):
    return np.split(_t(x), num_or_size_splits, axis=axis)


def tile(x, multiples, name=None):
    return np.tile(_t(x), [int(m) for m in np.asarray(multiples).reshape(-1)])


def reverse(x, axis, name=None):
    return np.flip(_t(x), axis=tuple(np.atleast_1d(axis)))


def fill(dims, value, name=None):
    dims = [int(d) for d in np.asarray(dims).reshape(-1)]
    return np.full(dims, _t(value))


def zeros(shape, dtype=np.float32, name=None):   # noqa: A002
    return np.zeros([int(d) for d in np.asarray(shape).reshape(-1)], dtype=dtype)


def ones(shape, dtype=np.float32, name=None):   # noqa: A002
    return np.ones([int(d) for d in np.asarray(shape).reshape(-1)], dtype=dtype)


def zeros_like(x, dtype=None, name=None):
    return np.zeros_like(_t(x), dtype=dtype)


def ones_like(x, dtype=None, name=None):
    return np.ones_like(_t(x), dtype=dtype)


def range(*a, **k):   # noqa: A001
    k.pop('name', None)
    a = [int(v) for v in a]
    return np.arange(*a, dtype=k.get('dtype', np.int32))


def meshgrid(*args, **kwargs):
    return np.meshgrid(*args, indexing=kwargs.get('indexing', 'xy'))


# ------------------------------------------------------------------------------------------- elementwise ----
def _bin(f):
    def op(x, y, name=None):
        x, y = _same(x, y)
        return f(x, y)
    return op


add = _bin(np.add)
subtract = _bin(np.subtract)
multiply = _bin(np.multiply)
maximum = _bin(np.maximum)
minimum = _bin(np.minimum)
greater = _bin(np.greater)
greater_equal = _bin(np.greater_equal)
less = _bin(np.less)
less_equal = _bin(np.less_equal)
equal = _bin(np.equal)
not_equal = _bin(np.not_equal)
logical_and = _bin(np.logical_and)
logical_or = _bin(np.logical_or)


def divide(x, y, name=None):
    x, y = _same(x, y)
    return x / y


def logical_not(x, name=None):
    return np.logical_not(x)


def negative(x, name=None):
    return np.negative(_t(x))


def exp(x, name=None):
    return np.exp(_t(x))


def log(x, name=None):
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.log(_t(x))


def sqrt(x, name=None):
    return np.sqrt(_t(x))


def square(x, name=None):
    return np.square(_t(x))


def abs(x, name=None):   # noqa: A001
    return np.abs(_t(x))


# -------------------------------------------------------------------------------------------- reductions ----
def _axis(axis):
    if axis is None:
        return None
    return tuple(int(a) for a in np.atleast_1d(axis))


def _red(f):
    def op(x, axis=None, keepdims=False, name=None, reduction_indices=None, keep_dims=None):
        if reduction_indices is not None:
            axis = reduction_indices
        if keep_dims is not None:
            keepdims = keep_dims
        x = _t(x)
        with np.errstate(invalid='ignore', divide='ignore'):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                r = f(x, axis=_axis(axis), keepdims=keepdims)
        return np.asarray(r, dtype=x.dtype) if np.issubdtype(x.dtype, np.floating) else np.asarray(r)
    return op


reduce_max = _red(np.max)
reduce_prod = _red(np.prod)
reduce_min = _red(np.min)
reduce_sum = _red(np.sum)
reduce_any = _red(np.any)
reduce_all = _red(np.all)


def reduce_mean(x, axis=None, keepdims=False, name=None):
    x = _t(x)
    if np.issubdtype(x.dtype, np.integer):          # TF: integer mean truncates
        return np.asarray(np.sum(x, axis=_axis(axis), keepdims=keepdims) // max(1, x.size), dtype=x.dtype)
    if x.size == 0:
        return np.asarray(np.nan, dtype=x.dtype)    # TF: mean of nothing is NaN
    return np.asarray(np.mean(x, axis=_axis(axis), keepdims=keepdims, dtype=x.dtype), dtype=x.dtype)


def count_nonzero(x, axis=None, keepdims=False, dtype=np.int64, name=None):
    return np.asarray(np.count_nonzero(_t(x), axis=_axis(axis)), dtype=dtype)


def argmax(x, axis=None, name=None, dimension=None, output_type=np.int64):
    if dimension is not None:
        axis = dimension
    return np.argmax(_t(x), axis=0 if axis is None else axis).astype(output_type)      # first occurrence, like TF


# ------------------------------------------------------------------------------ gather / scatter / select ----
def where(condition, x=None, y=None, name=None):
    condition = np.asarray(condition, dtype=np.bool_)
    if x is None and y is None:
        return np.argwhere(condition).astype(np.int64).reshape(-1, max(condition.ndim, 1))
    x, y = _same(_t(x), _t(y))
    if x.shape != y.shape:
        raise ValueError('tf shim: tf.where needs x and y of the same shape')
    if condition.shape != x.shape:
        # TF 1.x: a VECTOR condition selects whole rows of higher-rank x / y; anything else is an error
        if not (condition.ndim == 1 and x.ndim > 1 and condition.shape[0] == x.shape[0]):
            raise ValueError('tf shim: tf.where condition shape %s vs %s' % (condition.shape, x.shape))
        condition = condition.reshape((-1,) + (1,) * (x.ndim - 1))
    return np.where(condition, x, y)


def boolean_mask(tensor, mask, name=None, axis=None):
    tensor, mask = _t(tensor), np.asarray(mask, dtype=np.bool_)
    if mask.shape != tensor.shape[:mask.ndim]:
        raise ValueError('tf shim: boolean_mask shapes %s vs %s' % (mask.shape, tensor.shape))
    return tensor[mask]


def gather(params, indices, validate_indices=None, name=None, axis=0):
    params, indices = _t(params), np.asarray(indices)
    if not np.issubdtype(indices.dtype, np.integer):
        raise TypeError('tf shim: gather indices must be integers')
    if indices.size and (indices.min() < 0 or indices.max() >= params.shape[axis]):
        raise IndexError('tf shim: gather index out of range (TF CPU raises InvalidArgument)')
    return np.take(params, indices, axis=axis)


def gather_nd(params, indices, name=None):
    params, indices = _t(params), np.asarray(indices)
    return params[tuple(np.moveaxis(indices, -1, 0))]


def scatter_nd(indices, updates, shape, name=None):   # noqa: A002
    indices, updates = np.asarray(indices), _t(updates)
    out = np.zeros([int(s) for s in np.asarray(shape).reshape(-1)], dtype=updates.dtype)
    np.add.at(out, tuple(np.moveaxis(indices, -1, 0)), updates)     # duplicates accumulate
    return out


def sparse_to_dense(sparse_indices, output_shape, sparse_values, default_value=0, validate_indices=True, name=None):
    """TF 1.x: indices 0-D / (n,) / (n, rank); `validate_indices` demands lexicographically increasing, unique
    indices (InvalidArgument otherwise); without validation the CPU kernel writes in order — the LAST duplicate wins."""
    idx = np.asarray(sparse_indices)
    oshape = [int(s) for s in np.asarray(output_shape).reshape(-1)]
    vals = _t(sparse_values)
    if isinstance(default_value, (np.ndarray, np.generic)) and vals.dtype != np.asarray(default_value).dtype:
        raise TypeError('tf shim: sparse_to_dense default/value dtype mismatch')
    out = np.full(oshape, np.asarray(default_value).astype(vals.dtype))
    if idx.ndim == 0:
        idx = idx.reshape(1, 1)
    elif idx.ndim == 1:
        idx = idx.reshape(-1, 1) if len(oshape) == 1 else idx.reshape(1, -1)
    if idx.shape[1] != len(oshape):
        raise ValueError('tf shim: sparse_to_dense index rank %d vs output rank %d' % (idx.shape[1], len(oshape)))
    if idx.size and ((idx < 0).any() or (idx >= np.asarray(oshape)[None, :]).any()):
        raise IndexError('tf shim: sparse_to_dense index out of bounds')
    if validate_indices and idx.shape[0] > 1:
        flat = np.ravel_multi_index(tuple(idx.T), oshape)
        if not (np.diff(flat) > 0).all():
            raise ValueError('tf shim: sparse_to_dense indices out of order or repeated (validate_indices=True)')
    vals_b = np.broadcast_to(vals, (idx.shape[0],)) if vals.ndim == 0 else vals
    for row, v in zip(idx, vals_b):
        out[tuple(row)] = v
    return out


_Unique = collections.namedtuple('Unique', ['y', 'idx'])
_TopK = collections.namedtuple('TopKV2', ['values', 'indices'])


def unique(x, out_idx=np.int32, name=None):
    x = _t(x)
    assert x.ndim == 1
    _, first, inv = np.unique(x, return_index=True, return_inverse=True)
    order = np.argsort(first, kind='stable')               # first-occurrence order
    rank = np.empty_like(order)
    rank[order] = np.arange(order.shape[0])
    return _Unique(x[np.sort(first)], rank[inv].astype(out_idx))


def one_hot(indices, depth, on_value=None, off_value=None, axis=None, dtype=None, name=None):
    indices = np.asarray(indices)
    dtype = dtype or np.float32
    out = np.zeros(indices.shape + (int(depth),), dtype=dtype)
    ok = (indices >= 0) & (indices < depth)                # out-of-range rows stay all-zero (TF)
    it = np.nonzero(ok)
    out[it + (indices[ok],)] = 1
    return out


def cond(pred, true_fn=None, false_fn=None, strict=False, name=None, fn1=None, fn2=None):
    true_fn, false_fn = true_fn or fn1, false_fn or fn2
    r = true_fn() if builtins.bool(np.asarray(pred)) else false_fn()
    return _t(r)


def random_shuffle(value, seed=None, name=None):
    if _shuffle_hook is None:
        raise RuntimeError('tf shim: install a permutation with set_random_shuffle() first')
    caller = sys._getframe(1).f_code.co_name
    out = _shuffle_hook(_t(value), seed, caller)
    assert sorted(map(tuple, np.asarray(out).reshape(out.shape[0], -1).tolist())) == \
        sorted(map(tuple, np.asarray(value).reshape(value.shape[0], -1).tolist())), 'hook must return a permutation'
    return out


def assert_positive(x, message=None, **k):
    assert (np.asarray(x) > 0).all(), message


def assert_non_negative(x, message=None, **k):
    assert (np.asarray(x) >= 0).all(), message


# ------------------------------------------------------------------------------------------- scopes etc. ----
@contextlib.contextmanager
def _scope(*a, **k):
    yield None


name_scope = control_dependencies = _scope


class GraphKeys(object):
    GLOBAL_VARIABLES = 'variables'
    TRAINABLE_VARIABLES = 'trainable_variables'
    MODEL_VARIABLES = 'model_variables'
    REGULARIZATION_LOSSES = 'regularization_losses'


_VS = ['']            # stack of ABSOLUTE variable-scope names ('' = the root)
_variables = []       # every Variable, in creation order (TensorFlow's order of `tf.trainable_variables()`)
_var_by_name = {}
_scope_names = set()
_track_reg = [False]
_reg_names = []


class _Op(object):
    def __init__(self, name):
        self.name = name


class Variable(object):
    """A model variable: `.name` ('<scope>/<var>:0'), `.op.name`, `.value` (numpy, float32), `.collections`."""

    def __init__(self, full_name, value, collections):
        self.name = full_name + ':0'
        self.op = _Op(full_name)
        self.value = value
        self.collections = set(collections)
        self.shape = _Shape(np.shape(value))

    def __repr__(self):
        return '<Variable %s %s>' % (self.name, tuple(self.shape))


class _VarScope(object):
    def __init__(self, name):
        self.name = name
        self.original_name_scope = name + '/'


@contextlib.contextmanager
def variable_scope(name_or_scope=None, default_name=None, values=None, reuse=None, **k):
    """Relative names nest under the current scope; a _VarScope object re-enters its absolute name."""
    if isinstance(name_or_scope, _VarScope):
        full = name_or_scope.name
    else:
        nm = name_or_scope if name_or_scope is not None else default_name
        full = (_VS[-1] + '/' + nm) if _VS[-1] else nm
    _VS.append(full)
    try:
        yield _VarScope(full)
    finally:
        _VS.pop()


def get_variable_scope():
    return _VarScope(_VS[-1])


def reset_variables():
    """Forget every variable, scope name and regularised-variable name (between fixture generators)."""
    del _variables[:]
    del _reg_names[:]
    _var_by_name.clear()
    _scope_names.clear()
    del _VS[1:]


def track_regularizers(on):
    _track_reg[0] = bool(on)


def regularized_names():
    return list(_reg_names)


def create_variable(name, make_value, trainable=True, model_variable=False, regularizer=None):
    """The variable `name` of the CURRENT scope: created on first use (creation order is kept), returned again afterwards
    (`reuse`).  make_value: () -> float32 array.  With track_regularizers(True), `regularizer(value)` joins the
    regularization losses when the variable is created (TF: once per variable)."""
    full = (_VS[-1] + '/' + name) if _VS[-1] else name
    v = _var_by_name.get(full)
    if v is not None:
        return v
    cols = [GraphKeys.GLOBAL_VARIABLES]
    if trainable:
        cols.append(GraphKeys.TRAINABLE_VARIABLES)
    if model_variable:
        cols.append(GraphKeys.MODEL_VARIABLES)
    v = Variable(full, np.asarray(make_value(), np.float32), cols)
    _variables.append(v)
    _var_by_name[full] = v
    if regularizer is not None and _track_reg[0]:
        add_regularization_loss(regularizer(v.value))
        _reg_names.append(full)
    return v


def get_collection(key, scope=None):
    return [v for v in _variables if key in v.collections and (scope is None or v.name.startswith(scope))]


def trainable_variables():
    return get_collection(GraphKeys.TRAINABLE_VARIABLES)


def get_variables_in_module(module, collection=GraphKeys.TRAINABLE_VARIABLES):
    """snt.get_variables_in_module: the variables under the module's scope, creation order, as a TUPLE."""
    prefix = module.variable_scope.name + '/'
    return tuple(v for v in _variables if collection in v.collections and v.name.startswith(prefix))


def l2_regularizer(scale, scope=None):
    """tf.contrib.layers.l2_regularizer: scale * tf.nn.l2_loss(w) = scale * sum(w^2) / 2, float32."""
    sc = np.float32(scale)

    def reg(w):
        w = np.asarray(w, np.float32)
        return np.float32(sc * np.float32(np.sum(np.square(w), dtype=np.float32) / np.float32(2)))
    return reg


class _Inert(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Inert(self.__name__ + '.' + name)

    def __call__(self, *a, **k):
        return None


summary = _Inert('tensorflow.summary')
logging = _Inert('tensorflow.logging')
contrib = _Inert('tensorflow.contrib')


def _flatten(x, *a, **k):
    x = _t(x)
    return x.reshape(x.shape[0], -1)


contrib.layers = _Inert('tensorflow.contrib.layers')
contrib.layers.flatten = _flatten          # tf.contrib.layers.flatten (rcnn.py:192): (N, ...) -> (N, prod(...)), row-major
contrib.layers.l2_regularizer = l2_regularizer


def _initializer(*a, **k):
    return None       # variables come from `seeded_variable`; see the module docstring


truncated_normal_initializer = random_normal_initializer = zeros_initializer = _initializer
train = _Inert('tensorflow.train')
flags = _Inert('tensorflow.flags')
app = _Inert('tensorflow.app')
gfile = _Inert('tensorflow.gfile')


# ------------------------------------------------------------------------------------------------- tf.nn ----
class _NN(object):
    @staticmethod
    def top_k(input, k=1, sorted=True, name=None):   # noqa: A002
        """Descending values; equal values -> lower index first (TF 1.x TopK CPU kernel)."""
        x = _t(input)
        assert x.ndim == 1
        k = int(k)
        if k > x.shape[0]:
            raise ValueError('tf shim: top_k k=%d > n=%d (TF raises InvalidArgument)' % (k, x.shape[0]))
        key = -x.astype(np.float64) if np.issubdtype(x.dtype, np.floating) else -x.astype(np.int64)
        order = np.argsort(key, kind='stable')[:k]
        return _TopK(x[order], order.astype(np.int32))

    @staticmethod
    def softmax(logits, axis=-1, name=None, dim=None):
        x = _t(logits)
        axis = dim if dim is not None else axis
        m = x.max(axis=axis, keepdims=True)
        e = np.exp(x - m)
        return (e / e.sum(axis=axis, keepdims=True)).astype(x.dtype)

    @staticmethod
    def softmax_cross_entropy_with_logits(_sentinel=None, labels=None, logits=None, dim=-1, name=None):
        """TF xent kernel: loss = sum(labels * (log(sum(exp(z))) - z)), z = logits - max."""
        labels, logits = _same(_t(labels), _t(logits))
        z = logits - logits.max(axis=-1, keepdims=True)
        lse = np.log(np.exp(z).sum(axis=-1, keepdims=True))
        return (labels * (lse - z)).sum(axis=-1).astype(logits.dtype)

    softmax_cross_entropy_with_logits_v2 = softmax_cross_entropy_with_logits

    @staticmethod
    def max_pool(value, ksize, strides, padding, data_format='NHWC', name=None):
        x = _t(value)
        assert list(ksize) == [1, 2, 2, 1] and list(strides) == [1, 2, 2, 1] and padding.upper() == 'VALID', \
            'tf shim: only the 2x2/2 VALID pooling of roi_pool.py:83-87'
        R, H, W, C = x.shape
        H2, W2 = H // 2, W // 2
        return x[:, :H2 * 2, :W2 * 2].reshape(R, H2, 2, W2, 2, C).max(axis=(2, 4))

    @staticmethod
    def dropout(x, keep_prob, noise_shape=None, seed=None, name=None):
        assert float(keep_prob) == 1.0, 'tf shim: dropout only as the identity (keep_prob 1.0)'
        return _t(x)

    @staticmethod
    def relu(x, name=None):
        return np.maximum(_t(x), 0)

    @staticmethod
    def relu6(x, name=None):
        return np.minimum(np.maximum(_t(x), 0), 6)

    @staticmethod
    def zero_fraction(value, name=None):
        v = _t(value)
        return np.float32((v == 0).mean()) if v.size else np.float32(0)


nn = _NN()


# ---------------------------------------------------------------------------------------------- tf.image ----
class _Image(object):
    @staticmethod
    def non_max_suppression(boxes, scores, max_output_size, iou_threshold=0.5, name=None):
        from oracle import tfops
        boxes, scores = _t(boxes), _t(scores)
        assert boxes.dtype == np.float32 and scores.dtype == np.float32
        return tfops.non_max_suppression(boxes, scores, int(max_output_size), float(iou_threshold)).astype(np.int32)

    @staticmethod
    def crop_and_resize(image, boxes, box_ind, crop_size, method='bilinear', extrapolation_value=0, name=None):
        from oracle import tfops
        assert method == 'bilinear' and extrapolation_value == 0
        return tfops.crop_and_resize(_t(image), _t(boxes), np.asarray(box_ind), tuple(int(c) for c in crop_size))


image = _Image()


# --------------------------------------------------------------------------------------------- tf.losses ----
class _Losses(object):
    @staticmethod
    def add_loss(loss, loss_collection=None):
        _losses.append(_t(loss))

    @staticmethod
    def get_regularization_loss(scope=None, name=None):
        return np.float32(sum(_reg_losses)) if _reg_losses else np.float32(0)

    @staticmethod
    def get_total_loss(add_regularization_losses=True, name=None):
        tot = np.float32(0)
        for v in _losses:
            tot = np.float32(tot + v)
        if add_regularization_losses:
            for v in _reg_losses:
                tot = np.float32(tot + v)
        return tot


losses = _Losses()


def add_regularization_loss(value):
    _reg_losses.append(np.float32(value))


# ------------------------------------------------------------------------------------------------ sonnet ----
class AbstractModule(object):
    """snt.AbstractModule: `module(*args)` calls `_build(*args)` inside the module's variable scope, which is fixed where
    the module is CONSTRUCTED: '<scope at construction>/<name>', made unique with a numeric suffix like Sonnet does."""

    def __init__(self, _sentinel=None, custom_getter=None, name=None):
        self._module_name = name
        base = (_VS[-1] + '/' + name) if _VS[-1] else name
        full, i = base, 0
        while full in _scope_names:
            i += 1
            full = '%s_%d' % (base, i)
        _scope_names.add(full)
        self._scope_name = full

    @property
    def module_name(self):
        return self._module_name

    @property
    def variable_scope(self):
        return _VarScope(self._scope_name)

    @contextlib.contextmanager
    def _enter_variable_scope(self, reuse=None):
        _VS.append(self._scope_name)
        try:
            yield _VarScope(self._scope_name)
        finally:
            _VS.pop()

    def __call__(self, *args, **kwargs):
        with self._enter_variable_scope():
            return self._build(*args, **kwargs)


def seeded_variable(module_name, var, shape):
    """The variable `var` ('w' / 'b') of the Sonnet module called `module_name`: float32 values that are a fixed
    function of the name and the shape (legacy numpy RandomState seeded with a CRC of the name: the same on every numpy
    version) — O(1) outputs for O(1) inputs (w ~ N(0, 1 / fan_in)), non-zero biases so that a mis-ordered channel shows."""
    import zlib
    rs = np.random.RandomState(zlib.crc32(('%s/%s' % (module_name, var)).encode()) & 0x7FFFFFFF)
    shape = tuple(int(v) for v in shape)
    if var == 'w':
        fan_in = int(np.prod(shape[:-1]))
        return (rs.randn(*shape) / np.sqrt(fan_in)).astype(np.float32)
    return (rs.randn(*shape) * 0.25).astype(np.float32)


class _Shape(tuple):
    def as_list(self):
        return list(self)


class Tensor(np.ndarray):
    """An array that also answers the two static-shape calls the reference makes on graph tensors
    (`set_shape`, ssd.py:63; `.shape.as_list()`, ssd.py:122).  Any op on it returns a plain array."""

    def __new__(cls, a):
        return np.asarray(a).view(cls)

    def __array_finalize__(self, obj):
        pass

    def __array_wrap__(self, out, context=None, return_scalar=False):
        return np.asarray(out)

    # `ndarray <op> tensor`: python tries the reflected method of the SUBCLASS operand first.  TensorFlow converts the
    # ndarray to the tensor's dtype (ops.convert_to_tensor(value, dtype=tensor.dtype)): float -> int truncates toward zero.
    def _coerce(self, other):
        me = np.asarray(self)
        if isinstance(other, np.ndarray) and not isinstance(other, Tensor) and other.dtype != me.dtype:
            if np.issubdtype(me.dtype, np.integer) and np.issubdtype(other.dtype, np.floating):
                other = np.trunc(other)
            other = other.astype(me.dtype)
        return me, np.asarray(other)

    def __radd__(self, other):
        me, o = self._coerce(other)
        return o + me

    def __add__(self, other):
        me, o = self._coerce(other)
        return me + o

    @property
    def shape(self):
        return _Shape(np.asarray(self).shape)

    def set_shape(self, shape):
        have = list(np.asarray(self).shape)
        assert len(shape) == len(have) and all(a is None or a == b for a, b in zip(shape, have)), (shape, have)


def _conv2d_nhwc(x, w, padding):
    """Stride-1 NHWC x HWIO convolution in float32; SAME pads (k - 1) // 2 before and the rest after (TF's rule)."""
    x, w = np.asarray(x), np.asarray(w)
    assert x.dtype == np.float32 and w.dtype == np.float32 and x.ndim == 4
    kh, kw, cin, cout = w.shape
    assert x.shape[3] == cin
    if padding.upper() == 'SAME':
        pt, pl = (kh - 1) // 2, (kw - 1) // 2
        x = np.pad(x, ((0, 0), (pt, kh - 1 - pt), (pl, kw - 1 - pl), (0, 0)))
    else:
        assert padding.upper() == 'VALID'
    n, h, ww_, _ = x.shape
    oh, ow = h - kh + 1, ww_ - kw + 1
    y = np.zeros((n, oh, ow, cout), np.float32)
    for r in builtins.range(kh):
        for c in builtins.range(kw):
            y += (x[:, r:r + oh, c:c + ow, :].reshape(-1, cin) @ w[r, c]).reshape(n, oh, ow, cout)
    return y


class Conv2D(AbstractModule):
    """snt.Conv2D(output_channels, kernel_shape, stride=1, rate=1, padding='SAME', use_bias=True, ...): NHWC input,
    variables `w` (kh, kw, in, out) and `b` (out,)."""

    def __init__(self, output_channels, kernel_shape, stride=1, rate=1, padding='SAME', use_bias=True,
                 initializers=None, partitioners=None, regularizers=None, mask=None, data_format='NHWC',
                 custom_getter=None, name='conv_2d'):
        super(Conv2D, self).__init__(name=name)
        assert stride in (1, (1, 1), [1, 1]) and rate in (1, (1, 1), [1, 1]) and data_format == 'NHWC'
        ks = kernel_shape if isinstance(kernel_shape, (list, tuple)) else [kernel_shape, kernel_shape]
        self._kh, self._kw, self._cout = int(ks[0]), int(ks[1]), int(output_channels)
        self._padding, self._use_bias = padding, use_bias
        self._regularizers = regularizers or {}

    def _build(self, inputs):
        x = np.asarray(_t(inputs))
        shape = (self._kh, self._kw, x.shape[3], self._cout)
        self._w = create_variable('w', lambda: seeded_variable(self.module_name, 'w', shape),
                                  regularizer=self._regularizers.get('w')).value
        y = _conv2d_nhwc(x, self._w, self._padding)
        if self._use_bias:
            self._b = create_variable('b', lambda: seeded_variable(self.module_name, 'b', (self._cout,)),
                                      regularizer=self._regularizers.get('b')).value
            y = y + self._b
        return y


class Linear(AbstractModule):
    """snt.Linear(output_size, use_bias=True, ...): (N, in) @ w (in, out) + b (out,)."""

    def __init__(self, output_size, use_bias=True, initializers=None, partitioners=None, regularizers=None,
                 custom_getter=None, name='linear'):
        super(Linear, self).__init__(name=name)
        self._out, self._use_bias = int(output_size), use_bias
        self._regularizers = regularizers or {}

    def _build(self, inputs):
        x = np.asarray(_t(inputs))
        assert x.ndim == 2 and x.dtype == np.float32
        self._w = create_variable('w', lambda: seeded_variable(self.module_name, 'w', (x.shape[1], self._out)),
                                  regularizer=self._regularizers.get('w')).value
        y = x @ self._w
        if self._use_bias:
            self._b = create_variable('b', lambda: seeded_variable(self.module_name, 'b', (self._out,)),
                                      regularizer=self._regularizers.get('b')).value
            y = y + self._b
        return y


class EasyDict(dict):
    """easydict.EasyDict: attribute access, recursive for nested dicts."""

    def __init__(self, d=None, **kwargs):
        super(EasyDict, self).__init__()
        d = dict(d or {}, **kwargs)
        for k, v in d.items():
            setattr(self, k, v)

    def __setattr__(self, name, value):
        if isinstance(value, dict) and not isinstance(value, EasyDict):
            value = EasyDict(value)
        elif isinstance(value, (list, tuple)):
            value = type(value)(EasyDict(v) if isinstance(v, dict) and not isinstance(v, EasyDict) else v
                                for v in value)
        super(EasyDict, self).__setitem__(name, value)

    __setitem__ = __setattr__

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)


def install():
    """Registers the stand-ins as `tensorflow`, `sonnet` (+ the sub-module paths the reference imports from) and
    `easydict`; returns the tf module object."""
    me = sys.modules[__name__]
    sys.modules['tensorflow'] = me
    snt = _Inert('sonnet')
    snt.AbstractModule = AbstractModule
    snt.Conv2D, snt.Linear = Conv2D, Linear
    snt.get_variables_in_module = get_variables_in_module
    sys.modules['sonnet'] = snt
    for sub in ('sonnet.python', 'sonnet.python.modules', 'sonnet.python.modules.conv'):
        sys.modules[sub] = _Inert(sub)
    sys.modules['sonnet.python.modules.conv'].Conv2D = Conv2D
    ed = types.ModuleType('easydict')
    ed.EasyDict = EasyDict
    sys.modules['easydict'] = ed
    if not hasattr(np, 'int'):
        np.int = int
    return me
