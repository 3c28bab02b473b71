"""Unit tests of tests/golden/tf_numpy_shim.py: each op whose corner cases decide a label or an index in the
reference's graph code is checked against the TF 1.x documented behaviour (the examples of the TF API docs where one
exists).  The shim is test infrastructure; these tests are what makes the reference-executed fixtures trustworthy."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
import tf_numpy_shim as tf  # noqa: E402


def test_where_single_argument_returns_int64_coordinates():
    c = np.array([[True, False], [False, True], [True, True]])
    r = tf.where(c)
    assert r.dtype == np.int64 and r.tolist() == [[0, 0], [1, 1], [2, 0], [2, 1]]
    assert tf.where(np.array([False, True, True])).tolist() == [[1], [2]]
    assert tf.where(np.zeros((0,), bool)).shape == (0, 1)


def test_where_vector_condition_selects_rows():
    x, y = np.ones((3, 4), np.float32), np.zeros((3, 4), np.float32)
    r = tf.where(np.array([True, False, True]), x, y)
    assert r[:, 0].tolist() == [1, 0, 1] and (r == r[:, :1]).all()
    with pytest.raises(ValueError):
        tf.where(np.array([True, False]), x, y)


def test_strict_dtypes():
    with pytest.raises(TypeError):
        tf.add(np.ones(3, np.float32), np.ones(3, np.int32))
    with pytest.raises(TypeError):
        tf.where(np.array([True]), np.ones(1, np.float32), np.ones(1, np.float64))
    assert tf.maximum(np.ones(3, np.float32), 0.3).dtype == np.float32
    assert tf.less(np.float32(0.3) * np.ones(2, np.float32), 0.3).tolist() == [False, False]   # 0.3 rounds to fp32
    assert tf.cast(np.array([2.9, -2.9], np.float32), tf.int32).tolist() == [2, -2]              # truncation
    assert tf.to_int32(0.3 * 256).item() == 76


def test_sparse_to_dense_semantics():
    # TF docs: dense[sparse_indices[i]] = sparse_values[i]; scalar value broadcast; default elsewhere
    r = tf.sparse_to_dense(np.array([1, 3]), [5], True, default_value=False)
    assert r.tolist() == [False, True, False, True, False]
    # validate_indices=True: unsorted or repeated indices are an error (InvalidArgument in TF)
    with pytest.raises(ValueError):
        tf.sparse_to_dense(np.array([3, 1]), [5], True, default_value=False)
    with pytest.raises(ValueError):
        tf.sparse_to_dense(np.array([1, 1]), [5], True, default_value=False)
    # validate_indices=False: written in order, the LAST duplicate wins (rcnn_target.py:140-147, ssd/target.py:108-115)
    r = tf.sparse_to_dense(np.array([2, 0, 2]), [4], np.array([5., 6., 7.], np.float32), default_value=0.,
                           validate_indices=False)
    assert r.tolist() == [6., 0., 7., 0.]
    # (n, 1) index matrix (rcnn_target.py:181-187)
    r = tf.sparse_to_dense(np.array([[3], [0]]), np.array([4], np.int64), True, default_value=False,
                           validate_indices=False)
    assert r.tolist() == [True, False, False, True]
    with pytest.raises(IndexError):
        tf.sparse_to_dense(np.array([7]), [4], True, default_value=False)


def test_scatter_nd_accumulates_and_defaults_to_zero():
    # TF docs example: indices [[4],[3],[1],[7]], updates [9,10,11,12], shape [8] -> [0,11,0,10,9,0,0,12]
    r = tf.scatter_nd(np.array([[4], [3], [1], [7]]), np.array([9, 10, 11, 12], np.int32), [8])
    assert r.tolist() == [0, 11, 0, 10, 9, 0, 0, 12]
    r = tf.scatter_nd(np.array([[1], [1]]), np.array([[1., 2.], [3., 4.]], np.float32), [3, 2])
    assert r.tolist() == [[0, 0], [4, 6], [0, 0]]


def test_gather_and_gather_nd_shapes():
    p = np.arange(12, dtype=np.float32).reshape(4, 3)
    assert tf.gather(p, np.array([[2], [0]])).shape == (2, 1, 3)                  # indices shape + params.shape[1:]
    assert tf.gather_nd(p, np.array([[2], [0]])).tolist() == [[6, 7, 8], [0, 1, 2]]
    assert tf.gather_nd(p, np.array([[1, 2], [3, 0]])).tolist() == [5, 9]
    with pytest.raises(IndexError):
        tf.gather(p, np.array([4]))


def test_unique_first_occurrence_order():
    # TF docs example: x = [1,1,2,4,4,4,7,8,8] -> y = [1,2,4,7,8], idx = [0,0,1,2,2,2,3,4,4]
    u = tf.unique(np.array([1, 1, 2, 4, 4, 4, 7, 8, 8]))
    assert u.y.tolist() == [1, 2, 4, 7, 8] and u.idx.tolist() == [0, 0, 1, 2, 2, 2, 3, 4, 4]
    u = tf.unique(np.array([9, 3, 9, 1, 3]))
    assert u.y.tolist() == [9, 3, 1] and u.idx.tolist() == [0, 1, 0, 2, 1]


def test_top_k_order_and_ties():
    r = tf.nn.top_k(np.array([.2, .9, .9, .1, .5], np.float32), k=3)
    assert r.indices.tolist() == [1, 2, 4] and r.values.tolist() == pytest.approx([.9, .9, .5])
    v, i = tf.nn.top_k(np.array([5, 1, 7], np.int64), k=3)                        # tuple unpacking (rpn_target.py:163)
    assert v.tolist() == [7, 5, 1]
    assert tf.nn.top_k(np.array([-1., -1., -1.], np.float32), k=2).indices.tolist() == [0, 1]
    with pytest.raises(ValueError):
        tf.nn.top_k(np.zeros(2, np.float32), k=3)
    assert tf.nn.top_k(np.zeros(4, np.float32), k=0).indices.shape == (0,)


def test_one_hot_out_of_range_rows_are_zero():
    r = tf.one_hot(np.array([0, 2, -1, 3]), depth=3)
    assert r.tolist() == [[1, 0, 0], [0, 0, 1], [0, 0, 0], [0, 0, 0]] and r.dtype == np.float32


def test_reductions_and_empty_mean():
    assert np.isnan(tf.reduce_mean(np.zeros((0,), np.float32)))
    assert tf.reduce_mean(np.array([1., 2.], np.float32)).dtype == np.float32
    assert tf.reduce_sum(np.ones((2, 3), np.float32), [1]).tolist() == [3, 3]
    assert tf.argmax(np.array([[1, 5, 5], [7, 7, 0]]), axis=1).tolist() == [1, 0]  # first occurrence
    assert tf.count_nonzero(np.array([True, False, True])).item() == 2


def test_boolean_mask_cond_fill_tile_shape():
    assert tf.boolean_mask(np.arange(6).reshape(3, 2), np.array([True, False, True])).tolist() == [[0, 1], [4, 5]]
    assert tf.cond(np.array(3) > 2, lambda: np.float32(1), lambda: np.float32(2)) == 1
    assert tf.cond(np.array(False), true_fn=lambda: 1.0, false_fn=lambda: 0.0).dtype == np.float32
    assert tf.fill(tf.gather(tf.shape(np.zeros((5, 4))), [0]), -1.).tolist() == [-1.] * 5
    assert tf.fill([2], -1).dtype == np.int32
    assert tf.tile([3], [np.int32(2)]).tolist() == [3, 3]
    assert tf.shape(np.zeros((2, 3)), out_type=tf.int64).dtype == np.int64
    x1, x2 = tf.split(np.arange(8, dtype=np.float32).reshape(2, 4), 2, axis=1)
    assert x1.shape == (2, 2)


def test_softmax_cross_entropy_known_values():
    logits = np.array([[0., 0.], [np.log(3.), 0.]], np.float32)
    ce = tf.nn.softmax_cross_entropy_with_logits_v2(labels=tf.one_hot(np.array([1, 0]), 2), logits=logits)
    np.testing.assert_allclose(ce, [np.log(2.), np.log(4. / 3.)], rtol=1e-6)
    ce = tf.nn.softmax_cross_entropy_with_logits(labels=np.zeros((1, 2), np.float32), logits=logits[:1])
    assert ce.tolist() == [0.0]                                                   # all-zero label row: loss 0


def test_random_shuffle_hook_must_be_a_permutation():
    tf.set_random_shuffle(lambda v, seed, caller: v[::-1])
    assert tf.random_shuffle(np.array([[1], [2], [3]])).tolist() == [[3], [2], [1]]
    tf.set_random_shuffle(lambda v, seed, caller: v * 0)
    with pytest.raises(AssertionError):
        tf.random_shuffle(np.array([1, 2, 3]))
    tf.set_random_shuffle(None)
    with pytest.raises(RuntimeError):
        tf.random_shuffle(np.array([1, 2, 3]))


def test_losses_collection():
    tf.reset_losses()
    tf.losses.add_loss(np.float32(1.5))
    tf.add_regularization_loss(0.25)
    assert tf.losses.get_total_loss() == np.float32(1.75)
    tf.reset_losses()
    assert tf.losses.get_total_loss() == 0


def test_variable_scopes_modules_and_collections():
    """Round 6 (the reference's top-level composition is executed): a Sonnet module takes its variable scope where it is
    CONSTRUCTED and enters it when called; variables keep TensorFlow names and creation order; collections answer
    snt.get_variables_in_module / tf.get_collection; same-named modules are made unique."""
    tf.reset_variables()
    tf.reset_losses()
    try:
        class Outer(tf.AbstractModule):
            def __init__(self):
                super(Outer, self).__init__(name='outer')
                self.side = tf.Linear(3, name='side')                 # constructed OUTSIDE _build: scope 'side', not 'outer/side'

            def _build(self, x):
                a = tf.Linear(4, name='fc')(x)                        # 'outer/fc'
                b = tf.Linear(4, name='fc')(x)                        # 'outer/fc_1'
                with tf.variable_scope('extra'):
                    tf.create_variable('stat', lambda: np.zeros(2, np.float32), trainable=False, model_variable=True)
                return a + b + self.side(x)[:, :1]
        m = Outer()
        m(np.ones((2, 5), np.float32))
        names = [v.name for v in tf._variables]
        assert names == ['outer/fc/w:0', 'outer/fc/b:0', 'outer/fc_1/w:0', 'outer/fc_1/b:0', 'outer/extra/stat:0',
                         'side/w:0', 'side/b:0']
        assert m.variable_scope.name == 'outer' and m.side.variable_scope.name == 'side'
        assert [v.op.name for v in tf.get_variables_in_module(m)] == ['outer/fc/w', 'outer/fc/b', 'outer/fc_1/w', 'outer/fc_1/b']
        assert [v.name for v in tf.get_variables_in_module(m, tf.GraphKeys.MODEL_VARIABLES)] == ['outer/extra/stat:0']
        assert [v.name for v in tf.get_collection(tf.GraphKeys.MODEL_VARIABLES, scope='outer')] == ['outer/extra/stat:0']
        # (every module is connected ONCE in the fixtures: Sonnet's template machinery that re-uses the sub-modules of a second
        # connection is not restated)
        m.side(np.ones((2, 5), np.float32))                          # ... a layer called again re-uses its variables
        assert len(tf._variables) == 7
    finally:
        tf.reset_variables()


def test_regularizers_are_tracked_only_on_request():
    tf.reset_variables()
    tf.reset_losses()
    try:
        reg = tf.l2_regularizer(0.5)
        w = np.arange(6, dtype=np.float32).reshape(2, 3)
        assert reg(w) == np.float32(0.5 * (w ** 2).sum() / 2) and reg(w).dtype == np.float32
        tf.Linear(3, regularizers={'w': reg}, name='quiet')(np.ones((1, 2), np.float32))
        assert tf.regularized_names() == [] and tf.losses.get_regularization_loss() == 0       # round-5 behaviour by default
        tf.track_regularizers(True)
        lin = tf.Linear(3, regularizers={'w': reg}, name='loud')
        lin(np.ones((1, 2), np.float32))
        lin(np.ones((1, 2), np.float32))                                                      # once per VARIABLE, not per call
        assert tf.regularized_names() == ['loud/w']
        assert tf.losses.get_regularization_loss() == reg(lin._w)
    finally:
        tf.track_regularizers(False)
        tf.reset_variables()
        tf.reset_losses()


def test_ndarray_plus_tensor_takes_the_tensors_dtype():
    """fasterrcnn.py:299-302 adds the float64 numpy anchor reference to an int32 graph tensor: TensorFlow converts the
    ndarray to the TENSOR's dtype (truncation toward zero), which is why `all_anchors` is int32 (SURVEY.md appendix B.1,
    pinned by fasterrcnn_test.py:285-295)."""
    shifts = tf.expand_dims(np.array([[0, 0, 0, 0], [16, 0, 16, 0]], np.int32), axis=1)        # (2, 1, 4) graph tensor
    ref = np.array([[-22.627417, -11.3137085, 22.627417, 11.3137085]], np.float64)
    out = np.expand_dims(ref, axis=0) + shifts
    assert out.dtype == np.int32 and out.tolist() == [[[-22, -11, 22, 11]], [[-6, -11, 38, 11]]]
    assert (shifts + np.expand_dims(ref, axis=0)).dtype == np.int32
    same = tf.expand_dims(np.ones(3, np.float32), 0) + np.ones(3, np.float32)                 # equal dtypes: plain addition
    assert same.dtype == np.float32 and same.tolist() == [[2.0, 2.0, 2.0]]
