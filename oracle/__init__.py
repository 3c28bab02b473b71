"""CPU oracle for the Faster R-CNN / SSD train-step hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (numpy for the
integer/box arithmetic, torch-CPU fp32 for the dense, differentiable parts) of
the algorithms tryolabs/luminoth runs on its hot path.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
it; nothing under `luminoth_amd/` does.  The product path is the HIP library
(`luminoth_amd/csrc`) and fails loudly when that library is missing.

Parity pinning:
  * The reference (TensorFlow 1.x + Sonnet) cannot be imported in the build
    container, so the oracle is pinned against the golden vectors / known
    answers held in the reference's own unit tests (ported to
    `tests/test_oracle_*.py`, each citing the reference test file:line), and
    against `luminoth/utils/bbox_transform.py` (TF-free numpy twin) which IS
    importable by file path: `tests/golden/make_golden.py` generated
    `tests/golden/bbox_transform_golden.npz` from it.
  * PARITY UNPINNED rows (no reference test holds numbers for them): the slim
    ResNet/VGG backbone arithmetic (A2, A12), all SSD rows (S1-S6), the
    post-subsampling index choice of RPNTarget/RCNNTarget (TF's Philox
    `random_shuffle` is not reproducible outside TF; oracle and kernels share
    the counter-based hash in `oracle/rng.py`), and NMS on exact score ties
    (oracle: ties -> lower index, consistent with `tf.nn.top_k`).

All citations are relative to /root/reference/.
"""
