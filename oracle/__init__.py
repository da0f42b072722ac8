"""CPU restatement of the reference's style-transfer arithmetic.

TEST INFRASTRUCTURE ONLY.  Nothing under this directory is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker or the timed CPU baseline.

PARITY UNPINNED: the reference executes its arithmetic inside TensorFlow 1.x,
which is not installable here, and the reference repository holds no tests,
golden vectors or saved outputs for this path.  The functions below restate the
reference source text plus TensorFlow's documented op semantics (SAME padding,
conv2d_transpose as the gradient of conv2d, biased variance in tf.nn.moments,
TF1 Adam).  They are cross-checked against a second, loop-level restatement in
``oracle/loops.py`` but not against TensorFlow itself.
"""
