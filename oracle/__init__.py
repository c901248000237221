"""CPU oracle for the YOLO-ReT detection forward path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``yoloret_amd`` (the product) may import
this package.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

PARITY UNPINNED BY THE REFERENCE: the reference (prakharg24/yoloret) ships no
tests, no golden outputs and no checkpoints, and its arithmetic lives in
TensorFlow/Keras, which is absent here.  The oracle is therefore a restatement
of (a) the reference's own Python (``code/yolo3/model.py``,
``code/yolo3/efficientnet.py``, ``code/yolo.py``; cited per function) and
(b) the published behaviour of the TF/Keras ops it calls (SAME padding,
inference BatchNorm, MobileNetV2 graph, NonMaxSuppressionV3).  It is pinned by
- the hand-derived known-answer tests of SURVEY.md Appendix D
  (``tests/test_oracle_kat.py``),
- an independent PyTorch port of the MobileNetV2 / EfficientNet backbones
  (``transformers``; fixtures in ``tests/golden`` made by
  ``tests/golden/make_golden.py``),
- a second implementation of the conv stack on torch-CPU (``oracle/torch_ref.py``).
"""
