"""CPU oracle for the YOLO training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / the timed CPU baseline.
The product path (``custom-yolo-implmentation_amd/``) never imports this
package and raises if its HIP library is missing.

What it is: a plain-PyTorch fp32 (CPU) restatement of the reference's
algorithms for the path SURVEY.md section 8 names -- written functionally over
a flat ``{state_dict_key: tensor}`` store, each function citing the reference
``file:line`` it follows (paths relative to the reference checkout).

Parity pinning: ``tests/golden/gen_goldens.py`` imports the reference itself
(in the build container only, with a ``torchvision`` import stub because the
package is absent there) and writes input/output vectors to ``tests/golden``;
``tests/test_oracle_golden.py`` checks every oracle function against them.
The one third-party piece, ``torchvision.ops.nms`` (pinned 0.24.1 in the
reference's environment.yml:29, single call site src/utils/model_utils.py:264),
is absent from the reference checkout and from this image, so the greedy
suppression core is "parity unpinned": restated from its published semantics
and pinned by hand-computed known-answer cases in ``tests/test_nms_known.py``.
"""
