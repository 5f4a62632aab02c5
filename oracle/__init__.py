"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the Hierarchical Parallel Co-Attention path.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker /
the reported CPU baseline.  The product path (``visual-question-answering_amd``) never
imports this package and fails loudly when its HIP extension is missing.

Parity status: PINNED.  The restatement in ``coattn_oracle.py`` is checked against the
reference's own ``model.py`` classes (imported under a torchvision stub, see
``ref_import.py``) by ``make_golden.py``, which also writes the golden vectors under
``tests/golden/``.  The reference has no tests / fixtures of its own.
"""
