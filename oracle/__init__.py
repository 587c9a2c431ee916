"""CPU oracle for the StofNet inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker.  The shipped path
(``stofnet_amd``) never imports this package and fails loudly when its HIP
library is missing.

Parity status: PINNED.  Every function here is checked by
``tests/test_oracle_golden.py`` against golden vectors captured from the
reference itself (hahnec/stofnet imported on CPU, torch 2.10.0) by
``tests/golden/make_golden.py``.
"""
