"""CPU oracle for the Free Hunch guided-sampling hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  The product package
(``free-hunch_amd/``) never imports this package and fails loudly when its HIP
library is missing.

Every function restates, in plain numpy / scipy / torch-CPU, the algorithm of
one reference function and cites it (paths relative to the reference checkout).
The restatement is pinned against golden vectors produced by importing the
reference itself in the build container (``tests/golden/make_golden.py``);
see DESIGN.md "Oracle" for what is pinned and what is not (the third-party
``torch_dct`` boundary is pinned against SciPy only).
"""
