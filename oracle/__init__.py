"""CPU oracle for the DiffPIR sampling hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain restatement (torch-CPU fp32 / numpy) of the reference
algorithm for the restoration loop ``main_ddpir.py:341-470`` and everything it
calls.  Every function cites the reference file:line it follows.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the *checker*, never as the thing
shipped or measured.  Nothing under ``diffpir_amd/`` imports it; the product
path fails loudly when the HIP library is missing instead of falling back here.

Pinning: the reference ships no golden vectors or tests,
so the oracle is pinned against *outputs of the reference itself*, obtained by
importing the live modules from ``/root/reference`` in the build container
(``oracle/gen_golden.py``; ``oracle/ref_import.py``).  Those outputs are
committed under ``tests/golden/`` and re-checked on every ``pytest`` run
(``tests/test_oracle_golden.py``); when ``/root/reference`` is present the
restatement is additionally compared against the live modules
(``tests/test_oracle_vs_live_reference.py``).
"""
