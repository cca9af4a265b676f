"""CPU oracle for the DiffPIR sampling hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain restatement (torch-CPU fp32 / numpy) of the reference
algorithm for the restoration loop ``main_ddpir.py:341-470`` and everything it
calls.  Every function cites the reference file:line it follows.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the *checker*, never as the thing
shipped or measured.  Nothing under ``diffpir_amd/`` imports it; the product
path fails loudly when the HIP library is missing instead of falling back here.

Pinning: the reference ships no golden vectors or tests,
so the oracle is pinned against the reference ITSELF, EXECUTED in the build
container: ``oracle/ref_exec.py`` takes the restoration loop (``test_rho``,
``main_ddpir.py:249-536``) and the pieces of ``main()`` around it out of the
reference's source file with ``ast`` and runs them unmodified -- no restated
loop glue -- and runs the reference's whole ``main()`` on the images and kernel
files it ships.  The outputs are committed under ``tests/golden/`` (generators:
``oracle/gen_golden*.py``) and re-checked on every ``pytest`` run
(``tests/test_oracle_golden.py``); where ``/root/reference`` is present the
restatement is additionally compared with the reference executed live
(``tests/test_oracle_vs_live_reference.py``).
"""
