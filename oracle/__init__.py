"""CPU oracle for the SegLand PSPNet-POP hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``segland_amd/`` may import this package;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do,
and there only as the checker / the timed CPU baseline -- never as the product path.

Parity status: PINNED.  The restatement in ``oracle/pop_oracle.py`` is checked against
outputs of the reference itself (imported from /root/reference in the build container by
``tests/golden/make_golden.py``); the resulting vectors are committed under
``tests/golden/`` and re-checked by ``tests/test_oracle_golden.py`` on every run.
"""
