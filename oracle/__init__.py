"""CPU oracle for the BTSbot classifier hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``btsbot_amd`` may import this package: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do,
and there only as the checker.  See ``oracle/README.md`` for how the oracle is pinned.
"""
