"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the nelpy/ghost CWT hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker / CPU baseline -- never as the thing shipped.
See ``oracle/ghost_oracle.py`` for how parity is pinned.
"""
