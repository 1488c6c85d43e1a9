"""CPU oracle for the surfel-integration hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  Nothing under ``surfelmeshing_amd/``
does (``tests/test_no_oracle_in_product.py`` enforces it).

See ``oracle/smx_oracle.h`` for what is restated and how parity is pinned.
"""
from .binding import *  # noqa: F401,F403
