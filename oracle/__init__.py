"""oracle/ — CPU restatement of the MACARONS SCONE coverage-gain hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``macarons_amd/`` imports this package.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import / execute it, and there only as the checker (or the timed CPU baseline), never as
the thing shipped.

Every function cites the reference ``file:line`` (relative to the upstream MACARONS tree)
whose arithmetic it restates.  The restatement is pinned against golden vectors produced
by importing the real reference in the build container (``tests/golden/make_golden.py``);
``tests/test_oracle_golden.py`` checks every fixture.
"""
