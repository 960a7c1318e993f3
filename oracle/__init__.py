"""oracle/ — CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import anything
from this package, and only as the checker / the reported CPU baseline — never as the thing
shipped.  The product (`aid_amd`) does not import it and has no CPU fallback.
"""
