"""CPU oracle for the bitHTM per-timestep hot path.  TEST INFRASTRUCTURE ONLY.

This package is a NumPy restatement of the reference algorithm (cokwa/bitHTM,
`bithtm/networks.py`, `bithtm/projections.py`, `bithtm/regularizations.py`) under the
deterministic policies written down in DESIGN.md (stable top-k, documented float32 exp,
keyed counter-based random draws).  It exists to CHECK the HIP path; it is never the thing
shipped or measured.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg (the timed NumPy baseline and, with the same objects, the self-checks of
the bench line: the GPU's next steps against the baseline's, the sampled rows of the
configs[4] pool against the keyed generator) may import it -- there only as the checker.
Nothing under `bithtm_amd/` imports it.

Parity status: PINNED.  `tests/golden/generate_golden.py` runs the *unmodified* reference
(imported read-only from /root/reference in the build container) through its own
constructor hooks (`boosting=`, `inhibition=`: networks.py:16,22-24) and with
`np.random.rand` replaced by the keyed generator of `oracle.keyed_rng`; every per-step
output of that run is committed under `tests/golden/*.npz` and
`tests/test_oracle_golden.py` replays them through this oracle bit-for-bit.
`tests/test_oracle_vs_reference.py` repeats the comparison live whenever /root/reference
is importable.
"""

from .keyed_rng import draw24, STREAM_LEAST_USED, STREAM_GROWTH, STREAM_SEGMENT_JITTER  # noqa: F401
from .fexp import exp_f32  # noqa: F401
from .htm_oracle import (  # noqa: F401
    SPParams, TMParams, SpatialPoolerOracle, TemporalMemoryOracle, HTMOracle, stable_topk,
    canonical_synapses, populated_rows,
)
