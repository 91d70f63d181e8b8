#!/usr/bin/env python3
"""Records tests/golden/views_copy_custom.npz (run in the build container, where the reference is importable).

What it pins: the reference-layout view of the segment store (bithtm_amd.projections.SegmentProjectionView --
`output_edge`, `output_permanence`, `invalid_output_edge`, `get_output_edge_target`, `segment_bundle`) is what the
reference's own state importer consumes.  A learned state (the oracle's, stepped from a seed) is handed to the
UNMODIFIED `reference_implementations.TemporalMemory.copy_custom` (reference_implementations.py:48-88) through that
view; the textbook Temporal Memory then processes the next input (learning off) and its active cells, predictions,
bursting flags, matching / active segments and potentials are recorded -- after checking that they equal the
oracle's own next step.  The GPU test replays the same seed on the device, builds the view from device state, and
must reproduce view content and outputs (tests/test_reference_views.py)."""

import os
import sys
import zlib
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import HTMOracle  # noqa: E402
from oracle.ref_hooks import import_reference  # noqa: E402
from bithtm_amd.projections import SegmentProjectionView  # noqa: E402

CONFIG = dict(seed=77, input_dim=160, column_dim=2048, cell_dim=8, patterns=20, density=0.1, noise=0.02, steps=130)


def inputs(cfg):
    rng = np.random.RandomState(cfg["seed"] + 1)
    bank = rng.rand(cfg["patterns"], cfg["input_dim"]) < cfg["density"]
    return [bank[t % cfg["patterns"]] ^ (rng.rand(cfg["input_dim"]) < cfg["noise"]) for t in range(cfg["steps"] + 1)]


def view_digest(view, seg_cell):
    """Slot-order-free content of the view: per segment its owner and sorted (target, permanence bits) pairs."""
    out = []
    for s in range(view.output_dim):
        e = view.output_edge[s]
        ok = e != view.invalid_output_edge
        t = view.get_output_edge_target(e[ok]).astype(np.int64)
        order = np.argsort(t, kind="stable")
        out.append(zlib.crc32(np.int64(seg_cell[s]).tobytes() + t[order].tobytes() +
                              view.output_permanence[s][ok][order].astype(np.float32).tobytes()))
    return np.array(out, dtype=np.uint32)


def main(path=os.path.join(HERE, "views_copy_custom.npz")):
    cfg = CONFIG
    ref = import_reference()
    C, K = cfg["column_dim"], cfg["cell_dim"]
    np.random.seed(cfg["seed"])
    ora = HTMOracle(cfg["input_dim"], C, K, seed=cfg["seed"])
    xs = inputs(cfg)
    for t in range(cfg["steps"]):
        o_sp, o_tm = ora.step(xs[t])
    tm = ora.temporal_memory
    S = tm.S
    view = SegmentProjectionView(tm.presyn[:S], tm.perm[:S], C * K)
    custom = SimpleNamespace(
        column_dim=C, cell_dim=K, last_state=o_tm, flatten_cell=lambda cell: None if cell is None else cell[0] * K + cell[1],
        distal_projection=SimpleNamespace(segment_bundle=tm.seg_cell[:S, None], segment_projection=view))
    textbook = ref.reference_implementations.TemporalMemory(C, K)
    textbook.copy_custom(custom)                                        # the unmodified importer accepts the view
    assert textbook.segment_cell == tm.seg_cell[:S].tolist()
    assert sum(len(s) for s in textbook.segment_synapses) == int(tm.seg_nsyn[:S].sum())
    # next input, learning off, through the textbook TM and through the oracle
    nxt_sp, nxt_tm = ora.step(xs[cfg["steps"]], learning=False)
    got = textbook.process(SimpleNamespace(active_column=nxt_sp.active_column), learning=False)
    od = nxt_tm.distal_state
    assert set(got.active_cells) == set((nxt_tm.active_cell[0] * K + nxt_tm.active_cell[1]).tolist())
    assert np.array_equal(got.cell_prediction, nxt_tm.cell_prediction)
    assert np.array_equal(got.active_column_bursting, nxt_tm.active_column_bursting)
    assert got.matching_segments == set(od.matching_segment.tolist())
    assert got.active_segments == set(od.matching_segment[od.matching_segment_active].tolist())
    pot = np.array([got.segment_num_active_potential_synapses[s] for s in range(S)], dtype=np.int64)
    assert np.array_equal(pot, od.segment_potential)
    np.savez_compressed(
        path, **{k: np.asarray(v) for k, v in cfg.items()},
        segments=np.int64(S), view_digest=view_digest(view, tm.seg_cell[:S]), output_edges=view.output_edges[:, 0],
        next_active_column=nxt_sp.active_column,
        textbook_active_cells=np.sort(np.array(sorted(got.active_cells), dtype=np.int64)),
        textbook_prediction=np.packbits(got.cell_prediction.reshape(-1), bitorder="little"),
        textbook_bursting=got.active_column_bursting[:, 0],
        textbook_matching=np.array(sorted(got.matching_segments), dtype=np.int64),
        textbook_active_segments=np.array(sorted(got.active_segments), dtype=np.int64),
        textbook_potential=pot)
    print(f"wrote {path}: S={S}, {int(view.output_edges.sum())} synapses, {len(got.matching_segments)} matching segments")


if __name__ == "__main__":
    main(*sys.argv[1:])
