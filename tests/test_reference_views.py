"""The reference-layout view of the segment store: `PredictiveProjection.segment_projection`
gives `output_edge` / `output_permanence` / `invalid_output_edge` / `get_output_edge_target` / `input_edge` in the
packing of projections.py:60-68, so that the reference's `copy_custom` (reference_implementations.py:48-88) accepts a
bithtm_amd TemporalMemory.  tests/golden/views_copy_custom.npz was recorded with the unmodified reference
(tests/golden/generate_views_fixture.py); here the oracle (CPU) and the device (GPU) must reproduce it."""

import os
import sys

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)


def _fixture():
    z = np.load(os.path.join(GOLDEN, "views_copy_custom.npz"))
    g = {k: z[k] for k in z.files}
    cfg = {k: (float(g[k]) if k in ("density", "noise") else int(g[k])) for k in
           ("seed", "input_dim", "column_dim", "cell_dim", "patterns", "density", "noise", "steps")}
    return g, cfg


def _check_view(view, seg_cell, g):
    import generate_views_fixture as gen
    S = int(g["segments"])
    assert view.output_dim == S and view.invalid_output_edge == view.input_dim
    assert np.array_equal(view.output_edges[:, 0], g["output_edges"])
    assert np.array_equal(gen.view_digest(view, seg_cell), g["view_digest"])
    # the mirrored push form is consistent with the pull form (projections.py:40-44, :63-64)
    ok = view.output_edge != view.invalid_output_edge
    target, slot = view.unpack_output_edge(view.output_edge)
    seg = np.nonzero(ok)[0]
    assert np.array_equal(view.input_edge[target[ok], slot[ok]], 1 + seg)
    assert (view.input_edge != 0).sum() == ok.sum() and (view.input_edge[view.input_dim] == 0).all()
    assert (view.output_permanence[~ok] == -1.0).all() and (view.output_permanence[ok] >= 0).all()


def test_oracle_state_gives_the_recorded_view():
    import generate_views_fixture as gen
    from oracle import HTMOracle
    from bithtm_amd.projections import SegmentProjectionView
    g, cfg = _fixture()
    np.random.seed(cfg["seed"])
    ora = HTMOracle(cfg["input_dim"], cfg["column_dim"], cfg["cell_dim"], seed=cfg["seed"])
    for x in gen.inputs(cfg)[:cfg["steps"]]:
        ora.step(x)
    tm = ora.temporal_memory
    _check_view(SegmentProjectionView(tm.presyn[:tm.S], tm.perm[:tm.S], tm.N), tm.seg_cell[:tm.S], g)


@pytest.mark.skipif(not os.path.isdir("/root/reference/bithtm"), reason="reference checkout not present")
def test_fixture_regenerates_from_the_unmodified_reference(tmp_path):
    import generate_views_fixture as gen
    path = str(tmp_path / "v.npz")
    gen.main(path)
    a, b = np.load(path), np.load(os.path.join(GOLDEN, "views_copy_custom.npz"))
    assert set(a.files) == set(b.files) and all(np.array_equal(a[k], b[k]) for k in a.files)


@pytest.mark.gpu
def test_device_state_through_the_reference_layout_view():
    import generate_views_fixture as gen
    import bithtm_amd as B
    g, cfg = _fixture()
    C, K = cfg["column_dim"], cfg["cell_dim"]
    np.random.seed(cfg["seed"])
    htm = B.HierarchicalTemporalMemory(cfg["input_dim"], C, K, seed=cfg["seed"])
    xs = gen.inputs(cfg)
    for x in xs[:cfg["steps"]]:
        htm.process(x)
    tm = htm.temporal_memory
    view = tm.distal_projection.segment_projection
    _check_view(view, tm.distal_projection.segment_bundle[:, 0], g)
    assert tm.distal_projection.segment_bundle.shape == (int(g["segments"]), 1)
    # what copy_custom reads of the last State (reference_implementations.py:73-88)
    st = tm.last_state
    assert st.distal_state.segment_potential.shape == (int(g["segments"]),)
    assert tm.flatten_cell(st.active_cell).ndim == 1 and tm.flatten_cell(st.winner_cell).ndim == 1
    # the next step (learning off) equals what the textbook TM produced from the imported view
    sp_state, nxt = htm.process(xs[cfg["steps"]], learning=False)
    assert np.array_equal(sp_state.active_column, g["next_active_column"])
    assert np.array_equal(np.sort(tm.flatten_cell(nxt.active_cell)), g["textbook_active_cells"])
    assert np.array_equal(np.packbits(nxt.cell_prediction.reshape(-1), bitorder="little"), g["textbook_prediction"])
    assert np.array_equal(nxt.active_column_bursting[:, 0], g["textbook_bursting"])
    d = nxt.distal_state
    assert np.array_equal(d.matching_segment, g["textbook_matching"])
    assert np.array_equal(d.matching_segment[d.matching_segment_active], g["textbook_active_segments"])
    assert np.array_equal(d.segment_potential, g["textbook_potential"])
