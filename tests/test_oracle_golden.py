"""CPU: the oracle reproduces every golden vector recorded from the unmodified reference."""

import glob
import os

import numpy as np
import pytest

import golden_replay as gr
from oracle import SpatialPoolerOracle, exp_f32, stable_topk
from oracle.htm_oracle import sp_derived, SPParams

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRAJ = sorted(glob.glob(os.path.join(GOLDEN, "traj_*.npz")))


def test_fixtures_present():
    assert len(TRAJ) >= 4
    for name in ("ops_sp.npz", "harness_example.npz"):
        assert os.path.exists(os.path.join(GOLDEN, name))


@pytest.mark.parametrize("path", TRAJ, ids=[os.path.basename(p)[5:-4] for p in TRAJ])
def test_oracle_replays_trajectory(path):
    g = gr.load(path)
    n = gr.replay(g, gr.OracleImpl(g))
    assert n == int(g["steps"])
    # the fixture must exercise the learned (non-bursting) path, not only bursting
    assert g["predicted_columns_per_step"].max() > 0


def test_trajectories_cover_the_interesting_events():
    """Fixtures contain predicted columns, active segments, punish-able matching segments and a
    learning=False stretch; K in {4, 8, 16, 32}; an input_dim that is not a multiple of 32."""
    ks, any_false, any_active = set(), False, False
    for p in TRAJ:
        g = gr.load(p)
        ks.add(int(g["cell_dim"]))
        any_false |= bool((~g["learning"]).any())
        any_active |= bool(g["match_active"].any())
    assert {4, 8, 16, 32} <= ks and any_false and any_active
    assert any(int(gr.load(p)["input_dim"]) % 32 for p in TRAJ)


def test_dense_projection_known_answers():
    """projections.py:18-24 (no hooks involved)."""
    z = np.load(os.path.join(GOLDEN, "ops_sp.npz"))
    C, I = z["perm0"].shape
    sp = SpatialPoolerOracle(I, C, int(z["k"]), permanence=z["perm0"])
    assert np.array_equal(sp.overlaps(z["input"]), z["overlaps"])
    rows = z["updated_rows"]
    sp.permanence[rows] += np.where(z["input"], sp.d.delta_on, sp.d.delta_off)
    assert np.array_equal(sp.permanence[rows].view(np.int64), z["perm_after_rows"].view(np.int64))


def test_boosting_known_answers_and_exp_ulp_distance():
    """regularizations.py:15-21 with NumPy's own exp.  Duty cycles must be bit-exact; the boost
    factor uses the documented exp, whose distance to NumPy's (not correctly rounded) exp is
    bounded here: <= 2 float32 ulp on the factor."""
    z = np.load(os.path.join(GOLDEN, "ops_sp.npz"))
    C = z["perm0"].shape[0]
    k = int(z["k"])
    d = sp_derived(SPParams(), C, k)
    np.random.seed(int(z["seed"]))
    # replay the generator's RNG consumption to recover the same active sets
    np.random.randn(C, z["perm0"].shape[1])
    np.random.rand(z["perm0"].shape[1])
    np.random.choice(C, k, replace=False)
    duty = np.zeros(C, dtype=np.float32)
    worst = 0
    for j in range(len(z["duty_seq"])):
        act = np.sort(np.random.choice(C, k, replace=False))
        duty *= d.momentum32
        duty[act] += d.increment32
        assert np.array_equal(duty.view(np.int32), z["duty_seq"][j].view(np.int32))
        factor = exp_f32(d.coef32 * duty)
        ov = z["overlaps"]
        nz = ov > 0
        ref_factor = (z["boosted_seq"][j][nz] / ov[nz]).astype(np.float32)     # exact: product was exact
        worst = max(worst, int(np.abs(factor[nz].view(np.int32) - ref_factor.view(np.int32)).max()))
    assert worst <= 2, worst


def test_exp_is_correctly_rounded_against_float64_libm():
    x = np.linspace(-30.0, 0.0, 200001).astype(np.float32)
    want = np.exp(x.astype(np.float64)).astype(np.float32)
    got = exp_f32(x)
    assert np.array_equal(got.view(np.int32), want.view(np.int32))


def test_global_inhibition_known_answer():
    """regularizations.py:28-29 on tie-free input: same set as the stable policy."""
    z = np.load(os.path.join(GOLDEN, "ops_sp.npz"))
    assert np.array_equal(stable_topk(z["topk_values"], int(z["k"])), z["topk_sorted"])


def test_stable_topk_tie_rule():
    v = np.array([3.0, 5.0, 5.0, 1.0, 5.0, 3.0])
    assert stable_topk(v, 2).tolist() == [1, 2]
    assert stable_topk(v, 4).tolist() == [0, 1, 2, 4]
