"""GPU: the HIP path, driven through the Python drop-in classes and the C ABI, reproduces the
golden vectors of the unmodified reference and the oracle, bit for bit."""

import glob
import os

import numpy as np
import pytest

import golden_replay as gr

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRAJ = sorted(glob.glob(os.path.join(GOLDEN, "traj_*.npz")))


@pytest.mark.parametrize("path", TRAJ, ids=[os.path.basename(p)[5:-4] for p in TRAJ])
def test_hip_replays_reference_trajectory(path):
    from hip_impl import HipImpl
    g = gr.load(path)
    n = gr.replay(g, HipImpl(g))
    assert n == int(g["steps"])


def test_hip_vs_oracle_fresh_config_k32():
    from hip_impl import lockstep_vs_oracle
    lockstep_vs_oracle(seed=101, input_dim=1024, column_dim=4096, cell_dim=32, patterns=60, density=0.03,
                       noise=0.005, steps=330, store_every=40)


def test_hip_vs_oracle_learning_off_and_jumps():
    from hip_impl import lockstep_vs_oracle
    lockstep_vs_oracle(seed=102, input_dim=333, column_dim=2048, cell_dim=12, patterns=50, density=0.08,
                       noise=0.02, steps=300, store_every=50, jump=0.2, learning_schedule=lambda t: (t % 19) != 4)


def test_hip_small_slots_64():
    from hip_impl import lockstep_vs_oracle
    lockstep_vs_oracle(seed=103, input_dim=512, column_dim=4096, cell_dim=16, patterns=50, density=0.04,
                       noise=0.005, steps=260, store_every=50, segment_slots=64)


def test_harness_counters_match_reference():
    """example.py:50-57 counters (bursting / correct / incorrect columns) over 600 steps."""
    from hip_impl import make_htm
    z = np.load(os.path.join(GOLDEN, "harness_example.npz"))
    seed, I, C, K, P = int(z["seed"]), int(z["input_dim"]), int(z["column_dim"]), int(z["cell_dim"]), int(z["patterns"])
    np.random.seed(seed)
    perm = np.random.randn(C, I) * 0.1 + 0.0
    htm = make_htm(I, C, K, round(C * 0.02), seed, perm)
    rng = np.random.RandomState(seed + 1)
    bank = rng.rand(P, I) < float(z["density"])
    got = []
    for t in range(int(z["steps"])):
        prev_col_pred = htm.temporal_memory.last_state.cell_prediction.max(axis=1)
        x = bank[t % P] ^ (rng.rand(I) < float(z["noise"]))
        sp_state, tm_state = htm.process(x)
        burst = int(tm_state.active_column_bursting.sum())
        correct = int(prev_col_pred[sp_state.active_column].sum())
        got.append((burst, correct, int(prev_col_pred.sum() - correct)))
    assert np.array_equal(np.array(got, dtype=np.int32), z["counters"])
