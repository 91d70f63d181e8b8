"""Manual GPU fuzz (not collected by pytest): random small configurations, the HIP path against the
oracle in lock-step through process(), then batched pipelined runs and host-fed steps with nothing read in between
against the oracle's final state.  Every draw is seeded; a failure prints the configuration that reproduces it.

    python tests/fuzz_parity.py [--configs 40] [--seed 0]
"""

import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from hip_impl import make_htm, compare_with_oracle, compare_store_with_oracle  # noqa: E402
from oracle import HTMOracle, SPParams, TMParams  # noqa: E402
from bithtm_amd.engine import CapacityError  # noqa: E402


def draw_config(rng):
    K = int(rng.choice([1, 2, 4, 7, 8, 16, 31, 32, 33, 40, 48, 63, 64]))       # (above 32: two cell words per column, a wave per active column)
    C = int(rng.choice([64, 96, 256, 512, 1000, 2048, 4096, 6000]))
    I = int(rng.choice([17, 32, 64, 100, 257, 512, 1000]))
    k = max(1, int(round(C * rng.choice([0.01, 0.02, 0.05, 0.1]))))
    slots = int(rng.choice([64, 128, 128, 256]))
    tm = TMParams()
    if rng.rand() < 0.5:                           # lower thresholds: matching / recycling paths get busy early
        thr = int(rng.choice([3, 5, 8]))
        tm = TMParams(segment_activation_threshold=thr, segment_matching_threshold=max(2, thr - int(rng.randint(0, 3))),
                      segment_sampling_synapses=int(rng.choice([8, 16, 32])), permanence_initial=float(rng.choice([0.21, 0.45])),
                      permanence_punishment=float(rng.choice([0.01, 0.3])))
    sp = SPParams() if rng.rand() < 0.6 else SPParams(permanence_threshold=float(rng.choice([-0.05, 0.02])),
                                                       boost_intensity=float(rng.choice([0.1, 0.3, 1.0])))
    return dict(I=I, C=C, K=K, k=k, slots=slots, P=int(rng.choice([3, 7, 20])), density=float(rng.choice([0.05, 0.2, 0.5])),
                noise=float(rng.choice([0.0, 0.01, 0.05])), steps=int(rng.choice([30, 60, 90])), sp=sp, tm=tm)


def run_one(cfg, seed):
    I, C, K, k = cfg["I"], cfg["C"], cfg["K"], cfg["k"]
    np.random.seed(seed)
    ora = HTMOracle(I, C, K, active_columns=k, seed=seed, sp_params=cfg["sp"], tm_params=cfg["tm"])
    perm0 = ora.spatial_pooler.permanence.copy()
    cap = max(1 << 12, 64 * k * 8)
    htm = make_htm(I, C, K, k, seed, perm0.copy(), cfg["sp"], cfg["tm"], segment_capacity=cap, segment_slots=cfg["slots"])
    rng = np.random.RandomState(seed + 1)
    bank = rng.rand(cfg["P"], I) < cfg["density"]
    # (a) lock-step through process(), learning toggled now and then
    for t in range(cfg["steps"]):
        x = bank[t % cfg["P"]] ^ (rng.rand(I) < cfg["noise"])
        learning = (t % 11) != 7
        o_sp, o_tm = ora.step(x, learning=learning)
        h_sp, h_tm = htm.process(x, learning=learning)
        compare_with_oracle(t, o_sp, o_tm, h_sp, h_tm, K)
    compare_store_with_oracle(cfg["steps"] - 1, ora, htm)
    # (b) batched pipelined runs continue from there on the noise-free bank, in odd chunk lengths
    t = cfg["steps"]
    for n, graph in ((1, True), (2, False), (5, True), (19, True), (21, False)):
        for _ in range(n):
            o_sp, _ = ora.step(bank[t % cfg["P"]])
            t += 1
        htm.run(bank, n, use_graph=graph)
    assert np.array_equal(htm.engine.read_sp_fields()["active_column"], o_sp.active_column)
    compare_store_with_oracle(t - 1, ora, htm)
    # (c) host-fed steps with nothing read in between (each step's last launch rides in the next call's first), learning
    # toggled, then one more batched run straight behind them
    for i in range(23):
        x = bank[t % cfg["P"]] ^ (rng.rand(I) < cfg["noise"])
        learning = (i % 7) != 3
        o_sp, o_tm = ora.step(x, learning=learning)
        htm.process(x, learning=learning)
        t += 1
    for _ in range(3):
        o_sp, o_tm = ora.step(bank[t % cfg["P"]])
        t += 1
    htm.run(bank, 3)
    assert np.array_equal(htm.engine.read_sp_fields()["active_column"], o_sp.active_column)
    compare_store_with_oracle(t - 1, ora, htm)
    htm.engine.check_capacity()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.RandomState(args.seed)
    for i in range(args.configs):
        cfg = draw_config(rng)
        seed = int(rng.randint(1 << 20))
        shown = {k: v for k, v in cfg.items() if k not in ("sp", "tm")}
        try:
            run_one(cfg, seed)
        except CapacityError as e:                 # a fixed pool is a documented limit, not a parity failure
            print(f"skipped {i:3d} ({e}): seed={seed} {shown}", flush=True)
            continue
        except Exception:
            print(f"FAILED config {i}: seed={seed} {shown} sp={cfg['sp']} tm={cfg['tm']}", flush=True)
            raise
        print(f"ok {i:3d}: seed={seed} {shown}", flush=True)
    print("all configurations agree with the oracle")


if __name__ == "__main__":
    main()
