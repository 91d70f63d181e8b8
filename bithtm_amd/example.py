"""Harness for the MI355X engine with the command line, input model and per-step report of the
reference's example script (flags: example.py:22-30; random pattern bank: :34; flip noise: :52;
the three counters: :55-57), so that its output can be compared line by line.

    python -m bithtm_amd.example --epochs 8
    python -m bithtm_amd.example --epochs 8 --batched   # one C-ABI call per epoch, hipGraph replay
    python -m bithtm_amd.example --epochs 8 --use_reference_implementation   # example.py:30,36-37 (needs the user's `bithtm`)
"""

import argparse
import sys
import time

import numpy as np

from bithtm_amd import HierarchicalTemporalMemory

FLAGS = (
    # name, type, default  (the reference's flags, same names and defaults)
    ("epochs", int, 100),
    ("input_patterns", int, 100),
    ("input_dim", int, 1000),
    ("input_density", float, 0.2),
    ("input_noise_probability", float, 0.05),
    ("column_dim", int, 2048),
    ("cell_dim", int, 32),
)


def parse(argv):
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    for name, kind, default in FLAGS:
        ap.add_argument("--" + name, type=kind, default=default)
    ap.add_argument("--use_reference_implementation", action="store_true",
                    help="the reference's flag (example.py:30): swap the Temporal Memory for the textbook one of the user's own "
                         "`bithtm.reference_implementations` (example.py:7-12, 36-37); the Spatial Pooler stays on the device")
    ap.add_argument("--seed", type=int, default=0, help="seed of the Temporal Memory's keyed random draws")
    ap.add_argument("--batched", action="store_true",
                    help="run each epoch with HierarchicalTemporalMemory.run (no per-step read-back) and "
                         "report timesteps/s per epoch instead of the per-step counters")
    return ap.parse_args(argv)


def digits(n):
    """Field width the reference uses for a counter that can reach n - 1."""
    return int(np.ceil(np.log10(max(n - 1, 2))))


class Report:
    """Formats one line per timestep exactly like the reference's print statement."""

    def __init__(self, opts, active_columns):
        self.w_epoch, self.w_pattern = digits(opts.epochs), digits(opts.input_patterns)
        self.w_column, self.w_active = digits(opts.column_dim), digits(active_columns)

    def line(self, epoch, pattern, bursting, correct, incorrect):
        return (f"epoch {epoch:{self.w_epoch}d}, pattern {pattern:{self.w_pattern}d}: "
                f"bursting columns: {bursting:{self.w_active}d}, correct columns: {correct:{self.w_active}d}, "
                f"incorrect columns: {incorrect:{self.w_column}d}")


def column_counters(predicted_columns, sp_state, tm_state):
    """(bursting, correctly predicted, incorrectly predicted) columns of one timestep."""
    hit = int(predicted_columns[sp_state.active_column].sum())
    return int(tm_state.active_column_bursting.sum()), hit, int(predicted_columns.sum()) - hit


def run_stepwise(htm, bank, opts, out):
    report = Report(opts, htm.spatial_pooler.active_columns)
    for epoch in range(opts.epochs):
        for index, pattern in enumerate(bank):
            predicted_columns = htm.temporal_memory.last_state.cell_prediction.any(axis=1)
            flips = np.random.rand(opts.input_dim) < opts.input_noise_probability
            states = htm.process(pattern ^ flips)
            print(report.line(epoch, index, *column_counters(predicted_columns, *states)), file=out)


def run_batched(htm, bank, opts, out):
    width = digits(opts.epochs)
    for epoch in range(opts.epochs):
        noisy = bank ^ (np.random.rand(*bank.shape) < opts.input_noise_probability)
        began = time.time()
        htm.run(noisy, len(noisy))
        htm.engine.sync()
        rate = len(noisy) / (time.time() - began)
        print(f"epoch {epoch:{width}d}: {rate:.0f} timesteps/s, {htm.engine.info().segments} segments", file=out)


def main(argv=None, out=sys.stdout):
    opts = parse(argv)
    bank = np.random.rand(opts.input_patterns, opts.input_dim) < opts.input_density
    if opts.use_reference_implementation:
        # example.py:7-12: the same network with `temporal_memory=` the textbook implementation -- the user's package, imported
        # at the user's request (this package ships no copy of it and no CPU path of its own)
        if opts.batched:
            raise SystemExit("--batched runs the fused device step; --use_reference_implementation steps a host-side Temporal Memory")
        try:
            from bithtm.reference_implementations import TemporalMemory as ReferenceTemporalMemory
        except ImportError as e:
            raise SystemExit(f"--use_reference_implementation needs the reference's `bithtm` package on PYTHONPATH ({e})")
        htm = HierarchicalTemporalMemory(opts.input_dim, opts.column_dim, opts.cell_dim,
                                         temporal_memory=ReferenceTemporalMemory(opts.column_dim, opts.cell_dim))
    else:
        htm = HierarchicalTemporalMemory(opts.input_dim, opts.column_dim, opts.cell_dim, seed=opts.seed)
    began = time.time()
    (run_batched if opts.batched else run_stepwise)(htm, bank, opts, out)
    print(f"{time.time() - began} seconds.", file=out)


if __name__ == "__main__":
    main()
