"""What the reference's example.py loop costs as it stands (bithtm_amd/example.py: run_stepwise): one htm.process(x) per
timestep AND two States read back per step (predicted columns of the previous step, bursting / active columns of this one) --
the read-back of whole State objects (14 field reads and their conversions to the reference's bool / index arrays), not the
timestep, is what bounds it.  Prints the rate at 2 048 and 65 536 columns and the top of a profile of one epoch.

    python tools/example_loop_rate.py
"""
import io, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from bithtm_amd import example as E
from bithtm_amd import HierarchicalTemporalMemory
for cols in (2048, 65536):
    opts = E.parse(["--epochs", "3", "--column_dim", str(cols)])
    np.random.seed(0)
    bank = np.random.rand(opts.input_patterns, opts.input_dim) < opts.input_density
    htm = HierarchicalTemporalMemory(opts.input_dim, opts.column_dim, opts.cell_dim)
    out = io.StringIO()
    E.run_stepwise(htm, bank, E.parse(["--epochs", "1", "--column_dim", str(cols)]), out)      # warm
    t0 = time.perf_counter()
    E.run_stepwise(htm, bank, opts, out)
    dt = time.perf_counter() - t0
    n = opts.epochs * opts.input_patterns
    print(f"{cols} columns: example.py's stepwise loop (two States read per step): {n / dt:.0f} timesteps/s ({1e6 * dt / n:.0f} us/step)")
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); E.run_stepwise(htm, bank, E.parse(["--epochs", "1", "--column_dim", str(cols)]), out); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14); print("\n".join(s.getvalue().splitlines()[:32]))
