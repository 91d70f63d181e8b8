"""Counterpart of the reference's example.py (flags :22-30, input generation :34, per-step noise
:52, the three printed counters :55-57) running on the MI355X engine.

    python -m bithtm_amd.example --epochs 8
    python -m bithtm_amd.example --epochs 8 --batched      # one C-ABI call per epoch, hipGraph replay
"""

import argparse
import time

import numpy as np

from bithtm_amd import HierarchicalTemporalMemory


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--epochs', type=int, default=100)
    parser.add_argument('--input_patterns', type=int, default=100)
    parser.add_argument('--input_dim', type=int, default=1000)
    parser.add_argument('--input_density', type=float, default=0.2)
    parser.add_argument('--input_noise_probability', type=float, default=0.05)
    parser.add_argument('--column_dim', type=int, default=2048)
    parser.add_argument('--cell_dim', type=int, default=32)
    parser.add_argument('--seed', type=int, default=0, help='keyed random draws of the Temporal Memory')
    parser.add_argument('--batched', action='store_true',
                        help='no per-step read-back: run each epoch with HierarchicalTemporalMemory.run and '
                             'print timesteps/s per epoch instead of the per-step counters')
    args = parser.parse_args(argv)

    inputs = np.random.rand(args.input_patterns, args.input_dim) < args.input_density
    htm = HierarchicalTemporalMemory(args.input_dim, args.column_dim, args.cell_dim, seed=args.seed)

    def width(n):
        return int(np.ceil(np.log10(max(n - 1, 2))))
    ew, pw, cw, aw = width(args.epochs), width(args.input_patterns), width(args.column_dim), width(htm.spatial_pooler.active_columns)

    start_time = time.time()
    for epoch in range(args.epochs):
        if args.batched:
            noisy = inputs ^ (np.random.rand(*inputs.shape) < args.input_noise_probability)
            t0 = time.time()
            htm.run(noisy, len(noisy))
            htm.engine.sync()
            print(f'epoch {epoch:{ew}d}: {len(noisy) / (time.time() - t0):.0f} timesteps/s, '
                  f'{htm.engine.info().segments} segments')
            continue
        for input_index, curr_input in enumerate(inputs):
            prev_column_prediction = htm.temporal_memory.last_state.cell_prediction.max(axis=1)
            noisy_input = curr_input ^ (np.random.rand(args.input_dim) < args.input_noise_probability)
            sp_state, tm_state = htm.process(noisy_input)
            burstings = tm_state.active_column_bursting.sum()
            corrects = prev_column_prediction[sp_state.active_column].sum()
            incorrects = prev_column_prediction.sum() - corrects
            print(f'epoch {epoch:{ew}d}, pattern {input_index:{pw}d}: bursting columns: {burstings:{aw}d}, '
                  f'correct columns: {corrects:{aw}d}, incorrect columns: {incorrects:{cw}d}')
    print(f'{time.time() - start_time} seconds.')


if __name__ == '__main__':
    main()
