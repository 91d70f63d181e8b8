#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the UNMODIFIED reference.

Runs only in the build container (needs /root/reference, which never travels to the GPU
box).  Usage:   python tests/golden/generate_golden.py

What is recorded
  traj_*.npz   multi-step trajectories of the reference HierarchicalTemporalMemory
               (networks.py:131-149) run through its own `boosting=` / `inhibition=` hooks
               and the keyed `np.random.rand` (oracle/ref_hooks.py): inputs, per-step
               outputs of SpatialPooler.process / TemporalMemory.process, and the synapse
               store at checkpoints.  While recording, every step is also checked against
               (a) the oracle and (b) the reference's own textbook TemporalMemory
               (reference_implementations.py:211-256, state copied with copy_custom :48-88).
  ops_*.npz    single-operator known answers from reference components with NO hooks:
               DenseProjection.process/update, ExponentialBoosting.process/update (NumPy's
               own exp, kept to measure the ulp distance of the documented exp),
               GlobalInhibition on tie-free input.
  harness_example.npz   the three counters example.py:55-57 prints, per step.
"""

import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import refdiff  # noqa: E402
from oracle import canonical_synapses  # noqa: E402
from oracle.ref_hooks import import_reference, keyed_rand  # noqa: E402


def row_crc(a):
    a = np.ascontiguousarray(a)
    return np.array([zlib.crc32(a[i].tobytes()) for i in range(a.shape[0])], dtype=np.uint32)


def ragged(chunks, dtype):
    lens = np.array([len(c) for c in chunks], dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(lens)])
    data = np.concatenate([np.asarray(c, dtype=dtype) for c in chunks]) if len(chunks) else np.zeros(0, dtype)
    return data.astype(dtype), off


def textbook_agrees(ref, ref_htm, textbook, sp_state, tm_state, pre_state_copy):
    """SURVEY §4 probe: the textbook TM, started from the vectorised TM's pre-step state and
    given the post-learning synapse store, must reproduce active cells, predictions, bursting
    flags, active / matching segment sets and potentials (everything that involves no random
    tie-break)."""
    textbook.copy_custom(ref_htm.temporal_memory)          # post-learning store
    textbook.last_state.active_cells = pre_state_copy["active_cells"]
    textbook.last_state.winner_cells = pre_state_copy["winner_cells"]
    textbook.last_state.active_segments = pre_state_copy["active_segments"]
    textbook.last_state.matching_segments = pre_state_copy["matching_segments"]
    textbook.last_state.segment_num_active_potential_synapses = pre_state_copy["potentials"]
    textbook.last_state.cell_prediction = pre_state_copy["cell_prediction"]
    out = textbook.process(sp_state, learning=False)
    d = tm_state.distal_state
    ok = np.array_equal(out.cell_activation, tm_state.cell_activation)
    ok &= np.array_equal(out.cell_prediction, tm_state.cell_prediction)
    ok &= np.array_equal(out.active_column_bursting, tm_state.active_column_bursting)
    ok &= out.matching_segments == set(d.matching_segment.tolist())
    ok &= out.active_segments == set(d.matching_segment[d.matching_segment_active].tolist())
    pot = np.array([out.segment_num_active_potential_synapses.get(s, 0) for s in range(len(d.segment_potential))])
    ok &= np.array_equal(pot, d.segment_potential)
    return bool(ok)


def snapshot_for_textbook(textbook, ref_htm):
    textbook.copy_custom(ref_htm.temporal_memory)
    ls = textbook.last_state
    return dict(active_cells=list(ls.active_cells), winner_cells=list(ls.winner_cells),
                active_segments=set(ls.active_segments), matching_segments=set(ls.matching_segments),
                potentials=dict(ls.segment_num_active_potential_synapses),
                cell_prediction=ref_htm.temporal_memory.last_state.cell_prediction.copy())


def record_trajectory(ref, name, seed, input_dim, column_dim, cell_dim, patterns, density, noise,
                      steps, checkpoints, boosted_every, learning_schedule=None, textbook_range=(0, 0),
                      sp_params=None, tm_params=None, jump=0.0):
    K = cell_dim
    rec = dict(inputs=[], learning=[], active_column=[], overlaps=[], overlaps_crc=[], bursting=[], act_bits=[],
               pred_bits=[], winner=[], matching=[], match_pot=[], match_act=[], match_active=[],
               S=[], boosted_steps=[], boosted=[], textbook=[])
    ck = {}
    textbook = ref.reference_implementations.TemporalMemory(column_dim, cell_dim)
    if tm_params is not None:      # the textbook TM hard-codes the defaults (reference_implementations.py:24-32)
        for f in tm_params.__dataclass_fields__:
            setattr(textbook, f, getattr(tm_params, f))
    pre = {}

    # textbook probe needs the pre-step state; wrap the recorder around run_lockstep's loop
    def record(t, x, learning, sp_state, tm_state, ref_htm):
        rec["inputs"].append(np.packbits(x, bitorder="little"))
        rec["learning"].append(learning)
        rec["active_column"].append(sp_state.active_column.astype(np.int32))
        rec["overlaps_crc"].append(zlib.crc32(sp_state.overlaps.astype(np.int32).tobytes()))
        rec["bursting"].append(tm_state.active_column_bursting[:, 0].copy())
        rec["act_bits"].append(np.packbits(tm_state.cell_activation.reshape(-1), bitorder="little"))
        rec["pred_bits"].append(np.packbits(tm_state.cell_prediction.reshape(-1), bitorder="little"))
        rec["winner"].append((tm_state.winner_cell[0] * K + tm_state.winner_cell[1]).astype(np.int32))
        d = tm_state.distal_state
        rec["matching"].append(d.matching_segment.astype(np.int32))
        rec["match_pot"].append(d.segment_potential[d.matching_segment].astype(np.int16))
        rec["match_act"].append(d.matching_segment_activation.astype(np.int16))
        rec["match_active"].append(d.matching_segment_active.copy())
        rec["S"].append(len(d.segment_potential))
        if t % boosted_every == 0:
            rec["boosted_steps"].append(t)
            rec["boosted"].append(sp_state.boosted_overlaps.copy())
            rec["overlaps"].append(sp_state.overlaps.astype(np.int16))
        if textbook_range[0] <= t < textbook_range[1]:
            if "snap" in pre:
                rec["textbook"].append(textbook_agrees(ref, ref_htm, textbook, sp_state, tm_state, pre["snap"]))
            pre["snap"] = snapshot_for_textbook(textbook, ref_htm)
        if t in checkpoints:
            seg_cell, presyn, perm, nsyn, segcount = refdiff.reference_store(ref_htm.temporal_memory)
            canon = canonical_synapses(seg_cell, presyn, perm)
            ids, off = ragged([c[1] for c in canon], np.int32)
            perms, _ = ragged([c[2] for c in canon], np.float32)
            sp = ref_htm.spatial_pooler
            ck[f"ck{t}_seg_cell"] = np.array(seg_cell, dtype=np.int32)      # copies: these are views
            ck[f"ck{t}_seg_nsyn"] = np.array(nsyn, dtype=np.int32)
            ck[f"ck{t}_syn_off"] = off
            ck[f"ck{t}_syn_presyn"] = ids
            ck[f"ck{t}_syn_perm"] = perms
            ck[f"ck{t}_segcount"] = np.array(segcount, dtype=np.int32)
            ck[f"ck{t}_duty"] = sp.boosting.duty_cycle.copy()
            ck[f"ck{t}_sp_perm_crc"] = row_crc(sp.proximal_projection.permanence)
            ck[f"ck{t}_segment_potential"] = d.segment_potential.astype(np.int16)
            ck[f"ck{t}_max_jittered_potential"] = d.max_jittered_potential.copy()

    stats, ref_htm, ora = refdiff.run_lockstep(
        ref, seed, input_dim, column_dim, cell_dim, patterns, density, noise, steps,
        store_every=10, learning_schedule=learning_schedule, record=record,
        sp_params=sp_params, tm_params=tm_params, jump=jump)
    assert all(rec["textbook"]), "textbook TM disagreed"

    # initial SP permanence is regenerated from the seed by consumers; keep a digest + samples
    np.random.seed(seed)
    std, mean = (sp_params.permanence_std, sp_params.permanence_mean) if sp_params is not None else (0.1, 0.0)
    perm0 = np.random.randn(column_dim, input_dim) * std + mean                   # projections.py:16
    out = dict(
        seed=seed, input_dim=input_dim, column_dim=column_dim, cell_dim=cell_dim,
        active_columns=ref_htm.active_columns, patterns=patterns, density=density, noise=noise,
        steps=steps, checkpoints=np.array(sorted(checkpoints)),
        sp_perm0_crc=row_crc(perm0), sp_perm0_head=perm0[:4, :16].copy(),
        inputs=np.stack(rec["inputs"]), learning=np.array(rec["learning"]),
        active_column=np.stack(rec["active_column"]), overlaps=np.stack(rec["overlaps"]),
        overlaps_crc=np.array(rec["overlaps_crc"], dtype=np.uint32),
        bursting=np.stack(rec["bursting"]), act_bits=np.stack(rec["act_bits"]),
        pred_bits=np.stack(rec["pred_bits"]), S=np.array(rec["S"]),
        boosted_steps=np.array(rec["boosted_steps"]), boosted=np.stack(rec["boosted"]),
        textbook_checked_steps=len(rec["textbook"]), ambiguous_topk_steps=stats["ambiguous_topk"],
        final_slots=stats["slots"],
    )
    for key, dtype in (("winner", np.int32), ("matching", np.int32), ("match_pot", np.int16),
                       ("match_act", np.int16), ("match_active", np.bool_)):
        out[key], out[key + "_off"] = ragged(rec[key], dtype)
    for prefix, params in (("sp_", sp_params), ("tm_", tm_params)):
        if params is not None:
            for f in params.__dataclass_fields__:
                out[prefix + f] = getattr(params, f)
    out["jump"] = jump
    out["predicted_columns_per_step"] = (~out["bursting"]).sum(axis=1)
    out.update(ck)
    path = os.path.join(HERE, f"traj_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: {os.path.getsize(path) / 1e6:.2f} MB, S={stats['segments']}, slots={stats['slots']}, "
          f"textbook-checked steps={len(rec['textbook'])}, ambiguous top-k steps={stats['ambiguous_topk']}")


def record_ops(ref):
    """Known answers from reference components with no hooks at all."""
    np.random.seed(11)
    C, I, k = 512, 200, 10
    proj = ref.projections.DenseProjection(I, C)
    perm0 = proj.permanence.copy()
    x = np.random.rand(I) < 0.15
    overlaps = proj.process(x)
    rows = np.sort(np.random.choice(C, k, replace=False))
    proj.update(x, rows)
    boost = ref.regularizations.ExponentialBoosting(C, k)
    duty_seq, boosted_seq = [], []
    for _ in range(30):
        act = np.sort(np.random.choice(C, k, replace=False))
        boost.update(act)
        duty_seq.append(boost.duty_cycle.copy())
        boosted_seq.append(boost.process(overlaps))
    act_seq_seed = 11
    inhib = ref.regularizations.GlobalInhibition(k)
    vals = np.random.permutation(C).astype(np.float64) * 0.37       # tie-free
    top = np.sort(inhib.process(vals))
    np.savez_compressed(
        os.path.join(HERE, "ops_sp.npz"),
        perm0=perm0, input=x, overlaps=overlaps, updated_rows=rows, perm_after_rows=proj.permanence[rows].copy(),
        duty_seq=np.stack(duty_seq), boosted_seq=np.stack(boosted_seq), seed=act_seq_seed, k=k,
        topk_values=vals, topk_sorted=top)
    print("ops_sp.npz written")


def record_harness(ref):
    """example.py:34-65 with the deterministic hooks: the three printed counters per step
    (example.py defaults: 100 patterns, density 0.2, noise 0.05, 2048 x 32)."""
    seed, I, C, K, P, steps = 21, 1000, 2048, 32, 100, 600
    ref_htm, _ = refdiff.build_pair(ref, seed, I, C, K)
    rng = np.random.RandomState(seed + 1)
    bank = rng.rand(P, I) < 0.2
    counters = []
    with keyed_rand(seed, K) as patch:
        for t in range(steps):
            prev_col_pred = ref_htm.temporal_memory.last_state.cell_prediction.max(axis=1)   # example.py:50
            x = bank[t % P] ^ (rng.rand(I) < 0.05)                                           # example.py:52
            patch.step = t
            sp_state, tm_state = ref_htm.process(x)
            burst = int(tm_state.active_column_bursting.sum())                                # :55
            correct = int(prev_col_pred[sp_state.active_column].sum())                        # :56
            incorrect = int(prev_col_pred.sum() - correct)                                    # :57
            counters.append((burst, correct, incorrect))
    np.savez_compressed(os.path.join(HERE, "harness_example.npz"), seed=seed, input_dim=I, column_dim=C,
                        cell_dim=K, patterns=P, steps=steps, density=0.2, noise=0.05,
                        counters=np.array(counters, dtype=np.int32))
    c = np.array(counters)
    print("harness_example.npz written; mean bursting per epoch",
          [round(float(c[e * P:(e + 1) * P, 0].mean()), 1) for e in range(steps // P)])


def main():
    from oracle import SPParams, TMParams
    ref = import_reference()
    record_ops(ref)
    # K=8, small; a learning=False stretch in the middle
    record_trajectory(ref, "small_k8", seed=5, input_dim=300, column_dim=1024, cell_dim=8, patterns=60,
                      density=0.1, noise=0.03, steps=480, checkpoints={100, 300, 479}, boosted_every=20,
                      learning_schedule=lambda t: not (300 <= t < 310), textbook_range=(440, 480))
    # example.py defaults (config 1): I=1000 (not a multiple of 32), C=2048, K=32, 100 patterns
    record_trajectory(ref, "example_default", seed=7, input_dim=1000, column_dim=2048, cell_dim=32, patterns=100,
                      density=0.2, noise=0.05, steps=520, checkpoints={200, 519}, boosted_every=40,
                      textbook_range=(505, 520))
    # K=16 (config 5's cell count), sparse input like configs 2-5
    record_trajectory(ref, "sparse_k16", seed=9, input_dim=512, column_dim=4096, cell_dim=16, patterns=50,
                      density=0.04, noise=0.005, steps=400, checkpoints={399}, boosted_every=40)
    # non-default parameters everywhere + random jumps in the sequence: punishments, pruning,
    # segment recycling (projections.py:79-85)
    record_trajectory(
        ref, "stress_params", seed=13, input_dim=256, column_dim=1024, cell_dim=4, patterns=40, density=0.12,
        noise=0.02, steps=400, checkpoints={150, 399}, boosted_every=40, jump=0.15, textbook_range=(380, 400),
        sp_params=SPParams(permanence_mean=0.01, permanence_std=0.08, permanence_threshold=0.02,
                           permanence_increment=0.05, permanence_decrement=0.02, boost_intensity=0.5,
                           boost_momentum=0.95),
        tm_params=TMParams(permanence_initial=0.3, permanence_threshold=0.45, permanence_increment=0.12,
                           permanence_decrement=0.14, permanence_punishment=0.2, segment_activation_threshold=12,
                           segment_matching_threshold=9, segment_sampling_synapses=20))
    record_harness(ref)


if __name__ == "__main__":
    main()
