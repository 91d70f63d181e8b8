#!/usr/bin/env python3
"""Headline benchmark: HTM timesteps/s (SP + TM, learning on) at 65 536 columns x 32 cells.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full timestep of the hot path (SpatialPooler.process + TemporalMemory.process
with learning, networks.py:146-149) over one synthetic input.  The input bank (Bernoulli-sparse
patterns, cycled, with flip noise: BASELINE.md section 4) is resident in HBM before the timed
region.  Prints ONE JSON line (rank 0).

Besides the contract fields the line carries
  roofline      the dominant launch of the TIMED schedule: algorithmic bytes of its roles (formulas in DESIGN.md
                section 4 and in role_bytes below) / its average duration, measured here with HIP events on the
                engine's stream over an eager replay of the same schedule; every launch is listed;
  cpu_baseline  the NumPy oracle (a port of the reference's CPU path) timed on this box's host
                cores from the same learned state, on a bounded sample of steps;
  stress        BASELINE.json configs[4] as far as one GPU holds it: one rank's pre-populated shard (133.7 M
                segments) inside an 8-rank group, its segment scan against the HBM roofline.

The model is brought to the learned state (10 passes over the pattern bank) in untimed setup, whatever --warmup
says; `value` is the median of several repetitions of [W warm-up steps, exactly K timed steps].
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (this pool's driver supports dmabuf IPC only: RCCL across processes fails with `hipIpcGetMemHandle: invalid argument` without
# it.  The launcher exports it; keep it when somebody runs the line by hand.)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

WORKLOAD = dict(input_dim=1024, column_dim=65536, cell_dim=32, patterns=50, density=0.02, noise=0.005,
                noisy_copies=20, segment_slots=128)
# The same shape with a larger learned pool ("40 % of roofline and 10^4 steps/s coincide only if ... the
# segment pool is larger (S*E*8 ~ 300 MB, e.g. P ~ 350 patterns)"): 350 patterns, 10 passes of untimed pre-training.  The
# `large_pool` leg of the line; the headline stays P = 50.
LARGE_POOL = dict(WORKLOAD, patterns=350, noisy_copies=4, segment_capacity=2 << 20)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
PMC_FILES = ("r04_pmc_summary.json", "r03_pmc_summary.json")      # recorded PMC passes, newest first


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_inputs(w, seed=0):
    """seed 0; pattern bank rand(P, I) < d; SP permanences randn(C, I) * 0.1
    from the same stream; step t feeds pattern (t mod P) XOR flip noise."""
    np.random.seed(seed)
    P, I, C = w["patterns"], w["input_dim"], w["column_dim"]
    bank = np.random.rand(P, I) < w["density"]
    perm = np.random.randn(C, I) * 0.1
    noisy = np.empty((P * w["noisy_copies"], I), dtype=np.bool_)
    for r in range(w["noisy_copies"]):
        for p in range(P):
            noisy[r * P + p] = bank[p] ^ (np.random.rand(I) < w["noise"])
    return noisy, perm


def build_htm(w, perm, device, column_range=None):
    import bithtm_amd as B
    C, I, K = w["column_dim"], w["input_dim"], w["cell_dim"]
    k = round(C * 0.02)
    proximal = B.DenseProjection.__new__(B.DenseProjection)        # skip the RNG draw: perm is given
    proximal.input_dim, proximal.output_dim = I, C
    proximal.permanence_threshold, proximal.permanence_increment, proximal.permanence_decrement = 0.0, 0.03, 0.015
    proximal._engine, proximal._permanence = None, perm
    sp = B.SpatialPooler(I, C, k, proximal_projection=proximal)
    tm = B.TemporalMemory(C, K, distal_projection=B.PredictiveProjection(C * K, segment_slots=w["segment_slots"],
                                                                         segment_capacity=w.get("segment_capacity")), seed=0,
                          device=device)
    return B.HierarchicalTemporalMemory(I, C, K, active_columns=k, spatial_pooler=sp, temporal_memory=tm, device=device)


# the launches of the pipelined schedules (htm_pipeline.h) and the roles each holds: three per timestep (BITHTM_LEAN=1) ...
# (the last launch's name says which form of the scan it holds: small pools are scanned by blocks that are all resident at
# once, pools of more than ~295 k segments by the streaming form -- htm_engine.hip: scan_pool_is_large)
LEAN_KERNEL = {"tm_activate+sp_learn": "k_act_rows", "tm_mid+sp_overlap": "k_mid_overlap", "tm_learn+tm_scan+sp_emit": "k_learn_scan_emit",
               "tm_learn+tm_scan_large+sp_emit": "k_learn_scan_emit"}
LEAN_ROLES = {"tm_activate+sp_learn": ("tm_activate", "sp_rows", "tm_clear"), "tm_mid+sp_overlap": ("tm_mid", "sp_overlap"),
              "tm_learn+tm_scan+sp_emit": ("tm_learn", "tm_scan", "sp_emit"), "tm_learn+tm_scan_large+sp_emit": ("tm_learn", "tm_scan", "sp_emit")}
# ... two (the default: the first two in one, the middle role behind an in-launch fan-in of the activation blocks) ...
LEAN2_KERNEL = {"tm_activate+tm_mid+sp_learn+sp_overlap": "k_act_mid_rows", "tm_learn+tm_scan+sp_emit": "k_learn_scan_emit",
                "tm_learn+tm_scan_large+sp_emit": "k_learn_scan_emit"}
LEAN2_ROLES = {"tm_activate+tm_mid+sp_learn+sp_overlap": ("tm_activate", "sp_rows", "tm_clear", "tm_mid", "sp_overlap"),
               "tm_learn+tm_scan+sp_emit": ("tm_learn", "tm_scan", "sp_emit"), "tm_learn+tm_scan_large+sp_emit": ("tm_learn", "tm_scan", "sp_emit")}
# ... or four (BITHTM_LEAN=0)
LAUNCH_KERNEL = {"tm_activate+sp_emit": "k_open_emit", "tm_mid+sp_learn": "k_mid_rows",
                 "tm_learn+sp_overlap": "k_learn_overlap", "tm_scan+sp_select": "k_scan_sel", "tm_scan_large+sp_select": "k_scan_sel"}
LAUNCH_ROLES = {"tm_activate+sp_emit": ("tm_activate", "sp_emit"), "tm_mid+sp_learn": ("tm_mid", "sp_rows"),
                "tm_learn+sp_overlap": ("tm_learn", "sp_overlap"), "tm_scan+sp_select": ("tm_scan", "sp_select", "tm_clear"),
                "tm_scan_large+sp_select": ("tm_scan", "sp_select", "tm_clear")}


def launches_seen(prof, table):
    """The launches of a pipelined schedule among the profiled names: every launch of `table` that ran (the last launch has two
    names, one per form of the scan -- a run that crosses the threshold shows both), or None if a launch of the schedule is missing."""
    ran = {n: prof[n] for n in table if prof.get(n, (0, 0))[1]}
    kernels = {table[n] for n in table}
    return ran if {table[n] for n in ran} == kernels else None


def role_bytes(w, k, seg_nsyn, n_work, n_match):
    """Algorithmic HBM bytes per timestep of every role ('Kernels and rooflines'):
    what the role has to read and write once, whatever the kernel actually fetches."""
    C, I = w["column_dim"], w["input_dim"]
    W = ((I + 127) // 128) * 4                     # packed words per SP row
    nsyn = seg_nsyn.astype(np.int64)
    S, syn = len(nsyn), int(nsyn.sum())
    mean_syn = syn / max(S, 1)
    return {
        # mask rows + input + duty read + overlap / boosted / key writes (+ the select histogram: LDS, then a few KB of atomics)
        "sp_overlap": C * W * 4 + W * 4 + C * 4 + C * (4 + 8 + 8),
        # digit 1: the keys once + the digit-0 histogram copies
        "sp_select": C * 8 + 4 * 4096 * 4,
        # select finish + winner list: histogram copies, keys, per-block records (write + read), column bitmap, list
        "sp_emit": 4 * 4096 * 4 + C * 8 + 2 * 128 * ((C + 255) // 256) + C // 8 + 4 * k,
        # k winner rows: float64 read + write, mask row rewrite; duty cycle read + write, column bitmap
        "sp_rows": 2 * 8 * k * I + k * W * 4 + 2 * 4 * C + C // 8,
        # per winner column: previous prediction word, 32 x (cell maximum + segment count), six result words
        "tm_activate": k * (4 + 32 * 8 + 24),
        # lists of block 0 + classification: info word and owner cell of every segment, jitter / maximum / words of matching ones
        "tm_mid": 13 * k + 8 * S + 12 * n_match,
        # learning / punished rows: presynaptic ids + permanences, read and written; the info words for the sparse clear
        "tm_learn": int(16 * mean_syn * n_work) + 4 * S,
        # packed presynaptic ids of every segment + synapse count read and info word written per segment
        "tm_scan": 4 * syn + 8 * S,
        # dense per-column words of the coming step
        "tm_clear": 12 * C,
    }


def recorded_traffic(kernel, files=PMC_FILES, sum_nsyn=None, section=None, variant=None):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/r03_pmc_summary.json: rocprofv3 --pmc
    FETCH_SIZE and --pmc WRITE_SIZE in separate runs of this same command and schedule; KB units; FETCH_SIZE doubled as
    MI355X_MICROARCH.md section HBM prescribes for gfx950).  PMC counters cannot be read from inside this process, so this is a
    recorded value, not a live one -- and only used where the recorded pass saw the model in the state this run is in
    (the synapses in the pool, which the scan's traffic follows, within 5 %): otherwise traffic is null and says why."""
    for name in files:
        path = os.path.join(ROOT, "profiles", name)
        try:
            d = json.load(open(path))
            if section:                             # (a leg of the line recorded beside the headline: its own counters and state)
                d = d[section]
            # template kernels: "void k_scan_sel<true, 6>"; variant: what the instantiation's name must hold besides (the large-pool
            # form of the last launch is k_learn_scan_emit<*, 4, *>: a run that pre-trains through the threshold launches both)
            key = next(k for k in d["FETCH_SIZE"] if kernel in k and (variant is None or variant in k))
            f = d["FETCH_SIZE"][key]["mean_last150_KB"]
            wr = d["WRITE_SIZE"][key]["mean_last150_KB"]
            rec = d.get("state", {}).get("sum_nsyn")
            if sum_nsyn is not None and rec is not None and abs(rec - sum_nsyn) > 0.05 * sum_nsyn:
                return dict(traffic=None, traffic_source=f"profiles/{name} was recorded with {rec} synapses in the pool, this run has {sum_nsyn}: not comparable")
            return dict(traffic=int((2 * f + wr) * 1024), traffic_source=f"profiles/{name} (recorded PMC pass of this command, 2*FETCH_SIZE + WRITE_SIZE"
                                                                          + (f"; pool of {rec} synapses then, {sum_nsyn} now)" if rec is not None and sum_nsyn is not None else ")"))
        except Exception:
            continue
    return dict(traffic=None)


def cpu_baseline(w, htm, noisy, bank, start_step, sample_steps, run, chunk_plan=(5, 20, 5)):
    """Time the NumPy oracle on this host from the GPU's learned state (a port of the
    reference's CPU path: dense float64 `>=` + `&` + sum overlap, NumPy segment scan) -- and then let it check the
    bench's own call pattern at the full size: the GPU runs the same timesteps the way the timed region made its calls
    (streamed chunks with HTM_RUN_CONTINUE, graphs built ahead with htm_prepare) and must agree with the oracle at every
    chunk boundary and, at the end, on the whole state."""
    from oracle import HTMOracle
    eng = htm.engine
    C, I, K = w["column_dim"], w["input_dim"], w["cell_dim"]
    ora = HTMOracle(I, C, K, seed=0, permanence=eng.get_permanence())
    ora.spatial_pooler.duty_cycle = eng.read_duty_cycle().copy()
    ora.temporal_memory.import_state(eng.export_tm_state())
    n_bank = noisy.shape[0]
    states = [ora.step(noisy[start_step % n_bank])]     # untimed: page in / allocate
    t0 = time.perf_counter()
    for t in range(1, sample_steps + 1):
        states.append(ora.step(noisy[(start_step + t) % n_bank]))
    dt = time.perf_counter() - t0
    out = dict(value=sample_steps / dt, unit="timesteps/s", cores=1, kind="port",
               sample=f"{sample_steps} timesteps of the NumPy oracle from the GPU's learned state "
                      f"(S={ora.temporal_memory.S} segments), single-threaded NumPy")
    try:
        out.update(check_against_oracle(w, htm, noisy, bank, ora, states, run, chunk_plan))
    except AssertionError as e:
        log(f"[bench] PARITY FAILURE against the oracle: {e}")
        out.update(parity=f"FAILED: {e}")
    return out


def check_against_oracle(w, htm, noisy, bank, ora, states, run, chunk_plan=(5, 20, 5)):
    """The GPU's next len(states) timesteps, called as the timed region calls (run = its use_graph / pipeline flags),
    against the oracle's: Temporal Memory outputs at every chunk boundary, everything at the end.  The chunks: `chunk_plan`
    ([5 warm-up, 20 timed] as the driver's arguments make them, ...) and then the rest in one call -- long enough, with the
    default sample, to replay hipGraphs where the timed region does (htm_run launches calls of fewer than 64 steps eagerly):
    what each chunk did is asked of the library (htm_run_plan) and reported."""
    from bithtm_amd import _lib as L
    from bithtm_amd.engine import bool_to_words
    eng = htm.engine
    K, k = w["cell_dim"], htm.active_columns
    n_bank = noisy.shape[0]
    n = len(states)
    cont = bool(run["pipeline"])
    chunks, left = [], n
    for c in chunk_plan:
        if left > c:
            chunks.append(c)
            left -= c
    chunks.append(left)
    done, modes = 0, []
    for i, c in enumerate(chunks):
        last = i == len(chunks) - 1
        flags = dict(run, continuing=cont and not last)
        plan = eng.run_plan(c, **flags)
        modes.append(("graph" if plan["hip_graph"] else "eager") + ("+large-pool scan" if plan["scan_large"] else ""))
        eng.prepare(bank, n_bank, c, **flags)
        eng.run(bank, n_bank, c, **flags)
        done += c
        o_sp, o_tm = states[done - 1]
        od = o_tm.distal_state
        info = eng.check_capacity()
        assert info.segments == len(od.segment_potential), f"segment count after step {done}"
        assert np.array_equal(eng.read(L.F_ACTIVE_COLUMN, np.int32, k), o_sp.active_column), f"active columns of step {done}"
        assert np.array_equal(eng.read(L.F_CELL_ACTIVATION, np.uint32, w["column_dim"]), bool_to_words(o_tm.cell_activation)), f"cell activation of step {done}"
        assert np.array_equal(eng.read(L.F_CELL_PREDICTION, np.uint32, w["column_dim"]), bool_to_words(o_tm.cell_prediction)), f"cell prediction of step {done}"
        assert np.array_equal(eng.read(L.F_WINNER_CELL, np.int32, info.winner_cells), o_tm.winner_cell[0] * K + o_tm.winner_cell[1]), f"winner cells of step {done}"
        assert np.array_equal(eng.read(L.F_BURSTING, np.uint8, k).astype(bool), o_tm.active_column_bursting[:, 0]), f"bursting columns of step {done}"
        d = eng.read_distal()
        assert np.array_equal(d["matching_segment"], od.matching_segment), f"matching segments of step {done}"
        assert np.array_equal(d["matching_segment_activation"], od.matching_segment_activation), f"connected-active counts of step {done}"
        assert np.array_equal(d["matching_segment_active"], od.matching_segment_active), f"active segments of step {done}"
        assert np.array_equal(d["segment_potential"], od.segment_potential), f"segment potentials of step {done}"
        assert np.array_equal(d["max_jittered_potential"].view(np.int32), od.max_jittered_potential.view(np.int32)), f"per-cell maxima of step {done}"
    # the stream has ended: the Spatial Pooler's fields and the whole state
    o_sp, o_tm = states[-1]
    sp = eng.read_sp_fields()
    assert np.array_equal(sp["overlaps"], o_sp.overlaps), "overlaps of the last step"
    assert np.array_equal(sp["boosted_overlaps"].view(np.int64), o_sp.boosted_overlaps.view(np.int64)), "boosted overlaps of the last step"
    otm, osp = ora.temporal_memory, ora.spatial_pooler
    st = eng.read_store()
    S = otm.S
    assert st["S"] == S and np.array_equal(st["seg_cell"], otm.seg_cell[:S]) and np.array_equal(st["seg_nsyn"], otm.seg_nsyn[:S]), "segment owners / synapse counts"
    assert np.array_equal(st["segcount"], otm.segcount), "segments per cell"

    def canonical(presyn, perm, width):                # valid synapses first, by presynaptic id
        key = np.where(presyn >= 0, presyn.astype(np.int64), np.int64(1) << 40)
        order = np.argsort(key, axis=1, kind="stable")[:, :width]
        return np.take_along_axis(presyn, order, axis=1), np.take_along_axis(perm, order, axis=1).view(np.int32)
    width = int(max(st["seg_nsyn"].max(initial=0), 1))
    a_ps, a_pm = canonical(st["presyn"], st["perm"], width)
    b_ps, b_pm = canonical(otm.presyn[:S], otm.perm[:S], width)
    assert np.array_equal(a_ps, b_ps), "presynaptic cells of the segment store"
    assert np.array_equal(np.where(a_ps >= 0, a_pm, 0), np.where(b_ps >= 0, b_pm, 0)), "permanence bits of the segment store"
    assert np.array_equal(eng.read_duty_cycle().view(np.int32), osp.duty_cycle.view(np.int32)), "duty cycle"
    assert np.array_equal(eng.get_permanence().view(np.int64), osp.permanence.view(np.int64)), "Spatial Pooler permanences"
    called = ", ".join(f"{c} ({m})" for c, m in zip(chunks, modes))
    log(f"[bench] parity: {n} timesteps in chunks of {called} (streamed calls, prepared graphs) equal the oracle's at every chunk "
        f"boundary; whole state equal at the end (S={S})")
    return dict(parity="ok", parity_checked=f"{n} timesteps called as the timed region calls them (chunks of {called}; HTM_RUN_CONTINUE={cont}, "
                                            f"htm_prepare): TM outputs at every chunk boundary, SP outputs and the whole "
                                            f"state (segment store, permanence bits, duty cycle) at the end, bit for bit")


STRESS = dict(input_dim=1024, column_dim=262144, cell_dim=16, world=8, segments_per_cell=255, synapses=32,
              perm_lo=0.3, perm_hi=0.7, seed=0, segment_slots=64, patterns=8, density=0.02)
if os.environ.get("BITHTM_STRESS_SPC"):           # experiments: a smaller pool (segments per cell)
    STRESS["segments_per_cell"] = int(os.environ["BITHTM_STRESS_SPC"])


class LazyPermanence:
    """Rows [a, b) of a randn(C, I) * 0.1 matrix, generated when sliced (a rank only ever reads its own rows)."""

    def __init__(self, columns, inputs, seed):
        self.shape, self.seed = (columns, inputs), seed

    def __getitem__(self, rows):
        a, b, _ = rows.indices(self.shape[0])
        return np.random.RandomState(self.seed + a).randn(b - a, self.shape[1]) * 0.1


def check_stress_sample(eng, s, rows0, k, ranges=48, rows_per_range=256):
    """Self-check of the configs[4] leg (the oracle's keyed generator as the checker): `ranges` x `rows_per_range` rows spread
    over rank 0's pool are read back and compared with the rows the keyed generator gives those segment ids on the host --
    presynaptic cells, permanence bits, owner cells; their potentials against the last step's active cells as the host
    counts them; and the scan's verdict on each (the match word of k_tm_scan_wide: set, with that potential, exactly where
    the potential reaches the matching threshold)."""
    from oracle import populated_rows
    from bithtm_amd import _lib as L
    from bithtm_amd.engine import words_to_bool
    C, K, spc, n_syn = s["column_dim"], s["cell_dim"], s["segments_per_cell"], s["synapses"]
    act = words_to_bool(eng.read(L.F_CELL_ACTIVATION, np.uint32, C), K).reshape(-1)
    assert int(act.sum()) >= k, "the last step has active cells"
    starts = np.linspace(0, rows0 - rows_per_range, ranges).astype(np.int64)
    n_rows = n_match = 0
    for a in starts:
        gid = eng.read_rows(L.F_SEG_GID, np.int32, a, rows_per_range).astype(np.int64)
        assert np.array_equal(gid, np.arange(a, a + rows_per_range)), f"rows {a}..: global ids"
        cells, perms = populated_rows(C * K, gid, n_syn, s["perm_lo"], s["perm_hi"], s["seed"])
        presyn = eng.read_rows(L.F_SEG_PRESYN, np.int32, a, rows_per_range)
        perm = eng.read_rows(L.F_SEG_PERM, np.float32, a, rows_per_range)
        assert (presyn[:, n_syn:] == -1).all() and np.array_equal(presyn[:, :n_syn], cells), f"rows {a}..: presynaptic cells"
        assert np.array_equal(perm[:, :n_syn].view(np.int32), perms.view(np.int32)), f"rows {a}..: permanence bits"
        assert np.array_equal(eng.read_rows(L.F_SEG_CELL, np.int32, a, rows_per_range), gid // spc), f"rows {a}..: owner cells"
        pot = act[cells].sum(axis=1)
        assert np.array_equal(eng.read_rows(L.F_SEG_POTENTIAL, np.int32, a, rows_per_range), pot), f"rows {a}..: potentials"
        conn = (act[cells] & (perms >= np.float32(0.5))).sum(axis=1)
        matching = pot >= 15                           # PredictiveProjection defaults (projections.py:221-222)
        want = np.where(matching, pot.astype(np.uint32) | (conn.astype(np.uint32) << 12) | ((conn >= 15).astype(np.uint32) << 31), 0).astype(np.uint32)
        assert np.array_equal(eng.read_rows(L.F_MATCH_INFO, np.uint32, a, rows_per_range), want), f"rows {a}..: the scan's match words"
        n_rows += rows_per_range
        n_match += int(matching.sum())
    log(f"[bench] configs[4] leg: self-check of {n_rows} sampled rows ok ({n_match} of them matching)")
    # (a uniformly random pool matches nothing under the default thresholds -- SURVEY section 8d: a pure scan stress -- so of
    # the scan's verdict only "no match word where the potential is below the threshold" is seen here; what a MATCHING row's
    # word holds is pinned at this shape by tests/test_hip_populate.py, thresholds lowered until 3 % of the rows match)
    return dict(checked=True if n_match else "partial: rows, potentials and the non-matching verdicts (no sampled row matches under the default "
                                             "thresholds); matching rows of this kernel at this shape: tests/test_hip_populate.py",
                check=f"{n_rows} rows in {ranges} ranges over rank 0's pool: presynaptic cells, permanence bits and owner cells equal the "
                                    f"keyed generator's for those ids; potentials and the scan's match words equal the host's count against the "
                                    f"last step's active cells ({n_match} matching)")


def stress_leg(steps=8, warmup=3, check=True):
    """BASELINE.json configs[4] (262 144 columns x 16 cells, 255 segments per cell, 8 GPUs: the HBM-bound scan stress of
    SURVEY section 8d) as far as one GPU can hold it: the whole 8-rank group runs in this process (bithtm_amd.distributed.
    LocalGroup: every sharded kernel, the all-gather as device copies), with the pre-populated pool generated on the
    device for rank 0's cells only -- exactly one rank's shard of the pool, 133.7 M segments x 32 synapses.  Reported:
    rank 0's segment scan (the dominant kernel) against the HBM roofline, and rank 0's whole step."""
    import bithtm_amd as B
    from bithtm_amd.distributed import LocalGroup
    s = STRESS
    C, K, I, world, spc = s["column_dim"], s["cell_dim"], s["input_dim"], s["world"], s["segments_per_cell"]
    k = round(C * 0.02)
    own_cells = C // world * K
    rows0 = own_cells * spc
    growth = 64 * k                                   # ids for the segments the bursting columns grow meanwhile
    perm = LazyPermanence(C, I, 12345)

    def parts(r):
        return dict(distal=B.PredictiveProjection(C * K, segment_capacity=rows0 + growth, segment_slots=s["segment_slots"],
                                                  segment_capacity_local=(rows0 if r == 0 else 0) + growth))
    t0 = time.perf_counter()
    group = LocalGroup(world, I, C, K, active_columns=k, permanence=perm, make_parts=parts, seed=0)
    for e in group.engines:                           # every rank is told the same range: rank 0's cells
        e.populate(spc, synapses=s["synapses"], perm_lo=s["perm_lo"], perm_hi=s["perm_hi"], seed=s["seed"], cell_begin=0, cell_end=own_cells)
    rng = np.random.RandomState(7)
    group.upload_bank(rng.rand(s["patterns"], I) < s["density"])
    group.run(warmup)
    eng = group.engines[0]
    eng.sync()
    setup_s = time.perf_counter() - t0
    eng.profile(True)
    group.run(steps)
    prof = {n: 1e3 * ms / cnt for n, (ms, cnt) in eng.profile_read().items() if cnt}
    eng.profile(False)
    info = eng.check_capacity()
    rows = info.local_segments
    scan_bytes = (4 * s["synapses"] + 8) * rows0 + (4 * s["synapses"] + 8) * (rows - rows0)     # 4 B per synapse + 8 B per segment
    scan_name = next(n for n in ("tm_scan_wide", "tm_scan_large", "tm_scan") if n in prof)
    scan_us = prof[scan_name]
    ach = scan_bytes / scan_us / 1e3
    out = dict(workload=f"configs[4] on one GPU: rank 0 of {world} of {C} columns x {K} cells; its cells' {spc} segments x "
                        f"{s['synapses']} synapses each ({rows0} segments, generated on the device); all {world} ranks run in this process",
               kernel={"tm_scan_wide": "k_tm_scan_wide", "tm_scan_large": "k_tm_scan<true, 1>", "tm_scan": "k_tm_scan<true, 6>"}[scan_name] + " (the scan of rank 0)",
               segments=int(rows), bytes_per_launch=int(scan_bytes),
               avg_launch_us=round(scan_us, 1), achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
               rank0_launch_us={n: round(v, 1) for n, v in prof.items()}, rank0_step_us=round(sum(prof.values()), 1),
               exchange_bytes_per_rank=int(eng.shard_record_bytes()), setup_s=round(setup_s, 1))
    out.update(recorded_traffic("k_tm_scan_wide"))         # (a generated pool: the same in every run)
    if check:
        try:
            out.update(check_stress_sample(eng, s, rows0, k))
        except AssertionError as e:
            log(f"[bench] configs[4] leg: SELF-CHECK FAILED: {e}")
            out.update(checked=False, check_error=str(e))
    log(f"[bench] configs[4] leg: rank 0 scans {rows} segments in {scan_us:.0f} us = {ach:.0f} GB/s ({ach / HBM_PEAK_GBS:.1%} of peak); "
        f"its step {out['rank0_step_us']:.0f} us; setup {setup_s:.1f} s")
    del group
    return out


def measure(w, args, device, label, reps, cpu_steps, chunk_plan, pmc_section=None):
    """One workload through the bench protocol: untimed pre-training to the learned state, R repetitions of [W warm-up
    steps, exactly K timed steps] called as a streaming caller calls, per-launch HIP-event times of the timed schedule,
    the roofline object, and (cpu_steps > 0) the oracle timed from the same state + the parity check of the call pattern."""
    t_setup = time.perf_counter()
    noisy, perm = make_inputs(w)
    htm = build_htm(w, perm, device)
    del perm
    eng = htm.engine
    k = htm.active_columns
    bank = eng.upload_bank(noisy)
    n_bank = noisy.shape[0]
    # rocprofv3 crashes inside hipGraph replay on this image: under the profiler, launch eagerly
    under_profiler = "ROCP_TOOL_LIBRARIES" in os.environ
    if under_profiler and not args.no_graph:
        log(f"[bench] {label}: rocprofv3 detected: hipGraph replay switched off (eager launches of the same schedule)")
    use_graph = not args.no_graph and not under_profiler
    pipeline = not args.no_pipeline
    run = dict(learning=True, use_graph=use_graph, pipeline=pipeline)
    # Untimed setup: bring the model to the learned state the metric is quoted on (warm-up of
    # at least 10 passes over the pattern bank; predictions appear after four).  The W warm-up steps of the
    # command line come on top of it, before every timed repetition.  (In calls of one pass: the library sees the pool's
    # size between calls and switches the scan to its streaming form when the pool has outgrown the resident one.)
    pretrain = args.pretrain if args.pretrain >= 0 else 10 * w["patterns"]
    for a in range(0, pretrain, w["patterns"]):
        eng.run(bank, n_bank, min(w["patterns"], pretrain - a), **run)
        eng.sync()
    eng.check_capacity()
    log(f"[bench] {label}: setup {time.perf_counter() - t_setup:.1f}s incl. {pretrain} untimed pre-training steps; "
        f"S={eng.info().segments} segments")
    # R repetitions of [W warm-up steps, exactly K timed steps]; `value` is the median repetition
    # The calls are made the way a caller streaming its input in chunks makes them (HTM_RUN_CONTINUE, include/bithtm_hip.h):
    # the Spatial Pooler's look-ahead is kept across the call boundaries, so a timed region of K steps holds exactly K
    # Temporal Memory steps and K Spatial Pooler steps of the steady state, not a cold start plus a drain.
    stream = dict(run, continuing=pipeline and args.warmup > 0)
    plan = eng.run_plan(args.steps, **stream)          # what the timed call does: asked of the library, not assumed
    rates = []
    for r in range(reps):
        eng.run(bank, n_bank, args.warmup, **stream)
        eng.prepare(bank, n_bank, args.steps, **stream)  # every graph the timed call replays exists before the clock starts
        eng.sync()
        t0 = time.perf_counter()
        eng.run(bank, n_bank, args.steps, **stream)
        eng.sync()
        rates.append(args.steps / (time.perf_counter() - t0))
    eng.run(bank, n_bank, 4, **run)                      # untimed: ends the stream (no Spatial Pooler work left outstanding)
    eng.sync()
    info = eng.check_capacity()
    steps_per_s = float(np.median(rates))
    log(f"[bench] {label}: {reps} x {args.steps} timed steps ({'hipGraph replay' if plan['hip_graph'] else 'eager launches'}"
        f"{', large-pool scan' if plan['scan_large'] else ''}): median {steps_per_s:.0f} timesteps/s "
        f"(min {min(rates):.0f}, max {max(rates):.0f}); S={info.segments}; "
        f"select fallbacks so far: {info.select_fallbacks} of {info.step_index} steps, crowded bins cut to a sub-bin: {info.select_zoom_steps}")

    # per-launch device time (HIP events on the engine's stream) of the TIMED schedule, then of the same
    # workload with one role per launch; both eager (events cannot be read out of a graph replay)
    prof_steps = max(min(args.steps, 300), 50)
    eng.profile(True)
    eng.run(bank, n_bank, prof_steps, learning=True, use_graph=False, pipeline=pipeline)
    prof_timed = eng.profile_read()
    eng.run(bank, n_bank, prof_steps, learning=True, use_graph=False, pipeline=False)
    prof_roles = eng.profile_read()
    eng.profile(False)
    info = eng.check_capacity()
    store = eng.read_store()
    rb = role_bytes(w, k, store["seg_nsyn"], info.work_items, info.matching_segments)
    role_us = {n: 1e3 * ms / cnt for n, (ms, cnt) in prof_roles.items() if cnt}
    log(f"[bench] {label}: one role per launch, average launch (us): " +
        ", ".join(f"{n}={v:.1f}" for n, v in sorted(role_us.items(), key=lambda kv: -kv[1])))
    lean, four, two = launches_seen(prof_timed, LEAN_KERNEL), launches_seen(prof_timed, LAUNCH_KERNEL), launches_seen(prof_timed, LEAN2_KERNEL)
    if pipeline and (lean or four or two):
        ran, roles_of, kernel_of = (two, LEAN2_ROLES, LEAN2_KERNEL) if two else (lean, LEAN_ROLES, LEAN_KERNEL) if lean else (four, LAUNCH_ROLES, LAUNCH_KERNEL)
        launch_us = {n: 1e3 * ms / cnt for n, (ms, cnt) in ran.items()}
        launch_bytes = {n: sum(rb[r] for r in roles_of[n]) for n in ran}
    else:                                             # --no-pipeline (or a grid too large to pipeline): the roles ARE the launches
        per_role = {"sp_overlap": ("sp_overlap",), "sp_select": ("sp_select",), "sp_emit": ("sp_emit", "tm_activate", "tm_clear"),
                    "tm_mid": ("tm_mid", "sp_rows"), "tm_learn": ("tm_learn",), "tm_scan": ("tm_scan",), "tm_scan_large": ("tm_scan",)}
        launch_us = {n: v for n, v in role_us.items() if n in per_role}
        launch_bytes = {n: sum(rb[r] for r in per_role[n]) for n in launch_us}
        kernel_of = {"sp_overlap": "k_sp_overlap", "sp_select": "k_sel_pass", "sp_emit": "k_sp_emit", "tm_mid": "k_mid_rows",
                     "tm_learn": "k_tm_learn", "tm_scan": "k_tm_scan", "tm_scan_large": "k_tm_scan"}
    log(f"[bench] {label}: launches of the timed schedule (us): " + ", ".join(f"{n}={v:.1f}" for n, v in launch_us.items()) +
        f"; sum {sum(launch_us.values()):.1f} of {1e6 / steps_per_s:.1f} us per step")
    dominant = max(launch_us, key=lambda n: launch_us[n])
    achieved = launch_bytes[dominant] / (launch_us[dominant] * 1e-6) / 1e9
    whole = int(sum(launch_bytes.values()))
    roofline = dict(bound="hbm", kernel=f"{kernel_of[dominant]} ({dominant})", achieved=round(achieved, 1), peak=HBM_PEAK_GBS,
                    unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4), traffic=None,
                    bytes_per_launch=int(launch_bytes[dominant]), avg_launch_us=round(launch_us[dominant], 2),
                    # which clock `avg_launch_us` is: the begin / end timestamps hipExtLaunchKernelGGL stamps on the engine's
                    # stream, eager launches of the timed schedule, measured in this run (rocprofv3's per-dispatch duration of
                    # the same kernel, from a run of its own, is in profiles/: about a microsecond above)
                    clock="hip_events",
                    whole_step_bytes=whole, whole_step_frac=round(whole * steps_per_s / 1e9 / HBM_PEAK_GBS, 4),
                    launches={n: dict(kernel=kernel_of[n], us=round(launch_us[n], 2), bytes=int(launch_bytes[n]),
                                      frac=round(launch_bytes[n] / (launch_us[n] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4))
                              for n in launch_us})
    sum_nsyn = int(store["seg_nsyn"].astype(np.int64).sum())
    variant = ", 4, " if "tm_scan_large+sp_emit" in dominant else ", 6, " if "tm_scan+sp_emit" in dominant else None
    roofline.update(recorded_traffic(kernel_of[dominant], sum_nsyn=sum_nsyn, section=pmc_section, variant=variant))
    roofline["state"] = dict(step_index=int(info.step_index), segments=int(info.segments), sum_nsyn=sum_nsyn)
    try:                                               # rocprofv3's duration of the same kernel, where a pass of this command was kept
        rec = json.load(open(os.path.join(ROOT, "profiles", "r04_pipelined_kernel_us.json")))
        rec = rec[pmc_section] if pmc_section else rec
        key = next(k2 for k2 in rec if kernel_of[dominant] in k2 and (variant is None or variant in k2) and "mean_last300_us" in rec[k2])
        roofline["rocprofv3_us"] = round(rec[key]["mean_last300_us"], 2)
        roofline["rocprofv3_source"] = "profiles/r04_pipelined_kernel_us.json (mean of the last 300 launches, a run of its own)"
    except Exception:
        pass

    cpu = None
    if cpu_steps > 0:
        cpu = cpu_baseline(w, htm, noisy, bank, int(eng.info().step_index), cpu_steps, run, chunk_plan)
        log(f"[bench] {label}: cpu baseline: {cpu['value']:.2f} timesteps/s; parity {cpu.get('parity')}")
    return dict(steps_per_s=steps_per_s, rates=rates, roofline=roofline, cpu=cpu, role_us=role_us, role_bytes=rb, k=k,
                pretrain=pretrain, segments=int(info.segments), launches_per_step=len({kernel_of[n] for n in launch_us}),
                plan=plan, streamed=bool(stream["continuing"]), pipeline=pipeline)


def large_pool_leg(args, device):
    """The headline shape with a larger learned pool (LARGE_POOL: 350 patterns instead of 50, S ~ 0.65 M segments): where
    SURVEY section 8(d) says the two halves of the target -- >= 10^4 timesteps/s and >= 40 % of the HBM roofline -- can
    coincide on 65 536 x 32.  Same protocol as the headline (untimed pre-training of 10 passes, R x [W, K], the oracle's
    parity check of the call pattern, per-launch HIP events, recorded PMC traffic)."""
    w = dict(LARGE_POOL)
    m = measure(w, args, device, "large_pool", reps=max(3, min(7, -(-2000 // max(args.steps, 1)))),
                cpu_steps=0 if args.no_cpu_baseline else args.large_cpu_steps, chunk_plan=(5,), pmc_section="large_pool")
    rf = m["roofline"]
    out = dict(workload=f"65536 columns x 32 cells, SP + TM learning on, {w['patterns']} patterns (headline: {WORKLOAD['patterns']}): "
                        f"a learned pool of {m['segments']} segments",
               value=round(m["steps_per_s"], 1), unit="timesteps/s", ms_per_step=round(1e3 / m["steps_per_s"], 5),
               steps=args.steps, warmup=args.warmup, repetitions=[round(r, 1) for r in m["rates"]],
               patterns=w["patterns"], pretrain_steps=m["pretrain"], segments=m["segments"], hip_graph=m["plan"]["hip_graph"],
               scan_form="large-pool (streaming)" if m["plan"]["scan_large"] else "small-pool (resident)",
               launches_per_step=m["launches_per_step"], whole_step_frac=rf["whole_step_frac"], roofline=rf)
    if m["cpu"] is not None:
        out.update(parity=m["cpu"].get("parity"), parity_checked=m["cpu"].get("parity_checked"),
                   cpu_baseline={k2: m["cpu"][k2] for k2 in ("value", "unit", "cores", "kind", "sample")})
    return out


def host_fed_leg(w, device, steps=3000):
    """The loop the reference's example.py runs, unchanged but for the import: one `htm.process(x)` per timestep with a host
    input (bit-packing, one ctypes call, 128 bytes host -> device in the launch's arguments, the State objects of the
    reference's API; nothing read back), on the headline shape in its learned state.  A PCIe-inclusive rate: reported beside
    the line, never its `value` (the boundary's device-resident rate is)."""
    noisy, perm = make_inputs(w)
    htm = build_htm(w, perm, device)
    eng = htm.engine
    bank = eng.upload_bank(noisy)
    eng.run(bank, noisy.shape[0], 10 * w["patterns"], learning=True)        # untimed: the learned state, as the headline's
    for t in range(200):                                                      # warm-up of the host-fed path itself
        htm.process(noisy[t % len(noisy)])
    eng.sync()
    t0 = time.perf_counter()
    for t in range(steps):
        htm.process(noisy[t % len(noisy)])
    eng.sync()
    dt = time.perf_counter() - t0
    info = eng.check_capacity()
    return dict(value=round(steps / dt, 1), unit="timesteps/s", us_per_step=round(1e6 * dt / steps, 2), steps=steps,
                what="htm.process(x) per timestep from host memory (example.py's loop): three launches per step, no read-back",
                segments=int(info.segments))


def run_single(args):
    w = dict(WORKLOAD)
    if args.columns:
        w["column_dim"] = args.columns
    device = int(os.environ.get("LOCAL_RANK", "0"))
    import gc
    if args.large_pool_only:
        return large_pool_leg(args, device)
    reps = args.reps if args.reps > 0 else max(3, min(15, -(-4000 // max(args.steps, 1))))
    m = measure(w, args, device, "headline", reps, 0 if args.no_cpu_baseline else args.cpu_steps, (5, 20, 5))
    gc.collect()                                        # (the legs need the device to themselves: memory, and CU slots -- DESIGN.md)
    large = None
    if not args.no_large_pool and not args.columns:
        try:
            large = large_pool_leg(args, device)
        except Exception as e:                          # a report beside the headline: never a reason to lose the line
            log(f"[bench] large_pool leg failed: {e!r}")
            large = dict(error=repr(e))
        gc.collect()
    host_fed = None
    if not args.no_host_fed and not args.columns:
        try:
            host_fed = host_fed_leg(w, device)
            log(f"[bench] host-fed loop (htm.process per step): {host_fed['value']:.0f} timesteps/s")
        except Exception as e:
            log(f"[bench] host-fed leg failed: {e!r}")
            host_fed = dict(error=repr(e))
        gc.collect()
    stress = None
    if not args.no_stress and not args.columns:
        try:
            stress = stress_leg(check=not args.no_cpu_baseline)
        except Exception as e:
            log(f"[bench] configs[4] leg failed: {e!r}")
            stress = dict(error=repr(e))

    steps_per_s, k = m["steps_per_s"], m["k"]
    return dict(
        metric="HTM timesteps/sec (SP + TM, learning on), 65536 cols x 32 cells", value=round(steps_per_s, 1),
        unit="timesteps/s", n_gpus=1, steps=args.steps, warmup=args.warmup, ms_per_step=round(1e3 / steps_per_s, 5),
        higher_is_better=True, scaling="strong", vs_baseline=None, dtype="u32 bit-packed / f64 + f32 permanences",
        data="synthetic",
        config=dict(workload="configs[2]: 65536 columns x 32 cells, SP + TM learning on, 1 MI355X",
                    input_dim=w["input_dim"], column_dim=w["column_dim"], cell_dim=w["cell_dim"], active_columns=k,
                    patterns=w["patterns"], input_density=w["density"], flip_noise=w["noise"],
                    pretrain_steps=m["pretrain"], segments=m["segments"], segment_slots=w["segment_slots"],
                    # what the TIMED call did (htm_run_plan): calls of fewer than 64 steps launch eagerly whatever was asked
                    hip_graph=m["plan"]["hip_graph"], hip_graph_requested=not args.no_graph, pipelined=m["plan"]["pipelined"],
                    launches_per_step=m["launches_per_step"], repetitions=reps, streamed_calls=m["streamed"]),
        repetitions=[round(r, 1) for r in m["rates"]],
        roofline=m["roofline"], cpu_baseline=m["cpu"], large_pool=large, host_fed=host_fed, stress=stress,
        role_us_one_per_launch={n: round(v, 2) for n, v in m["role_us"].items()},
        role_bytes={n: int(v) for n, v in m["role_bytes"].items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--columns", type=int, default=0, help="override column_dim (debugging)")
    ap.add_argument("--pretrain", type=int, default=-1, help="untimed pre-training steps (default: 10 passes over the pattern bank)")
    ap.add_argument("--reps", type=int, default=0, help="timed repetitions (default: enough for about 4000 timed steps, 3..15)")
    ap.add_argument("--cpu-steps", type=int, default=100, help="timesteps of the NumPy oracle timed for cpu_baseline (and checked against the GPU)")
    ap.add_argument("--large-cpu-steps", type=int, default=69, help="... of the large_pool leg's parity check")
    ap.add_argument("--no-large-pool", action="store_true", help="skip the large_pool leg (the headline shape with 350 patterns)")
    ap.add_argument("--large-pool-only", action="store_true", help="run only the large_pool leg and print its object (profiling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stress", action="store_true", help="skip the configs[4] leg (one rank's pre-populated shard)")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the host-fed leg (example.py's loop: htm.process(x) per timestep)")
    ap.add_argument("--stress-only", action="store_true", help="run only the configs[4] leg and print its object (profiling)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="one role per launch (what the profiled replay always does)")
    args = ap.parse_args()
    if args.stress_only:
        print(json.dumps(stress_leg(check=not args.no_cpu_baseline)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started by hand without the launcher: start one process per GPU as a CHILD (this process has
        # not touched the GPU) and pass its exit code on
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if args.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1:
        from bench_sharded import run_sharded
        out = run_sharded(args)
    else:
        out = run_single(args)
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
