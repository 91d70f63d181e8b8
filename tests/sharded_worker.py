"""One rank of the world_size-N CPU rehearsal of the column-sharded timestep (gloo).
Launched by tests/test_sharded_gloo.py; not collected by pytest.

Every rank runs (a) its shard of the sharded oracle, exchanging the wire-format records with
torch.distributed.all_gather, and (b) the unsharded oracle, and asserts after every step that its
shard of every result is bit-identical."""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from oracle import HTMOracle, SPParams, TMParams, canonical_synapses  # noqa: E402
from oracle.sharded import ShardedHTMOracle, pack_record, unpack_record, record_nbytes  # noqa: E402


def main():
    cfg = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if cfg == "default":
        I, C, K, P, density, noise, steps, jump, spp, tmp, seed = 200, 1024, 8, 40, 0.1, 0.02, 260, 0.0, None, None, 41
    else:   # non-default parameters, random jumps: punishments, pruning, dead segments, recycling
        I, C, K, P, density, noise, steps, jump, seed = 256, 1024, 4, 40, 0.12, 0.02, 320, 0.15, 13
        spp = SPParams(permanence_mean=0.01, permanence_std=0.08, permanence_threshold=0.02, permanence_increment=0.05,
                       permanence_decrement=0.02, boost_intensity=0.5, boost_momentum=0.95)
        tmp = TMParams(permanence_initial=0.3, permanence_threshold=0.45, permanence_increment=0.12,
                       permanence_decrement=0.14, permanence_punishment=0.2, segment_activation_threshold=12,
                       segment_matching_threshold=9, segment_sampling_synapses=20)
    np.random.seed(seed)
    std, mean = (spp.permanence_std, spp.permanence_mean) if spp else (0.1, 0.0)
    perm = np.random.randn(C, I) * std + mean
    full = HTMOracle(I, C, K, seed=seed, sp_params=spp, tm_params=tmp, permanence=perm)
    # "stress": odd ranks hand over whole threshold bins (variable candidate counts, as the HIP engine does when they fit) with
    # their hot lists, even ranks exactly their top-min(k, own columns) and no hot list: the global result must not care.
    # "default": all ranks the former -- the global top-k is then settled among the hot lists wherever they hold k keys
    part = ShardedHTMOracle(rank, world, I, C, K, seed=seed, sp_params=spp, tm_params=tmp, permanence=perm,
                            offer="bin" if (rank % 2 or cfg == "default") else "exact")
    c0, c1 = part.c0, part.c1
    cap = part.cap
    nbytes = record_nbytes(cap)
    offered = 0

    def all_gather(rec):
        send = torch.from_numpy(pack_record(rec, K, cap))
        recv = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(recv, send)
        return [unpack_record(r.numpy(), cap, K) for r in recv]

    rng = np.random.RandomState(seed + 1)
    bank = rng.rand(P, I) < density
    recycled_total = dead_total = 0
    thr = full.temporal_memory.params.segment_matching_threshold
    for t in range(steps):
        idx = int(rng.randint(P)) if (jump > 0 and rng.rand() < jump) else t % P
        x = bank[idx] ^ (rng.rand(I) < noise)
        learning = (t % 23) != 5
        f_sp, f_tm = full.step(x, learning=learning)
        out, rec = part.step(x, all_gather, learning=learning)
        dead_total += len(part.dead_out)
        offered += len(rec.col) - part.n_cand

        def eq(name, a, b):
            assert np.array_equal(np.asarray(a), np.asarray(b)), f"rank {rank} step {t}: {name}"
        eq("active_column", out.active_column, f_sp.active_column)
        eq("boosted (own)", rec.boosted_all.view(np.int64), f_sp.boosted_overlaps[c0:c1].view(np.int64))
        eq("overlaps (own)", rec.overlaps, f_sp.overlaps[c0:c1])
        eq("winner cells", out.winner_flat, f_tm.winner_cell[0] * K + f_tm.winner_cell[1])
        eq("activation", out.cell_activation, f_tm.cell_activation)
        eq("bursting", out.bursting, f_tm.active_column_bursting[:, 0])
        eq("prediction (own)", out.cell_prediction[c0:c1], f_tm.cell_prediction[c0:c1])
        ftm, ptm = full.temporal_memory, part.tm
        assert ptm.S == ftm.S, f"rank {rank} step {t}: S {ptm.S} vs {ftm.S}"
        S = ftm.S
        eq("seg_cell", ptm.seg_cell[:S], ftm.seg_cell[:S])
        owned = part.owns_cell(ftm.seg_cell[:S])
        eq("nsyn (own)", ptm.seg_nsyn[:S][owned], ftm.seg_nsyn[:S][owned])
        # deaths of this step travel with the NEXT exchange (the owner applies its own then, too): until then
        # those ids still read alive; every other "< threshold" flag must already agree, on every rank
        pending = [None] * world
        dist.all_gather_object(pending, part.dead_out.tolist())
        settled = np.ones(S, dtype=np.bool_)
        for ids in pending:
            settled[np.asarray(ids, dtype=np.int64)] = False
        eq("dead flags (settled)", part.dead[:S][settled], (ftm.seg_nsyn[:S] < thr)[settled])
        fd, pd = f_tm.distal_state, out.distal_state
        mine = owned[fd.matching_segment]
        eq("matching (own)", pd.matching_segment, fd.matching_segment[mine])
        eq("matching active (own)", pd.matching_segment_active, fd.matching_segment_active[mine])
        eq("potential (own)", pd.segment_potential[owned], fd.segment_potential[owned])
        cells = slice(c0 * K, c1 * K)
        eq("cell max (own)", pd.max_jittered_potential[cells].view(np.int32), fd.max_jittered_potential[cells].view(np.int32))
        eq("segcount (own)", ptm.segcount[cells], ftm.segcount[cells])
        if t % 20 == 0 or t == steps - 1:
            ids = np.flatnonzero(owned)
            a = canonical_synapses(ptm.seg_cell[ids], ptm.presyn[ids], ptm.perm[ids])
            b = canonical_synapses(ftm.seg_cell[ids], ftm.presyn[ids], ftm.perm[ids])
            for (ca, ia, pa), (cb, ib, pb) in zip(a, b):
                assert ca == cb and np.array_equal(ia, ib) and np.array_equal(pa.view(np.int32), pb.view(np.int32)), \
                    f"rank {rank} step {t}: synapses of an owned segment"
            eq("SP permanence (own)", part.permanence.view(np.int64), full.spatial_pooler.permanence[c0:c1].view(np.int64))
            eq("duty (own)", part.duty.view(np.int32), full.spatial_pooler.duty_cycle[c0:c1].view(np.int32))
    # the run must have exercised the cross-rank parts of the protocol
    tot = torch.tensor([dead_total, int((f_tm.cell_prediction.any(axis=1)).sum()), offered, part.hot_selects])
    dist.all_reduce(tot)
    if rank == 0:
        print(f"OK world={world} cfg={cfg} steps={steps} S={full.temporal_memory.S} dead_reported={int(tot[0])} "
              f"candidates_beyond_the_cut={int(tot[2])}")
        if cfg != "default":
            assert int(tot[0]) > 0, "stress run reported no dead segments: protocol path untested"
        assert int(tot[2]) > 0, "no rank ever offered more than its exact cut: variable counts untested"
        if cfg == "default":
            assert int(tot[3]) > world * steps // 2, "the hot lists hardly ever settled the global top-k: that path is untested"
        print(f"global top-k settled among the hot lists in {int(tot[3]) // world} of {steps} steps")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
