"""CPU, world_size 2 and 4 over gloo: the column-sharded timestep (oracle/sharded.py, the protocol
the multi-GPU HIP path implements) equals the unsharded one bit for bit on every rank."""

import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, cfg):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sharded_worker.py"), cfg], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=600)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out[-3000:]}"
    assert "OK world=" in outs[0]


@pytest.mark.parametrize("cfg", ["default", "stress"])
def test_two_ranks(cfg):
    _run(2, cfg)


def test_four_ranks_stress():
    _run(4, "stress")


def test_wire_format_round_trip():
    import numpy as np
    from types import SimpleNamespace
    from oracle.sharded import pack_record, unpack_record, record_nbytes, cand_cap, DEAD_CAP
    rng = np.random.RandomState(0)
    kl, K = 41, 12
    rec = SimpleNamespace(boosted=rng.rand(kl), col=np.sort(rng.choice(5000, kl, replace=False)).astype(np.int64),
                          bursting=rng.rand(kl) < 0.5, win=rng.rand(kl, K) < 0.2, unacc=rng.rand(kl, K) < 0.1,
                          dead=np.array([5, 77, 1234567], dtype=np.int64), hot_slot=None, hot_floor=0.7)
    for cap in (kl, kl + 23):                      # a full record, and one with free candidate slots
        for hot in (None, np.flatnonzero(rec.boosted > 0.7), np.arange(kl)):       # no hot list, some hot candidates, all
            rec.hot_slot = hot
            buf = pack_record(rec, K, cap)
            assert len(buf) == record_nbytes(cap) and len(buf) % 16 == 0
            back = unpack_record(buf, cap, K)
            for f in ("boosted", "col", "bursting", "win", "unacc", "dead"):
                assert np.array_equal(getattr(back, f), getattr(rec, f)), f
            assert (back.hot_slot is None) == (hot is None) and (hot is None or np.array_equal(back.hot_slot, hot))
            assert back.hot_floor == 0.7
    with pytest.raises(OverflowError):
        pack_record(rec, K, kl - 1)
    rec.dead = np.arange(DEAD_CAP + 1)
    with pytest.raises(OverflowError):
        pack_record(rec, K)
    # 20 B per candidate; BASELINE.json configs[3] (65 536 columns, k = 1 311) sharded 8-way
    # (+ a quarter more slots than the rank must offer: room for the threshold bin of its local select)
    # and 10 B for its place in the hot list
    assert cand_cap(1311, 65536 // 8) == 1638 and record_nbytes(1638) <= 50 * 1024
    assert cand_cap(1311, 65536 // 2) == 1966       # (below 8 shards: half as many again)
