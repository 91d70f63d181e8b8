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
    from oracle.sharded import pack_record, unpack_record, record_nbytes, DEAD_CAP
    rng = np.random.RandomState(0)
    c_local, K = 96, 12
    rec = SimpleNamespace(boosted=rng.rand(c_local), act=rng.rand(c_local, K) < 0.5, win=rng.rand(c_local, K) < 0.2,
                          unacc=rng.rand(c_local, K) < 0.1, bursting=rng.rand(c_local) < 0.5,
                          dead=np.array([5, 77, 1234567], dtype=np.int64))
    buf = pack_record(rec, K)
    assert len(buf) == record_nbytes(c_local) and len(buf) % 16 == 0
    back = unpack_record(buf, c_local, K)
    for f in ("boosted", "act", "win", "unacc", "bursting", "dead"):
        assert np.array_equal(getattr(back, f), getattr(rec, f)), f
    rec.dead = np.arange(DEAD_CAP + 1)
    with pytest.raises(OverflowError):
        pack_record(rec, K)
