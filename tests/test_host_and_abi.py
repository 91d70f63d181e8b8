"""CPU: host-side logic and the C-ABI surface (no compute calls, no GPU needed)."""

import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from bithtm_amd import _lib
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "bithtm_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:const\s+)?[A-Za-z_][A-Za-z0-9_\s\*]*?\b(htm_[a-z_]+)\s*\(", header, flags=re.M))
    declared -= {"htm_handle", "htm_status", "htm_config", "htm_info", "htm_field"}
    assert len(declared) >= 18
    from bithtm_amd import _lib
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.htm_abi_version() == _lib.ABI_VERSION == 3


def test_create_rejects_bad_config_without_touching_a_gpu(lib):
    import ctypes as C
    from bithtm_amd import _lib
    cfg = _lib.HtmConfig()
    h = C.c_void_p()
    assert lib.htm_create(C.byref(cfg), C.byref(h)) == -1          # struct_bytes mismatch
    assert b"size mismatch" in lib.htm_last_error(None)
    cfg.struct_bytes = C.sizeof(_lib.HtmConfig)
    cfg.column_dim, cfg.active_columns, cfg.enable_tm, cfg.cell_dim = 64, 4, 1, 33
    cfg.segment_slots, cfg.segment_capacity, cfg.segment_sampling_synapses = 128, 16, 32
    assert lib.htm_create(C.byref(cfg), C.byref(h)) == -1
    assert b"cell_dim" in lib.htm_last_error(None)


def test_pack_and_unpack_bits():
    from bithtm_amd.engine import pack_bits, words_to_bool, bool_to_words
    rng = np.random.RandomState(0)
    x = rng.rand(1000) < 0.3
    w = pack_bits(x, 32)
    assert w.dtype == np.uint32 and w.shape == (32,)
    for i in (0, 1, 31, 32, 33, 999):
        assert bool((w[i >> 5] >> (i & 31)) & 1) == bool(x[i])
    assert w[31] >> 8 == 0                      # bits beyond input_dim are zero
    m = rng.rand(50, 13) < 0.5
    assert np.array_equal(words_to_bool(bool_to_words(m), 13), m)


def test_config_scalars_equal_the_oracles_derivation():
    """The binding forms the implicit scalars with the reference's own Python expressions."""
    import inspect
    from bithtm_amd import engine
    from oracle.htm_oracle import SPParams, TMParams, sp_derived, tm_derived
    src = inspect.getsource(engine.Engine.__init__)
    for needle in ("1.0 * (inc + dec) - dec", "0.0 * (inc + dec) - dec", "np.float32(-(boosting.intensity / density))",
                   "np.float32(1.0 - boosting.momentum)", "1.0 * (a - b) + b", "0.0 * (a - b) + b"):
        assert needle in src
    d = sp_derived(SPParams(), 2048, 41)
    assert d.delta_on == 1.0 * (0.03 + 0.015) - 0.015 and d.delta_off == -0.015
    t = tm_derived(TMParams())
    assert t.learn_active == 0.1 and t.learn_inactive == -0.1 and t.punish_active == -0.01 and t.punish_inactive == 0.0


def test_plugin_arguments_are_type_checked():
    import bithtm_amd as B

    class Foreign:
        pass
    with pytest.raises(TypeError):
        B.SpatialPooler(10, 64, 2, boosting=Foreign())
    with pytest.raises(TypeError):
        B.TemporalMemory(64, 4, distal_projection=Foreign())
    sp = B.SpatialPooler(10, 64, 2, proximal_projection=B.DenseProjection(10, 64, permanence_threshold=0.1))
    assert sp.proximal_projection.permanence.shape == (64, 10)
    assert B.SpatialPooler.compute is B.SpatialPooler.process
    assert B.HierarchicalTemporalMemory.compute is B.HierarchicalTemporalMemory.process


def test_dense_projection_draws_like_the_reference():
    """projections.py:16: same expression => same matrix under the same seed."""
    import bithtm_amd as B
    np.random.seed(5)
    want = np.random.randn(32, 20) * 0.1 + 0.0
    np.random.seed(5)
    got = B.DenseProjection(20, 32).permanence
    assert np.array_equal(want, got)


def test_device_twins_match_the_oracle_sources():
    """The constants of the device exp / RNG headers are the oracle's."""
    from oracle import fexp
    h = open(os.path.join(ROOT, "bithtm_amd", "csrc", "htm_fexp.h")).read()
    for c in (fexp.LOG2E, fexp.LN2_HI, fexp.LN2_LO) + fexp.TAYLOR:
        assert c.hex().replace("0x1.", "0x1.") in h, c.hex()
    r = open(os.path.join(ROOT, "bithtm_amd", "csrc", "htm_rng.h")).read()
    for c in ("0x7FEB352Du", "0x846CA68Bu", "0x9E3779B9u"):
        assert c in r
