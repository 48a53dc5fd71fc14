"""CPU suite: the oracle (C restatement) against the reference-produced golden fixtures and the
known-answer hash table of SURVEY.md 8(c).  No GPU, no /root/reference needed."""
import numpy as np
import pytest

import golden_cases as G
from conftest import assert_bit_equal


@pytest.mark.parametrize("path", G.step_files(), ids=lambda p: p.split("/")[-1])
def test_oracle_steps_match_golden(oracle, path):
    G.check_steps(oracle, path)


@pytest.mark.parametrize("path", G.ops_files(), ids=lambda p: p.split("/")[-1])
def test_oracle_ops_match_golden(oracle, path):
    G.check_ops(oracle, path)


@pytest.mark.parametrize("path", G.channel_files(), ids=lambda p: p.split("/")[-1])
def test_oracle_generic_advect_matches_golden(oracle, path):
    G.check_channels(oracle, path)


def test_fixture_inventory():
    assert len(G.step_files()) >= 7 and len(G.ops_files()) >= 6 and len(G.channel_files()) >= 4


def test_lcg_recipe_and_hash_helpers_agree(oracle):
    v, c = G.lcg_fields(13, 9, 4711, 50.0)
    vo, co = oracle.lcg_fields(13, 9, 4711, 50.0)
    assert_bit_equal(v, vo, "lcg v")
    assert_bit_equal(c, co, "lcg colour")
    assert G.fnv1a64(v) == "%016x" % oracle.fnv1a64(v)


@pytest.mark.parametrize("case", list(G.KAT), ids=lambda c: f"{c[0]}x{c[1]}_i{c[2]}")
def test_oracle_known_answer_hashes(oracle, case):
    dim_x, dim_y, iters, vamp, seed = case
    h_v0, h_c0, rows = G.KAT[case]
    h = lambda a: "%016x" % oracle.fnv1a64(a)
    v, c = oracle.lcg_fields(dim_x, dim_y, seed, vamp)
    assert (h(v), h(c)) == (h_v0, h_c0)
    for want in rows:
        v, d, p, c = oracle.step(v, c, np.float32(1 / 30.0), 1.0, iters, np.float32(1.96))
        assert (h(v), h(d), h(p), h(c)) == want
