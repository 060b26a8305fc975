"""CPU: the oracle against the golden vectors generated from the reference
(tests/golden/make_golden.py): FASTA, -G graph dump and 17-digit trace, byte
for byte; MSA vectors of the reference's MultipleSequenceAlignmentSP."""
import gzip
import hashlib
import json
import os

import pytest

import sc_testlib as T

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INDEX = json.load(open(os.path.join(GOLD, "index.json")))


def load_case(name, d):
    meta = INDEX[name]
    args = T.make_case(meta["seed"], d)
    assert args[:-2] == meta["argv"]
    # the seeded generator must reproduce the inputs the reference saw
    assert hashlib.sha256(open(args[-2], "rb").read()).hexdigest() == meta["fasta_sha256"]
    assert hashlib.sha256(open(args[-1], "rb").read()).hexdigest() == meta["sam_sha256"]
    cdir = os.path.join(GOLD, name)
    exp_fa = open(os.path.join(cdir, "expected.fa")).read()
    exp_g = gzip.open(os.path.join(cdir, "expected.graph.gz")).read().decode()
    exp_tr = gzip.open(os.path.join(cdir, "expected.trace.gz")).read().decode()
    return args, exp_fa, exp_g, exp_tr


@pytest.mark.parametrize("name", sorted(INDEX))
def test_oracle_matches_reference(name, tmp_path, oracle_bin):
    args, exp_fa, exp_g, exp_tr = load_case(name, str(tmp_path))
    fa, tr = T.run_oracle(args, str(tmp_path), trace=True)
    g, _ = T.run_oracle(args, str(tmp_path), graph=True)
    assert fa == exp_fa
    assert g == exp_g
    assert tr == exp_tr            # long double, 17 significant digits: identical text


@pytest.mark.parametrize("name", sorted(INDEX)[:6])
def test_oracle_counting_edge_support_is_the_same(name, tmp_path, oracle_bin, monkeypatch):
    """SC_ORACLE_FAST_SUPPORT=1 (edge support by counting instead of the reference's double loop over two read pools,
    oracle/o_graph.h) must not change a byte: it exists only so that the oracle can finish the unthinned configs[3]
    case (tests/golden/config4_full_D100000), where the double loop and the reference itself cannot."""
    args, exp_fa, exp_g, exp_tr = load_case(name, str(tmp_path))
    monkeypatch.setenv("SC_ORACLE_FAST_SUPPORT", "1")
    fa, tr = T.run_oracle(args, str(tmp_path), trace=True)
    g, _ = T.run_oracle(args, str(tmp_path), graph=True)
    assert fa == exp_fa and g == exp_g and tr == exp_tr


def test_oracle_msa_vectors(oracle_bin):
    cases = json.loads(gzip.open(os.path.join(GOLD, "msa_vectors.json.gz")).read())
    assert len(cases) >= 100
    for c in cases:
        rows = T.oracle_msa(c["seqs"])
        assert rows == c["rows"], c["seqs"]
        assert len(rows[0]) == c["ncol"]
