"""-m gpu: the HIP path (through the C-ABI) against the oracle on seeded inputs."""
import os
import random

import pytest

import sc_testlib as T

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", list(range(0, 16)))
def test_region_parity(seed, tmp_path, oracle_bin):
    d = str(tmp_path)
    args = T.make_case(seed, d)
    exp_fa, exp_tr = T.run_oracle(args, d, trace=True)
    exp_g, _ = T.run_oracle(args, d, graph=True)
    tf = os.path.join(d, "product.trace")
    got_fa = T.run_product(args, trace_file=tf)
    got_g = T.run_product(args, graph=True)
    assert got_g == exp_g                      # graph topology, levels, labels, read counts: bit-exact
    assert got_fa == exp_fa                    # consensus FASTA: bit-exact
    T.compare_traces(open(tf).read(), exp_tr)  # per-level candidates identical, abundances to 1e-9


def test_msa_kernel_random(oracle_bin):
    from rambl_amd import capi
    rng = random.Random(5)
    with capi.Context(0, 1) as ctx:
        for it in range(60):
            n = rng.randint(2, 40)
            alpha = "ACGT" if it % 5 else "ACGTacgtN-"
            seqs = ["".join(rng.choice(alpha) for _ in range(rng.randint(1, 12))) for _ in range(n)]
            seqs.sort(key=len, reverse=True)
            assert ctx.msa_align(seqs) == T.oracle_msa(seqs), seqs


def test_config2_full_size_matches_reference_fasta(tmp_path):
    """BASELINE.json configs[1] at full size (10 000 x 150 bp reads, 1 500 bp gene):
    the consensus FASTA must equal the reference's own output (tests/golden/
    config2_full, produced by oracle/_ref in the build container), and a second
    run must reproduce it bit for bit."""
    import hashlib
    import json
    from rambl_amd import synth
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config2_full")
    meta = json.load(open(os.path.join(gold, "meta.json")))
    fa, sam, _ = synth.config2(str(tmp_path))
    assert hashlib.sha256(open(fa, "rb").read()).hexdigest() == meta["fasta_sha256"]
    assert hashlib.sha256(open(sam, "rb").read()).hexdigest() == meta["sam_sha256"]
    args = meta["argv"] + [fa, sam]
    got = T.run_product(args)
    assert got == open(os.path.join(gold, "expected.fa")).read()
    assert T.run_product(args) == got
    # domain property: the abundant contigs are the planted strains (generator truth), up to the window ends
    assert got.count(">") >= 3
