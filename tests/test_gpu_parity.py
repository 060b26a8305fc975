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
