"""CPU: rambl.py stage 3 (seed genes from the phylogeny, find_seed_otus.py) -- the product's linear-time version against
the literal restatement (oracle/seed_otus_oracle.py: ete2-style node objects, scipy linkage / fcluster) on random trees,
a hand-checked case, and the CPython 2.7 facts the printed order rests on.  Parity with the reference is unpinned."""
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import seed_otus_oracle as O  # noqa: E402  (test infrastructure)
from rambl_amd import stage3  # noqa: E402


def test_python27_dict_order_and_hash_known_answers():
    # well-known CPython 2.7 (64-bit) facts: hash('a'), hash('abc'); {'a':1,'b':2,'c':3} prints a, c, b; one/two/three prints three, two, one
    assert stage3._py27_hash("a") == 12416037344 and stage3._py27_hash("abc") == 1453079729188098211
    for keys, order in ((["a", "b", "c"], ["a", "c", "b"]), (["one", "two", "three"], ["three", "two", "one"])):
        d = stage3.Py27StrDict()
        for k in keys:
            d.add(k)
        assert d.keys() == order == O.py27_dict_key_order(keys)
    rng = random.Random(4)
    keys = ["%d" % rng.randint(1, 4000000) for _ in range(3000)]
    d = stage3.Py27StrDict()
    for k in keys:
        d.add(k)
    assert d.keys() == O.py27_dict_key_order(keys) and sorted(d.keys()) == sorted(set(keys))


def _write_case(d, newick, abun, cover, sizes, bed):
    open(os.path.join(d, "t.nwk"), "w").write(newick)
    with open(os.path.join(d, "abun.txt"), "w") as f:
        for g in abun:
            f.write("%s\t1\t%d\t%f\t%f\n" % (g, sizes[g], abun[g], cover[g]))
    with open(os.path.join(d, "genes.fai"), "w") as f:
        for g, n in sizes.items():
            f.write("%s\t%d\t0\t60\t61\n" % (g, n))
    with open(os.path.join(d, "mask.bed"), "w") as f:
        for g, a, b in bed:
            f.write("%s\t%d\t%d\t5.0\n" % (g, a, b))
    return [os.path.join(d, x) for x in ("t.nwk", "abun.txt", "mask.bed", "genes.fai")]


def test_hand_checked_case(tmp_path):
    # ((A:0.02,B:0.03)0.9:0.5,(C:0.01,(D:0.2,E:0.01):0.01):0.5,Z:0.4);  Z has no abundance and is pruned.
    # A-B: 0.05 <= 0.1 -> merge (abundance 30 + 50; representative B: 30 < 50).  D-E: 0.21 -> no merge; so C|(D,E) and the
    # root stay unmerged.  Seeds: the clade {A, B} (coverage of the union of their masks) and the single genes C, D, E
    # where abundant and covered enough.
    nwk = "((A:0.02,B:0.03)0.9:0.5,(C:0.01,(D:0.2,E:0.01):0.01):0.5,Z:0.4);"
    abun = {"A": 30.0, "B": 50.0, "C": 12.0, "D": 9.0, "E": 40.0}
    cover = {"A": 0.9, "B": 0.7, "C": 0.8, "D": 0.9, "E": 0.5}
    sizes = {g: 100 for g in "ABCDEZ"}
    bed = [("A", 1, 60), ("B", 41, 90), ("C", 1, 80), ("D", 1, 90), ("E", 1, 50)]
    files = _write_case(str(tmp_path), nwk, abun, cover, sizes, bed)
    got = stage3.find_seed_otus(*files, sim_thres=0.9, depth_thres=10, gene_cover=0.6)
    rows = {l.split("\t")[0]: l.split("\t") for l in got}
    assert set(rows) == {"A", "C"}                       # D: 9 < 10; E: coverage 0.5 < 0.6; the clade {A,B}: gene A (best covered)
    assert rows["A"][1:6] == ["80.000000", "0.900000", "30.000000", "0.900000", "2"]
    assert rows["C"][1:6] == ["12.000000", "0.800000", "12.000000", "0.800000", "1"]
    assert got == O.find_seed_otus(*files, sim_thres=0.9, depth_thres=10, gene_cover_thres=0.6)


def _random_newick(rng, n_leaves):
    names = ["%d" % (100000 + k) for k in range(n_leaves)]
    rng.shuffle(names)
    nodes = [(nm, True) for nm in names]
    while len(nodes) > 1:
        k = min(len(nodes), rng.choice([2, 2, 2, 2, 3, 4]))
        picked = [nodes.pop(rng.randrange(len(nodes))) for _ in range(k)]
        parts = []
        for text, leaf in picked:
            bl = "" if rng.random() < 0.05 else ":%s" % repr(round(rng.choice([0.001, 0.01, 0.03, 0.06, 0.2]) * rng.uniform(0.5, 1.5), 5))
            parts.append(text + bl)
        sup = "" if rng.random() < 0.3 else repr(round(rng.random(), 3))
        nodes.append(("(" + ",".join(parts) + ")" + sup, False))
    return nodes[0][0] + ";", names


@pytest.mark.parametrize("seed", range(12))
def test_random_trees_match_the_literal_restatement(seed, tmp_path):
    rng = random.Random(500 + seed)
    n = rng.choice([5, 17, 60, 200, 400])
    nwk, names = _random_newick(rng, n)
    abun, cover, sizes, bed = {}, {}, {}, []
    for g in names:
        sizes[g] = rng.choice([900, 1200, 1500, 1501])
        if rng.random() < 0.6:
            abun[g] = rng.choice([0.0, 0.5, 8.0, 25.0, 25.0, 120.0]) * rng.choice([1.0, 1.0, 1.37])
            cover[g] = round(rng.random(), 3)
            for _ in range(rng.randint(0, 3)):
                a = rng.randint(1, sizes[g] - 10)
                bed.append((g, a, min(sizes[g], a + rng.randint(5, 900))))
    files = _write_case(str(tmp_path), nwk, abun, cover, sizes, bed)
    for sim, dth, cov, ratio in ((0.9, 10, 0.6, None), (0.97, 1, 0.2, None), (0.8, 10, 0.0, 0.01)):
        try:
            exp = O.find_seed_otus(*files, sim_thres=sim, depth_thres=dth, gene_cover_thres=cov, depth_ratio=ratio)
        except (ValueError, AttributeError):
            # a clade left with one child reaches scipy's linkage, or nothing is left of the tree: the reference dies there too
            with pytest.raises(ValueError):
                stage3.find_seed_otus(*files, sim_thres=sim, depth_thres=dth, gene_cover=cov, depth_ratio=ratio)
            continue
        assert stage3.find_seed_otus(*files, sim_thres=sim, depth_thres=dth, gene_cover=cov, depth_ratio=ratio) == exp


def test_stage3_scales_linearly(tmp_path):
    """20 000 leaves in seconds (the reference's name search per distance makes it quadratic)."""
    import time
    rng = random.Random(77)
    nwk, names = _random_newick(rng, 20000)
    abun = {g: rng.choice([0.0, 5.0, 30.0]) for g in names}
    cover = {g: 0.7 for g in names}
    sizes = {g: 1500 for g in names}
    files = _write_case(str(tmp_path), nwk, abun, cover, sizes, [(g, 1, 1200) for g in names[::3]])
    t0 = time.time()
    out = stage3.find_seed_otus(*files)
    assert time.time() - t0 < 30 and len(out) > 10
