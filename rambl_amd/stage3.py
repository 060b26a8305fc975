"""Stage 3 of rambl.py: seed genes from the phylogeny and the abundance profile.

Mirror of /root/reference/scripts/find_seed_otus.py (`find_seed_otus.py [-T tax] -s SIM -c COVER -d DEPTH TREE ABUNDANCE
MASK GENE_INDEX GENE_ALIGN`, launched by rambl.py:122-143; the alignment file is loaded by the reference and never
used, :190-211 is dead): leaves without abundance are pruned from the tree (:155-188), clusters of sequence-similar
genes are propagated bottom-up -- the children of a node merge when their representative genes lie within 1 - SIM of
each other in tree distance (:213-286) -- and every merged clade that is abundant and covered enough reports its
best-covered gene as a seed (:373-420).  The seeds are the regions stage 5 shards over the GPUs.

This is tree bookkeeping on at most ~200 000 leaves: it runs on the host, in linear time.  The only numeric kernel
the reference has here, a tree distance per pair of sibling representatives, needs no search (the common ancestor of
two genes under different children of a node is that node) and is O(depth) per pair, so nothing in this stage pays for a
trip to the device.  What the reference takes from its libraries is kept as it is: ete2's deletion of nodes without
branch-length preservation and its collapse of single-child parents, scipy's `linkage` applied to the ROWS of the
upper-triangular distance matrix (single linkage, flat clusters at 1 - SIM), numpy's cyclic `resize` of coverage masks,
`%f` formatting, and the key order of a CPython 2.7 dict for the printed list.  Parity is unpinned (the reference
needs Python 2 with ete2 and holds no fixture); tests compare this module with a literal restatement kept with the
test infrastructure, on random trees.
"""
import csv
import math
import sys

import numpy as np


class _Tree:
    """Index-based rooted tree: parent, ordered children, branch length, name per node (node 0 = root)."""

    def __init__(self):
        self.parent, self.children, self.dist, self.name = [], [], [], []

    def new(self, parent):
        self.parent.append(parent)
        self.children.append([])
        self.dist.append(1.0)               # ete2 TreeNode defaults
        self.name.append("NoName")
        k = len(self.parent) - 1
        if parent >= 0:
            self.children[parent].append(k)
        return k

    def postorder(self, root=0):
        """ete2's stack walk: a node's children are snapshotted when the node is first met."""
        stack = [root]
        while stack:
            x = stack.pop()
            if x < 0:
                yield ~x
            elif self.children[x]:
                stack.append(~x)
                stack.extend(reversed(self.children[x]))
            else:
                yield x

    def preorder(self, root=0):
        stack = [root]
        while stack:
            x = stack.pop()
            yield x
            stack.extend(reversed(self.children[x]))

    def levelorder_below(self, root):
        out, k = list(self.children[root]), 0
        while k < len(out):
            out.extend(self.children[out[k]])
            k += 1
        return out

    def delete(self, x, collapse=True):
        """ete2 TreeNode.delete(prevent_nondicotomic=True, preserve_branch_length=False)."""
        p = self.parent[x]
        if p >= 0:
            for ch in self.children[x]:
                self.parent[ch] = p
                self.children[p].append(ch)
            self.children[p].remove(x)
            self.parent[x] = -1
            if collapse and len(self.children[p]) < 2:
                self.delete(p, collapse=False)


def parse_newick(text):
    """Newick format 0: names on the leaves, support values (or names) on internal nodes, `:length` anywhere."""
    text = text.strip()
    if not text.endswith(";"):
        raise ValueError("newick text must end with ';'")
    t = _Tree()
    cur = t.new(-1)
    i, n = 0, len(text)
    fresh = True                              # the current node has not received its label yet
    while i < n:
        ch = text[i]
        if ch == "(":
            cur = t.new(cur)
            fresh = True
            i += 1
        elif ch == ",":
            cur = t.new(t.parent[cur])
            fresh = True
            i += 1
        elif ch == ")":
            cur = t.parent[cur]
            fresh = True
            i += 1
        elif ch == ";":
            break
        else:
            j = i
            while j < n and text[j] not in ",();":
                j += 1
            tok = text[i:j]
            nm, _, ds = tok.partition(":")
            nm = nm.strip().strip("'\"")
            if nm and fresh:
                if t.children[cur]:
                    try:
                        float(nm)             # a support value
                    except ValueError:
                        t.name[cur] = nm
                else:
                    t.name[cur] = nm
            if ds.strip():
                t.dist[cur] = float(ds)
            fresh = False
            i = j
    return t


def _py27_hash(s):
    b = s.encode("latin-1")
    if not b:
        return 0
    x = b[0] << 7
    for c in b:
        x = ((1000003 * x) ^ c) & 0xFFFFFFFFFFFFFFFF
    x ^= len(b)
    return 0xFFFFFFFFFFFFFFFE if x == 0xFFFFFFFFFFFFFFFF else x


class Py27StrDict:
    """Key order of a CPython 2.7 dict of str keys (64-bit, no hash randomisation, no deletions): open addressing with
    the perturbed probe sequence of dictobject.c, growth by 4x (2x above 50 000 keys) at two thirds full."""

    def __init__(self):
        self.slots = [None] * 8

    def _put(self, slots, key, h):
        mask = len(slots) - 1
        i, perturb = h & mask, h
        while slots[i & mask] is not None:
            if slots[i & mask][0] == key:
                return False
            i = (5 * i + perturb + 1) & 0xFFFFFFFFFFFFFFFF
            perturb >>= 5
        slots[i & mask] = (key, h)
        return True

    def add(self, key):
        if not self._put(self.slots, key, _py27_hash(key)):
            return
        used = sum(1 for e in self.slots if e is not None)
        if used * 3 >= len(self.slots) * 2:
            size = 8
            while size <= (2 if used > 50000 else 4) * used:
                size <<= 1
            new = [None] * size
            for e in self.slots:
                if e is not None:
                    self._put(new, e[0], e[1])
            self.slots = new

    def keys(self):
        return [e[0] for e in self.slots if e is not None]


def find_seed_otus(tree_file, abun_file, mask_file, index_file, align_file=None, sim_thres=0.9, depth_thres=10.0, gene_cover=0.6,
                   depth_ratio=None, taxonomy=None):
    """-> the lines find_seed_otus.py prints: gene, clade abundance, clade coverage, seed abundance, seed coverage,
    leaves of the clade, taxonomy."""
    t = parse_newick(open(tree_file).read())
    abun, cover = {}, {}
    with open(abun_file) as f:
        for row in csv.reader(f, delimiter="\t"):
            abun[row[0]] = float(row[3])            # :39-55
            cover[row[0]] = float(row[4])           # :57-75
    size = {}
    with open(index_file) as f:
        for line in f:
            fld = line.split()
            size.setdefault(fld[0], int(fld[1]))
    mask = {g: np.zeros(n) for g, n in size.items()}                        # :78-103
    with open(mask_file) as f:
        for line in f:
            fld = line.split()
            mask[fld[0]][int(fld[1]) - 1:int(fld[2])] = 1
    tax = {}
    if taxonomy is not None:
        for line in open(taxonomy):
            g, tx = line.rstrip().split("\t")
            tax[g] = tx
    n_nodes = len(t.parent)
    node_id = [0] * n_nodes                                                  # :147-151
    for count, x in enumerate(t.postorder()):
        node_id[x] = count

    # ---- remove_null_nodes, :155-188
    prunable = [False] * n_nodes
    order = list(t.postorder())
    was_leaf = [not c for c in t.children]
    for x in order:
        if not t.children[x]:
            prunable[x] = not (abun.get(t.name[x], 0) > 0)
        else:
            prunable[x] = all(prunable[c] for c in t.children[x])
    for x in order:                       # the walk was fixed before the first deletion, as ete2's stack fixes it
        if prunable[x]:
            t.delete(x)

    # ---- gene_tree_cluster bottom-up, :213-286
    dissim = 1. - sim_thres
    merged = [0] * n_nodes
    cen_gene = [None] * n_nodes           # representative gene and summed abundance of a merged clade
    cen_abun = [0.0] * n_nodes
    leaf_of = {}
    for x in t.preorder():
        if not t.children[x]:
            leaf_of[t.name[x]] = x

    def up_to(d, x, anc):                 # ete2 adds the branch lengths of both walks into one running sum
        while x != anc:
            d += t.dist[x]
            x = t.parent[x]
        return d

    for x in t.postorder():
        ch = t.children[x]
        if not ch:
            if not was_leaf[x]:
                # every gene below the root was pruned: the reference dies here (an inner node without `gene_set`)
                raise ValueError("no gene of the tree has any abundance")
            cen_gene[x], cen_abun[x], merged[x] = t.name[x], abun.get(t.name[x], 0), 1
            continue
        if any(merged[c] == 0 for c in ch):
            continue
        k = len(ch)
        if k < 2:
            raise ValueError("a clade with a single child reaches the clustering: scipy's linkage refuses one observation")
        # rows of the script's upper-triangular matrix: row i holds d(i, j) for j > i; the common ancestor of two genes
        # under different children of x is x, so the tree distance is two walks up to x (second gene first, as ete2 adds)
        rows = [[0.0] * k for _ in range(k)]
        for a in range(k):
            for b in range(a + 1, k):
                rows[a][b] = up_to(up_to(0.0, leaf_of[cen_gene[ch[b]]], x), leaf_of[cen_gene[ch[a]]], x)
        # single linkage cut at `dissim` = connected components of the Euclidean row distances <= dissim
        comp = list(range(k))

        def find(i):
            while comp[i] != i:
                comp[i] = comp[comp[i]]
                i = comp[i]
            return i

        for a in range(k):
            for b in range(a + 1, k):
                s = 0.0
                for c in range(k):
                    d = rows[a][c] - rows[b][c]
                    s += d * d
                if math.sqrt(s) <= dissim:
                    comp[find(a)] = find(b)
        roots = {find(i) for i in range(k)}
        if len(roots) == 1:
            g, a = cen_gene[ch[0]], cen_abun[ch[0]]
            for c in ch[1:]:
                if a < cen_abun[c]:           # against the sum so far, as the script compares
                    g = cen_gene[c]
                a = a + cen_abun[c]
            cen_gene[x], cen_abun[x], merged[x] = g, a, 1

    # ---- thresholds and report, :356-420
    total = 0
    for x in t.preorder():
        if not t.children[x]:
            total += abun.get(t.name[x], 0.0)
    abun_thres = max([depth_thres, 0.0001 * total])
    if depth_ratio is not None:
        abun_thres = depth_ratio * total
    seeds, lines = Py27StrDict(), {}
    visited = set()
    for x in t.preorder():
        if node_id[x] in visited:
            continue
        visited.add(node_id[x])
        if merged[x] != 1 or not (cen_abun[x] >= abun_thres):
            continue
        gs = []
        gm = np.zeros(1)
        if t.name[x] in cover:
            gs.append((cover[t.name[x]], t.name[x]))
        if t.name[x] in mask:
            gm = mask[t.name[x]]
        below = t.levelorder_below(x)
        for y in below:
            nm = t.name[y]
            if nm in cover:
                gs.append((cover[nm], nm))
            if nm in mask:
                ln = max(gm.shape[0], mask[nm].shape[0])
                gm = np.resize(gm, ln) + np.resize(mask[nm], ln)
            visited.add(node_id[y])
        gs.sort(reverse=True)
        frac = np.sum(gm > 0) / (len(gm) + 0.)
        if frac >= gene_cover:
            g = gs[0][1]
            if g not in lines:
                n_leaves = sum(1 for y in below if not t.children[y]) if t.children[x] else 1
                lines[g] = "%s\t%f\t%f\t%f\t%f\t%d\t%s" % (g, cen_abun[x], frac, abun.get(g, 0.0), cover.get(g, 0.0), n_leaves,
                                                           tax[cen_gene[x]] if taxonomy is not None else "")
                seeds.add(g)
    return [lines[g] for g in seeds.keys()]


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="Find seed OTUs based on the phylogeny tree and abundance profile")
    ap.add_argument("tree")
    ap.add_argument("abun")
    ap.add_argument("mask")
    ap.add_argument("seq")
    ap.add_argument("align")
    ap.add_argument("-s", dest="sim_thres", type=float, default=0.9)
    ap.add_argument("-d", dest="depth_thres", type=float, default=10)
    ap.add_argument("-c", dest="gene_cover", type=float, default=0.6)
    ap.add_argument("-r", dest="depth_ratio", type=float, default=None)
    ap.add_argument("-T", dest="taxonomy", default=None)
    ap.add_argument("-v", dest="verbose", action="store_true")
    a = ap.parse_args(argv)
    for line in find_seed_otus(a.tree, a.abun, a.mask, a.seq, a.align, a.sim_thres, a.depth_thres, a.gene_cover, a.depth_ratio, a.taxonomy):
        sys.stdout.write(line + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
