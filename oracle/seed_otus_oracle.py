"""TEST INFRASTRUCTURE ONLY (oracle) -- parity unpinned.

Literal restatement of rambl.py stage 3, /root/reference/scripts/find_seed_otus.py:155-438 (remove_null_nodes :155-188,
gene_tree_cluster :213-286, find_seed_otus :309-438), together with the behaviour it takes from its libraries:

  ete2 Tree        Newick format 0; node defaults name "NoName", dist 1.0; `traverse('postorder')` works on a stack
                   that snapshots a node's children when the node is first met; `delete()` re-hangs the children on
                   the parent, drops the node's own branch length, and deletes a parent left with one child
                   (prevent_nondicotomic); `get_distance(name1, name2)` finds both nodes by a search from the ROOT and
                   adds the branch lengths second node first; `get_descendants()` is level order.
  scipy hierarchy  `linkage(dist)` on the square, upper-triangular matrix the script builds takes its ROWS as
                   observations (Euclidean distances between rows, single linkage); `fcluster(.., 'distance')`.
  numpy            `np.resize` repeats the shorter mask cyclically.
  CPython 2.7      the seed genes are printed in the key order of a dict of str (64-bit build, no hash randomisation).

The reference needs Python 2 with ete2 / skbio and cannot be imported here; it holds no fixture for this stage, so
nothing pins this restatement to outputs of the reference.  Only tests/ import it; the product (rambl_amd/stage3.py)
is written separately (index-based, linear-time distances) and is compared with it on random trees.
"""
import csv
import itertools

import numpy as np
from scipy.cluster import hierarchy


class Node(object):
    def __init__(self):
        self.name = "NoName"
        self.dist = 1.0
        self.support = 1.0
        self.children = []
        self.up = None

    # ---- ete2 TreeNode behaviour used by the script
    def is_leaf(self):
        return len(self.children) == 0

    def get_children(self):
        return self.children

    def add_child(self, ch):
        ch.up = self
        self.children.append(ch)

    def remove_child(self, ch):
        self.children.remove(ch)
        ch.up = None

    def delete(self, prevent_nondicotomic=True):
        parent = self.up
        if parent:
            for ch in list(self.children):
                parent.add_child(ch)
            parent.remove_child(self)
        if prevent_nondicotomic and parent and len(parent.children) < 2:
            parent.delete(prevent_nondicotomic=False)

    def traverse(self, strategy):
        if strategy == "postorder":
            to_visit = [self]
            while to_visit:
                node = to_visit.pop(-1)
                if isinstance(node, list):
                    yield node[1]
                elif not node.is_leaf():
                    to_visit.extend(reversed(node.children + [[1, node]]))
                else:
                    yield node
        elif strategy == "preorder":
            to_visit = [self]
            while to_visit:
                node = to_visit.pop(0)
                yield node
                to_visit = list(node.children) + to_visit
        else:                                   # levelorder
            to_visit = [self]
            while to_visit:
                node = to_visit.pop(0)
                yield node
                to_visit.extend(node.children)

    def iter_leaves(self):
        for n in self.traverse("preorder"):
            if n.is_leaf():
                yield n

    def get_leaves(self):
        return list(self.iter_leaves())

    def get_descendants(self):
        return [n for n in self.traverse("levelorder") if n is not self]

    def get_tree_root(self):
        r = self
        while r.up is not None:
            r = r.up
        return r

    def get_distance(self, name1, name2):
        root = self                          # with both targets given, ete searches below the node it is called on
        found = {}
        for n in root.traverse("levelorder"):
            if n.name in (name1, name2):
                if n.name in found:
                    raise ValueError("Ambiguous node name: " + n.name)
                found[n.name] = n
        t1, t2 = found[name1], found[name2]
        anc = set()
        x = t1
        while x is not None:
            anc.add(id(x))
            x = x.up
        a = t2
        while id(a) not in anc:
            a = a.up
        dist = 0.0
        for n in (t2, t1):
            cur = n
            while cur is not a:
                dist += cur.dist
                cur = cur.up
        return dist


def parse_newick(text):
    """Newick format 0 (names on leaves, supports on internal nodes, branch lengths), quotes stripped."""
    text = text.strip()
    if not text.endswith(";"):
        raise ValueError("newick must end with ;")
    pos = [0]

    def label(node, leaf):
        s = pos[0]
        while pos[0] < len(text) and text[pos[0]] not in ",();":
            pos[0] += 1
        tok = text[s:pos[0]]
        nm, _, ds = tok.partition(":")
        nm = nm.strip().strip("'\"")
        if nm:
            if leaf:
                node.name = nm
            else:
                try:
                    node.support = float(nm)
                except ValueError:
                    node.name = nm
        if ds.strip():
            node.dist = float(ds)

    def sub():
        node = Node()
        if text[pos[0]] == "(":
            pos[0] += 1
            while True:
                node.add_child(sub())
                if text[pos[0]] == ",":
                    pos[0] += 1
                    continue
                if text[pos[0]] == ")":
                    pos[0] += 1
                    break
                raise ValueError("bad newick at %d" % pos[0])
            label(node, False)
        else:
            label(node, True)
        return node

    root = sub()
    return root


def py27_str_hash(s):
    """CPython 2.7 string_hash, 64-bit, PYTHONHASHSEED unset."""
    if not s:
        return 0
    b = s.encode("latin-1") if not isinstance(s, bytes) else s
    x = (b[0] << 7) & 0xFFFFFFFFFFFFFFFF
    for c in b:
        x = ((1000003 * x) ^ c) & 0xFFFFFFFFFFFFFFFF
    x ^= len(b)
    if x == 0xFFFFFFFFFFFFFFFF:
        x = 0xFFFFFFFFFFFFFFFE
    return x                                    # as an unsigned 64-bit pattern


def py27_dict_key_order(keys):
    """Order in which a CPython 2.7 dict holding `keys` (inserted in this order, none deleted) iterates."""
    mask, fill = 7, 0
    table = [None] * 8

    def insert(tab, m, key, h):
        i = h & m
        perturb = h
        while tab[i & m] is not None:
            if tab[i & m][0] == key:
                return False
            i = ((i << 2) + i + perturb + 1) & 0xFFFFFFFFFFFFFFFF
            perturb >>= 5
        tab[i & m] = (key, h)
        return True

    for k in keys:
        h = py27_str_hash(k)
        if insert(table, mask, k, h):
            fill += 1
            if fill * 3 >= (mask + 1) * 2:
                minused = (2 if fill > 50000 else 4) * fill
                newsize = 8
                while newsize <= minused:
                    newsize <<= 1
                new = [None] * newsize
                for e in table:
                    if e is not None:
                        insert(new, newsize - 1, e[0], e[1])
                table, mask = new, newsize - 1
    return [e[0] for e in table if e is not None]


def find_seed_otus(tree_file, abun_file, mask_file, index_file, sim_thres=0.9, depth_thres=10.0, gene_cover_thres=0.6,
                   depth_ratio=None, taxonomy_file=None):
    """-> output lines of the script, in its print order."""
    tree = parse_newick(open(tree_file).read())
    for count, node in enumerate(tree.traverse("postorder")):
        node.id = count
    gene_abun, gene_cover = {}, {}
    with open(abun_file) as f:
        for row in csv.reader(f, delimiter="\t"):
            gene_abun[row[0]] = float(row[3])
            gene_cover[row[0]] = float(row[4])
    gene_mask = {}
    with open(index_file) as f:
        for line in f:
            fields = line.split()
            gene_mask.setdefault(fields[0], np.zeros(int(fields[1])))
    with open(mask_file) as f:
        for line in f:
            fields = line.split()
            gene_mask[fields[0]][int(fields[1]) - 1:int(fields[2])] = 1      # KeyError where the defaultdict(array) of the script fails too
    gene_tax = {}
    if taxonomy_file is not None:
        for line in open(taxonomy_file):
            g, tax = line.rstrip().split("\t")
            gene_tax[g] = tax
    for node in tree.iter_leaves():
        node.gene_set = [(node.name, gene_abun.get(node.name, 0))]
    # remove_null_nodes
    for t in tree.iter_leaves():
        t.prunable = not (t.gene_set[0][1] > 0)
    for t in tree.traverse("postorder"):
        if t.is_leaf():
            continue
        prunable = True
        for c in t.get_children():
            prunable &= c.prunable
        t.prunable = prunable
    for t in tree.traverse("postorder"):
        if t.prunable:
            t.delete()
    # gene_tree_cluster, bottom up
    dissim = 1. - sim_thres
    for t in tree.traverse("postorder"):
        if t.is_leaf():
            t.centroid = list(t.gene_set)
            t.merged = 1
            continue
        children = t.get_children()
        if any(ch.merged == 0 for ch in children):
            t.merged = 0
            continue
        sizes = [len(ch.centroid) for ch in children]
        if any(s > 1 for s in sizes):
            t.centroid = [c for ch in children for c in ch.centroid]
            continue
        n = int(np.sum(sizes))
        dist = np.zeros((n, n))
        for c1, c2 in itertools.combinations(range(len(children)), 2):
            d1, d2 = int(np.sum(sizes[:c1])), int(np.sum(sizes[:c2]))
            dist[d1][d2] = t.get_distance(str(children[c1].centroid[0][0]), str(children[c2].centroid[0][0]))
        Z = hierarchy.linkage(dist)
        C = hierarchy.fcluster(Z, t=dissim, criterion="distance") - 1
        centroid = [None] * len(np.unique(C))
        for c in range(len(children)):
            d = int(np.sum(sizes[:c]))
            for i in range(sizes[c]):
                k = C[d + i]
                if centroid[k] is None:
                    centroid[k] = tuple(children[c].centroid[i])
                else:
                    g, a = centroid[k]
                    if a < children[c].centroid[i][1]:
                        g = children[c].centroid[i][0]
                    centroid[k] = (g, a + children[c].centroid[i][1])
        t.centroid = centroid
        t.merged = 1 if len(np.unique(C)) == 1 else 0
    total = 0
    for node in tree.iter_leaves():
        total += gene_abun.get(node.name, 0.0)
    abun_thres = max([depth_thres, 0.0001 * total])
    if depth_ratio is not None:
        abun_thres = depth_ratio * total
    seed, order = {}, []
    visited = set()
    for t in tree.traverse("preorder"):
        if t.id in visited:
            continue
        visited.add(t.id)
        if t.merged == 1 and t.centroid[0][1] >= abun_thres:
            gs = []
            gm = np.zeros(1)
            if t.name in gene_cover:
                gs.append((gene_cover[t.name], t.name))
            if t.name in gene_mask:
                gm = gene_mask[t.name]
            for nd in t.get_descendants():
                if nd.name in gene_cover:
                    gs.append((gene_cover[nd.name], nd.name))
                if nd.name in gene_mask:
                    gm_l = max([gm.shape[0], gene_mask[nd.name].shape[0]])
                    gm = np.resize(gm, gm_l) + np.resize(gene_mask[nd.name], gm_l)
                visited.add(nd.id)
            gs = sorted(gs, reverse=True)
            cover_frac = np.sum(gm > 0) / (len(gm) + 0.)
            if cover_frac >= gene_cover_thres:
                g = gs[0][1]
                if g not in seed:
                    order.append(g)
                    seed[g] = "%s\t%f\t%f\t%f\t%f\t%d\t%s" % (g, t.centroid[0][1], cover_frac, gene_abun.get(g, 0.0), gene_cover.get(g, 0.0),
                                                              len(t.get_leaves()), gene_tax[t.centroid[0][0]] if taxonomy_file is not None else "")
    return [seed[g] for g in py27_dict_key_order(order)]
