"""Independent pure-Python re-derivation of pcl::octree::OctreePointCloud (PCL 1.10) that builds an
EXPLICIT pointer tree exactly the way PCL does (re-rooting on bounding-box growth, createLeafRecursive
by depth mask, depth-first leaf iteration over children 0..7).  The oracle (oracle/dmsa_oracle.cpp)
never builds a tree — it carries keys + integer shifts — so agreement between the two is a real check
of the restated semantics, not a tautology.  Small clouds only."""
import math

import numpy as np

EPS = float(np.finfo(np.float32).eps)


class Branch:
    __slots__ = ("child",)

    def __init__(self):
        self.child = [None] * 8


class Leaf:
    __slots__ = ("idx",)

    def __init__(self):
        self.idx = []


class PclOctree:
    def __init__(self, resolution: float):
        self.res = float(resolution)
        self.defined = False
        self.mn = [0.0, 0.0, 0.0]
        self.mx = [0.0, 0.0, 0.0]
        self.depth = 0
        self.root = Branch()
        self.events = 0

    def _adopt(self, p):
        while True:
            lo = [float(p[a]) < self.mn[a] for a in range(3)]
            hi = [float(p[a]) >= self.mx[a] for a in range(3)]
            if not (any(lo) or any(hi) or not self.defined):
                return
            if self.defined:
                child_idx = ((not hi[0]) << 2) | ((not hi[1]) << 1) | (not hi[2])
                new_root = Branch()
                new_root.child[child_idx] = self.root
                self.root = new_root
                side = float(1 << self.depth) * self.res
                for a in range(3):
                    if not hi[a]:
                        self.mn[a] -= side
                self.depth += 1
                side = float(1 << self.depth) * self.res - EPS
                for a in range(3):
                    self.mx[a] = self.mn[a] + side
                self.events += 1
            else:
                for a in range(3):
                    self.mn[a] = float(p[a]) - self.res / 2
                    self.mx[a] = float(p[a]) + self.res / 2
                mk = [int(math.ceil((self.mx[a] - self.mn[a] - EPS) / self.res)) for a in range(3)]
                mv = max(max(mk), 2)
                self.depth = max(min(32, int(math.ceil(math.log(mv) / math.log(2.0) - EPS))), 0)
                side = float(1 << self.depth) * self.res
                for a in range(3):
                    over = (side - (self.mx[a] - self.mn[a])) / 2.0
                    if over > EPS:
                        self.mn[a] -= over
                        self.mx[a] += over
                self.defined = True

    def add(self, i, p):
        if not all(math.isfinite(float(v)) for v in p[:3]):
            return
        self._adopt(p)
        key = [int((float(p[a]) - self.mn[a]) / self.res) for a in range(3)]
        node = self.root
        mask = 1 << (self.depth - 1)
        while True:
            c = (4 if key[0] & mask else 0) | (2 if key[1] & mask else 0) | (1 if key[2] & mask else 0)
            if mask > 1:
                if node.child[c] is None:
                    node.child[c] = Branch()
                node = node.child[c]
                mask >>= 1
            else:
                if node.child[c] is None:
                    node.child[c] = Leaf()
                node.child[c].idx.append(i)
                return

    def leaves_depth_first(self):
        out = []

        def rec(node, depth_left):
            for c in range(8):
                ch = node.child[c]
                if ch is None:
                    continue
                if isinstance(ch, Leaf):
                    out.append(list(ch.idx))
                else:
                    rec(ch, depth_left - 1)

        if self.defined:
            rec(self.root, self.depth)
        return out


def pcl_leaves(xyz, resolution):
    t = PclOctree(resolution)
    for i in range(xyz.shape[0]):
        t.add(i, xyz[i])
    return t, t.leaves_depth_first()
