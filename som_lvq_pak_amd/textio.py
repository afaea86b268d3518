"""Reader/writer for SOM_PAK/LVQ_PAK text files (.dat / .cod), Python side.

Format (reference datafile.c:112-189 header, :552-748 rows, :396-447 output):
  line 1   "<dim> [topol [xdim ydim neigh]]"
  '#' lines are comments anywhere; a row is <dim> numbers ('x' = masked component)
  followed by any of: labels, "weight=N", "fixed=X,Y".
  Output rows print each value with "%g " and then the labels.

This is the Python mirror used by the tests and bench; the C command-line tools
have their own reader (som_lvq_pak_amd/host/pakio.c).
"""
import numpy as np

TOPOL = {"data": 1, "lvq": 2, "hexa": 3, "rect": 4}
NEIGH = {"bubble": 1, "gaussian": 2}
TOPOL_STR = {v: k for k, v in TOPOL.items()}
NEIGH_STR = {v: k for k, v in NEIGH.items()}


class LabelTable:
    """string <-> int, 1-based in order of first appearance; 0 = no label
    (reference labels.c:75, labels.h:26)."""

    def __init__(self):
        self.names = [""]
        self.index = {}

    def to_index(self, s):
        if s not in self.index:
            self.index[s] = len(self.names)
            self.names.append(s)
        return self.index[s]

    def to_label(self, i):
        return self.names[i]


class Entries:
    def __init__(self):
        self.dim = 0
        self.topol = 0
        self.neigh = 0
        self.xdim = 0
        self.ydim = 0
        self.points = None      # float32 [n, dim]
        self.mask = None        # uint8 [n, dim] or None
        self.labels = None      # list of lists of int
        self.weight = None      # int16 [n]
        self.fixed = None       # int16 [n, 2], -1 = none

    @property
    def first_label(self):
        return np.array([l[0] if l else 0 for l in self.labels], dtype=np.int32)


def read_entries(path, table=None, skip_empty=True):
    table = table if table is not None else LabelTable()
    e = Entries()
    rows, masks, labels, weights, fixed = [], [], [], [], []
    any_mask = False
    header_done = False
    with open(path) as f:
        for line in f:
            s = line.strip()
            if not s or s.startswith("#"):
                continue
            tok = s.split()
            if not header_done:
                e.dim = int(tok[0])
                if len(tok) > 1:
                    e.topol = TOPOL.get(tok[1].lower(), 0)
                if len(tok) >= 5:
                    e.xdim, e.ydim = int(tok[2]), int(tok[3])
                    e.neigh = NEIGH.get(tok[4].lower(), 0)
                header_done = True
                continue
            vals = np.zeros(e.dim, dtype=np.float32)
            m = np.zeros(e.dim, dtype=np.uint8)
            for i in range(e.dim):
                if tok[i] == "x":
                    m[i] = 1
                else:
                    vals[i] = np.float32(float(tok[i]))
            if skip_empty and m.all():
                continue                      # datafile.c:677-686
            lab, w, fx = [], 0, (-1, -1)
            for t in tok[e.dim:]:
                if t.startswith("weight="):
                    w = int(t[7:])
                elif t.startswith("fixed="):
                    a, b = t[6:].split(",")
                    fx = (int(a), int(b))
                else:
                    lab.append(table.to_index(t))
            any_mask |= bool(m.any())
            rows.append(vals); masks.append(m); labels.append(lab)
            weights.append(w); fixed.append(fx)
    e.points = np.stack(rows) if rows else np.zeros((0, e.dim), dtype=np.float32)
    e.mask = np.stack(masks) if any_mask else None
    e.labels = labels
    e.weight = np.array(weights, dtype=np.int16)
    e.fixed = np.array(fixed, dtype=np.int16).reshape(-1, 2)
    return e, table


def fmt_g(v):
    """C's "%g" for a float32 value (datafile.c:431)."""
    return "%g" % float(np.float32(v))


def write_entries(path, e, table=None):
    with open(path, "w") as f:
        hdr = str(e.dim)
        if e.topol > 1:
            hdr += " " + TOPOL_STR[e.topol]
            if e.topol >= 3:
                hdr += " %d %d %s" % (e.xdim, e.ydim, NEIGH_STR.get(e.neigh, "bubble"))
        f.write(hdr + "\n")
        for r in range(e.points.shape[0]):
            # write_entry, datafile.c:420-447: "%g " per value, "%s " per label, newline
            line = ""
            for i in range(e.dim):
                if e.mask is not None and e.mask[r, i]:
                    line += "x "
                else:
                    line += fmt_g(e.points[r, i]) + " "
            if e.labels is not None and table is not None:
                for l in e.labels[r]:
                    if l:
                        line += table.to_label(l) + " "
            f.write(line + "\n")
