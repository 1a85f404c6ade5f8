"""Synthetic catalogue / log / simulator-weight generators.

There is no dataset and no DIEN checkpoint in the reference tree (README.md:22-24,89-138 point to
external downloads), so benches and tests run on synthetic inputs written in the reference's own file
formats:

* catalogue  -> ``item_info.csv`` schema parsed by ``rl4rs/env/slate.py:28-65``
  (header, space separated ``item_id item_vec price location special_item``, no trailing newline)
* log        -> ``@``-joined records parsed by ``rl4rs/utils/datautil.py:20-32``
* weights    -> the variable set of ``rl4rs/nets/dien.py:8-45`` / ``rl4rs/nets/utils.py`` (see
  ``rl4rs_amd.nets.dien.DienWeights``)

All generators use ``numpy.random.RandomState`` (legacy, bit-stable across numpy versions) so committed
golden fixtures can be regenerated.
"""
import numpy as np

# hard-coded layer boundaries of rl4rs/env/slate.py:60-64
LAYER_BOUNDS = ((1, 40), (40, 148), (148, 284))


def make_catalog_text(action_size=284, item_dim=40, n_special=113, seed=1234):
    """Return the text of a synthetic ``item_info.csv`` (ids 1..action_size-1)."""
    rs = np.random.RandomState(seed)
    n = action_size - 1
    vecs = np.round(rs.randn(n, item_dim), 4)
    prices = np.round(rs.uniform(7.0, 1500.0, size=n), 1)
    ids = np.arange(1, action_size)
    # special items: a fixed subset of ids >= 62 (the real catalogue has 113, all >= 62)
    cand = ids[ids >= min(62, action_size - 1)]
    n_special = min(n_special, len(cand))
    special = set(rs.choice(cand, size=n_special, replace=False).tolist())
    lines = ["item_id item_vec price location special_item"]
    for k, i in enumerate(ids):
        loc = 1 if i < 40 else (2 if i < 148 else 3)
        vec = ",".join(repr(float(x)) for x in vecs[k])
        lines.append("%d %s %s %d %d" % (i, vec, repr(float(prices[k])), loc, 2 if i in special else 0))
    return "\n".join(lines)


def special_ids_from_text(catalog_text):
    """ids whose ``special_item`` column is 2 (slate.py:58)."""
    out = []
    for line in catalog_text.split('\n')[1:]:
        f = line.split(' ')
        if int(f[4]) == 2:
            out.append(int(f[0]))
    return out


def make_records(n, pages=1, page_items=9, action_size=284, hash_size=100000, seed=1000,
                 illegal_frac=0.05, max_hist=128, special_ids=()):
    """Return ``n`` synthetic log records (list of str).

    exposed_items per page: 3 distinct ids from each location layer with at most ONE special item per
    page (legal by construction, like logged slates); a fraction ``illegal_frac`` of records gets a
    duplicate / wrong-layer / second-special item injected so that ``get_violation`` fires.
    """
    rs = np.random.RandomState(seed)
    special = set(int(x) for x in special_ids)
    out = []
    T = pages * page_items
    per_layer = page_items // 3
    for r in range(n):
        exposed = []
        for _ in range(pages):
            page = []
            for (lo, hi) in LAYER_BOUNDS:
                hi = min(hi, action_size)
                plain = [i for i in range(lo, hi) if i not in special]
                page.extend(rs.choice(plain, size=per_layer, replace=False).tolist())
            if special and rs.rand() < 0.5:          # one special item somewhere it is legal
                pos = int(rs.randint(0, len(page)))
                lo, hi = LAYER_BOUNDS[pos // per_layer]
                cands = [i for i in range(lo, min(hi, action_size)) if i in special]
                if cands:
                    page[pos] = int(rs.choice(cands))
            exposed.extend(page)
        if rs.rand() < illegal_frac:
            k = rs.randint(0, 4)
            pos = rs.randint(1, T)
            if k == 0:      # adjacent duplicate
                exposed[pos] = exposed[pos - 1]
            elif k == 1:    # wrong layer
                exposed[pos] = int(rs.randint(1, action_size))
            elif k == 2:    # distance-2 duplicate
                if pos >= 2:
                    exposed[pos] = exposed[pos - 2]
            elif special:   # two different special items on the first page
                sp = sorted(special)
                exposed[page_items - 1] = sp[-1]
                exposed[page_items - 2] = sp[-2]
        feedback = (rs.rand(T) < 0.3).astype(int).tolist()
        hist_len = int(rs.randint(1, max_hist + 1))
        hist = rs.randint(1, action_size, size=hist_len).tolist()
        ucat = rs.randint(0, hash_size, size=10).tolist()
        udense = np.round(np.abs(rs.randn(32) * 10.0), 4).tolist()
        portrait = ",".join([str(int(x)) for x in ucat] + [repr(float(x)) for x in udense])
        # item_feature is parsed (datautil.py:31) but never used by the env state (slate.py:70-72)
        item_feature = ";".join(",".join("0.0" for _ in range(2)) for _ in range(page_items))
        rec = "@".join([
            str(2992008 + r), str(r + 1), "1",
            ",".join(map(str, exposed)),
            ",".join(map(str, feedback)),
            ",".join(map(str, hist)),
            portrait,
            item_feature,
            "1",
        ])
        out.append(rec)
    return out


def write_text(path, text):
    with open(path, "w") as f:
        f.write(text)


def write_records(path, records):
    with open(path, "w") as f:
        f.write("\n".join(records) + "\n")
