"""Catalogue tables.  ORACLE — test infrastructure only (see oracle/__init__.py).

Follows ``rl4rs/env/slate.py:28-53`` (get_iteminfo_from_file) and ``:55-65`` (get_mask_from_file).
"""
import numpy as np


class Catalog(object):
    """item_vec[A,D] f64, price[A] f64, action_emb[A,E] f64, location_mask[4,A], special_items."""

    def __init__(self, iteminfo_file, action_size, action_emb_size=32):
        # slate.py:30-31: split on '\n', drop header, split on ' ' (the file has no trailing newline)
        rows = open(iteminfo_file, 'r').read().split('\n')[1:]
        rows = [x.split(' ') for x in rows]
        dim = len(rows[0][1].split(','))
        self.action_size = action_size
        self.item_dim = dim
        # slate.py:32-46: dict keyed by str(id); id '0' = zero vector / price 0
        max_id = max(int(r[0]) for r in rows)
        size = max(action_size, max_id + 1)
        self.item_vec = np.zeros((size, dim), dtype=np.float64)
        self.price = np.zeros((size,), dtype=np.float64)
        self.location = np.zeros((size,), dtype=np.int64)
        self.known = np.zeros((size,), dtype=bool)
        self.known[0] = True
        for (itemid, item_vec, price, location, is_special) in rows:
            i = int(itemid)
            self.item_vec[i] = list(map(float, item_vec.split(',')))
            self.price[i] = float(price)
            self.location[i] = int(location)
            self.known[i] = True
        # slate.py:47-52: action_emb rows 1.. = last E dims of item_vec, L2-normalised, in FILE ORDER
        self.action_emb = np.zeros((action_size, action_emb_size))
        item_vecs = np.array([list(map(float, r[1].split(',')))[-action_emb_size:] for r in rows])
        self.action_emb[1:] = np.einsum('ij,i->ij', item_vecs, 1.0 / np.linalg.norm(item_vecs, axis=1))
        # slate.py:58-64
        self.special_items = [int(r[0]) for r in rows if int(r[4]) == 2]
        self.location_mask = np.zeros((4, action_size), dtype=np.int64)
        self.location_mask[0, 1:40] = 1
        self.location_mask[1, 40:148] = 1
        self.location_mask[2, 148:] = 1
        self.location_mask[3, 0] = 1
        self.is_special = np.zeros((size,), dtype=bool)
        self.is_special[self.special_items] = True
