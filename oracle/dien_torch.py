"""DIEN scorer on torch-CPU (float32 by default, all host cores) - the NN half of bench.py's ``cpu_baseline`` "vectorised"
leg; with ``dtype=torch.float64`` the multi-threaded fp64 checker of the full-size GPU parity tests (the numpy oracle is
single-threaded: hundreds of envs per check need the host's cores).
ORACLE - test infrastructure only (see oracle/__init__.py).

Same topology and cell equations as ``oracle/dien.py`` (rl4rs/nets/dien.py:8-45, rl4rs/nets/utils.py:16-25,48-54,100-129;
PARITY UNPINNED for the same reason), written with torch ops so that the recurrences run on the host's BLAS threads the
way the reference's TF-CPU session would: one [R, 3E] x [.., 2N] matmul pair per step, no per-row python.  Checked against
the numpy oracle in tests/test_oracle_dien.py."""
import numpy as np
import torch


class TorchDien(object):
    def __init__(self, weights, config, workers=1, dtype=torch.float32):
        self.config = config
        self.workers = workers            # row-parallel python threads (bench.py: one per host core)
        self.dtype = dtype
        self.np_dtype = np.float64 if dtype == torch.float64 else np.float32
        self.w = dict((k, torch.from_numpy(np.ascontiguousarray(v, dtype=self.np_dtype))) for k, v in weights.items())
        self.seq_num = config['seq_num']

    def _cell_loop(self, X, Wg, bg, Wc, bc, N, att=None, keep=False):
        R, L, E = X.shape
        h = torch.zeros((R, N), dtype=self.dtype)
        out = torch.empty((R, L, N), dtype=self.dtype) if keep else None
        # the input half of both matmuls for all steps at once ([x, h] W = x W[:E] + h W[E:])
        Xg = (X.reshape(R * L, E) @ Wg[:E]).reshape(R, L, -1) + bg
        Xc = (X.reshape(R * L, E) @ Wc[:E]).reshape(R, L, -1) + bc
        Wgh, Wch = Wg[E:], Wc[E:]
        for t in range(L):
            g = torch.sigmoid(Xg[:, t] + h @ Wgh)
            r, u = g[:, :N], g[:, N:]
            c = torch.tanh(Xc[:, t] + (r * h) @ Wch)
            if att is not None:
                u = (1.0 - att[:, t:t + 1]) * u
            h = u * h + (1.0 - u) * c
            if keep:
                out[:, t] = h
        return out if keep else h

    def features(self, seq, dense, cat):
        w = self.w
        seq = torch.as_tensor(np.asarray(seq).astype(np.int64))
        cat = torch.as_tensor(np.asarray(cat).astype(np.int64))
        dense = torch.as_tensor(np.asarray(dense, dtype=self.np_dtype))
        R = cat.shape[0]
        Ec = w['cat_emb'][cat]                                                   # [R, Cn, E]
        att = torch.softmax(Ec @ Ec.transpose(1, 2), dim=-1) @ Ec
        c = torch.cat([att.mean(dim=1), Ec.reshape(R, -1)], dim=1)
        elu = torch.nn.functional.elu
        d = elu(elu(dense @ w['dense_w1'] + w['dense_b1']) @ w['dense_w2'] + w['dense_b2'])
        q = w['seq_emb'][cat[:, -10:]].mean(dim=1)
        finals = []
        E = q.shape[1]
        for i in range(self.seq_num):
            X = w['seq_emb'][seq[:, i, :]]
            H1 = self._cell_loop(X, w['gru%d_gate_w' % i], w['gru%d_gate_b' % i], w['gru%d_cand_w' % i], w['gru%d_cand_b' % i], E,
                                 keep=True)
            # LocalActivationUnit on [q, k, q-k, q*k] without materialising the [R, L, 4E] concat:
            # [q,k,q-k,q*k] W1 = q (W1a + W1c) + k (W1b - W1c) + (q*k) W1d
            W1 = w['att%d_w1' % i]
            Wa, Wb, Wc_, Wd = W1[:E], W1[E:2 * E], W1[2 * E:3 * E], W1[3 * E:]
            pre = (q @ (Wa + Wc_))[:, None, :] + H1 @ (Wb - Wc_) + (q[:, None, :] * H1) @ Wd + w['att%d_b1' % i]
            h1 = torch.sigmoid(pre)
            h2 = torch.sigmoid(h1 @ w['att%d_w2' % i] + w['att%d_b2' % i])
            s = (h2 @ w['att%d_w3' % i] + w['att%d_b3' % i])[..., 0]
            N = w['augru%d_cand_w' % i].shape[1]
            finals.append(self._cell_loop(H1, w['augru%d_gate_w' % i], w['augru%d_gate_b' % i], w['augru%d_cand_w' % i],
                                          w['augru%d_cand_b' % i], N, att=s))
        return torch.cat(finals + [d, c], dim=1)

    def _map_rows(self, fn, seq, dense, cat, workers):
        """Rows are independent: split them over ``workers`` python threads, each running single-threaded torch ops (they
        release the GIL).  The 64-step recurrences are chains of small matmuls that one intra-op thread pool cannot spread
        over a 128-core host (measured on the bench box, R = 4096: 234 GFLOP/s at 16 threads, 63 GFLOP/s at 128); row-parallel
        workers scale."""
        n = len(cat)
        workers = max(1, min(int(workers), (n + 31) // 32))
        step = (n + workers - 1) // workers
        chunks = [(seq[lo:lo + step], dense[lo:lo + step], cat[lo:lo + step]) for lo in range(0, n, step)]
        if len(chunks) == 1:
            return fn(*chunks[0])
        from concurrent.futures import ThreadPoolExecutor
        nt = torch.get_num_threads()
        torch.set_num_threads(1)
        try:
            with ThreadPoolExecutor(len(chunks)) as ex:
                outs = list(ex.map(lambda c: fn(*c), chunks))
        finally:
            torch.set_num_threads(nt)
        return np.concatenate(outs, axis=0)

    def _obs1(self, seq, dense, cat):
        with torch.no_grad():
            return torch.nn.functional.elu(self.features(seq, dense, cat) @ self.w['obs_w'] + self.w['obs_b']).numpy()

    def _prob1(self, seq, dense, cat):
        with torch.no_grad():
            o = torch.nn.functional.elu(self.features(seq, dense, cat) @ self.w['obs_w'] + self.w['obs_b'])
            return torch.softmax(o @ self.w['out_w'] + self.w['out_b'], dim=1)[:, 1].numpy().astype(self.np_dtype)

    def obs(self, seq, dense, cat):
        return self._map_rows(self._obs1, seq, dense, cat, self.workers)

    def prob(self, seq, dense, cat):
        return self._map_rows(self._prob1, seq, dense, cat, self.workers)
