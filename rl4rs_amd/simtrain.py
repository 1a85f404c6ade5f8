"""Supervised training of the simulator from the logs, on the device (all four simulator families - dien, dnn, widedeep, lstm).

The reference builds its supervised set in ``script/data_preprocess.py:91-131``: one sample per (page record, slot j) with
``category = user_cat(10) + [sequence_id] + exposed_items(9) + [item_j]``, ``dense = user_dense(32) + item_feature(9 x 40)
+ item_feature_j`` and ``label = user_feedback[j]`` - exactly the complete-state rows the env scores at the reward step
(``SlateState.get_complete_states``, slate.py:117-131) when the logged slate is replayed.  So a training batch is: sample
records (``RecDataBase``), replay ``offline_action`` through the device state machine, take the complete-state rows and the
logged feedback.  The model, loss and optimiser are those of ``script/supervised_train.py:37-42`` with
any ``model_type`` (``DeviceSimTrainer`` / ``rl4rs_simtrain_*`` for dnn / widedeep / lstm, ``DeviceDienTrainer`` /
``rl4rs_dientrain_*`` for dien).
"""
import numpy as np
import torch

from . import device as D
from collections import OrderedDict


class SimulatorTrainer(object):
    def __init__(self, sim, weights=None, minibatch=256, seed=0, lr=1e-3, dropout_rate=0.2):
        """sim: a ``SlateRecEnv`` (its config names the log / catalogue files and the model sizes)."""
        cfg = sim.config
        self.algo = cfg.get('algo', 'dien')
        if self.algo != 'dien' and self.algo not in D.SIMTRAIN_ORDER:
            raise NotImplementedError("config['algo'] must be dien, dnn, widedeep or lstm (got %r)" % (self.algo,))
        self.sim = sim
        self.minibatch, self.seed, self.lr, self.dropout_rate = int(minibatch), int(seed), lr, dropout_rate
        if weights is None:
            weights = sim.model.weights
        if self.algo == 'dien':
            self.trainer = D.DeviceDienTrainer(cfg, weights, max_batch=self.minibatch)
        else:
            self.trainer = D.DeviceSimTrainer(cfg, weights, max_batch=self.minibatch)
        self.P = int(cfg.get('page_items', 9))

    def dataset_from_logs(self):
        """One cache window of the log -> (dense [B*P, Dn] f32, cat [B*P, Cn] i32, labels [B*P] i32, seqs) on the device;
        seqs = the seq_num sequence inputs [B*P, maxlen] i32 (user history, then the constant [0] sequence of page 1,
        data_preprocess.py:103-105)."""
        data = self.sim._recData
        data.reset()
        samples = data.sample(self.sim.batch_size)
        env = samples._live()
        assert not samples.is_seq, "simulator training replays one page per record (SlateState)"
        for _ in range(self.P):
            env.act_discrete(env.offline_action())
        env.build_complete()
        dense = env.snapshot(D.BUF_C_DENSE)
        cat = env.snapshot(D.BUF_C_CATEGORY)
        labels = samples._feedback[:, :self.P].reshape(-1).to(torch.int32).contiguous()
        hist = env.snapshot(D.BUF_SEQ0).repeat_interleave(self.P, dim=0).contiguous()
        seqs = [hist] + [torch.zeros_like(hist) for _ in range(int(self.sim.config['seq_num']) - 1)]
        return dense, cat, labels, seqs

    def fit(self, windows=1, epochs=1):
        """``epochs`` shuffled passes of minibatch SGD (Adam) over each of ``windows`` cache windows; returns the losses."""
        losses = []
        rs = np.random.RandomState(self.seed)
        for _ in range(windows):
            dense, cat, labels, seqs = self.dataset_from_logs()
            n = dense.shape[0]
            for _ in range(epochs):
                perm = torch.from_numpy(rs.permutation(n)).to(dense.device)
                d, c, y = dense[perm], cat[perm], labels[perm]
                q = [x[perm] for x in seqs] if self.algo != 'dnn' else None
                for lo in range(0, n - self.minibatch + 1, self.minibatch):
                    hi = lo + self.minibatch
                    losses.append(self.trainer.step(d[lo:hi], c[lo:hi], y[lo:hi], None if q is None else [x[lo:hi] for x in q],
                                                    lr=self.lr, dropout_rate=self.dropout_rate, seed=self.seed))
        return [float(x.item()) for x in losses]

    def fit_tfrecord(self, filenames, steps, seed=None):
        """``model.fit(featureutil.read_tfrecord(files), steps_per_epoch=steps)`` of script/supervised_train.py:34-42: `steps`
        Adam steps on shuffled batches of the reference's TFRecord training set (rl4rs_amd/utils/tfrecord.py reads it)."""
        from .utils.datautil import FeatureUtil
        fu = FeatureUtil(dict(self.sim.config, batch_size=self.minibatch))
        it = fu.read_tfrecord(filenames, is_pred=False, seed=self.seed if seed is None else seed)
        dev = self.trainer.device
        losses = []
        for _ in range(steps):
            (seq, dense, cat, _), target = next(it)
            seqs = None if self.algo == 'dnn' else [torch.from_numpy(np.ascontiguousarray(seq[:, s])).to(dev)
                                                    for s in range(seq.shape[1])]
            losses.append(self.trainer.step(torch.from_numpy(dense).to(dev), torch.from_numpy(cat).to(dev),
                                            torch.from_numpy(target.argmax(axis=1).astype(np.int32)).to(dev), seqs, lr=self.lr,
                                            dropout_rate=self.dropout_rate, seed=self.seed))
        return [float(x.item()) for x in losses]

    def export_weights(self):
        """Trained parameters as numpy arrays (``rl4rs_amd.nets.simnets.simnet_spec`` names) - what ``model_file`` takes."""
        return dict((k, v.cpu().numpy()) for k, v in self.trainer.weights().items())

    def save(self, model_file):
        """``saver.save(sess, model_file)`` (supervised_train.py:44-46): an ``.npz`` of this package's weight names, or -
        any other path - a TF checkpoint prefix under the reference's graph variable names (``utils.tfckpt``)."""
        w = self.export_weights()
        if str(model_file).endswith('.npz'):
            np.savez(model_file, **w)
        else:
            from .utils import tfckpt
            tfckpt.save_simulator_weights(model_file, w, self.sim.config, self.algo)

    def install(self):
        """Put the trained weights into the simulator the env scores with."""
        self.sim.model.weights = OrderedDict((k, np.ascontiguousarray(v, dtype=np.float32))
                                                     for k, v in self.export_weights().items())
        if self.sim.model.device_net is not None:
            self.sim.model.device_net.close()
            self.sim.model.device_net = None
