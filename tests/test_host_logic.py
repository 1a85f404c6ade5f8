"""CPU: host-side logic of the product (parsers, sampler, FeatureUtil list path, API surface) against the
golden fixtures / the oracle.  No GPU, no compute calls into the library."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, load_scenario


def test_catalog_tables_match_reference_golden():
    from rl4rs_amd.data import CatalogTables
    for name, path in (('slate_discrete', 'catalog_synth.csv'), ('real_discrete', 'item_info_real.csv')):
        g = np.load(os.path.join(GOLDEN, name + '.npz'))
        t = CatalogTables(os.path.join(GOLDEN, path), 284)
        assert np.array_equal(t.action_emb, g['action_emb'])          # bit-exact float64
        assert t.item_vec.dtype == np.float32 and t.item_vec.shape == (284, 40)
        assert not t.item_vec[0].any() and t.price[0] == 0.0
        assert t.location_mask.sum(1).tolist() == [39, 108, 136, 1]
    assert len(t.special_items) == 113 and t.price.max() == 1478.1
    d = t.item_info_dict()
    assert d['0']['price'] == 0.0 and len(d['1']['item_vec']) == 40 and len(d) == 284


def test_record_columns_match_oracle_parser():
    from rl4rs_amd.data import RecordColumns
    from oracle.records import ParsedRecords, pad_sequences
    for name in ('slate_discrete', 'seq36_discrete', 'real_discrete'):
        m, cfg, records, g = load_scenario(name)
        c = RecordColumns(records, 64)
        p = ParsedRecords(records)
        assert np.array_equal(c.exposed, np.array(p.exposed))
        assert np.array_equal(c.feedback, np.array(p.feedback))
        assert np.array_equal(c.history, pad_sequences(p.history, 64))
        assert np.array_equal(c.user_cat, p.user_cat)
        assert np.array_equal(c.user_dense, p.user_dense.astype(np.float32))
        assert c.users == p.users
        # the un-acted state of the reference = user features padded
        assert np.array_equal(c.user_dense, g['dense_init'][:, :32])
        assert np.array_equal(c.user_cat, g['cat_init'][:, :10])
        assert np.array_equal(c.history, g['seq_init'][:, 0])


class _CaptureState(object):
    def __init__(self, config, records, **kw):
        self.records = records
        self.kw = kw


def test_recdatabase_cache_wrap_and_rng(tmp_path):
    """base.py:82-100: cache window, wrap-around (skip line 0, take line 1), np.random.choice stream."""
    from rl4rs_amd.env.base import RecDataBase
    lines = ['rec%d' % i for i in range(7)]
    p = tmp_path / 'log.csv'
    p.write_text('\n'.join(lines) + '\n')
    cfg = {'sample_file': str(p), 'maxlen': 64, 'cache_size': 5, 'is_eval': False}
    db = RecDataBase(cfg, _CaptureState)
    db.reset()
    assert db.sample_list == lines[:5]
    db.reset()
    assert db.sample_list == lines[5:7] + [lines[1], lines[2], lines[3]]
    db.reset(reset_file=True)
    assert db.sample_list == lines[:5]
    db.seed(42)
    st = db.sample(3)
    np.random.seed(42)
    assert list(st.records) == list(np.random.choice(lines[:5], 3))
    assert list(st.records.rows) == [lines.index(x) for x in st.records]
    # eval mode: the first B cached lines, cache_size must equal B (base.py:93-96)
    ev = RecDataBase(dict(cfg, is_eval=True), _CaptureState)
    ev.reset()
    assert list(ev.sample(5).records) == lines[:5]
    with pytest.raises(AssertionError):
        ev.sample(4)


def test_featureutil_list_path_matches_reference_golden():
    """feature_extraction on the reference's nested-list rows (datautil.py:34-69)."""
    from rl4rs_amd.utils.datautil import FeatureUtil
    from rl4rs_amd.env.slate import SlateState
    m, cfg, records, g = load_scenario('slate_discrete')
    fu = FeatureUtil(cfg)
    rows = SlateState.records_to_state(records)
    (seq, dense, cat, labels), y = fu.feature_extraction(rows)
    assert np.array_equal(seq, g['seq_init']) and seq.dtype == np.int32
    assert np.array_equal(dense, g['dense_init']) and dense.dtype == np.float32
    assert np.array_equal(cat, g['cat_init']) and cat.dtype == np.int32
    assert labels.shape == (len(records), 9) and y == [0] * len(records)
    parts = FeatureUtil.record_split(records[0])
    assert len(parts) == 9 and isinstance(parts[3], list) and isinstance(parts[6][0], float)
    with pytest.raises(ValueError):
        FeatureUtil.record_split('a@b')


def test_single_elem_support_semantics():
    from rl4rs_amd.env.base import single_elem_support

    @single_elem_support
    def f(x):
        return x
    assert f([5]) == 5
    assert f(np.array([[1, 2]])).tolist() == [1, 2]
    assert f(([[1, 2]], [3.0], [0], [{}])) == [[1, 2], 3.0, 0, {}]
    assert f([1, 2]) == [1, 2]
    out = f((np.zeros((2, 3)), [0.0, 1.0], [0, 0], [{}, {}]))
    assert isinstance(out, tuple) and out[0].shape == (2, 3)


def test_reference_import_paths_and_spaces():
    import rl4rs  # noqa: F401
    from rl4rs.env import RecState, RecSimBase, RecDataBase, RecEnvBase
    from rl4rs.env.slate import SlateRecEnv, SlateState
    from rl4rs.env.seqslate import SeqSlateRecEnv, SeqSlateState
    from rl4rs.utils.datautil import FeatureUtil  # noqa: F401
    from rl4rs.utils.rllib_vector_env import MyVectorEnvWrapper  # noqa: F401
    from rl4rs_amd.env.base import _observation_space, _action_space
    assert issubclass(SeqSlateState, SlateState) and issubclass(SlateState, RecState)
    assert issubclass(SeqSlateRecEnv, SlateRecEnv) and issubclass(SlateRecEnv, RecSimBase)
    for cls in (RecState, RecSimBase):
        with pytest.raises(TypeError):
            cls({}, [])
    assert RecDataBase and RecEnvBase
    sp = _observation_space({'support_rllib_mask': True}, {'action_mask': np.zeros(284), 'obs': np.zeros(256)})
    assert sp.spaces['action_mask'].shape == (284,) and sp.spaces['obs'].shape == (256,)
    assert _observation_space({}, np.zeros(256)).shape == (256,)
    assert _action_space({'support_conti_env': True, 'action_emb_size': 32}).shape == (32,)
    assert _action_space({'action_size': 284}).n == 284


def test_synthetic_logs_are_legal_by_construction(tmp_path):
    from rl4rs_amd import synth
    from oracle.state import OracleState
    text = synth.make_catalog_text(seed=5)
    p = str(tmp_path / 'c.csv')
    synth.write_text(p, text)
    recs = synth.make_records(64, pages=4, seed=3, illegal_frac=0.0, special_ids=synth.special_ids_from_text(text))
    cfg = {"maxlen": 64, "batch_size": 64, "action_size": 284, "dense_feature_num": 432, "category_feature_num": 21,
           "max_steps": 36, "iteminfo_file": p}
    st = OracleState(cfg, recs, seq=True)
    for t in range(36):
        st.act(st.offline_action)
    cfg2 = dict(cfg, max_steps=9)
    st2 = OracleState(cfg2, recs)
    for t in range(9):
        st2.act(st2.offline_action)
    assert st2.get_violation().all()


def test_native_record_parser_matches_python_parser():
    """rl4rs_parse_records (host C++ in librl4rs_hip.so) == RecordColumns on golden + synthetic records; errors."""
    import time
    from rl4rs_amd.build import build_lib
    build_lib()
    from rl4rs_amd import synth, _lib
    from rl4rs_amd.data import RecordColumns, parse_records_native
    recs = []
    for name in ('slate_discrete', 'seq36_discrete', 'real_discrete'):
        recs.append(load_scenario(name)[2])
    text = synth.make_catalog_text(seed=9)
    recs.append(synth.make_records(500, pages=4, seed=77, special_ids=synth.special_ids_from_text(text)))
    for rr in recs:
        a = RecordColumns(rr, 64)
        b = parse_records_native(rr, 64)
        for f in ('exposed', 'feedback', 'history', 'user_cat', 'user_dense', 'exposed_len'):
            assert np.array_equal(getattr(a, f), getattr(b, f)), f
        assert b.user_dense.dtype == np.float32 and b.history.dtype == np.int32
    # narrower log_steps truncates like RecordColumns(log_steps=...)
    a = RecordColumns(recs[3], 64, log_steps=9)
    b = parse_records_native(recs[3], 64, log_steps=9)
    assert np.array_equal(a.exposed, b.exposed) and np.array_equal(a.exposed_len, b.exposed_len)
    for bad in ('a@b', recs[0][0].replace('@', '#', 1), recs[0][0].replace(',', ';', 1)):
        with pytest.raises(_lib.Rl4rsHipError):
            parse_records_native([bad], 64, log_steps=9)
    big = synth.make_records(4000, pages=1, seed=5)
    def best_of(fn, n=3):                   # best of three: a busy host must not decide this
        best = float('inf')
        for _ in range(n):
            t = time.time(); fn(big, 64); best = min(best, time.time() - t)
        return best
    assert best_of(parse_records_native) < best_of(RecordColumns)


def test_fileutil_lookup_helpers(tmp_path):
    """rl4rs/utils/fileutil.py:7-24: glob under each directory of a pathsep-separated search path; newest by ctime, '' if none."""
    import os
    import time
    from rl4rs.utils.fileutil import find_match_files, find_newest_files
    a, b = tmp_path / 'a', tmp_path / 'b'
    a.mkdir()
    b.mkdir()
    (a / 'train.tfrecord-0').write_text('x')
    time.sleep(0.02)
    (b / 'train.tfrecord-1').write_text('y')
    (b / 'other.txt').write_text('z')
    path = os.pathsep.join([str(a), str(b)])
    found = sorted(os.path.basename(p) for p in find_match_files('train.tfrecord*', path))
    assert found == ['train.tfrecord-0', 'train.tfrecord-1']
    assert os.path.basename(find_newest_files('train.tfrecord*', path)) == 'train.tfrecord-1'
    assert find_newest_files('nothing*', path) == ''
    assert list(find_match_files('*.txt', str(a))) == []


def test_logstore_width_is_the_longest_record(tmp_path):
    """the exposed/feedback table is as wide as the LONGEST record of the file, not the first parsed one (a later,
    longer record would be truncated and change offline_reward, slate.py:164-174)."""
    from rl4rs_amd import synth
    from rl4rs_amd.env.base import LogStore
    text = synth.make_catalog_text(seed=4)
    sp = synth.special_ids_from_text(text)
    recs = synth.make_records(5, pages=1, seed=1, hash_size=2000, special_ids=sp) + \
        synth.make_records(3, pages=2, seed=2, hash_size=2000, special_ids=sp)
    path = os.path.join(str(tmp_path), 'log.csv')
    synth.write_records(path, recs)
    store = LogStore(path, 64)
    store._ensure_width([0])
    assert store.log_steps == 18
    store.ensure([0, 6], 'cpu')                         # host tensors: the parser and the table logic need no GPU
    assert store.exposed_len[0] == 9 and store.exposed_len[6] == 18
    assert (store._dev['exposed'][0, 9:] == 0).all() and (store._dev['exposed'][6] > 0).all()


def test_lazy_stats_is_a_mapping_that_resolves_once_and_pickles_as_a_dict():
    """rl4rs_amd.train.LazyStats: what Trainer.train_iteration returns (the numbers stay on the device until read)."""
    import pickle
    from rl4rs_amd.train import LazyStats

    class FakeTrainer(object):
        calls = 0

        def _resolve(self, token):
            FakeTrainer.calls += 1
            return {'policy_loss': 1.5, 'kl': 0.25, 'iteration': token}

    st = LazyStats(FakeTrainer(), 7)
    assert FakeTrainer.calls == 0                       # nothing waited for yet
    assert st['policy_loss'] == 1.5 and st.get('missing', 3) == 3 and 'kl' in st and len(st) == 3
    assert sorted(st.keys()) == ['iteration', 'kl', 'policy_loss'] and dict(st)['iteration'] == 7
    assert FakeTrainer.calls == 1                       # one resolution serves every access
    back = pickle.loads(pickle.dumps(st))
    assert type(back) is dict and back == {'policy_loss': 1.5, 'kl': 0.25, 'iteration': 7} and st == back


def test_offline_action_list_is_a_list_that_forgets_its_device_copy_when_changed():
    """rl4rs_amd.device.OfflineActionList: what offline_action hands out in the reference-shaped modes - a plain list to every
    consumer (equality, json, numpy, pickle / deepcopy give plain lists), carrying a reference to the device copy of the same
    ids that ANY in-place change drops (a modified list must be converted and uploaded like any other)."""
    import copy
    import json
    import pickle
    from rl4rs_amd.device import OfflineActionList
    dev = object()
    a = OfflineActionList([3, 1, 2], dev=dev, tag=(1, 0))
    assert isinstance(a, list) and a == [3, 1, 2] and a._dev is dev and a._tag == (1, 0)
    assert json.dumps(a) == '[3, 1, 2]' and np.asarray(a).tolist() == [3, 1, 2]
    assert type(pickle.loads(pickle.dumps(a))) is list and type(copy.deepcopy(a)) is list
    b = a + [4]                                       # new objects are plain lists, the original keeps its reference
    assert type(b) is list and a._dev is dev
    assert a[1:] == [1, 2] and a._dev is dev          # reading does not drop it
    for change in (lambda x: x.__setitem__(0, 9), lambda x: x.append(1), lambda x: x.extend([1]), lambda x: x.insert(0, 1),
                   lambda x: x.pop(), lambda x: x.remove(1), lambda x: x.reverse(), lambda x: x.sort(), lambda x: x.clear(),
                   lambda x: x.__iadd__([5]), lambda x: x.__imul__(2), lambda x: x.__delitem__(0),
                   lambda x: x.__setitem__(slice(0, 1), [7])):
        c = OfflineActionList([3, 1, 2], dev=dev, tag=(1, 0))
        change(c)
        assert c._dev is None, change


def test_scorer_kernel_option_names():
    """config['scorer_kernels'] parsing (names of rl4rs_dien_cfg.kernel_opts bits) - no GPU needed."""
    from rl4rs_amd.device import parse_dien_opts
    from rl4rs_amd import _lib
    assert parse_dien_opts(None) == () and parse_dien_opts('') == ()
    assert parse_dien_opts(' augru_h16 , no_din16,augru_h16') == ('augru_h16', 'no_din16')
    assert parse_dien_opts(['DIN_V1']) == ('din_v1',)
    with pytest.raises(ValueError, match='scorer_kernels'):
        parse_dien_opts('augru_rows48')
    bits = [_lib.DIEN_OPTS[k] for k in _lib.DIEN_OPTS]
    assert sorted(bits) == [1 << i for i in range(len(bits))]          # one distinct bit each, contiguous from bit 0
    # the header's enum and the binding agree (names and values)
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'rl4rs_hip.h')).read()
    for name, bit in _lib.DIEN_OPTS.items():
        m = re.search(r'RL4RS_DIEN_OPT_%s = 1 << (\d+)' % name.upper(), hdr)
        assert m and (1 << int(m.group(1))) == bit, name
    for name, val in _lib.STEP_WANT.items():
        m = re.search(r'RL4RS_STEP_WANT_%s = (\d+)' % name.upper(), hdr)
        assert m and int(m.group(1)) == val, name


def test_cache_window_with_blank_lines_matches_the_line_by_line_loop(tmp_path):
    """base.py:82-90 read line by line (readline().rstrip(); on an empty read: seek(0), skip one line, take the next) against
    the bulk form RecDataBase.sample_cache uses (slices of the blank-free runs), on files with blank lines in odd places."""
    from rl4rs_amd.env.base import RecDataBase
    rng = np.random.RandomState(5)
    for trial in range(30):
        n = int(rng.randint(3, 40))
        lines = ['rec%d  ' % i if rng.rand() > 0.15 else ('' if rng.rand() > 0.5 else '   ') for i in range(n)]
        lines[1] = 'rec1'                                  # the line taken after a wrap
        text = '\n'.join(lines) + ('\n' if rng.rand() > 0.3 else '')
        p = tmp_path / ('log%d.csv' % trial)
        p.write_text(text)
        file_lines = text.split('\n')

        class Loop(object):                                # the reference's loop over a file object
            def __init__(self):
                self.c = 0

            def readline(self):
                if self.c >= len(file_lines):
                    return ''
                self.c += 1
                return file_lines[self.c - 1].rstrip()

            def window(self, num):
                out = []
                for _ in range(num):
                    tmp = self.readline()
                    if len(tmp) < 1:
                        self.c = 0
                        self.readline()
                        tmp = self.readline()
                    out.append(tmp)
                return out

        cache = int(rng.randint(1, 25))
        db = RecDataBase({'sample_file': str(p), 'maxlen': 64, 'cache_size': cache, 'is_eval': False}, _CaptureState)
        ref = Loop()
        for _ in range(6):
            db.reset()
            want = ref.window(cache)
            assert db.sample_list == want, (trial, lines, cache)
            assert [file_lines[r].rstrip() if r >= 0 else '' for r in db.sample_rows] == want


def test_ranks_sharing_device(monkeypatch):
    """rl4rs_amd/dist.py::ranks_sharing_device: 1 in the product configuration (one process per GPU), ceil(local ranks / GPUs) when a
    dry run puts more ranks on a node than it has GPUs (the trainer then shrinks the persistent PPO pass's co-residency budget)."""
    import torch
    from rl4rs_amd import dist as rdist
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE'):
        monkeypatch.delenv(k, raising=False)
    assert rdist.ranks_sharing_device() == 1
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: True)
    for world, n_dev, want in ((8, 8, 1), (8, 1, 8), (8, 3, 3), (2, 1, 2), (1, 1, 1), (4, 8, 1)):
        monkeypatch.setenv('WORLD_SIZE', str(world))
        monkeypatch.setattr(torch.cuda, 'device_count', lambda n=n_dev: n)
        assert rdist.ranks_sharing_device() == want, (world, n_dev)
    monkeypatch.setenv('WORLD_SIZE', '16')                  # two nodes of eight: LOCAL_WORLD_SIZE decides
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '8')
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 8)
    assert rdist.ranks_sharing_device() == 1
