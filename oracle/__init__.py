"""CPU oracle for the rl4rs batched env.step() hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it; the product package ``rl4rs_amd`` never does
(``tests/test_no_oracle_in_product.py`` enforces that).

It is a vectorised numpy restatement of the reference algorithm, each function citing the reference
file:line it follows:

* ``oracle.catalog``  – ``rl4rs/env/slate.py:28-65`` (catalogue + masks)
* ``oracle.records``  – ``rl4rs/utils/datautil.py:20-32``, ``rl4rs/env/slate.py:67-83``
* ``oracle.state``    – ``SlateState`` / ``SeqSlateState`` (``rl4rs/env/slate.py``, ``seqslate.py``)
* ``oracle.dien``     – DIEN scorer (``rl4rs/nets/dien.py``, ``rl4rs/nets/utils.py`` + deepctr 0.9.0 /
  TF 1.15 cell equations restated from their published definitions)
* ``oracle.simnets``  – the dnn / widedeep / lstm simulators (``rl4rs/nets/dnn.py``, ``widedeep.py``, ``lstm.py``,
  ``rl4rs/nets/utils.py:7-97``; keras GRU v1 cell restated from its published definition)
* ``oracle.env``      – ``RecSimBase._step`` order (``rl4rs/env/base.py:157-170``) and the two
  ``forward`` reward rules (``slate.py:281-308``, ``seqslate.py:136-160``)
* ``oracle.policy``   – action-masked policy net (``rl4rs/nets/rllib/rllib_mask_model.py:41-62``)

Pinning status
--------------
* state machine / features / masks / K-NN / violation / offline action+reward / reward reduction:
  PINNED against golden vectors captured from the reference's own numpy state machine imported in the
  build container (``tests/golden/make_golden.py``) and against the tutorial known answers
  (SURVEY.md §8c).
* DIEN / dnn / widedeep / lstm arithmetic and policy net: **parity unpinned** – the arithmetic lives in deepctr==0.9.0 +
  tensorflow-gpu==1.15.0 / ray==1.5.1, none of which are vendored or installable here, and the
  reference holds no test vector or checkpoint for them.  The restatement follows the reference's own
  topology call sites and the libraries' published cell equations; it is the checker for the HIP
  scorer with seeded synthetic weights.
"""
