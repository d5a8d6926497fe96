"""Bookkeeping of the batched-W^T table of the training path (_train._WtTable), on the CPU: which entries it would hand
to avsep_op_transpose_many.  nn.Module.to() / .cpu() / .double() swap ``p.data`` on the same Parameter object, so the
table must look at where the storage IS, not at the identity of the tensor (a stale row pointing at
host memory, a peer device or a half-sized buffer is a GPU fault)."""
import gc

import torch

from av_separation import _train as tr


def _table():
    return tr._WtTable(torch.device("cpu"))       # device equality is all the bookkeeping needs; nothing is launched


def _live(t):
    return [w for w, _ in t._descriptors()]


def test_entry_follows_a_same_device_storage_swap():
    t = _table()
    w = torch.nn.Parameter(torch.randn(8, 32))
    t.register(w)
    assert _live(t) == [w] and t.ptrs[0] == w.data_ptr()
    w.data = w.data.clone()
    assert [x.data_ptr() for x in _live(t)] == [w.data_ptr()] and t.ptrs[0] == w.data_ptr()
    w.data = torch.randn(16, 64)                   # another shape: the W^T buffer is re-made
    (lw, buf), = t._descriptors()
    assert lw is w and tuple(buf.shape) == (64, 32)


def test_entry_is_dropped_when_dtype_or_layout_changes():
    t = _table()
    a, b, c = (torch.nn.Parameter(torch.randn(8, 32)) for _ in range(3))
    for w in (a, b, c):
        t.register(w)
    assert len(_live(t)) == 3
    a.data = a.data.double()                       # Module.double()
    b.data = torch.randn(32, 8).t()                # a strided view
    assert _live(t) == [c]
    assert t.get(a) is None and t.get(b) is None
    a.data = a.data.float()                        # back to float32: usable again once a forward re-registers it
    t.register(a)
    assert set(map(id, _live(t))) == {id(a), id(c)}


def test_weight_on_another_device_is_never_listed():
    t = tr._WtTable(torch.device("meta"))          # a table of some OTHER device
    w = torch.nn.Parameter(torch.randn(8, 32))     # a CPU weight
    t.register(w)
    assert t._descriptors() == [] and t.get(w) is None
    # and an entry registered while on the table's device is dropped when Module.to() moves it away
    t2 = _table()
    t2.register(w)
    assert len(_live(t2)) == 1
    t2.device = torch.device("meta")               # same situation seen from the table: weight and table devices differ
    assert t2._descriptors() == [] and t2.get(w) is None


def test_dead_models_leave_no_slots():
    t = _table()
    keep = torch.nn.Parameter(torch.randn(8, 32))
    t.register(keep)
    for _ in range(5):
        w = torch.nn.Parameter(torch.randn(8, 32))
        t.register(w)
        del w
    gc.collect()
    assert _live(t) == [keep]
    assert len(t.weights) == len(t.bufs) == len(t.ptrs) == len(t.index) == 1 and t.index[id(keep)] == 0


def test_get_survives_the_compaction_its_own_refresh_triggers(monkeypatch):
    """get() refreshes a stale table, the refresh drops dead slots and renumbers the live ones: the position looked up
    before the refresh must not be used after it."""
    t = _table()
    dead = [torch.nn.Parameter(torch.randn(8, 32)) for _ in range(3)]
    for w in dead:
        t.register(w)
    keep = torch.nn.Parameter(torch.randn(8, 32))
    t.register(keep)                                # position 3
    del dead, w
    gc.collect()
    monkeypatch.setattr(tr, "_ck", lambda rc, what="": rc)            # no library on the CPU box: the launch is a no-op

    class _Lib:
        def avsep_op_transpose_many(self, *a):
            return 0
    monkeypatch.setattr(tr, "_lib", lambda: _Lib())
    monkeypatch.setattr(torch.cuda, "current_stream", lambda dev=None: type("S", (), {"cuda_stream": 0})())
    buf = t.get(keep)
    assert buf is not None and tuple(buf.shape) == (32, 32) and t.index[id(keep)] == 0 and not t.stale
