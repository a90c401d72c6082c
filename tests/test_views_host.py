"""host logic of katome_amd.device: close() of an owner whose zero-copy views are still alive (no GPU needed)"""
import gc

from katome_amd.device import _DevArray, _ViewOwner


class _Owner(_ViewOwner):
    def __init__(self):
        self.destroyed = 0

    def _destroy(self):
        self.destroyed += 1


def test_close_is_put_off_until_the_last_view_is_gone():
    o = _Owner()
    a = _DevArray(0x1000, (4,), "<i8", o)
    b = _DevArray(0x2000, (4,), "<i4", o)
    assert o._views == 2
    o.close()
    assert o.destroyed == 0 and o._close_pending
    del a
    gc.collect()
    assert o._views == 1 and o.destroyed == 0
    del b
    gc.collect()
    assert o._views == 0 and o.destroyed == 1 and not o._close_pending


def test_close_without_views_is_immediate_and_views_without_close_leave_the_owner_alone():
    o = _Owner()
    o.close()
    assert o.destroyed == 1
    p = _Owner()
    v = _DevArray(0x1000, (1,), "|u1", p)
    del v
    gc.collect()
    assert p._views == 0 and p.destroyed == 0


def test_views_of_an_inner_builder_count_against_the_sharded_builder():
    from katome_amd.shard import _InnerBuilder
    o = _Owner()
    inner = _InnerBuilder(0, 31, True, 0, parent=o)
    v = _DevArray(0x1000, (1,), "|u1", inner)
    assert o._views == 1
    o.close()
    assert o.destroyed == 0
    del v, inner
    gc.collect()
    assert o.destroyed == 1
