"""haphic_amd/containers.py: an array-backed table must be indistinguishable from the defaultdict the reference's loops build — for ANY
sequence of dict operations, applied to a LinkTable / PairLists over a stand-in session and to the plain defaultdict holding the same
items: same return values, same exceptions, same final contents in the same order, and a pickle that loads as the plain defaultdict.
(What the seams of run() do with a table that is still frozen is pinned by tests/test_seam_containers.py.)"""
import copy
import pickle
from array import array
from collections import defaultdict

import numpy as np
from hypothesis import example, given, settings
from hypothesis import strategies as st

from haphic_amd import containers

NAMES = ['ctg%d' % k for k in range(12)]


class FakeSession:
    """the part of cluster.IngestSession the containers use"""

    def __init__(self, i, j, v, lists=None):
        self.i, self.j, self.v = np.asarray(i, np.int32), np.asarray(j, np.int32), np.asarray(v)
        self.lists = lists
        self.thawed = 0

    def n_keys(self, kind):
        return len(self.i)

    def link_arrays(self, kind):
        return self.i, self.j, self.v, NAMES

    def pair_items(self, kind):
        keys = [(NAMES[a], NAMES[b]) for a, b in zip(self.i.tolist(), self.j.tolist())]
        return zip(keys, [array('i', x) for x in self.lists])

    def note_thawed(self):
        self.thawed += 1


def _keys(draw_pairs):
    seen, i, j = set(), [], []
    for a, b in draw_pairs:
        if a != b and (a, b) not in seen:
            seen.add((a, b))
            i.append(a)
            j.append(b)
    return i, j


pairs = st.lists(st.tuples(st.integers(0, 11), st.integers(0, 11)), max_size=30)
key = st.tuples(st.sampled_from(NAMES), st.sampled_from(NAMES))
value = st.one_of(st.integers(-5, 10 ** 12), st.floats(allow_nan=False, allow_infinity=False, width=32))
op = st.one_of(
    st.tuples(st.just('get'), key), st.tuples(st.just('getitem'), key), st.tuples(st.just('set'), key, value), st.tuples(st.just('del'), key),
    st.tuples(st.just('in'), key), st.tuples(st.just('len')), st.tuples(st.just('iter')), st.tuples(st.just('items')), st.tuples(st.just('values')),
    st.tuples(st.just('pop'), key), st.tuples(st.just('setdefault'), key, value), st.tuples(st.just('imul'), key, value), st.tuples(st.just('popitem')),
    st.tuples(st.just('update'), st.lists(st.tuples(key, value), max_size=4)), st.tuples(st.just('bool')), st.tuples(st.just('eq')), st.tuples(st.just('copy')),
    st.tuples(st.just('pickle')), st.tuples(st.just('repr')), st.tuples(st.just('clear')), st.tuples(st.just('keys')), st.tuples(st.just('reversed')))


def _apply(d, o, other):
    try:
        if o[0] == 'get':
            return d.get(o[1])
        if o[0] == 'getitem':
            return d[o[1]]                                   # a defaultdict: inserts the default
        if o[0] == 'set':
            d[o[1]] = o[2]
            return None
        if o[0] == 'del':
            del d[o[1]]
            return None
        if o[0] == 'in':
            return o[1] in d
        if o[0] == 'len':
            return len(d)
        if o[0] == 'iter':
            return list(d)
        if o[0] == 'items':
            return list(d.items())
        if o[0] == 'values':
            return list(d.values())
        if o[0] == 'keys':
            return list(d.keys())
        if o[0] == 'reversed':
            return list(reversed(d))
        if o[0] == 'pop':
            return d.pop(o[1])
        if o[0] == 'setdefault':
            return d.setdefault(o[1], o[2])
        if o[0] == 'imul':
            d[o[1]] *= o[2]
            return d[o[1]]
        if o[0] == 'popitem':
            return d.popitem()
        if o[0] == 'update':
            d.update(o[1])
            return None
        if o[0] == 'bool':
            return bool(d)
        if o[0] == 'eq':
            return (d == other, other == d, d != other)
        if o[0] == 'copy':
            c = copy.copy(d)
            return (type(c) is defaultdict or isinstance(c, defaultdict), list(c.items()), c.default_factory)
        if o[0] == 'pickle':
            c = pickle.loads(pickle.dumps(d))
            return (type(c) is defaultdict, list(c.items()), [type(v) for v in c.values()], c.default_factory)
        if o[0] == 'repr':
            return repr(d).split('(', 1)[1]                  # the class name differs, the rest must not
        if o[0] == 'clear':
            d.clear()
            return None
    except Exception as e:                                   # noqa: BLE001 — the exception type is part of the behaviour
        return ('raised', type(e).__name__)
    raise AssertionError(o)


@settings(max_examples=300, deadline=None)
@given(pairs, st.lists(st.integers(1, 10 ** 9), min_size=30, max_size=30), st.lists(op, max_size=12), st.booleans())
def test_link_table_behaves_like_the_defaultdict(draw_pairs, counts, ops, as_float):
    i, j = _keys(draw_pairs)
    v = np.asarray(counts[:len(i)], np.float64 if as_float else np.int64)
    session = FakeSession(i, j, v)
    table = containers.LinkTable(session, 'full')
    plain = defaultdict(int)
    for a, b, c in zip(i, j, v.tolist()):
        plain[(NAMES[a], NAMES[b])] = c
    assert table.frozen and len(table) == len(plain) and bool(table) == bool(plain) and table.frozen
    twin = defaultdict(int, plain)
    for o in ops:
        assert _apply(table, o, twin) == _apply(plain, o, twin), o
        if o[0] not in ('len', 'bool', 'copy', 'pickle'):
            assert not table.frozen, o                        # everything but these four turns it into the real dict
    assert list(table.items()) == list(plain.items()) and table == plain and not table.frozen
    assert [type(x) for x in table.values()] == [type(x) for x in plain.values()]
    assert table.default_factory is int and isinstance(table, defaultdict) and session.thawed == 1
    back = pickle.loads(pickle.dumps(table))
    assert type(back) is defaultdict and list(back.items()) == list(plain.items())


@settings(max_examples=100, deadline=None)
@given(pairs, st.lists(st.lists(st.integers(0, 2 ** 31 - 1), max_size=6), min_size=30, max_size=30), st.lists(op, max_size=8))
@example([], [[]] * 30, [('update', [(('ctg0', 'ctg0'), 0)])])         # round 5's red case: an op may store a non-array under any NAMES key
@example([(0, 1)], [[1, 2]] * 30, [('set', ('ctg0', 'ctg0'), 7), ('setdefault', ('ctg1', 'ctg1'), 1.5)])
def test_pair_lists_behave_like_the_defaultdict(draw_pairs, lists, ops):
    i, j = _keys(draw_pairs)
    session = FakeSession(i, j, np.zeros(len(i), np.int64), lists[:len(i)])
    table = containers.PairLists(session, 'clm', 'i')
    plain = defaultdict(lambda: array('i'))
    for a, b, x in zip(i, j, lists):
        plain[(NAMES[a], NAMES[b])] = array('i', x)
    assert table.frozen and len(table) == len(plain)
    ops = [o for o in ops if o[0] not in ('pickle', 'copy', 'imul', 'repr')]      # the factory is a lambda (never pickled by run()), arrays do not scale
    twin = defaultdict(lambda: array('i'), plain)
    for o in ops:
        assert _apply(table, o, twin) == _apply(plain, o, twin), o
    assert list(table.items()) == list(plain.items())
    absent = ('absent', 'absent')                             # a key no strategy above can write
    assert table[absent] == array('i') and absent in table    # still a defaultdict of arrays


def test_slices_as_arrays():
    flat = np.arange(40, dtype=np.int64)
    ptr = np.array([0, 2, 2, 5, 10])
    got = list(containers.slices_as_arrays('l', flat, ptr, 4))
    assert got == [array('l', range(0, 8)), array('l'), array('l', range(8, 20)), array('l', range(20, 40))]
    assert list(containers.slices_as_arrays('i', flat.astype(np.int32), ptr, 2)) == [array('i', range(0, 4)), array('i'), array('i', range(4, 10)), array('i', range(10, 20))]
