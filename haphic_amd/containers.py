"""What the S5 mirrors hand back to the reference's run() (:2869-2935): dict-shaped containers that stay ARRAYS.

The reference's loops return `defaultdict`s keyed by name tuples (:1605-1620) and its run() then walks them through
output_pickle :710, output_clm :376, normalize_by_nlinks :718, filter_fragments :741, dict_to_matrix :310.  At 100k contigs /
500 M pairs those dicts hold 1.6e8 + 1.2e8 + ~3e8 keys: building them as Python objects costs minutes around a link-matrix
build that takes 40 ms on the device.  So the mirrors return

    LinkTable  (full_link_dict, flank_link_dict, HT_link_dict)   — a defaultdict subclass backed by an IngestSession
    PairLists  (clm_dict, ctg_coord_dict)                        — the same for the per-contig-pair arrays

whose entries exist only as the device tables of the ingest handle (and, once someone asks, as numpy arrays in dict order).
The seams that run() calls next (cluster.filter_fragments / normalize_by_nlinks / dict_to_matrix / output_pickle / output_clm)
recognise a table that is still `frozen` and work on the arrays with no per-key Python.  ANY other access — item lookup,
iteration, assignment, deletion, as remove_allelic_HiC_links :474-689 or the --remove_concentrated_links loop :2888-2891 do —
thaws the container first: the real dict is filled with C-speed constructors (`dict.update(zip(...))`), the object turns into a
plain defaultdict subclass, and from then on every mirror takes its generic dict path.  Pickling never needs the subclass:
`__reduce_ex__` describes a plain `collections.defaultdict`, which is what HapHiC sort / reassign unpickle.
"""
from array import array
from collections import defaultdict
from itertools import repeat

import numpy as np


THAW_LOG = []          # (kind, keys, seconds) of every thaw of this process: what the reference's own Python costs when it touches a table


class Thawed(defaultdict):
    """A LinkTable / PairLists after its entries became real dict entries: a defaultdict in all but the class name."""

    def __reduce_ex__(self, protocol):
        return defaultdict, (self.default_factory,), None, None, iter(dict.items(self))

    __reduce__ = lambda self: self.__reduce_ex__(2)             # noqa: E731

    @property
    def frozen(self):
        return False


def _thawing(name):
    def method(self, *args, **kwargs):
        self._thaw()
        return getattr(self, name)(*args, **kwargs)
    method.__name__ = name
    return method


class _Frozen(defaultdict):
    """Base of the array-backed containers.  `_source_items()` yields the (key, value) pairs in dict insertion order."""

    frozen = True

    def __init__(self, factory, session, kind):
        defaultdict.__init__(self, factory)
        self._session = session
        self._kind = kind

    def _n(self):
        raise NotImplementedError

    def _source_items(self):
        raise NotImplementedError

    def _thaw(self):
        if type(self) is Thawed:
            return
        import gc
        import time
        t0 = time.perf_counter()
        session = self._session
        kind = self._kind
        collect = gc.isenabled()
        gc.disable()                      # millions of fresh tuples: the generational collector would walk them again and again
        try:
            dict.update(self, self._source_items())
        except BaseException:             # a MemoryError half-way: stay frozen (and retryable), never a truncated plain dict
            dict.clear(self)
            raise
        finally:
            if collect:
                gc.enable()
        self.__dict__.clear()
        self.__class__ = Thawed
        session.note_thawed()
        THAW_LOG.append((kind, dict.__len__(self), time.perf_counter() - t0))

    # answered from the arrays
    def __len__(self):
        return self._n()

    def __bool__(self):
        return self._n() > 0

    def __reduce_ex__(self, protocol):
        return defaultdict, (self.default_factory,), None, None, iter(self._source_items())

    __reduce__ = lambda self: self.__reduce_ex__(2)             # noqa: E731

    def __copy__(self):
        out = defaultdict(self.default_factory)
        out.update(self._source_items())
        return out

    copy = __copy__


for _name in ('__getitem__', '__setitem__', '__delitem__', '__contains__', '__iter__', '__reversed__', '__missing__', 'keys', 'values',
              'items', 'get', 'pop', 'popitem', 'setdefault', 'update', 'clear', '__eq__', '__ne__', '__repr__', '__or__', '__ror__',
              '__ior__'):
    setattr(_Frozen, _name, _thawing(_name))


class LinkTable(_Frozen):
    """full_link_dict / flank_link_dict / HT_link_dict (:1605-1616): {(name, name): links}.  kind = 'full' | 'flank' | 'HT'."""

    def __init__(self, session, kind):
        _Frozen.__init__(self, int, session, kind)

    def _n(self):
        return self._session.n_keys(self._kind)

    def arrays(self):
        """(i, j, value, names): ids into `names` and the values (int64 counts, or float64 after a weighting step) in dict order"""
        return self._session.link_arrays(self._kind)

    def _source_items(self):
        i, j, v, names = self.arrays()
        table = np.empty(len(names), object)                  # the names gathered by numpy's take on an object array: 1.5 x the rate
        table[:] = names                                       # of map(names.__getitem__, ids) (measured; the dict insert itself is ~half the time)
        return zip(zip(table[i].tolist(), table[j].tolist()), v.tolist())


class PairLists(_Frozen):
    """clm_dict (update_clm_dict :395-401) / ctg_coord_dict (record_coord_pairs :454-471): {(ctg, ctg): array}.
    kind = 'clm' | 'crd'."""

    def __init__(self, session, kind, code):
        _Frozen.__init__(self, lambda: array(code), session, kind)

    def _n(self):
        return self._session.n_keys('full')

    def _source_items(self):
        return self._session.pair_items(self._kind)


def slices_as_arrays(code, flat, ptr, width):
    """array(code) objects over flat[width * ptr[k] : width * ptr[k + 1]] for every k, built by C-level iterators only
    (slice objects -> memoryview slices -> array(code, bytes) = array.frombytes)."""
    view = memoryview(np.ascontiguousarray(flat)).cast('B')
    step = width * flat.dtype.itemsize
    bounds = (np.asarray(ptr, np.int64) * step).tolist()
    return map(array, repeat(code), map(bytes, map(view.__getitem__, map(slice, bounds[:-1], bounds[1:]))))
