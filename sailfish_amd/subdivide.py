"""
Slab domain decomposition helpers.

Counterpart of the reference's ``sailfish/subdivide.py``: the mesh is cut along
axis 0 into contiguous slabs (``subdivide``, reference lines 35-60), reductions
over patches launch everything first and block afterwards (``lazy_reduce``,
lines 13-32), and the solution is gathered to one host array with the guard
zones stripped (``concat_on_host``, lines 63-122).
"""
import numpy as np

from .kernel.system import to_host  # re-exported; reference: subdivide.py:6-10

__all__ = ["to_host", "lazy_reduce", "partition", "subdivide", "concat_on_host"]


def lazy_reduce(reduction, block, launches, contexts):
    """
    Start every launch, then resolve the returned tokens one by one under
    their execution context and reduce the results.
    """
    tokens = [launch() for launch in launches]
    values = []
    for token, context in zip(tokens, contexts):
        with context:
            values.append(block(token))
    return reduction(values)


def partition(elements, num_parts):
    """Sizes of `num_parts` nearly equal parts; the first ``elements % num_parts`` get one extra."""
    base, extra = divmod(elements, num_parts)
    return [base + (1 if k < extra else 0) for k in range(num_parts)]


def subdivide(interval, num_parts):
    """Contiguous, non-overlapping ``(a, b)`` index ranges covering `interval`."""
    if isinstance(interval, (tuple, list)):
        start, stop = interval
    else:
        start, stop = 0, interval
    ranges = []
    for size in partition(stop - start, num_parts):
        ranges.append((start, start + size))
        start += size
    return ranges


def _strip(array, guards):
    index = tuple(slice(g, array.shape[axis] - g) for axis, g in enumerate(guards))
    return array[index]


def concat_on_host(arrays, num_guard=None, rank=None):
    """
    Concatenate patch arrays (possibly on different devices) along axis 0 into
    one float64 host array, dropping `num_guard` zones per side.

    `num_guard` is an int for 1-d grids and ``(ngi, ngj)`` for 2-d grids, as in
    the reference.  `rank` is the number of grid axes; by default every axis
    but the last.
    """
    if rank is None:
        rank = len(arrays[0].shape) - 1
    if rank == 1:
        guards = (num_guard or 0,)
    elif rank == 2:
        guards = tuple(num_guard or (0, 0))
    else:
        raise ValueError(f"concatenation for arrays of rank {rank} not supported")

    pieces = [to_host(_strip(a, guards)) for a in arrays]
    trailing = {p.shape[1:] for p in pieces}
    if len(trailing) != 1:
        raise ValueError("got distinct values")
    return np.concatenate(pieces, axis=0).astype(np.float64, copy=False)
