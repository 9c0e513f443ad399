"""Result arrays in recycled page-locked memory.

The reference's calls take NumPy arrays and return fresh ones (`MGCMTSolver.py:326 return v`, `:431`).  A fresh
pageable array costs more than the PCIe transfer that fills it: its pages are faulted in on first touch and unmapped
again when the caller drops it, and the DMA engine cannot write to it directly (csrc/transfer.hip stages such
transfers through pinned chunks).  `empty(n)` returns an ordinary float64 ndarray whose memory is a buffer of
`mgcmt_host_alloc` (include/mgcmt_hip.h): the download is one DMA, and when the last reference to the array (or to any
view of it) goes away the buffer returns to a free list for the next result of that size.  A result that is passed
back in as the next call's `v0` uploads as one DMA too.

Page-locked memory is a bounded resource: buffers in use plus buffers on the free list never exceed
MGCMT_PINNED_POOL_BYTES (default 8 GiB; 0 disables the pool); beyond that — and for arrays below the staging threshold
of transfer.hip, where none of this matters — `empty` is `numpy.empty`."""
import collections
import ctypes
import os
import threading
import weakref

import numpy as np

from . import _lib

MIN_BYTES = 16 << 20
_lock = threading.Lock()
_free = {}            # (library, nbytes) -> [address, ...]  (library: the binding a buffer came from; tests switch it)
_libs = {}            # id -> library object
_total = 0            # bytes allocated from mgcmt_host_alloc and not yet given back to it
_stats = {"allocated": 0, "reused": 0, "fallback": 0}


def _limit():
    try:
        return int(os.environ.get("MGCMT_PINNED_POOL_BYTES", 8 << 30))
    except ValueError:
        return 8 << 30


_returned = collections.deque()   # (key, address) of dropped results, not yet on the free list


def _release(key, address):
    """Finalizer of a result array.  It may run inside ANY allocation of this thread (a cyclic-GC pass), also one made
    while `_lock` is held — so it takes no lock: deque.append is atomic, and the entries are folded into `_free` by the
    next `empty()` / `drain()` / `stats()` under the lock."""
    _returned.append((key, address))


def _fold_returned():
    """dropped results -> free list (lock held)"""
    while True:
        try:
            key, address = _returned.popleft()
        except IndexError:
            return
        _free.setdefault(key, []).append(address)


def _trim(need, limit):
    """free-list buffers back to the allocator until `need` more bytes fit under the limit (lock held)"""
    global _total
    if limit >= 0 and _total - sum(k[1] * len(v) for k, v in _free.items()) + need > limit:
        return False  # (would not fit even with every free buffer given back: keep them)
    for key in sorted(_free, key=lambda k: -k[1]):
        lst = _free[key]
        while lst and _total + need > limit:
            _libs[key[0]].mgcmt_host_free(ctypes.c_void_p(lst.pop()))
            _total -= key[1]
    return _total + need <= limit


def empty(n):
    """float64 ndarray of n elements; contents undefined"""
    global _total
    n = int(n)
    nbytes = 8 * n
    limit = _limit()
    if nbytes < MIN_BYTES or limit <= 0:
        return np.empty(n, dtype=np.float64)
    address = None
    lib = _lib.lib()
    key = (id(lib), nbytes)
    with _lock:
        _fold_returned()
        _libs[id(lib)] = lib
        lst = _free.get(key)
        if lst:
            address = lst.pop()
            _stats["reused"] += 1
        elif _total + nbytes <= limit or _trim(nbytes, limit):
            p = ctypes.c_void_p()
            if lib.mgcmt_host_alloc(nbytes, ctypes.byref(p)) == 0 and p.value:
                address = p.value
                _total += nbytes
                _stats["allocated"] += 1
    if address is None:
        _stats["fallback"] += 1
        return np.empty(n, dtype=np.float64)
    buf = (ctypes.c_double * n).from_address(address)
    fin = weakref.finalize(buf, _release, key, address)
    fin.atexit = False
    return np.frombuffer(buf, dtype=np.float64)


def stats():
    with _lock:
        _fold_returned()
        return dict(_stats, pinned_bytes=_total, free_buffers=sum(len(v) for v in _free.values()))


def drain():
    """give every buffer on the free list back to the allocator (tests; a caller that wants the memory back)"""
    with _lock:
        _fold_returned()
        _trim(0, -1)
