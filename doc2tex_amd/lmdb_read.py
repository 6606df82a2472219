"""Read-only access to an LMDB environment without liblmdb (the on-disk format in front of the training loop).

The reference reads its datasets through `lmdb.open(root, readonly=True, lock=False, ...)`, `env.begin(write=False)` and
`txn.get(key)` (doc2tex/data/lmdb_dataset.py:15-24,37,51-57; py-lmdb pinned at lmdb==1.6.2 in envs/requirements.txt:34, a
binding of LMDB 0.9.x).  That module is absent from this image and the reference holds no `.mdb` file, so this is a
restatement of LMDB's published data-file layout (mdb.c of LMDB 0.9: `MDB_page`, `MDB_node`, `MDB_meta`, `MDB_db`) and it
is UNPINNED: nothing here could be checked against a file written by liblmdb.  tests/test_lmdb.py checks it against an
independent writer of the same layout (tests/lmdb_writer.py) -- internal consistency, not parity.

Layout (little-endian, 64-bit `size_t` / `pgno_t`, page size taken from the meta page):
  page header, 16 bytes      pgno u64 | pad u16 | flags u16 | lower u16, upper u16   (overflow pages: page count u32 instead)
  flags                      P_BRANCH 0x01, P_LEAF 0x02, P_OVERFLOW 0x04, P_META 0x08
  node pointers              u16 offsets from the page start, from byte 16; count = (lower - 16) / 2, sorted by key
  node, 8-byte header        lo u16 | hi u16 | flags u16 | ksize u16 | key | data
                             leaf: data size = lo | hi << 16; F_BIGDATA 0x01: the data field is the u64 number of an
                             overflow page whose payload starts 16 bytes in and runs over consecutive pages
                             branch: child page = lo | hi << 16 | flags << 32; the first node's key is empty (= lowest)
  meta pages 0 and 1         header | magic 0xBEEFC0DE u32 | version u32 | address u64 | mapsize u64 | MDB_db free |
                             MDB_db main | last_pg u64 | txnid u64; the one with the larger txnid is current
  MDB_db, 48 bytes           pad u32 (the page size, in the free DB's record) | flags u16 | depth u16 | branch_pages u64 |
                             leaf_pages u64 | overflow_pages u64 | entries u64 | root u64 (all ones: empty)
Keys compare as byte strings (memcmp, then length): the default comparator, which is what the reference's databases use.
Only the unnamed main database is read; sub-databases and duplicate-sorted data (F_SUBDATA / F_DUPDATA) raise.
"""
import io
import mmap
import os
import struct

P_BRANCH, P_LEAF, P_OVERFLOW, P_META = 0x01, 0x02, 0x04, 0x08
F_BIGDATA, F_SUBDATA, F_DUPDATA = 0x01, 0x02, 0x04
MAGIC, VERSION = 0xBEEFC0DE, 1
PAGEHDR = 16
INVALID = (1 << 64) - 1


class LmdbFormatError(ValueError):
    pass


class Environment:
    """`lmdb.open(path, readonly=True, lock=False)` for reading: a memory map of `<path>/data.mdb` (or of `path` itself when
    it is a file: the `subdir=False` form)."""

    def __init__(self, path, **_ignored):
        self.path = path
        f = os.path.join(path, "data.mdb") if os.path.isdir(path) else path
        self._fh = io.open(f, "rb")  # (this module defines its own `open`, as lmdb does)
        size = os.fstat(self._fh.fileno()).st_size
        if size < 2 * 512:
            raise LmdbFormatError(f"{f}: too small to hold two meta pages")
        self._mm = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        m0 = self._meta(0, None)
        self.psize = m0["psize"]
        if self.psize < 512 or self.psize & (self.psize - 1) or size < 2 * self.psize:
            raise LmdbFormatError(f"{f}: bad page size {self.psize}")
        m1 = self._meta(1, self.psize)
        self.meta = m1 if m1["txnid"] > m0["txnid"] else m0

    def _meta(self, which, psize):
        off = 0 if which == 0 else psize
        pgno, _pad, flags, _lo, _up = struct.unpack_from("<QHHHH", self._mm, off)
        if not flags & P_META:
            raise LmdbFormatError(f"page {which} is not a meta page (flags {flags:#x})")
        magic, version, _addr, mapsize = struct.unpack_from("<IIQQ", self._mm, off + PAGEHDR)
        if magic != MAGIC:
            raise LmdbFormatError(f"bad magic {magic:#x}")
        if version != VERSION:
            raise LmdbFormatError(f"unsupported data version {version}")
        dbs = []
        for i in range(2):
            pad, dflags, depth, branch, leaf, over, entries, root = struct.unpack_from("<IHHQQQQQ", self._mm, off + PAGEHDR + 24 + 48 * i)
            dbs.append(dict(pad=pad, flags=dflags, depth=depth, branch_pages=branch, leaf_pages=leaf, overflow_pages=over,
                            entries=entries, root=root))
        last_pg, txnid = struct.unpack_from("<QQ", self._mm, off + PAGEHDR + 24 + 96)
        return dict(psize=dbs[0]["pad"], mapsize=mapsize, free=dbs[0], main=dbs[1], last_pg=last_pg, txnid=txnid)

    def begin(self, write=False, **_ignored):
        if write:
            raise NotImplementedError("doc2tex_amd.lmdb_read is read-only")
        return Transaction(self)

    def stat(self):
        m = self.meta["main"]
        return dict(psize=self.psize, depth=m["depth"], branch_pages=m["branch_pages"], leaf_pages=m["leaf_pages"],
                    overflow_pages=m["overflow_pages"], entries=m["entries"])

    def close(self):
        if self._mm is not None:
            self._mm.close()
            self._fh.close()
            self._mm = None

    def __bool__(self):  # the reference tests `if not self.env` (lmdb_dataset.py:26)
        return self._mm is not None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def open(path, **kwargs):  # noqa: A001 - the name the reference calls (lmdb.open)
    return Environment(path, **kwargs)


class Transaction:
    def __init__(self, env):
        self.env = env
        self._mm = env._mm
        self._ps = env.psize
        self._root = env.meta["main"]["root"]
        self._last = env.meta["last_pg"]

    # -- pages ------------------------------------------------------------
    def _page(self, pgno):
        if pgno > self._last:
            raise LmdbFormatError(f"page {pgno} beyond the last page {self._last}")
        off = pgno * self._ps
        got, _pad, flags, lower, upper = struct.unpack_from("<QHHHH", self._mm, off)
        if got != pgno:
            raise LmdbFormatError(f"page {pgno} carries number {got}")
        return off, flags, (lower - PAGEHDR) >> 1

    def _node(self, off, i):
        ptr = struct.unpack_from("<H", self._mm, off + PAGEHDR + 2 * i)[0]
        lo, hi, flags, ksize = struct.unpack_from("<HHHH", self._mm, off + ptr)
        return off + ptr, lo, hi, flags, ksize

    def _key(self, noff, ksize):
        return bytes(self._mm[noff + 8:noff + 8 + ksize])

    def _data(self, noff, lo, hi, flags, ksize):
        if flags & (F_SUBDATA | F_DUPDATA):
            raise NotImplementedError("sub-databases / duplicate-sorted data are not read")
        size = lo | (hi << 16)
        d = noff + 8 + ksize
        if flags & F_BIGDATA:
            pg = struct.unpack_from("<Q", self._mm, d)[0]
            off = pg * self._ps
            got, _pad, pflags, pages = struct.unpack_from("<QHHI", self._mm, off)
            if got != pg or not pflags & P_OVERFLOW or PAGEHDR + size > pages * self._ps:
                raise LmdbFormatError(f"bad overflow page {pg}")
            return bytes(self._mm[off + PAGEHDR:off + PAGEHDR + size])
        return bytes(self._mm[d:d + size])

    # -- lookups ----------------------------------------------------------
    def get(self, key, default=None):
        key = bytes(key)
        if self._root == INVALID:
            return default
        pgno = self._root
        for _ in range(64):  # a tree deeper than this is a cycle
            off, flags, n = self._page(pgno)
            if flags & P_BRANCH:
                lo_i, hi_i = 1, n - 1  # node 0 holds the empty (lowest) key: the last node whose key <= `key`
                child = 0
                while lo_i <= hi_i:
                    mid = (lo_i + hi_i) >> 1
                    noff, _l, _h, _f, ks = self._node(off, mid)
                    if self._key(noff, ks) <= key:
                        child = mid
                        lo_i = mid + 1
                    else:
                        hi_i = mid - 1
                _noff, l, h, f, _ks = self._node(off, child)
                pgno = l | (h << 16) | (f << 32)
            elif flags & P_LEAF:
                lo_i, hi_i = 0, n - 1
                while lo_i <= hi_i:
                    mid = (lo_i + hi_i) >> 1
                    noff, l, h, f, ks = self._node(off, mid)
                    k = self._key(noff, ks)
                    if k == key:
                        return self._data(noff, l, h, f, ks)
                    if k < key:
                        lo_i = mid + 1
                    else:
                        hi_i = mid - 1
                return default
            else:
                raise LmdbFormatError(f"page {pgno}: neither branch nor leaf (flags {flags:#x})")
        raise LmdbFormatError("tree deeper than 64 levels")

    def cursor(self):
        return self.items()

    def items(self):
        """(key, value) pairs in key order (depth-first over the tree)."""
        if self._root == INVALID:
            return
        stack = [self._root]
        while stack:
            off, flags, n = self._page(stack.pop())
            if flags & P_BRANCH:
                kids = []
                for i in range(n):
                    _noff, l, h, f, _ks = self._node(off, i)
                    kids.append(l | (h << 16) | (f << 32))
                stack.extend(reversed(kids))
            else:
                for i in range(n):
                    noff, l, h, f, ks = self._node(off, i)
                    yield self._key(noff, ks), self._data(noff, l, h, f, ks)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False
