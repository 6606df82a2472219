"""Test-side writer of LMDB's data-file layout (see doc2tex_amd/lmdb_read.py for the layout and its source): builds a
`data.mdb` from a dict so that the reader can be exercised on leaf-only trees, multi-level trees and overflow pages.
Written independently of the reader (top-down page filling as mdb.c does it: pointers grow up from byte 16, nodes grow down
from the end of the page), but from the same published description -- a consistency check, not a parity pin."""
import os
import struct

P_BRANCH, P_LEAF, P_OVERFLOW, P_META = 0x01, 0x02, 0x04, 0x08
F_BIGDATA = 0x01
INVALID = (1 << 64) - 1


def _even(n):
    return (n + 1) & ~1


class _Page:
    def __init__(self, psize, flags):
        self.psize, self.flags, self.nodes = psize, flags, []
        self.used = 0

    def fits(self, node):
        return 16 + 2 * (len(self.nodes) + 1) + self.used + _even(len(node)) <= self.psize

    def add(self, node):
        self.nodes.append(node)
        self.used += _even(len(node))

    def render(self, pgno):
        buf = bytearray(self.psize)
        upper = self.psize
        ptrs = []
        for node in self.nodes:
            upper -= _even(len(node))
            buf[upper:upper + len(node)] = node
            ptrs.append(upper)
        lower = 16 + 2 * len(ptrs)
        assert lower <= upper
        struct.pack_into("<QHHHH", buf, 0, pgno, 0, self.flags, lower, upper)
        for i, p in enumerate(ptrs):
            struct.pack_into("<H", buf, 16 + 2 * i, p)
        return bytes(buf)


def write_lmdb(root, items, psize=4096, subdir=True):
    """items: {bytes: bytes}.  Returns the path of the data file."""
    if subdir:
        os.makedirs(root, exist_ok=True)
        path = os.path.join(root, "data.mdb")
    else:
        path = root
    nodemax = (((psize - 16) // 2) & ~1) - 2
    pages = {}  # pgno -> bytes
    next_pg = [2]

    def alloc(n=1):
        p = next_pg[0]
        next_pg[0] += n
        return p

    leaf_pages, overflow_pages, branch_pages = 0, 0, 0
    level = []  # (first key, pgno) of the pages of the current level
    cur = _Page(psize, P_LEAF)
    first = None

    def flush_leaf():
        nonlocal cur, first, leaf_pages
        if cur.nodes:
            pg = alloc()
            pages[pg] = cur.render(pg)
            level.append((first, pg))
            leaf_pages += 1
        cur, first = _Page(psize, P_LEAF), None

    for k in sorted(items):
        v = items[k]
        if 8 + len(k) + len(v) > nodemax:
            n = -(-(16 + len(v)) // psize)
            pg = alloc(n)
            body = bytearray(n * psize)
            struct.pack_into("<QHHI", body, 0, pg, 0, P_OVERFLOW, n)
            body[16:16 + len(v)] = v
            for j in range(n):
                pages[pg + j] = bytes(body[j * psize:(j + 1) * psize])
            overflow_pages += n
            node = struct.pack("<HHHH", len(v) & 0xFFFF, len(v) >> 16, F_BIGDATA, len(k)) + k + struct.pack("<Q", pg)
        else:
            node = struct.pack("<HHHH", len(v) & 0xFFFF, len(v) >> 16, 0, len(k)) + k + v
        if not cur.fits(node):
            flush_leaf()
        if first is None:
            first = k
        cur.add(node)
    flush_leaf()
    depth = 1 if level else 0
    while len(level) > 1:
        up = []
        cur = _Page(psize, P_BRANCH)
        first = None
        for k, pg in level:
            key = b"" if not cur.nodes else k  # the first node of a branch page carries no key
            node = struct.pack("<HHHH", pg & 0xFFFF, (pg >> 16) & 0xFFFF, (pg >> 32) & 0xFFFF, len(key)) + key
            if not cur.fits(node):
                bp = alloc()
                pages[bp] = cur.render(bp)
                up.append((first, bp))
                branch_pages += 1
                cur, first = _Page(psize, P_BRANCH), None
                node = struct.pack("<HHHH", pg & 0xFFFF, (pg >> 16) & 0xFFFF, (pg >> 32) & 0xFFFF, 0)
            if first is None:
                first = k
            cur.add(node)
        bp = alloc()
        pages[bp] = cur.render(bp)
        up.append((first, bp))
        branch_pages += 1
        level = up
        depth += 1
    root_pg = level[0][1] if level else INVALID
    last_pg = next_pg[0] - 1

    def meta(pgno, txnid, root, entries, depth_, counts):
        buf = bytearray(psize)
        struct.pack_into("<QHHHH", buf, 0, pgno, 0, P_META, 0, 0)
        struct.pack_into("<IIQQ", buf, 16, 0xBEEFC0DE, 1, 0, 1 << 30)
        struct.pack_into("<IHHQQQQQ", buf, 16 + 24, psize, 0, 0, 0, 0, 0, 0, INVALID)  # free DB: empty; pad = page size
        struct.pack_into("<IHHQQQQQ", buf, 16 + 24 + 48, 0, 0, depth_, counts[0], counts[1], counts[2], entries, root)
        struct.pack_into("<QQ", buf, 16 + 24 + 96, last_pg if root != INVALID else 1, txnid)
        return bytes(buf)

    with open(path, "wb") as f:
        f.write(meta(0, 0, INVALID, 0, 0, (0, 0, 0)))  # the older meta page: an empty environment
        f.write(meta(1, 1, root_pg, len(items), depth, (branch_pages, leaf_pages, overflow_pages)))
        for pg in range(2, last_pg + 1):
            f.write(pages[pg])
    return path
