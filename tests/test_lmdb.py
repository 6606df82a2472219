"""doc2tex_amd.lmdb_read + data.LMDB_Dataset.  UNPINNED: liblmdb / py-lmdb are absent from the image and the
reference ships no `.mdb` file, so these tests check the reader against an independently written file of the published
layout (tests/lmdb_writer.py) and the dataset class against the reference's documented behaviour -- consistency, not parity."""
import io
import os
import random
import struct

import numpy as np
import pytest

from doc2tex_amd import lmdb_read
from doc2tex_amd.data import LMDB_Dataset
from lmdb_writer import write_lmdb


def _png(h, w, seed):
    from PIL import Image
    rng = np.random.default_rng(seed)
    buf = io.BytesIO()
    Image.fromarray(rng.integers(0, 256, (h, w), dtype=np.uint8), mode="L").save(buf, format="PNG")
    return buf.getvalue()


@pytest.mark.parametrize("n,psize", [(0, 4096), (1, 4096), (40, 4096), (3000, 4096), (500, 16384)])
def test_reader_round_trip(tmp_path, n, psize):
    """empty environment, a single leaf page, a two-level and a three-level tree, inline values and overflow pages (values
    above the node limit of (page - 16) / 2), a second page size"""
    random.seed(n)
    items = {}
    for i in range(1, n + 1):
        items[b"image-%09d" % i] = bytes(random.getrandbits(8) for _ in range(random.choice([0, 10, 700, 2030, 2040, 9000])))
        items[b"label-%09d" % i] = ("\\frac{%d}{x}" % i).encode()
        items[b"name-%09d" % i] = b"img%05d.png" % i
        items[b"width-%09d" % i] = struct.pack("<i", 100 + i)
    if n:
        items[b"num-samples"] = str(n).encode()
    write_lmdb(str(tmp_path), items, psize=psize)
    with lmdb_read.open(str(tmp_path), readonly=True, lock=False, readahead=False, meminit=False) as env:
        assert env and env.stat()["entries"] == len(items) and env.stat()["psize"] == psize
        txn = env.begin(write=False)
        for k, v in items.items():
            assert txn.get(k) == v
        for k in (b"", b"a", b"image-000000000", b"image-%09d" % (n + 1), b"label", b"zzz", b"num-samples-"):
            if k not in items:
                assert txn.get(k) is None and txn.get(k, b"dflt") == b"dflt"
        assert list(txn.items()) == sorted(items.items())
        if n >= 3000:
            assert env.stat()["depth"] >= 3 and env.stat()["overflow_pages"] > 0
        with pytest.raises(NotImplementedError):
            env.begin(write=True)


def test_reader_takes_the_newer_meta_page_and_rejects_garbage(tmp_path):
    p = write_lmdb(str(tmp_path / "a"), {b"k": b"v"})
    raw = bytearray(open(p, "rb").read())
    # swap the transaction ids: page 0 (an empty tree) becomes the current snapshot
    struct.pack_into("<Q", raw, 16 + 24 + 96 + 8, 5)
    os.makedirs(tmp_path / "b")
    open(tmp_path / "b" / "data.mdb", "wb").write(raw)
    assert lmdb_read.open(str(tmp_path / "a")).begin().get(b"k") == b"v"
    assert lmdb_read.open(str(tmp_path / "b")).begin().get(b"k") is None
    raw[16] ^= 0xFF  # magic
    open(tmp_path / "b" / "data.mdb", "wb").write(raw)
    with pytest.raises(lmdb_read.LmdbFormatError):
        lmdb_read.open(str(tmp_path / "b"))
    open(tmp_path / "b" / "data.mdb", "wb").write(b"\0" * 100)
    with pytest.raises(lmdb_read.LmdbFormatError):
        lmdb_read.open(str(tmp_path / "b"))


def test_dataset_mirrors_the_reference_class(tmp_path):
    """data/lmdb_dataset.py:45-93: 1-based keys, (uint8 image, label, (None, None), name); a corrupted image becomes a
    blank imgW x imgH image with the label '[dummy_label]'; `downsample` resizes only when both quotients stay above
    min_dimension."""
    items = {b"num-samples": b"4"}
    shapes = {1: (40, 200), 2: (64, 256), 3: (32, 64), 4: (20, 30)}
    for i, (h, w) in shapes.items():
        items[b"image-%09d" % i] = _png(h, w, i) if i != 3 else b"not an image"
        items[b"label-%09d" % i] = ("x^{%d}" % i).encode()
        items[b"name-%09d" % i] = ("f%d.png" % i).encode()
    write_lmdb(str(tmp_path), items)
    cfg = {"rgb": False, "imgH": 48, "imgW": 160}
    ds = LMDB_Dataset(str(tmp_path), cfg)
    assert len(ds) == 4 and "Number of samples: 4" in repr(ds)
    img, label, size, name = ds[0]
    assert img.dtype == np.uint8 and img.shape == (40, 200) and label == "x^{1}" and size == (None, None) and name == "f1.png"
    from PIL import Image
    assert np.array_equal(img, np.asarray(Image.open(io.BytesIO(items[b"image-000000001"])).convert("L")))
    img, label, _, name = ds[2]
    assert img.shape == (48, 160) and not img.any() and label == "[dummy_label]" and name == "f3.png"
    rgb = LMDB_Dataset(str(tmp_path), {"rgb": True, "imgH": 48, "imgW": 160})
    assert rgb[1][0].shape == (64, 256, 3)
    down = LMDB_Dataset(str(tmp_path), {**cfg, "downsample": 2, "min_dimension": (16, 32)})
    assert down[1][0].shape == (32, 128)   # 64x256 / 2, both above the minimum
    assert down[3][0].shape == (20, 30)    # 20 / 2 < 16: left alone
