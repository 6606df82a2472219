"""doc2tex_amd.data.PrefetchLoader against the reference's contract (doc2tex/data/prefetcher.py:6-53): same batches in the
same order, `input` on the device, targets / names untouched, pass-through attributes -- on the CPU here (no stream), and on
the GPU with a side stream feeding the recognizer."""
import pytest
import torch

from doc2tex_amd.data import PrefetchLoader


class _Loader:
    sampler, dataset = "the sampler", "the dataset"

    def __init__(self, n, shape=(2, 1, 8, 16)):
        g = torch.Generator().manual_seed(5)
        self.batches = [(torch.rand(shape, generator=g), [f"label{i}a", f"label{i}b"], (f"img{i}a.png", f"img{i}b.png")) for i in range(n)]

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


def test_prefetch_loader_on_cpu_yields_the_same_batches():
    src = _Loader(5)
    pl = PrefetchLoader(src, "cpu")
    out = list(pl)
    assert len(pl) == 5 and pl.sampler == "the sampler" and pl.dataset == "the dataset" and not pl.is_cuda
    assert len(out) == 5
    for (x, t, n), (rx, rt, rn) in zip(out, src.batches):
        assert torch.equal(x, rx) and t is rt and n is rn


def test_prefetch_loader_single_batch_and_exhaustion():
    out = list(PrefetchLoader(_Loader(1), "cpu"))
    assert len(out) == 1 and out[0][1] == ["label0a", "label0b"]


@pytest.mark.gpu
def test_prefetch_loader_feeds_the_recognizer():
    """Batches staged through pinned memory on a side stream arrive intact (bit-identical) and in order while the consumer
    runs the engine on them; every yielded tensor lives on the device."""
    from conftest import engine_model
    from doc2tex_amd import synth
    cfg, m = engine_model("T2", 6)
    imgs = [synth.synth_images(3, 48, 64, seed=40 + i) for i in range(6)]

    class L(_Loader):
        def __init__(self):
            self.batches = [(x, [str(i)] * 3, (f"{i}.png",) * 3) for i, x in enumerate(imgs)]

    text = torch.full((3, 1), 1, dtype=torch.long, device="cuda")
    seen = []
    with torch.no_grad():
        ref = [m(x.cuda(), text, is_train=False)[0].cpu() for x in imgs]
        for x, t, n in PrefetchLoader(L(), "cuda"):
            assert x.is_cuda and x.shape == (3, 1, 48, 64)
            i = int(t[0])
            assert torch.equal(x.cpu(), imgs[i]) and n[0] == f"{i}.png"
            seen.append((i, m(x, text, is_train=False)[0].cpu()))
    assert [i for i, _ in seen] == list(range(6))
    for i, p in seen:
        assert torch.equal(p, ref[i])


def test_lmdb_dataset_shim_names_the_replacement():
    from doc2tex_amd.data import LMDB_Dataset
    with pytest.raises(ImportError, match="PrefetchLoader"):
        LMDB_Dataset("/nonexistent")
