"""Input-side helpers of the training / evaluation loops (the GPU-facing part).

`PrefetchLoader` mirrors doc2tex/data/prefetcher.py:6-53: it wraps any iterable of `(input, target, names)` batches and
moves `input` to the device on a side stream while the previous batch is being consumed, with the reference's exact
hand-over (`current_stream().wait_stream(side)` before a batch is yielded) and the same `__len__` / `.sampler` /
`.dataset` pass-throughs.  Two MI355X-side additions that do not change what the consumer sees:
  * the host tensor is staged through pinned memory (a pageable `non_blocking=True` copy is synchronous on ROCm), from a
    small ring of pinned buffers so that staging batch i+1 never overwrites batch i while its copy is still in flight;
  * the yielded tensor is recorded on the consumer's stream (`record_stream`), so the caching allocator cannot hand its
    memory back while kernels of the consumer still read it.
`LMDB_Dataset` mirrors doc2tex/data/lmdb_dataset.py:12-102 (same constructor, keys, return tuple and dummy-image rule); it
opens the environment with the real `lmdb` module when that is importable and with doc2tex_amd.lmdb_read otherwise -- a
read-only restatement of LMDB's data-file layout that could NOT be pinned here (no liblmdb, no `.mdb` file: see that
module's header and DESIGN.md section 8).
"""
import io
from functools import cached_property

import numpy as np
import torch
from torch.utils.data import Dataset


def _lmdb_module():
    try:
        import lmdb  # the reference's dependency (envs/requirements.txt:34), authoritative when present
        return lmdb
    except ImportError:
        from . import lmdb_read
        return lmdb_read


class LMDB_Dataset(Dataset):
    """doc2tex/data/lmdb_dataset.py:12-102.  Keys (data/data_const.py:5-12, written by tools/lmdb_builders/
    create_lmdb_dataset.py:72-96): `num-samples`, `image-%09d` (an encoded image file), `label-%09d`, `name-%09d`, counted
    from 1.  `config` is the reference's dataset dict: `rgb`, `imgH`, `imgW`, optional `downsample` + `min_dimension`."""

    N_SAMPLES, IMAGE, PATH, LABEL = "num-samples", "image", "name", "label"

    def __init__(self, root, config):
        self.root = root
        self.config = config
        self.env = _lmdb_module().open(root, max_readers=32, readonly=True, lock=False, readahead=False, meminit=False)
        self.txn = self.env.begin(write=False)

    @cached_property
    def dataset_samples(self):
        return int(self.txn.get(self.N_SAMPLES.encode()))

    @cached_property
    def filtered_index_list(self):
        return [index + 1 for index in range(self.dataset_samples)]

    def _get_new_size(self, index):
        return None, None

    def __len__(self):
        return len(self.filtered_index_list)

    def __getitem__(self, index):
        from PIL import Image
        assert index <= len(self), f"index range error {index} with length of dataset {len(self)}"
        value = self.filtered_index_list[index]
        label = self.txn.get(f"{self.LABEL}-%09d".encode() % value).decode("utf-8")
        imgbuf = self.txn.get(f"{self.IMAGE}-%09d".encode() % value)
        img_name = self.txn.get(f"{self.PATH}-%09d".encode() % value).decode("utf-8")
        buf = io.BytesIO()
        buf.write(imgbuf)
        buf.seek(0)
        try:
            img = Image.open(buf).convert("RGB" if self.config["rgb"] else "L")
        except IOError:
            print(f"Corrupted image for {value}")
            # dummy image and dummy label for a corrupted entry (lmdb_dataset.py:68-75)
            img = Image.new("RGB" if self.config["rgb"] else "L", (self.config["imgW"], self.config["imgH"]))
            label = "[dummy_label]"
        if self.config.get("downsample", None) is not None:
            ori_h, ori_w = img.size[::-1]
            ratio = self.config["downsample"]
            if ori_h / ratio >= self.config["min_dimension"][0] and ori_w / ratio >= self.config["min_dimension"][1]:
                # the reference hands the float quotients to Image.resize (lmdb_dataset.py:84-88): a TypeError on
                # Python >= 3.10 ('float' object cannot be interpreted as an integer); truncated here so the branch runs
                ori_h, ori_w = int(ori_h / ratio), int(ori_w / ratio)
                img = img.resize((ori_w, ori_h), resample=Image.LANCZOS)
        img = np.asarray(img).astype("uint8")
        new_h, new_w = self._get_new_size(index)
        return (img, label, (new_h, new_w), img_name)

    def __repr__(self) -> str:
        return self.__class__.__name__ + ": (" + f"Number of samples: {len(self)}, Data path: {self.root}" + ")"


class PrefetchLoader:
    def __init__(self, loader, device: str):
        self.loader = loader
        self.device = device
        dev = torch.device(device) if not isinstance(device, torch.device) else device
        self.is_cuda = torch.cuda.is_available() and dev.type == "cuda"  # the reference compares with the string "cuda"
        self._pinned = []

    def _stage(self, x, slot):
        """Pinned copy of a host tensor (ring of three buffers); device tensors pass through."""
        if not isinstance(x, torch.Tensor) or x.is_cuda or x.is_pinned():
            return x
        while len(self._pinned) < 3:
            self._pinned.append(None)
        buf = self._pinned[slot % 3]
        if buf is None or buf.shape != x.shape or buf.dtype != x.dtype:
            buf = self._pinned[slot % 3] = torch.empty(x.shape, dtype=x.dtype).pin_memory()
        buf.copy_(x)
        return buf

    def __iter__(self):
        first = True
        input, target, name = None, None, None
        stream = torch.cuda.Stream(device=self.device) if self.is_cuda else None
        slot = 0
        for next_input, next_target, next_names in self.loader:
            if stream is not None:
                with torch.cuda.stream(stream):
                    next_input = self._stage(next_input, slot).to(device=self.device, non_blocking=True)
                slot += 1
            else:
                next_input = next_input.to(device=self.device, non_blocking=True)
            if not first:
                yield input, target, name
            else:
                first = False
            if stream is not None:
                torch.cuda.current_stream(self.device).wait_stream(stream)
                if isinstance(next_input, torch.Tensor) and next_input.is_cuda:
                    next_input.record_stream(torch.cuda.current_stream(self.device))
            input = next_input
            target = next_target
            name = next_names
        yield input, target, name

    def __len__(self):
        return len(self.loader)

    @property
    def sampler(self):
        return self.loader.sampler

    @property
    def dataset(self):
        return self.loader.dataset
