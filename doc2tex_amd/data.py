"""Input-side helpers of the training / evaluation loops (the GPU-facing part).

`PrefetchLoader` mirrors doc2tex/data/prefetcher.py:6-53: it wraps any iterable of `(input, target, names)` batches and
moves `input` to the device on a side stream while the previous batch is being consumed, with the reference's exact
hand-over (`current_stream().wait_stream(side)` before a batch is yielded) and the same `__len__` / `.sampler` /
`.dataset` pass-throughs.  Two MI355X-side additions that do not change what the consumer sees:
  * the host tensor is staged through pinned memory (a pageable `non_blocking=True` copy is synchronous on ROCm), from a
    small ring of pinned buffers so that staging batch i+1 never overwrites batch i while its copy is still in flight;
  * the yielded tensor is recorded on the consumer's stream (`record_stream`), so the caching allocator cannot hand its
    memory back while kernels of the consumer still read it.
The LMDB reader behind it in the reference (data/lmdb_dataset.py:45-93) is NOT rebuilt: the `lmdb` module is absent from the
image and nothing could pin a re-implementation of its file format.
"""
import torch


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
