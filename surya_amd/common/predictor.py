"""Plugin seam of the predictors (behaviour of surya/common/predictor.py:9-57 and surya/common/load.py:8-24)."""
from __future__ import annotations

import gc
from contextlib import contextmanager
from typing import Any, Optional

from ..settings import settings


class ModelLoader:
    def __init__(self, checkpoint: Optional[str] = None):
        self.checkpoint = checkpoint

    def model(self, device=None, dtype=None) -> Any:
        raise NotImplementedError()

    def processor(self, device=None, dtype=None) -> Any:
        raise NotImplementedError()


class BasePredictor:
    model_loader_cls = ModelLoader
    batch_size: Optional[int] = None
    default_batch_sizes = {"cpu": 1, "mps": 1, "cuda": 1}
    disable_tqdm: bool = settings.DISABLE_TQDM
    torch_dtype = None

    def __init__(self, checkpoint: Optional[str] = None, device=None, dtype=None):
        if device is None:
            device = settings.TORCH_DEVICE_MODEL
        if dtype is None:
            dtype = self.torch_dtype
        self.model = None
        self.processor = None
        loader = self.model_loader_cls(checkpoint)
        self.model = loader.model(device, dtype)
        self.processor = loader.processor()

    def to(self, device_dtype=None):
        if not self.model:
            raise ValueError("Model not loaded")
        self.model.to(device_dtype)

    def get_batch_size(self):
        bs = self.batch_size
        if bs is None:
            bs = self.default_batch_sizes["cpu"]
            if settings.TORCH_DEVICE_MODEL in self.default_batch_sizes:
                bs = self.default_batch_sizes[settings.TORCH_DEVICE_MODEL]
        return bs

    def __call__(self, *args, **kwargs):
        raise NotImplementedError()


@contextmanager
def gc_paused():
    """The cyclic garbage collector off for the duration of a call that builds ~10^5-10^6 acyclic result objects (characters,
    polygons). With it on, every few hundred allocations trigger a collection that walks the ever-growing set of live result
    objects: measured 903 -> 383 us per assembled line (2842 lines, ~42 characters each). Reference counting still frees
    every temporary; the collector's previous state is restored on exit.

    On exit everything allocated meanwhile still sits in the YOUNGEST generation, so the first collection after gc.enable()
    would walk all ~10^6 result objects at once (measured 180 ms for 2842 lines, landing in whatever the caller does next).
    gc.freeze() + gc.unfreeze() splices those lists into the permanent generation and back into the OLDEST one in O(1): nothing
    is exempt from collection, the objects are simply where a survivor of three collections would be, and are next visited by
    a full collection at the collector's usual long-lived threshold."""
    was = gc.isenabled() and settings.SURYA_AMD_PAUSE_GC
    if was:
        gc.disable()
    try:
        yield
    finally:
        if was:
            if gc.get_freeze_count() == 0:        # leave an application's own frozen set alone
                gc.freeze()
                gc.unfreeze()
            gc.enable()


def sharded_over_ranks(images, run_local, model_device, group=None, chunk: int = 1):
    """Multi-GPU for predictors whose inputs are independent images (detection pages, layout pages, table crops): the
    images of ONE call are dealt over the ranks of the initialised process group, each rank runs `run_local` on its share and the
    per-image results (small objects) are all-gathered back in the caller's order on every rank. Every rank must pass the same images
    (checked: count, sizes and a pixel probe). `chunk` > 1 deals whole runs of `chunk` consecutive images (the caller's batches) instead
    of single images -- for predictors whose result for an image depends on its batch mates (table recognition appends the columns of
    ALL the batch's tables to every row prompt, surya/table_rec/__init__.py:196-222). Returns None when there is nothing to shard (world
    size 1): the caller runs locally."""
    import zlib
    import numpy as np
    from .. import dist as sdist
    rank, world = sdist.world_info(group)
    if not sdist.collectives_on(group):
        return None
    dev = sdist.collective_device(model_device, group)
    sizes = np.asarray([im.size for im in images], np.int64).reshape(-1, 2)
    probe = b"".join(im.tobytes()[:4096] for im in images[:: max(1, len(images) // 16)])
    sdist.assert_same_inputs([len(images), zlib.crc32(sizes.tobytes()), zlib.crc32(probe)], group, dev)
    n_chunks = (len(images) + chunk - 1) // chunk
    mine = [i for c in sdist.shard_indices(n_chunks, world, rank) for i in range(c * chunk, min((c + 1) * chunk, len(images)))]
    local = run_local([images[i] for i in mine]) if mine else []
    return sdist.gather_objects(local, mine, len(images), group)
