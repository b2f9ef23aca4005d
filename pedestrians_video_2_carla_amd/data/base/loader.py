"""``DeviceLoader``: what stands between a stored subset and the train step.

The reference builds ``torch.utils.data.DataLoader(dataset, batch_size, num_workers=32, pin_memory=bool(gpus), shuffle=train)``
(data/base/base_datamodule.py:334-359); its workers run ``BaseDataset.__getitem__`` per clip on the CPU
(base_dataset.py:206-234: augmentation, deformation, two normalisations, confidence handling, node map) and the default
collate stacks the clips. Here the per-clip chain is ONE device launch per batch (K11, ``DeviceProjection2DPipeline``), so
the host side only has to move raw arrays:

    host arrays (``load_subset``: HDF5 / npz, or any dict of numpy arrays)
      -> rows of the batch gathered into PINNED staging buffers (two sets, used alternately)
      -> asynchronous H2D copies on a copy stream, one batch AHEAD of the consumer
      -> K11 on the consumer's stream -> (frames, targets, meta) with the reference's keys
      -> ``Trainer.train_step`` (graph mode copies it into its static buffers and runs the per-batch hook)

Batch order: ``shuffle`` draws one permutation per epoch from a seeded host generator (DataLoader's RandomSampler);
``drop_last`` (default True: a captured step has one batch shape) drops the ragged tail the reference would keep.
Under data-parallel training rank r of W takes every W-th index of the epoch's order, and every rank gets the SAME number
of clips, as with DistributedSampler: the order is truncated to floor(n / W) * W indices under ``drop_last`` and otherwise
padded with its own head to ceil(n / W) * W -- each step carries a collective, so a rank with one batch fewer would leave
the others waiting in it.
"""
from typing import Dict, Iterable, Iterator, Optional, Tuple

import numpy as np
import torch

from pedestrians_video_2_carla_amd.data.carla import reference as ref


class DeviceLoader:
    def __init__(self, projection_2d: np.ndarray, targets: Dict[str, np.ndarray], meta: Dict[str, Iterable], pipeline,
                 batch_size: int, device, shuffle: bool = False, drop_last: bool = True, seed: int = 22742,
                 rank: int = 0, world_size: int = 1, target_keys: Optional[Iterable[str]] = None):
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('DeviceLoader feeds the HIP input pipeline: it needs a GPU device')
        self.pipeline, self.batch_size, self.shuffle, self.drop_last = pipeline, int(batch_size), shuffle, drop_last
        self.rank, self.world_size, self.seed, self.epoch = rank, world_size, seed, 0
        keys = list(targets.keys()) if target_keys is None else [k for k in target_keys if k in targets]
        self._host = {'projection_2d': torch.from_numpy(np.ascontiguousarray(projection_2d, dtype=np.float32))}
        for k in keys:
            v = np.ascontiguousarray(targets[k])
            self._host['targets/' + k] = torch.from_numpy(v.astype(np.float32) if v.dtype.kind == 'f' else v)
        self.n = self._host['projection_2d'].shape[0]
        if any(t.shape[0] != self.n for t in self._host.values()):
            raise RuntimeError('every array of a subset holds one row per clip')
        self._meta = {k: (list(v) if not isinstance(v, np.ndarray) else v) for k, v in meta.items()}
        if 'age' in self._meta and 'gender' in self._meta:       # per-clip reference skeleton as an index (projection.py:52-71)
            self._host['meta/skel_type'] = ref.skeleton_types_from_meta(
                {'age': self._meta['age'], 'gender': self._meta['gender']}, batch_size=self.n, strict=True).to(torch.int32)
        # two sets of pinned staging buffers + their device twins
        self._pinned = [{k: torch.empty((self.batch_size,) + tuple(v.shape[1:]), dtype=v.dtype).pin_memory()
                         for k, v in self._host.items()} for _ in range(2)]
        self._copy_stream = torch.cuda.Stream(device=self.device)
        self._slot_events = [None, None]

    def _per_rank(self) -> int:
        """Clips per rank and epoch: the same on every rank (torch DistributedSampler's num_samples)."""
        w = max(int(self.world_size), 1)
        return self.n // w if self.drop_last else -(-self.n // w)

    def __len__(self) -> int:
        per_rank = self._per_rank()
        return per_rank // self.batch_size if self.drop_last else -(-per_rank // self.batch_size)

    def _order(self) -> torch.Tensor:
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g)
        else:
            order = torch.arange(self.n)
        total = self._per_rank() * max(int(self.world_size), 1)
        if total > order.numel():                    # pad with the head of the same order (repeated for a tiny subset)
            reps = -(-total // max(order.numel(), 1))
            order = order.repeat(reps)[:total] if order.numel() else order
        else:
            order = order[:total]
        return order[self.rank::self.world_size]

    def _stage(self, idx: torch.Tensor, slot: int):
        """rows idx -> pinned buffers of `slot` -> device tensors (async on the copy stream); returns (tensors, event)."""
        n = idx.numel()
        out = {}
        prev = self._slot_events[slot]
        if prev is not None:
            prev.synchronize()          # the DMA engine has finished reading this pinned set (issued two batches ago)
        with torch.cuda.stream(self._copy_stream):
            for k, src in self._host.items():
                pin = self._pinned[slot][k][:n]
                torch.index_select(src, 0, idx, out=pin)
                out[k] = pin.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        self._slot_events[slot] = ev
        return out, ev, idx

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, Dict[str, torch.Tensor], Dict[str, Iterable]]]:
        order = self._order()
        self.epoch += 1
        chunks = list(order.split(self.batch_size))
        if chunks and self.drop_last and chunks[-1].numel() < self.batch_size:
            chunks.pop()
        staged = self._stage(chunks[0], 0) if chunks else None
        for i in range(len(chunks)):
            cur = staged
            tensors, ev, idx = cur
            # the NEXT batch's rows go to the other pinned set while this one is consumed; a pinned set is reused two batches
            # later, once the event of its previous use has completed (_stage waits for it on the host)
            staged = self._stage(chunks[i + 1], (i + 1) & 1) if i + 1 < len(chunks) else None
            torch.cuda.current_stream(self.device).wait_event(ev)
            for t in tensors.values():
                t.record_stream(torch.cuda.current_stream(self.device))
            raw = tensors.pop('projection_2d')
            targets = {k[8:]: v for k, v in tensors.items() if k.startswith('targets/')}
            sel = idx.tolist()
            meta = {k: ([v[j] for j in sel] if isinstance(v, list) else v[sel]) for k, v in self._meta.items()}
            if 'meta/skel_type' in tensors:
                meta['skel_type'] = tensors['meta/skel_type']
            frames, projection_targets = self.pipeline(raw, targets, meta)
            yield frames, {**targets, **projection_targets}, meta
