"""The thin slice of ``BaseDataModule`` the train step depends on (reference data/base/base_datamodule.py).

Also here: ``save_predictions`` (the predict-mode writer, reference :560-630, in the common subset format of
``data/base/subset_io.py``) and ``get_dataloader`` (reference :334-359) returning a ``DeviceLoader`` over a stored subset.

Kept: constructor kwargs (data_nodes, input_nodes, clip_length, batch_size, transform), ``transform`` /
``transform_callable`` (``_setup_data_transform`` :202-209, default hips_neck_bbox :276), the batch contract
``(frames (B,T,J,2), targets: Dict[str,Tensor], meta: Dict[str,list])`` and ``node remap``
(base_dataset.py:156-191) as a device op. Out of scope: HDF5 subsets, pandas, md5 settings digests, DataLoader workers.
"""
from typing import Callable, Dict, Optional, Type, Union

import torch

from pedestrians_video_2_carla_amd.data.base.base_transforms import BaseTransforms
from pedestrians_video_2_carla_amd.data.base.skeleton import Skeleton, get_common_indices
from pedestrians_video_2_carla_amd.transforms.pose.normalization import Normalizer
from pedestrians_video_2_carla_amd.transforms.pose.normalization.bbox_extractor import BBoxExtractor
from pedestrians_video_2_carla_amd.transforms.pose.normalization.hips_neck_bbox_fallback_extractor import \
    HipsNeckBBoxFallbackExtractor
from pedestrians_video_2_carla_amd.transforms.pose.normalization.hips_neck_extractor import HipsNeckExtractor


class BaseDataModule(object):
    def __init__(self,
                 data_nodes: Type[Skeleton],
                 input_nodes: Type[Skeleton] = None,
                 clip_length: Optional[int] = 30,
                 batch_size: Optional[int] = 64,
                 transform: Optional[Union[BaseTransforms, str, Callable]] = BaseTransforms.hips_neck_bbox,
                 **kwargs):
        self.clip_length = clip_length
        self.batch_size = batch_size
        self.data_nodes = data_nodes
        self.input_nodes = input_nodes if input_nodes is not None else data_nodes
        self.kwargs = kwargs
        if isinstance(transform, str):
            transform = BaseTransforms[transform.lower()]
        self.transform, self.transform_callable = self._setup_data_transform(transform)

    def _setup_data_transform(self, transform):
        if not isinstance(transform, BaseTransforms):
            return BaseTransforms.user_defined, transform
        table = {
            BaseTransforms.none: None,
            BaseTransforms.hips_neck: Normalizer(HipsNeckExtractor(self.data_nodes)),
            BaseTransforms.bbox: Normalizer(BBoxExtractor(self.data_nodes)),
            BaseTransforms.hips_neck_bbox: Normalizer(HipsNeckBBoxFallbackExtractor(self.data_nodes)),
            BaseTransforms.user_defined: transform,
        }
        return transform, table[transform]

    @property
    def hparams(self) -> Dict:
        return {'data_nodes': self.data_nodes.__name__, 'input_nodes': self.input_nodes.__name__,
                'clip_length': self.clip_length, 'batch_size': self.batch_size, 'transform': self.transform.name}

    # ---- node remap (data_nodes -> input_nodes), zero fill: base_dataset.py:156-191 ---------------------------------
    def map_nodes(self, tensor: torch.Tensor) -> torch.Tensor:
        """(B,T,len(data_nodes),C) -> (B,T,len(input_nodes),C) on device (p2c_remap_nodes)."""
        if self.data_nodes is self.input_nodes:
            return tensor
        from pedestrians_video_2_carla_amd import ops
        input_indices, data_indices = get_common_indices(input_nodes=self.data_nodes, output_nodes=self.input_nodes)
        return ops.remap_nodes(tensor, len(self.input_nodes), list(data_indices), list(input_indices))

    # ---- loader (base_datamodule.py:334-359 ``get_dataloader``) -------------------------------------------------------------
    def get_dataloader(self, subset, device, shuffle: bool = False, drop_last: bool = True, is_training: Optional[bool] = None,
                       rank: int = 0, world_size: int = 1, seed: int = 22742, **pipeline_kwargs):
        """``subset``: path of a stored subset (``subset_io``) or the (projection_2d, targets, meta) host arrays themselves.
        Returns a ``DeviceLoader`` whose batches went through the device input pipeline (K11) with this data module's nodes
        and transform; ``pipeline_kwargs`` = the ``Projection2DMixin`` keywords (noise, missing_joint_probabilities,
        augment_flip, augment_rotate, needs_confidence)."""
        from pedestrians_video_2_carla_amd.data.base.loader import DeviceLoader
        from pedestrians_video_2_carla_amd.data.base.projection_2d_pipeline import DeviceProjection2DPipeline
        from pedestrians_video_2_carla_amd.data.base.subset_io import load_subset
        projection_2d, targets, meta = load_subset(subset) if isinstance(subset, str) else subset
        pipeline = DeviceProjection2DPipeline(self.data_nodes, self.input_nodes, transform=self.transform,
                                              is_training=shuffle if is_training is None else is_training, seed=seed,
                                              **pipeline_kwargs)
        return DeviceLoader(projection_2d, targets, meta, pipeline, self.batch_size, device, shuffle=shuffle,
                            drop_last=drop_last, seed=seed, rank=rank, world_size=world_size)

    # ---- predict-mode writer (base_datamodule.py:560-630 ``save_predictions``) ------------------------------------------------
    def save_predictions(self, run_id: str, outputs, crucial_keys, outputs_key: str, outputs_dir: str,
                         predict_set_name: str = 'predict', prefer_hdf5: bool = True) -> str:
        """Store what ``flow.predict_step`` returned -- an iterable of (sliced_data, batch_meta) -- as a subset the next model
        can train on. Returns the directory. What is saved follows the reference: the tensor under ``outputs_key`` becomes
        ``projection_2d``; every target key plus the flow's crucial keys except ``projection_2d_*`` become ``targets/*``
        (the prediction wins over the target of the same name); meta is concatenated.
        Reference quirk kept: the de-normalisation branch tests ``outputs_key == "projections_2d_transformed"`` (sic, plural:
        base_datamodule.py:599), which no flow ever passes, so normalised predictions are stored as they are."""
        import itertools
        import os

        import numpy as np

        from pedestrians_video_2_carla_amd.data.base.subset_io import save_subset
        from pedestrians_video_2_carla_amd.transforms.pose.normalization.denormalizer import DeNormalizer
        outputs = list(outputs)
        if not outputs:
            raise ValueError('no predictions to save')
        predictions_output_dir = os.path.join(outputs_dir, 'Predictions', run_id)
        os.makedirs(predictions_output_dir, exist_ok=True)
        meta_keys = list(outputs[0][1].keys())
        targets_keys = set(outputs[0][0]['targets'].keys()).union(set(crucial_keys))
        targets_keys = [k for k in targets_keys if not k.startswith('projection_2d_')]
        host = lambda t: t.detach().cpu().numpy()     # noqa: E731
        projections_2d, targets, meta = [], {k: [] for k in targets_keys}, {k: [] for k in meta_keys}
        for sliced_data, batch_meta in outputs:
            if outputs_key == 'projections_2d_transformed':
                projections_2d.append(host(DeNormalizer()(sliced_data[outputs_key][..., :2],
                                                         sliced_data['targets']['projection_2d_scale'],
                                                         sliced_data['targets']['projection_2d_shift'])))
            else:
                projections_2d.append(host(sliced_data[outputs_key]))
            for k in targets_keys:
                v = sliced_data[k] if sliced_data.get(k) is not None else sliced_data['targets'][k]
                targets[k].append(host(v))
            for k in meta_keys:
                v = batch_meta[k]
                meta[k].append(host(v) if isinstance(v, torch.Tensor) else v)
        projections_2d = np.concatenate(projections_2d, axis=0)
        targets = {k: np.concatenate(v, axis=0) for k, v in targets.items()}
        meta = {k: (list(itertools.chain(*v)) if isinstance(v[0], list) else np.concatenate(v, axis=0)) for k, v in meta.items()}
        save_subset(predictions_output_dir, predict_set_name, projections_2d, targets, meta, prefer_hdf5=prefer_hdf5)
        return predictions_output_dir
