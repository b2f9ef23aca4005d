"""``BaseModel``: the plugin contract every movements / trajectory model implements.

Mirrors reference modules/flow/base_model.py:10-202 -- same constructor kwargs (``{prefix}_lr``,
``{prefix}_enable_lr_scheduler``, ``{prefix}_scheduler_*``, ``{prefix}_weight_decay``, ``input_nodes``/``data_nodes``),
same ``hparams`` keys, same optimizer factory (AdamW lr 1e-4 -- 5e-2 with a scheduler --, weight_decay 1e-8).
"""
import logging
from enum import Enum
from typing import Dict

import torch
from torch.optim.lr_scheduler import CosineAnnealingWarmRestarts, ReduceLROnPlateau, StepLR

from pedestrians_video_2_carla_amd.data.base.skeleton import get_skeleton_name_by_type, get_skeleton_type_by_name

_SCHEDULER_DEFAULTS = dict(scheduler_type='ReduceLROnPlateau', scheduler_gamma=0.98, scheduler_step_size=1,
                           scheduler_min_lr=1e-8, scheduler_patience=50, scheduler_cooldown=20, weight_decay=1e-8)


class BaseModel(torch.nn.Module):
    def __init__(self, prefix: str, **kwargs):
        super().__init__()
        self._prefix = prefix
        self._hparams = {}

        def opt(name, default=None):
            return kwargs.get(f'{prefix}_{name}', default)

        self.enable_lr_scheduler = opt('enable_lr_scheduler')
        lr = opt('lr')
        self.learning_rate = lr if lr is not None else (5e-2 if self.enable_lr_scheduler else 1e-4)
        self.lr_scheduler_type = opt('scheduler_type', _SCHEDULER_DEFAULTS['scheduler_type'])
        self.lr_scheduler_gamma = opt('scheduler_gamma', _SCHEDULER_DEFAULTS['scheduler_gamma'])
        self.lr_scheduler_step_size = opt('scheduler_step_size', _SCHEDULER_DEFAULTS['scheduler_step_size'])
        self.lr_scheduler_min_lr = opt('scheduler_min_lr', _SCHEDULER_DEFAULTS['scheduler_min_lr'])
        self.lr_scheduler_patience = opt('scheduler_patience', _SCHEDULER_DEFAULTS['scheduler_patience'])
        self.lr_scheduler_cooldown = opt('scheduler_cooldown', _SCHEDULER_DEFAULTS['scheduler_cooldown'])
        self.lr_weight_decay = opt('weight_decay', _SCHEDULER_DEFAULTS['weight_decay'])

        nodes = kwargs.get('input_nodes', None)
        if nodes is None:
            nodes = kwargs.get('data_nodes')
        self.input_nodes = get_skeleton_type_by_name(nodes) if isinstance(nodes, str) else nodes

    # ---- introspection used by the flows ----
    @property
    def hparams(self) -> Dict:
        p = self._prefix
        base = {
            f'{p}_model_name': type(self).__name__,
            f'{p}_output_type': self.output_type.name,
            f'{p}_enable_lr_scheduler': self.enable_lr_scheduler,
            f'{p}_lr': self.learning_rate,
            f'{p}_scheduler_type': self.lr_scheduler_type,
            f'{p}_scheduler_gamma': self.lr_scheduler_gamma,
            f'{p}_scheduler_step_size': self.lr_scheduler_step_size,
            f'{p}_scheduler_min_lr': self.lr_scheduler_min_lr,
            f'{p}_scheduler_patience': self.lr_scheduler_patience,
            f'{p}_scheduler_cooldown': self.lr_scheduler_cooldown,
            f'{p}_weight_decay': self.lr_weight_decay,
            'input_nodes': get_skeleton_name_by_type(self.input_nodes) if self.input_nodes is not None else None,
        }
        try:
            return {**base, **self._hparams}
        except AttributeError as e:  # pragma: no cover - same tolerance as the reference
            logging.getLogger(__name__).warning('AttributeError: %s. Skipping non-base hparams.', e)
            return base

    @property
    def output_type(self) -> Enum:
        raise NotImplementedError()

    needs_targets = property(lambda self: False)
    needs_confidence = property(lambda self: False)
    needs_graph = property(lambda self: False)

    @staticmethod
    def add_model_specific_args(parent_parser, prefix: str):
        group = parent_parser.add_argument_group('Base Model')
        group.add_argument(f'--{prefix}_lr', default=None, type=float)
        group.add_argument(f'--{prefix}_enable_lr_scheduler', default=False, action='store_true')
        group.add_argument(f'--{prefix}_scheduler_type', default='ReduceLROnPlateau', type=str,
                           choices=['ReduceLROnPlateau', 'StepLR', 'CosineAnnealingWarmRestarts'])
        for name, typ in (('scheduler_gamma', float), ('scheduler_step_size', int), ('scheduler_min_lr', float),
                          ('scheduler_patience', int), ('scheduler_cooldown', int), ('weight_decay', float)):
            group.add_argument(f'--{prefix}_{name}', default=_SCHEDULER_DEFAULTS[name], type=typ)
        if 'input_nodes' not in [a.dest for a in group._actions]:
            group.add_argument('--input_nodes', type=get_skeleton_type_by_name, default=None,
                               help='Input nodes for the model (data module output); defaults to data_nodes.')
        return parent_parser

    def configure_optimizers(self):
        optimizer = torch.optim.AdamW(self.parameters(), lr=self.learning_rate, weight_decay=self.lr_weight_decay)
        config = {'optimizer': optimizer}
        if self.enable_lr_scheduler:
            kind = self.lr_scheduler_type
            if kind == 'ReduceLROnPlateau':
                config['lr_scheduler'] = {
                    'scheduler': ReduceLROnPlateau(optimizer, mode='min', min_lr=self.lr_scheduler_min_lr,
                                                   factor=self.lr_scheduler_gamma, patience=self.lr_scheduler_patience,
                                                   cooldown=self.lr_scheduler_cooldown),
                    'interval': 'epoch', 'monitor': 'val_loss/primary'}
            elif kind == 'StepLR':
                config['lr_scheduler'] = {'scheduler': StepLR(optimizer, step_size=self.lr_scheduler_step_size,
                                                              gamma=self.lr_scheduler_gamma)}
            elif kind == 'CosineAnnealingWarmRestarts':
                config['lr_scheduler'] = {'scheduler': CosineAnnealingWarmRestarts(
                    optimizer, T_0=self.lr_scheduler_step_size, eta_min=self.lr_scheduler_min_lr)}
            else:
                raise ValueError('Unknown lr scheduler type: {}'.format(kind))
        return config

    def forward(self, x, *args, **kwargs):
        raise NotImplementedError()
