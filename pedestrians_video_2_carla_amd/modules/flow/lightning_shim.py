"""Minimal stand-in for ``pl.LightningModule`` when pytorch_lightning is not installed.

The flows only need: nn.Module behaviour, ``self.log`` / ``self.log_dict``, ``self.trainer`` (for
``trainer.datamodule.transform_callable``, pose_lifting.py:167-170), ``self.device``, ``self.global_step``,
``save_hyperparameters`` / ``hparams``. With Lightning present the real base class is used and nothing here matters.
"""
import torch

try:  # pragma: no cover - Lightning is absent in the build image
    import pytorch_lightning as pl
    LightningModuleBase = pl.LightningModule
    HAVE_LIGHTNING = True
except ImportError:
    HAVE_LIGHTNING = False

    class LightningModuleBase(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.trainer = None
            self._hparams = {}
            self.logged = {}          # name -> last logged value (tensor, not synced to the host)
            self.global_step = 0

        @property
        def hparams(self):
            return self._hparams

        def save_hyperparameters(self, params=None, **kwargs):
            self._hparams.update(params or {})

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device('cpu')

        def log(self, name, value, **kwargs):
            self.logged[name] = value.detach() if isinstance(value, torch.Tensor) else value

        def log_dict(self, values, **kwargs):
            for k, v in values.items():
                self.log(k, v, **kwargs)
