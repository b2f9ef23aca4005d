"""LinearAEResidual / LinearAEResidualLeaky: residual bottleneck autoencoder that outputs absolute locations + rotations
(reference modules/movements/linear_ae/linear_ae_residual.py:9-111, linear_ae_residual_leaky.py:5-17).

Per-frame MLP 52 -> 256 -> 128 -> 64 -> 32 (BatchNorm1d + activation + Dropout(0.5) after every layer but the first), a
parallel 52 -> 32 shortcut added at the bottleneck, decoder 32 -> 64 -> 128 -> 256 -> 26*9; output = (locations (B,T,J,3),
rotation_6d_to_matrix of the other six features). Module names (``_encoder``, ``_residual_bottleneck``, ``_decoder``)
follow the reference so that its checkpoints load unchanged. BatchNorm / Dropout make this a library-op model (SURVEY.md
section 8f rank 4): it plugs into the HIP pose head through the ``absolute_loc_rot`` output type; Adam(1e-4) as in the reference.
"""
import torch
from torch import nn

from pedestrians_video_2_carla_amd.modules.flow.output_types import MovementsModelOutputType
from pedestrians_video_2_carla_amd.modules.movements.movements import MovementsModel
from pedestrians_video_2_carla_amd.transforms.rotation_conversions import rotation_6d_to_matrix


class LinearAEResidual(MovementsModel):
    def __init__(self, linear_size=256, activation_cls=nn.ReLU, **kwargs):
        super().__init__(**kwargs)
        self._input_size = len(self.input_nodes) * 2
        self._output_nodes_len, self._output_features = len(self.output_nodes), 9
        self._output_size = self._output_nodes_len * self._output_features

        def block(i, o, drop=True):
            return [nn.Linear(i, o), nn.BatchNorm1d(o), activation_cls()] + ([nn.Dropout(0.5)] if drop else [])

        s = linear_size
        self._encoder = nn.Sequential(nn.Linear(self._input_size, s), *block(s, s // 2), *block(s // 2, s // 4),
                                      *block(s // 4, s // 8))
        self._residual_bottleneck = nn.Sequential(*block(self._input_size, s // 8, drop=False))
        self._decoder = nn.Sequential(*block(s // 8, s // 4), *block(s // 4, s // 2), nn.Linear(s // 2, s),
                                      nn.Linear(s, self._output_size))
        self._hparams.update({'linear_size': linear_size})
        self.apply(self.init_weights)

    @property
    def output_type(self) -> MovementsModelOutputType:
        return MovementsModelOutputType.absolute_loc_rot

    @staticmethod
    def init_weights(m):
        if type(m) == nn.Linear:
            torch.nn.init.kaiming_normal_(m.weight)

    @staticmethod
    def add_model_specific_args(parent_parser):
        parent_parser = MovementsModel.add_model_specific_args(parent_parser)
        group = parent_parser.add_argument_group('LinearAEResidual Lightning Module')
        group.add_argument('--linear_size', default=256, type=int)
        return parent_parser

    def forward(self, x, *args, **kwargs):
        original_shape = x.shape
        x = x.reshape(-1, self._input_size)
        bottleneck = self._encoder(x) + self._residual_bottleneck(x)
        x = self._decoder(bottleneck).view(*original_shape[0:2], self._output_nodes_len, self._output_features)
        return x[..., :3], rotation_6d_to_matrix(x[..., 3:])

    def configure_optimizers(self):
        return {'optimizer': torch.optim.Adam(self.parameters(), lr=1e-4)}


class LinearAEResidualLeaky(LinearAEResidual):
    def __init__(self, **kwargs):
        super().__init__(**{**kwargs, 'activation_cls': nn.LeakyReLU})
