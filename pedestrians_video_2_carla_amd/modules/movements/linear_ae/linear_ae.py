"""LinearAE movements model (reference modules/movements/linear_ae/linear_ae.py:5-59).

Per-frame MLP 2P -> P -> P/2 -> P/4 -> O/4 -> O/2 -> O with ReLU between (P = input joints * 2, O = output joints *
output_features). Attribute names are kept (``__encoder`` / ``__decoder`` inside class ``LinearAE``) so state_dict keys
(``_LinearAE__encoder.0.weight`` ...) match reference checkpoints.
"""
from torch import nn

from pedestrians_video_2_carla_amd.modules.movements.movements import MovementsModel, MovementsModelOutputTypeMixin


def _mlp(sizes, last_activation):
    layers = []
    for i, (a, b) in enumerate(zip(sizes[:-1], sizes[1:])):
        layers.append(nn.Linear(a, b))
        if last_activation or i < len(sizes) - 2:
            layers.append(nn.ReLU())
    return nn.Sequential(*layers)


class LinearAE(MovementsModelOutputTypeMixin, MovementsModel):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.__n_out = len(self.output_nodes)
        self.__in = len(self.input_nodes) * 2                  # (x, y) per joint
        out = self.__n_out * self.output_features
        self.__encoder = _mlp([self.__in, self.__in // 2, self.__in // 4, self.__in // 8], last_activation=True)
        self.__decoder = _mlp([self.__in // 8, out // 4, out // 2, out], last_activation=False)

    @staticmethod
    def add_model_specific_args(parent_parser):
        parent_parser = MovementsModel.add_model_specific_args(parent_parser)
        MovementsModelOutputTypeMixin.add_cli_args(parent_parser.add_argument_group('LinearAE Model'))
        return parent_parser

    def forward(self, x, *args, **kwargs):
        lead = x.shape[0:2]
        h = self.__decoder(self.__encoder(x.view((-1, self.__in))))
        return self._format_output(h.view(*lead, self.__n_out, self.output_features))
