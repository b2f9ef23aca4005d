"""Seq2Seq movements model: LSTM encoder over the clip, LSTM decoder unrolled frame by frame
(reference modules/movements/seq2seq/seq2seq.py:21-363; after Sutskever et al. 2014, arXiv:1409.3215).

Parity notes:
  * the recurrent layers are ``nn.LSTM`` (the "GRU" of BASELINE.json config 3 does not exist in the reference);
  * ``_decode_frame`` does not carry the new hidden/cell state forward: every decoded frame starts from the ENCODER
    state and only the previous output is fed back (seq2seq.py:272-288). Kept, checkpoints depend on it.
Module / parameter names (``encoder.rnn``, ``decoder.rnn``, ``decoder.fc_out``) match the reference state_dict.
"""
import warnings
from enum import Enum
from typing import Dict, Iterable, Tuple, Union

import torch
from torch import Tensor, nn

from pedestrians_video_2_carla_amd.modules.flow.output_types import MovementsModelOutputType
from pedestrians_video_2_carla_amd.modules.movements.movements import MovementsModel, MovementsModelOutputTypeMixin
from pedestrians_video_2_carla_amd.transforms.rotation_conversions import matrix_to_rotation_6d


class TeacherMode(Enum):
    no_force = 0
    clip_force = 1
    frames_force = 2


def _stack(in_size, hid_dim, n_layers, dropout, bidirectional):
    if isinstance(hid_dim, int):
        return nn.LSTM(in_size, hid_dim, num_layers=n_layers, dropout=dropout, bidirectional=bidirectional)
    sizes = [in_size] + list(hid_dim)
    return nn.Sequential(*[nn.LSTM(a, b, num_layers=1, dropout=dropout, bidirectional=bidirectional)
                           for a, b in zip(sizes[:-1], sizes[1:])])


def _fused_ok(rnn, x) -> bool:
    """The HIP recurrence covers one nn.LSTM stack (uni- or bidirectional), fp32 on the GPU, hidden size 16 / 32 / 48 / 64 / 96 / 128
    (128: the reference's configs/compare/carla-recorded_autoencoder_tests.yaml:38). Per-layer hidden sizes (an nn.Sequential of LSTMs, which the reference cannot run either: seq2seq.py:38-45 hands the
    first layer's tuple to the second) and CPU parity runs take nn.LSTM."""
    ok = (isinstance(rnn, nn.LSTM) and rnn.proj_size == 0 and x.is_cuda
          and x.dtype == torch.float32 and rnn.hidden_size in (16, 32, 48, 64, 96, 128))
    if not ok and x.is_cuda and x.dtype == torch.float32 and isinstance(rnn, nn.LSTM):
        _warn_fallback(rnn)
    return ok


_WARNED = set()


def _warn_fallback(rnn):
    """Once per shape: an fp32 LSTM stack on the GPU that the HIP recurrence does not cover runs through the framework's RNN
    (MIOpen: about 15x slower per step at cfg3's sizes) -- e.g. the reference's own ``hidden_size: 128`` configs
    (configs/compare/carla-recorded_autoencoder_tests.yaml:38). Not silent."""
    key = (rnn.hidden_size, rnn.proj_size)
    if key in _WARNED:
        return
    _WARNED.add(key)
    warnings.warn(f'Seq2Seq: nn.LSTM(hidden_size={rnn.hidden_size}, proj_size={rnn.proj_size}) is outside the HIP recurrence '
                  f'(hidden sizes 16 / 32 / 48 / 64 / 96 / 128, no projection): this stack runs on the framework RNN path, roughly an order of '
                  f'magnitude slower per step', RuntimeWarning, stacklevel=3)


def _run_stack(rnn: nn.LSTM, x: Tensor, hidden: Tensor = None, cell: Tensor = None, input_map=None):
    """nn.LSTM.forward semantics (layer loop, inter-layer dropout in training) on the fused layer op.
    x (T,B,I); hidden / cell (num_layers * D,B,H) or None, D = 2 for a bidirectional stack (state order l0, l0_reverse, l1,
    ...; the reverse direction is the same recurrence over the time-flipped sequence, the next layer reads [forward | reverse]).
    Returns (out (T,B,D*H), hidden, cell).
    ``input_map`` = (weight (4H,I'), bias (4H)) replaces layer 0's input projection (weight_ih_l0, bias_ih_l0 + bias_hh_l0)
    -- used when a linear front end has been folded into it."""
    from pedestrians_video_2_carla_amd import ops
    hs, cs = [], []
    dirs = ('', '_reverse') if rnn.bidirectional else ('',)
    for k in range(rnn.num_layers):
        outs = []
        for di, suffix in enumerate(dirs):
            idx = k * len(dirs) + di
            h0 = hidden[idx].contiguous() if hidden is not None else None  # None = nn.LSTM's zero initial state: the kernel
            c0 = cell[idx].contiguous() if cell is not None else None      # reads nothing and no gradient is produced for it
            b_ih = getattr(rnn, f'bias_ih_l{k}{suffix}', None) if rnn.bias else None
            b_hh = getattr(rnn, f'bias_hh_l{k}{suffix}', None) if rnn.bias else None
            w_ih = getattr(rnn, f'weight_ih_l{k}{suffix}')
            if k == 0 and input_map is not None and di == 0:
                w_ih, b_ih, b_hh = input_map[0], input_map[1], None
            xin = x if di == 0 or x.shape[0] == 1 else x.flip(0)
            o, hT, cT = ops.lstm_layer(xin, h0, c0, w_ih, getattr(rnn, f'weight_hh_l{k}{suffix}'), b_ih, b_hh)
            outs.append(o if di == 0 or x.shape[0] == 1 else o.flip(0))
            hs.append(hT), cs.append(cT)
        x = outs[0] if len(outs) == 1 else torch.cat(outs, dim=-1)
        if rnn.dropout > 0 and rnn.training and k < rnn.num_layers - 1:
            x = torch.nn.functional.dropout(x, rnn.dropout, True)
    return x, torch.stack(hs), torch.stack(cs)


class Encoder(nn.Module):
    def __init__(self, hid_dim=64, n_layers=2, dropout=0.2, input_size=26 * 2, bidirectional=True):
        super().__init__()
        self.hid_dim, self.n_layers, self.input_size = hid_dim, n_layers, input_size
        self.rnn = _stack(input_size, hid_dim, n_layers, dropout, bidirectional)

    def forward(self, x):
        x = x.reshape(*x.shape[0:2], self.input_size)
        if _fused_ok(self.rnn, x):
            _, hidden, cell = _run_stack(self.rnn, x)
            return hidden, cell
        _, (hidden, cell) = self.rnn(x)
        return hidden, cell


class Decoder(nn.Module):
    def __init__(self, hid_dim=64, n_layers=2, dropout=0.2, output_size=26 * 6, bidirectional=False):
        super().__init__()
        self.hid_dim, self.n_layers, self.output_size = hid_dim, n_layers, output_size
        self.rnn = _stack(output_size, hid_dim, n_layers, dropout, bidirectional)
        self.fc_out = nn.Linear(hid_dim * 2 if bidirectional else hid_dim, output_size)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, hidden, cell):
        if _fused_ok(self.rnn, x):
            from pedestrians_video_2_carla_amd import ops
            output, hidden, cell = _run_stack(self.rnn, x.unsqueeze(0), hidden, cell)
            return ops.dense(output.squeeze(0), self.fc_out.weight, self.fc_out.bias), hidden, cell      # (K16 / K12)
        output, (hidden, cell) = self.rnn(x.unsqueeze(0), (hidden, cell))
        return self.fc_out(output.squeeze(0)), hidden, cell


class Seq2Seq(MovementsModelOutputTypeMixin, MovementsModel):
    def __init__(self,
                 hidden_size: Union[int, Iterable[int]] = 64,
                 num_layers: int = 2,
                 p_dropout: float = 0.2,
                 teacher_mode: TeacherMode = TeacherMode.no_force,
                 teacher_force_ratio: float = 0.2,
                 teacher_force_drop: float = 0.02,
                 input_features: int = 2,
                 invert_sequence: bool = False,
                 bidirectional: bool = False,
                 input_size: int = None,
                 **kwargs):
        super().__init__(**kwargs)
        if input_size is not None and input_features is not None:
            warnings.warn('Both input_size and input_features were specified, using input_size.')
        self.input_size = input_size if input_size is not None else input_features * len(self.input_nodes)
        self.output_size = self.output_features * len(self.output_nodes)
        self.teacher_mode = teacher_mode if isinstance(teacher_mode, TeacherMode) else TeacherMode[teacher_mode]
        forcing = self.teacher_mode != TeacherMode.no_force
        self.teacher_force_ratio = teacher_force_ratio if forcing else 0.0
        self.teacher_force_drop = teacher_force_drop if forcing else 0.0
        if not isinstance(hidden_size, int):
            assert len(hidden_size) == num_layers, 'hidden_size must be an int or a list of num_layers ints'
        self.encoder = Encoder(hid_dim=hidden_size, n_layers=num_layers, dropout=p_dropout,
                               input_size=self.input_size, bidirectional=bidirectional)
        self.decoder = Decoder(hid_dim=hidden_size if isinstance(hidden_size, int) else hidden_size[::-1],
                               n_layers=num_layers, dropout=p_dropout, output_size=self.output_size,
                               bidirectional=bidirectional)
        self.invert_sequence = invert_sequence
        self._hparams.update({
            'hidden_size': hidden_size, 'num_layers': num_layers, 'p_dropout': p_dropout,
            'teacher_mode': self.teacher_mode.name, 'teacher_force_ratio': self.teacher_force_ratio,
            'teacher_force_drop': self.teacher_force_drop, 'invert_sequence': self.invert_sequence,
            'bidirectional': bidirectional,
        })

    @property
    def needs_targets(self) -> bool:
        return self.teacher_mode != TeacherMode.no_force

    @staticmethod
    def add_model_specific_args(parent_parser):
        parent_parser = MovementsModel.add_model_specific_args(parent_parser)
        group = parent_parser.add_argument_group('Seq2Seq Movements Module')
        MovementsModelOutputTypeMixin.add_cli_args(group)
        group.add_argument('--num_layers', default=2, type=int)
        group.add_argument('--hidden_size', default=64, type=int)
        group.add_argument('--p_dropout', default=0.2, type=float)
        group.add_argument('--teacher_mode', default=TeacherMode.no_force, choices=list(TeacherMode),
                           type=TeacherMode.__getitem__)
        group.add_argument('--teacher_force_ratio', default=0.2, type=float)
        group.add_argument('--teacher_force_drop', default=0.02, type=float)
        group.add_argument('--invert_sequence', default=False, type=lambda v: str(v).lower() in ('1', 'true'))
        group.add_argument('--bidirectional', default=False, type=lambda v: str(v).lower() in ('1', 'true'))
        return parent_parser

    def forward(self, x: Tensor, targets: Dict[str, Tensor] = None, *args, **kwargs) -> Tensor:
        original_shape = x.shape
        batch_size, clip_length = original_shape[:2]
        hidden, cell = self._encode(x)
        needs_forcing, forced, force_idx = self._teacher_forcing(targets)
        if type(self)._decode_frame is Seq2Seq._decode_frame and self._decoder_loop_fusable(x):
            # K7c: the T decoder steps (frozen encoder state, output fed back, forced frames replaced in the launch) are ONE HIP
            # launch
            out = self._fused_decoder(hidden, cell, clip_length, force_idx if needs_forcing else None,
                                      forced if needs_forcing else None)
            return self._format_output(original_shape, out, batch_first=True)
        step_in = torch.zeros((batch_size, self.decoder.output_size), device=x.device, dtype=x.dtype)     # <sos>
        outputs = []
        for t in range(clip_length):
            step_in, out = self._decode_frame(hidden, cell, step_in, needs_forcing,
                                              force_idx[t] if needs_forcing else None,
                                              forced[t] if needs_forcing else None)
            outputs.append(out)
        return self._format_output(original_shape, torch.stack(outputs, 0))

    def _decode_frame(self, hidden: Tensor, cell: Tensor, step_in: Tensor, needs_forcing: bool, force_indices: Tensor,
                      target: Tensor) -> Tuple[Tensor, Tensor]:
        """One decoded frame -> (next input, output) (reference seq2seq.py:272-288; the hook the residual variants override).
        NB: (hidden, cell) are the encoder's for every frame -- see module docstring. ``force_indices`` (B,) bool and
        ``target`` (B,O) replace the reference's boolean-mask writes by ``torch.where`` (no host sync)."""
        out, _, _ = self.decoder(step_in, hidden, cell)
        if needs_forcing:
            # the reference writes the forced rows INTO the decoder output (``input = output; input[idx] = target``, one
            # tensor under two names, seq2seq.py:283-288): the forced values are also what ``outputs[t]`` receives, so those
            # rows carry zero loss and zero gradient. Same here: one tensor is both the next input and the frame's output.
            out = torch.where(force_indices.unsqueeze(-1), target, out)
        return out, out

    def _encode(self, x: Tensor) -> Tuple[Tensor, Tensor]:
        from pedestrians_video_2_carla_amd import ops
        rnn = self.encoder.rnn
        if (type(self)._format_input is Seq2Seq._format_input and isinstance(rnn, nn.LSTM) and _fused_ok(rnn, x)
                and ops.encoder_stack_supported(rnn, x, self.invert_sequence)):
            # the 2-layer encoder over the batch-first rows as one explicit launch sequence (ops.EncoderStackFunction)
            return ops.encoder_stack(x.reshape(*x.shape[:2], self.encoder.input_size), rnn,
                                     drop_state=self._kernel_drop_state(x.device) if (rnn.dropout > 0 and rnn.training) else None)
        return self.encoder(self._format_input(x))

    def _decoder_loop_fusable(self, x: Tensor) -> bool:
        from pedestrians_video_2_carla_amd import ops
        rnn = self.decoder.rnn
        return (_fused_ok(rnn, x) and rnn.bias and not rnn.bidirectional and isinstance(self.decoder.fc_out, nn.Linear)
                and ops.decoder_loop_supported(rnn.hidden_size, rnn.num_layers, self.decoder.output_size))

    def _kernel_drop_state(self, device) -> Tensor:
        """The state of this model's in-kernel dropout streams (ops.dropout_state; site 0 = encoder, 1 = decoder), or None when
        the framework's dropout is asked for (P2C_TORCH_DROPOUT=1)."""
        from pedestrians_video_2_carla_amd import ops
        if not ops.kernel_dropout_enabled():
            return None
        st = getattr(self, '_drop_state', None)
        if st is None or st.device != device:
            st = self._drop_state = ops.dropout_state(device)
        return st

    def _fused_decoder(self, hidden: Tensor, cell: Tensor, clip_length: int, force_idx: Tensor = None,
                       forced: Tensor = None) -> Tensor:
        from pedestrians_video_2_carla_amd import ops
        rnn, fc = self.decoder.rnn, self.decoder.fc_out
        # (the decoder state is the encoder's for every frame: its recurrent terms k_l = b_ih_l + b_hh_l + W_hh_l hidden_l are
        # per-clip constants, formed inside the launch)
        drop = None
        state = self._kernel_drop_state(hidden.device) if (rnn.dropout > 0 and rnn.training) else None
        if state is not None:                      # nn.LSTM's inter-layer dropout, its mask drawn inside the decoder kernels
            drop = (state, float(rnn.dropout), 1)
        elif rnn.dropout > 0 and rnn.training:     # ... or one mask tensor for all frames from the framework's generator:
            shape = (clip_length, hidden.shape[1], rnn.hidden_size)       # dropout(ones) = mask / keep in ONE launch
            ones = getattr(self, '_drop_ones', None)
            if ones is None or ones.shape != shape or ones.device != hidden.device or ones.dtype != hidden.dtype:
                ones = self._drop_ones = torch.ones(shape, device=hidden.device, dtype=hidden.dtype)
            drop = torch.nn.functional.dropout(ones, rnn.dropout, True)
        return ops.decoder_stack(hidden, cell, rnn, fc, clip_length, drop, force_idx, forced)            # (B,T,O)

    def _format_output(self, original_shape, outputs, batch_first: bool = False):
        if not batch_first:
            outputs = outputs.permute(1, 0, 2)
        outputs = outputs.reshape(*original_shape[:2], len(self.output_nodes), self.output_features)
        return super()._format_output(outputs)

    def _format_input(self, x: Tensor) -> Tensor:
        x = x.permute(1, 0, *range(2, x.dim()))          # sequence first
        return x.flip(0) if self.invert_sequence else x

    def _teacher_forcing(self, targets) -> Tuple[bool, Tensor, Tensor]:
        needs = (self.training and self.teacher_mode != TeacherMode.no_force and targets is not None
                 and self.teacher_force_ratio > 0)
        if not needs:
            return False, None, None
        if self.output_type == MovementsModelOutputType.pose_changes:
            target = matrix_to_rotation_6d(targets['pose_changes'])
        else:
            target = targets['projection_2d_transformed']
        B, T = target.shape[:2]
        target = target.permute(1, 0, *range(2, target.dim())).reshape(T, B, self.decoder.output_size)
        if self.teacher_mode == TeacherMode.clip_force:
            idx = (torch.rand((1, B), device=target.device) < self.teacher_force_ratio).repeat(T, 1)
        else:
            idx = torch.rand((T, B), device=target.device) < self.teacher_force_ratio
        return True, target, idx

    def training_epoch_end(self, *args, **kwargs) -> Dict[str, float]:
        if self.teacher_mode == TeacherMode.no_force:
            return {}
        current = self.teacher_force_ratio
        self.teacher_force_ratio = max(0.0, current - self.teacher_force_drop) if current > self.teacher_force_drop else 0
        return {'teacher_force_ratio/{}'.format(self.teacher_mode.name): current}
