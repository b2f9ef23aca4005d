"""Seq2SeqResidualA / B / C: Seq2SeqEmbeddings with a residual connection around the decoder step
(reference modules/movements/seq2seq/seq2seq_residual_a.py:6-31, _residual_b.py:6-31, _residual_c.py:8-44).

They override ``_decode_frame`` only; the encoder keeps the fused path (folded embeddings + K7b recurrence), the decoder
runs frame by frame through the fused LSTM layer op (the one-launch decoder loop K7c implements the plain feedback only).
Teacher forcing: the reference writes ``input[force_indices] = target (+/x) force_input`` with a boolean mask; here the
same selection is a ``torch.where`` (no host sync). SURVEY.md section 8f rank 4.
"""
from typing import Tuple

import torch
from torch import Tensor

from pedestrians_video_2_carla_amd.transforms.rotation_conversions import matrix_to_rotation_6d, rotation_6d_to_matrix

from .seq2seq_embeddings import Seq2SeqEmbeddings


class Seq2SeqResidualA(Seq2SeqEmbeddings):
    """Version A: the residual sum is both the next input and the returned output."""

    def _decode_frame(self, hidden, cell, step_in, needs_forcing, force_indices, target) -> Tuple[Tensor, Tensor]:
        output, _, _ = self.decoder(step_in, hidden, cell)
        residual_output = output + step_in
        nxt = residual_output
        if needs_forcing:
            nxt = torch.where(force_indices.unsqueeze(-1), target + step_in, residual_output)
            residual_output = nxt         # the reference forces in place: the returned tensor aliases the next input
        return nxt, residual_output


class Seq2SeqResidualB(Seq2SeqEmbeddings):
    """Version B: the residual sum feeds the next frame, the "pure" decoder output is returned."""

    def _decode_frame(self, hidden, cell, step_in, needs_forcing, force_indices, target) -> Tuple[Tensor, Tensor]:
        output, _, _ = self.decoder(step_in, hidden, cell)
        nxt = output + step_in
        if needs_forcing:
            nxt = torch.where(force_indices.unsqueeze(-1), target + step_in, nxt)
        return nxt, output


class Seq2SeqResidualC(Seq2SeqEmbeddings):
    """Version C: like B, but the residual composes rotations, R(next) = R(input) @ R(output) in the 6-D representation
    (the reference marks it as work in progress: the all-zero <sos> input has no valid rotation)."""

    @staticmethod
    def _compose(a: Tensor, b: Tensor) -> Tensor:
        shape = a.shape
        prod = torch.bmm(rotation_6d_to_matrix(a.reshape(-1, 6)), rotation_6d_to_matrix(b.reshape(-1, 6)))
        return matrix_to_rotation_6d(prod).reshape(shape)

    def _decode_frame(self, hidden, cell, step_in, needs_forcing, force_indices, target) -> Tuple[Tensor, Tensor]:
        output, _, _ = self.decoder(step_in, hidden, cell)
        nxt = self._compose(step_in, output)
        if needs_forcing:
            nxt = torch.where(force_indices.unsqueeze(-1), self._compose(step_in, target), nxt)
        return nxt, output
