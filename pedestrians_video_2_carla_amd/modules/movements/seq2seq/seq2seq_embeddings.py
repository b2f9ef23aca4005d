"""Seq2SeqEmbeddings: one Linear(2, E) per joint in front of the Seq2Seq encoder
(reference modules/movements/seq2seq/seq2seq_embeddings.py:6-78).

The reference loops over 26 ``nn.Linear(2, 64)`` and writes 26 slices. On the GPU ``_format_input`` is ONE grouped HIP
launch (K7a, ``ops.joint_embeddings`` -> ``p2c_embed_fwd/_bwd``) that writes the sequence-first (T,B,J*E) tensor the
encoder LSTM consumes, time-reversed when ``invert_sequence``; with the flat trainer the 26 weights / biases and their
gradients are read and written in place in the flat buffers. Host tensors / non-fp32 (the CPU parity pipeline) take the
batched contraction (T,B,J,2) x (J,E,2) -> (T,B,J,E). Parameters stay in the ``embeddings.{i}.weight/bias`` ModuleList so
reference checkpoints load unchanged.
"""
import torch
from torch import nn

from .seq2seq import Seq2Seq


class Seq2SeqEmbeddings(Seq2Seq):
    def __init__(self, single_joint_embeddings_size=64, **kwargs):
        super().__init__(**{**kwargs, 'input_features': single_joint_embeddings_size})
        self.single_joint_embeddings_size = single_joint_embeddings_size
        self.grad_sink = False     # set by the flat trainer: gradients go straight into the flat gradient buffer
        self.embeddings = nn.ModuleList([nn.Linear(2, single_joint_embeddings_size)
                                         for _ in range(len(self.input_nodes))])
        self._hparams.update({'single_joint_embeddings_size': single_joint_embeddings_size})

    @staticmethod
    def add_model_specific_args(parent_parser):
        parent_parser = Seq2Seq.add_model_specific_args(parent_parser)
        group = parent_parser.add_argument_group('Seq2SeqEmbeddings Movements Module')
        group.add_argument('--single_joint_embeddings_size', default=64, type=int)
        return parent_parser

    def _format_input(self, x):
        joints = x.shape[2]
        assert joints == len(self.input_nodes) == len(self.embeddings)
        E = self.single_joint_embeddings_size
        if x.is_cuda and x.dtype == torch.float32 and E % 4 == 0 and x.shape[-1] <= 4 and not x.requires_grad:
            from pedestrians_video_2_carla_amd import ops
            ws, bs = [e.weight for e in self.embeddings], [e.bias for e in self.embeddings]
            sinks = None
            if self.grad_sink and torch.is_grad_enabled() and all(p.grad is not None for p in ws + bs):
                sinks = [g for pair in zip((w.grad for w in ws), (b.grad for b in bs)) for g in pair]
            emb = ops.joint_embeddings(x.contiguous(), ws, bs, flip=self.invert_sequence, sinks=sinks)
            return emb
        weight = torch.stack([e.weight for e in self.embeddings])        # (J, E, 2)
        bias = torch.stack([e.bias for e in self.embeddings])            # (J, E)
        emb = torch.einsum('btjc,jec->tbje', x, weight) + bias           # sequence first
        return emb.flip(0) if self.invert_sequence else emb
