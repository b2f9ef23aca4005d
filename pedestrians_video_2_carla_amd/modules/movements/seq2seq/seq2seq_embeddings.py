"""Seq2SeqEmbeddings: one Linear(2, E) per joint in front of the Seq2Seq encoder
(reference modules/movements/seq2seq/seq2seq_embeddings.py:6-78).

The reference loops over 26 ``nn.Linear(2, 64)`` and writes 26 slices. On the GPU ``_format_input`` is ONE grouped HIP
launch (K7a, ``ops.joint_embeddings`` -> ``p2c_embed_fwd/_bwd``) that writes the sequence-first (T,B,J*E) tensor the
encoder LSTM consumes, time-reversed when ``invert_sequence``; with the flat trainer the 26 weights / biases and their
gradients are read and written in place in the flat buffers. Host tensors / non-fp32 (the CPU parity pipeline) take the
batched contraction (T,B,J,2) x (J,E,2) -> (T,B,J,E). Parameters stay in the ``embeddings.{i}.weight/bias`` ModuleList so
reference checkpoints load unchanged.

``fold_embeddings`` (default on): nothing non-linear sits between the embeddings and the encoder's first input projection
(nn.LSTM applies its dropout BETWEEN layers only, seq2seq.py:36-58), so ``W_ih0 . concat_j(W_j x_j + b_j)`` is a
(4H x 2J) map of the raw keypoints: W_eff[:, 2j:2j+2] = W_ih0[:, jE:(j+1)E] W_j, b_eff = b_ih0 + b_hh0 + sum_j W_ih0[:, j] b_j.
The fused encoder path composes that map per step (26 products of 256x64 by 64x2 -- autograd carries the gradients back to
the embedding and LSTM parameters) and runs the encoder on the (T,B,52) keypoints: the (T,B,1664) embedding tensor, its
8192x1664x256 GEMM and the two gradient GEMMs of the same size disappear (cfg3: 1.02 -> see DESIGN.md section 7). Same function;
fp32 rounding differs by the re-association only.
"""
import torch
from torch import nn

from .seq2seq import Seq2Seq


def _block_view(tensors):
    """One strided view (n, *shape) over n equally spaced contiguous views of the same buffer (the trainer's flat
    parameter / gradient buffers hold embeddings.0.weight, embeddings.0.bias, embeddings.1.weight, ... back to back),
    or None."""
    from pedestrians_video_2_carla_amd import ops
    t0 = tensors[0]
    step = ops._uniform_stride(tensors)
    if step is None or any(t.untyped_storage().data_ptr() != t0.untyped_storage().data_ptr() for t in tensors):
        return None
    return torch.as_strided(t0.detach(), (len(tensors),) + tuple(t0.shape), (step,) + tuple(t0.stride()))


class _FoldedInputMap(torch.autograd.Function):
    """(w_eff (G, J*C), b_eff (G)) = W_ih0 composed with the J embeddings, plus the two LSTM biases -- ``p2c_fold_fwd`` /
    ``p2c_fold_bwd`` (K7a'), one launch each way, with the embedding parameters and their gradients addressed as ONE strided
    block each (views of the trainer's flat buffers). The backward adds dW_j / db_j into the gradient block and, inside the
    trainer's ``grad_sinks`` context, d W_ih0 and the two bias gradients straight into their ``.grad``. (As framework ops the
    same map is 6 launches forward and 11 backward of 4-15 us each.)"""

    @staticmethod
    def forward(ctx, w_ih, b_ih, b_hh, anchor, W, b, gW, gb):
        import ctypes
        from pedestrians_video_2_carla_amd import _lib, ops
        G = w_ih.shape[0]
        J, E, C = W.shape
        if W.stride(1) != C or W.stride(2) != 1 or b.stride(1) != 1 or gW.stride() != W.stride() or gb.stride() != b.stride():
            raise RuntimeError('folded input map: the embedding blocks must be (J, E, C) / (J, E) views with dense rows')
        w_ih, b_ih, b_hh = (ops._require_device(t, n) for t, n in ((w_ih, 'weight_ih_l0'), (b_ih, 'bias_ih_l0'), (b_hh, 'bias_hh_l0')))
        w_eff = torch.empty(G, J * C, dtype=torch.float32, device=w_ih.device)
        b_eff = torch.empty(G, dtype=torch.float32, device=w_ih.device)
        with torch.cuda.device(w_ih.device):
            _lib.check(_lib.lib().p2c_fold_fwd(w_ih.data_ptr(), W.data_ptr(), b.data_ptr(), W.stride(0), b.stride(0),
                                                b_ih.data_ptr(), b_hh.data_ptr(), w_eff.data_ptr(), b_eff.data_ptr(),
                                                G, J, E, C, ops._stream()), 'p2c_fold_fwd')
        ctx.save_for_backward(w_ih, b_ih, b_hh)
        ctx.blocks = (W, b, gW, gb)
        return w_eff, b_eff

    @staticmethod
    def backward(ctx, g_eff, g_b):
        from pedestrians_video_2_carla_amd import _lib, ops
        w_ih, b_ih, b_hh = ctx.saved_tensors
        W, b, gW, gb = ctx.blocks
        G = w_ih.shape[0]
        J, E, C = W.shape
        g_eff = torch.zeros(G, J * C, dtype=torch.float32, device=w_ih.device) if g_eff is None else ops._require_device(g_eff, 'grad w_eff')
        g_b = torch.zeros(G, dtype=torch.float32, device=w_ih.device) if g_b is None else ops._require_device(g_b, 'grad b_eff')
        sw, si, sh = ops._sink(w_ih), ops._sink(b_ih), ops._sink(b_hh)
        if sw is not None and not sw.is_contiguous():
            sw = None
        g_w = sw if sw is not None else torch.empty_like(w_ih)
        with torch.cuda.device(w_ih.device):
            _lib.check(_lib.lib().p2c_fold_bwd(w_ih.data_ptr(), W.data_ptr(), b.data_ptr(), W.stride(0), b.stride(0),
                                                g_eff.data_ptr(), g_b.data_ptr(), g_w.data_ptr(), int(sw is not None),
                                                gW.data_ptr(), gb.data_ptr(), ops._ptr(si), ops._ptr(sh),
                                                G, J, E, C, ops._stream()), 'p2c_fold_bwd')
        return (None if sw is not None else g_w, None if si is not None else g_b, None if sh is not None else g_b,
                None, None, None, None, None)


class Seq2SeqEmbeddings(Seq2Seq):
    def __init__(self, single_joint_embeddings_size=64, **kwargs):
        super().__init__(**{**kwargs, 'input_features': single_joint_embeddings_size})
        self.single_joint_embeddings_size = single_joint_embeddings_size
        self.grad_sink = False     # set by the flat trainer: gradients go straight into the flat gradient buffer
        self.fold_embeddings = True
        self.embeddings = nn.ModuleList([nn.Linear(2, single_joint_embeddings_size)
                                         for _ in range(len(self.input_nodes))])
        self._hparams.update({'single_joint_embeddings_size': single_joint_embeddings_size})

    @staticmethod
    def add_model_specific_args(parent_parser):
        parent_parser = Seq2Seq.add_model_specific_args(parent_parser)
        group = parent_parser.add_argument_group('Seq2SeqEmbeddings Movements Module')
        group.add_argument('--single_joint_embeddings_size', default=64, type=int)
        return parent_parser

    def _encode(self, x):
        from .seq2seq import _fused_ok, _run_stack
        rnn = self.encoder.rnn
        if not (self.fold_embeddings and _fused_ok(rnn, x) and rnn.bias and not rnn.bidirectional):
            return super()._encode(x)
        B, T, J = x.shape[:3]
        E, G = self.single_joint_embeddings_size, 4 * rnn.hidden_size
        assert J == len(self.input_nodes) == len(self.embeddings)
        ws, bs = [e.weight for e in self.embeddings], [e.bias for e in self.embeddings]
        blocks = None
        if self.grad_sink and torch.is_grad_enabled() and all(p.grad is not None for p in ws + bs):
            blocks = [_block_view(t) for t in (ws, bs, [p.grad for p in ws], [p.grad for p in bs])]
        if blocks is not None and all(v is not None for v in blocks):
            w_eff, b_eff = _FoldedInputMap.apply(rnn.weight_ih_l0, rnn.bias_ih_l0, rnn.bias_hh_l0, ws[0], *blocks)
        else:                                                                     # any parameter layout: autograd
            weight, bias = torch.stack(ws), torch.stack(bs)                       # (J, E, C), (J, E)
            w_ih = rnn.weight_ih_l0.view(G, J, E)
            w_eff = (w_ih.unsqueeze(-1) * weight.unsqueeze(0)).sum(2).reshape(G, J * weight.shape[-1])
            b_eff = (w_ih * bias.unsqueeze(0)).sum((1, 2)) + rnn.bias_ih_l0 + rnn.bias_hh_l0
        from pedestrians_video_2_carla_amd import ops
        if ops.encoder_stack_supported(rnn, x, self.invert_sequence):             # batch-first rows, one explicit launch sequence
            return ops.encoder_stack(x.reshape(B, T, -1), rnn, input_map=(w_eff, b_eff),
                                     drop_state=self._kernel_drop_state(x.device) if (rnn.dropout > 0 and rnn.training) else None)
        seq = x.permute(1, 0, 2, 3).reshape(T, B, -1)                             # sequence first, raw keypoints
        if self.invert_sequence:
            seq = seq.flip(0)
        _, hidden, cell = _run_stack(rnn, seq.contiguous(), input_map=(w_eff, b_eff))
        return hidden, cell

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
