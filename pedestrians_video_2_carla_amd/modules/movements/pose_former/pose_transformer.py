"""``PoseTransformer``: the spatial-temporal transformer the reference's PoseFormer plugin wraps
(modules/movements/pose_former/pose_former.py:6,62-76 binds ``third_party.pose_former.model_poseformer.PoseTransformer``).

The third-party source is an EMPTY git submodule in the reference checkout (.gitmodules:5-8, no commit pinned), so this is
the build's own module written from the published architecture (Zheng et al., "3D Human Pose Estimation with Spatial and
Temporal Transformers", ICCV 2021, section 3 + the public repository's layer list) -- arithmetic parity with the third-party
code is UNPINNED. Parameter names follow the published checkpoint layout so that a state_dict of the
original loads: ``Spatial_patch_to_embedding``, ``Spatial_pos_embed``, ``Temporal_pos_embed``, ``Spatial_blocks.N.*``,
``blocks.N.*`` (``norm1``, ``attn.qkv``, ``attn.proj``, ``norm2``, ``mlp.fc1``, ``mlp.fc2``), ``Spatial_norm``,
``Temporal_norm``, ``weighted_mean``, ``head.0`` / ``head.1``.

    x (B, F, J, C) 2-D keypoints of F = num_frame frames
      per frame : joints are tokens -- Linear(C, E) + learned joint positions -> `depth` pre-norm transformer blocks (E wide)
      per window: frames are tokens of width J*E + learned frame positions -> `depth` blocks -> LayerNorm
      -> a learned weighted mean over the F frames (Conv1d(F, 1, 1)) -> LayerNorm + Linear(J*E, 3 J)
    returns (B, 1, J, 3): the 3-D pose of the CENTRE frame.

Attention: fp32 on the GPU through K14 (``ops.small_attention``, csrc/p2c_attn.hip: one launch each way, a workgroup per 26- or
9-token sequence); under bf16 autocast, with attention dropout, or on the host through
``torch.nn.functional.scaled_dot_product_attention``.
"""
import torch
import torch.nn.functional as F
from torch import nn


def _fused(x) -> bool:
    """fp32 tensors on the GPU outside autocast take the build's own dense kernels (K16 / K12)."""
    return x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled()


def _linear(layer: nn.Linear, x, scale=None, residual=None):
    """``layer(x)`` [``* scale`` per sample ``+ residual``]; fp32 on the GPU it runs through ``ops.dense``: K16 (csrc/p2c_gemm.hip,
    fp32 MFMA with the bias, the stochastic-depth factor and the residual add in its epilogue) forward and for the input
    gradient, K12 (p2c_atb) for the weight + bias gradient. The spatial blocks contract 546 624 rows (cfg5: 2 336 windows x 9
    frames x 26 joints) into 32..96 x 32..64 outputs -- pure streaming; the temporal blocks (21 024 rows, 832 x 2 496 outputs)
    are the step's 2.8 TFLOP."""
    if _fused(x):
        from pedestrians_video_2_carla_amd import ops
        shp = x.shape
        rows = x.reshape(-1, shp[-1])
        res = None if residual is None else residual.reshape(-1, layer.out_features)
        y = ops.dense(rows, layer.weight, layer.bias, scale, rows.shape[0] // shp[0] if scale is not None else 1, res)
        return y.view(*shp[:-1], layer.out_features)
    y = layer(x)
    if scale is not None:
        y = y * scale.view(-1, *([1] * (y.ndim - 1)))
    return y if residual is None else y + residual


def _add_param(x, p):
    """``x + p`` for a learned p broadcast over the batch; fp32 on the GPU through ``ops.add_row_parameter`` (its gradient is a
    K12 column sum: a framework reduction in a captured backward replays wrong on this stack, see ops.py)."""
    if _fused(x):
        from pedestrians_video_2_carla_amd import ops
        return ops.add_row_parameter(x, p)
    return x + p


def _norm(layer: nn.LayerNorm, x):
    """``layer(x)``; fp32 on the GPU through K15 (``ops.layer_norm``, csrc/p2c_norm.hip): G lanes per row instead of a workgroup
    pass per row -- 18 LayerNorms per step over 546 624 x 32 or 21 024 x 832 rows."""
    if (x.is_cuda and x.dtype == torch.float32 and layer.elementwise_affine      # (autocast keeps layer_norm in fp32 as well)
            and layer.bias is not None and len(layer.normalized_shape) == 1):
        from pedestrians_video_2_carla_amd import ops
        if ops.layer_norm_supported(x, layer.normalized_shape[0]):
            return ops.layer_norm(x, layer.weight, layer.bias, layer.eps)
    return layer(x)


class _DropPath(nn.Module):
    """Stochastic depth: in training a sample's residual branch is dropped with probability p (survivors scaled by 1/(1-p))."""

    def __init__(self, p: float = 0.0):
        super().__init__()
        self.p = float(p)

    def forward(self, x):
        f = self.factor(x)
        return x if f is None else x * f.view(-1, *([1] * (x.ndim - 1)))

    def factor(self, x):
        """(samples,) survivor factor 0 or 1 / (1 - p), or None when nothing is dropped: the fused layers apply it in the GEMM
        epilogue that also adds the residual."""
        if self.p == 0.0 or not self.training:
            return None
        keep = 1.0 - self.p
        return x.new_empty(x.shape[0]).bernoulli_(keep).div_(keep)


def _combine(y, scale, residual):
    if scale is not None:
        y = y * scale.view(-1, *([1] * (y.ndim - 1)))
    return y if residual is None else y + residual


class _Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias, qk_scale, attn_drop, proj_drop):
        super().__init__()
        self.num_heads, self.scale = num_heads, qk_scale if qk_scale is not None else (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = attn_drop
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x, scale=None, residual=None):
        """attention(x) [* scale per sample + residual]."""
        B, N, C = x.shape
        qkv = _linear(self.qkv, x).reshape(B, N, 3, self.num_heads, C // self.num_heads)
        if x.is_cuda and (self.attn_drop == 0.0 or not self.training):
            from pedestrians_video_2_carla_amd import ops
            if ops.small_attention_supported(N, self.num_heads, C // self.num_heads):      # K14: one launch each way
                # (fp32 arithmetic also under bf16 autocast: the scores of 4-wide heads gain nothing from bf16 MFMA)
                att = ops.small_attention(qkv.float(), self.scale)
                if self.proj_drop.p == 0.0 or not self.training:       # projection, drop-path factor and residual: one launch
                    return _linear(self.proj, att, scale, residual)
                y = self.proj_drop(_linear(self.proj, att))
                return _combine(y, scale, residual)
        q, k, v = qkv.permute(2, 0, 3, 1, 4)
        out = F.scaled_dot_product_attention(q, k, v, dropout_p=self.attn_drop if self.training else 0.0, scale=self.scale)
        return _combine(self.proj_drop(_linear(self.proj, out.transpose(1, 2).reshape(B, N, C))), scale, residual)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden, drop):
        super().__init__()
        self.fc1, self.act, self.fc2, self.drop = nn.Linear(dim, hidden), nn.GELU(), nn.Linear(hidden, dim), nn.Dropout(drop)

    def forward(self, x, scale=None, residual=None):
        """mlp(x) [* scale per sample + residual]; fp32 on the GPU without dropout: ``ops.mlp_gelu`` -- two K16 launches forward
        (bias + GELU, then bias + factor + residual in the epilogues), two backward (gelu' in the epilogue of fc2's input
        gradient)."""
        if _fused(x) and (self.drop.p == 0.0 or not self.training) and isinstance(self.act, nn.GELU) \
                and self.act.approximate == 'none':
            from pedestrians_video_2_carla_amd import ops
            shp = x.shape
            rows = x.reshape(-1, shp[-1])
            res = None if residual is None else residual.reshape(-1, self.fc2.out_features)
            y = ops.mlp_gelu(rows, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, scale,
                             rows.shape[0] // shp[0] if scale is not None else 1, res)
            return y.view(*shp[:-1], self.fc2.out_features)
        return _combine(self.drop(_linear(self.fc2, self.drop(self.act(_linear(self.fc1, x))))), scale, residual)


class _Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, qkv_bias, qk_scale, drop, attn_drop, drop_path, norm_layer):
        super().__init__()
        self.norm1, self.norm2 = norm_layer(dim), norm_layer(dim)
        self.attn = _Attention(dim, num_heads, qkv_bias, qk_scale, attn_drop, drop)
        self.drop_path = _DropPath(drop_path)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio), drop)

    def _one_node(self, x) -> bool:
        """fp32 on the GPU, no dropout inside the block, every layer with a bias: the whole block is one autograd node."""
        if not _fused(x) or x.ndim != 3:
            return False
        a, m = self.attn, self.mlp
        live = self.training
        if live and (a.attn_drop != 0.0 or a.proj_drop.p != 0.0 or m.drop.p != 0.0):
            return False
        if not (isinstance(m.act, nn.GELU) and m.act.approximate == 'none' and isinstance(self.norm1, nn.LayerNorm)
                and isinstance(self.norm2, nn.LayerNorm) and self.norm1.elementwise_affine and self.norm2.elementwise_affine
                and self.norm1.bias is not None and self.norm2.bias is not None and a.qkv.bias is not None
                and a.proj.bias is not None and m.fc1.bias is not None and m.fc2.bias is not None):
            return False
        from pedestrians_video_2_carla_amd import ops
        return ops.transformer_block_supported(x, a.num_heads)

    def forward(self, x, factors=None):
        """``factors`` = this block's two (samples,) survivor factors when the caller drew them for the whole stack at once
        (``_stack_factors``), else they are drawn here."""
        # (two draws per block and sample, attention first: the order of x + drop_path(attn) ; x + drop_path(mlp))
        f1, f2 = factors if factors is not None else (self.drop_path.factor(x), self.drop_path.factor(x))
        if self._one_node(x):
            from pedestrians_video_2_carla_amd import ops
            return ops.transformer_block(x, f1, f2, self.attn.num_heads, self.attn.scale, self.norm1, self.attn.qkv,
                                         self.attn.proj, self.norm2, self.mlp.fc1, self.mlp.fc2)
        x = self.attn(_norm(self.norm1, x), f1, x)
        return self.mlp(_norm(self.norm2, x), f2, x)


def _stack_factors(blocks, keep, x):
    """Stochastic-depth factors of a whole stack of blocks from ONE draw: row 2 i / 2 i + 1 of a (2 * live blocks, samples)
    uniform draw serves the attention / MLP half of the i-th block that drops anything; floor(keep + u) / keep is 1 / keep with
    probability keep, else 0 (the same Bernoulli factor the per-block draw gives, four launches per stack instead of two per
    half). ``keep`` = the stack's (2 * live, 1) keep probabilities (a buffer of the model)."""
    if keep is None or not blocks[0].training:
        return [None] * len(blocks)
    f = torch.rand(keep.shape[0], x.shape[0], device=x.device, dtype=x.dtype).add_(keep).floor_().div_(keep)
    out, row = [], 0
    for b in blocks:
        if b.drop_path.p == 0.0:
            out.append((None, None))
        else:
            out.append((f[row], f[row + 1]))
            row += 2
    return out


class PoseTransformer(nn.Module):
    def __init__(self, num_frame=9, num_joints=17, in_chans=2, embed_dim_ratio=32, depth=4, num_heads=8, mlp_ratio=2.,
                 qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.2, norm_layer=None,
                 compute_dtype: torch.dtype = torch.float32):
        super().__init__()
        # fp32 by default (parameters, GEMMs, softmax). torch.bfloat16: the blocks run under autocast -- bf16 MFMA GEMMs and
        # attention with fp32 accumulation, fp32 parameters and LayerNorm statistics (opt-in: the arithmetic is parity-unpinned
        # either way, but every number quoted for the head behind it is an fp32 number)
        self.compute_dtype = compute_dtype
        norm_layer = norm_layer or (lambda d: nn.LayerNorm(d, eps=1e-6))
        embed_dim = embed_dim_ratio * num_joints
        self.num_frame, self.num_joints = num_frame, num_joints
        self.Spatial_patch_to_embedding = nn.Linear(in_chans, embed_dim_ratio)
        self.Spatial_pos_embed = nn.Parameter(torch.zeros(1, num_joints, embed_dim_ratio))
        self.Temporal_pos_embed = nn.Parameter(torch.zeros(1, num_frame, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [float(v) for v in torch.linspace(0, drop_path_rate, depth)]          # stochastic-depth decay rule
        mk = lambda dim, i: _Block(dim, num_heads, mlp_ratio, qkv_bias, qk_scale, drop_rate, attn_drop_rate, dpr[i], norm_layer)  # noqa: E731
        self.Spatial_blocks = nn.ModuleList([mk(embed_dim_ratio, i) for i in range(depth)])
        self.blocks = nn.ModuleList([mk(embed_dim, i) for i in range(depth)])
        self.Spatial_norm, self.Temporal_norm = norm_layer(embed_dim_ratio), norm_layer(embed_dim)
        self.weighted_mean = nn.Conv1d(in_channels=num_frame, out_channels=1, kernel_size=1)
        self.head = nn.Sequential(nn.LayerNorm(embed_dim), nn.Linear(embed_dim, num_joints * 3))

    def forward(self, x):
        if self.compute_dtype != torch.float32 and x.is_cuda:
            with torch.autocast('cuda', dtype=self.compute_dtype):
                return self._forward(x).float()
        return self._forward(x)

    def spatial_is_deterministic(self) -> bool:
        """No random op acts on the per-frame (spatial) half: it then depends on the frame alone, not on the window."""
        if not self.training:
            return True
        return self.pos_drop.p == 0 and all(b.drop_path.p == 0 and b.attn.attn_drop == 0 and b.attn.proj_drop.p == 0
                                            and b.mlp.drop.p == 0 for b in self.Spatial_blocks)

    def forward_clip(self, x, n_windows: int):
        """All sliding windows of a clip batch at once, the per-frame half run once per FRAME: x (B, T, J, C) ->
        (B * n_windows, 1, J, 3), window w of clip b = frames w .. w + num_frame - 1. The spatial blocks see one frame at a
        time (joints are the tokens, no frame index enters), so a frame's features are the same in each of the up to
        ``num_frame`` windows that contain it: T frames are encoded instead of n_windows * num_frame (81 vs 657 per clip at
        cfg5). Equal to ``forward`` on the unfolded windows whenever the spatial half is deterministic (eval mode, or no
        dropout / stochastic depth); in training with stochastic depth the random drops would be shared by the windows of a
        frame instead of drawn per (window, frame) -- callers opt in to that (``PoseFormer(share_spatial=True)``)."""
        if self.compute_dtype != torch.float32 and x.is_cuda:
            with torch.autocast('cuda', dtype=self.compute_dtype):
                return self._forward_clip(x, n_windows).float()
        return self._forward_clip(x, n_windows)

    def _spatial(self, frames):
        t = _add_param(_linear(self.Spatial_patch_to_embedding, frames), self.Spatial_pos_embed)
        t = self.pos_drop(t)
        for blk, f in zip(self.Spatial_blocks, _stack_factors(self.Spatial_blocks, self._keep_of(self.Spatial_blocks, t), t)):
            t = blk(t, f)
        return _norm(self.Spatial_norm, t)

    def _keep_of(self, blocks, x):
        """(2 * live blocks, 1) keep probabilities of a stack on x's device (cached: no host-to-device copy inside a step)."""
        ps = [1.0 - b.drop_path.p for b in blocks if b.drop_path.p != 0.0 for _ in (0, 1)]
        if not ps:
            return None
        key = (id(blocks), x.device, x.dtype)
        cache = self.__dict__.setdefault('_keep_cache', {})
        if key not in cache or cache[key][0] != ps:
            cache[key] = (ps, torch.tensor(ps, device=x.device, dtype=x.dtype).unsqueeze(1))
        return cache[key][1]

    def _forward_clip(self, x, n_windows: int):
        B, T, J, C = x.shape
        feats = self._spatial(x.reshape(B * T, J, C)).reshape(B, T, -1)                     # (B, T, J*E), once per frame
        t = feats.unfold(1, self.num_frame, 1)[:, :n_windows].permute(0, 1, 3, 2)            # (B, W, F, J*E) view
        t = _add_param(t.reshape(B * n_windows, self.num_frame, -1), self.Temporal_pos_embed)
        return self._temporal(t, J)

    def _forward(self, x):
        B, Fr, J, C = x.shape
        t = _add_param(self._spatial(x.reshape(B * Fr, J, C)).reshape(B, Fr, -1), self.Temporal_pos_embed)
        return self._temporal(t, J)

    def _temporal(self, t, J):
        B = t.shape[0]
        t = self.pos_drop(t)
        for blk, f in zip(self.blocks, _stack_factors(self.blocks, self._keep_of(self.blocks, t), t)):
            t = blk(t, f)
        # the learned mean over the frames, Conv1d(F, 1, kernel 1), written as the weighted sum it is (the convolution library
        # spends seconds searching kernels for this shape at the first step and then runs four launches for it); fp32 on the GPU its
        # backward runs through K12 (ops.frame_mean), not through framework reductions
        wm = self.weighted_mean
        n = _norm(self.Temporal_norm, t)
        if _fused(n):
            from pedestrians_video_2_carla_amd import ops
            t = ops.frame_mean(n, wm.weight, wm.bias).unsqueeze(1)                                              # (B, 1, J*E)
        else:
            t = ((n * wm.weight.view(1, -1, 1)).sum(1) + wm.bias).unsqueeze(1)
        return _linear(self.head[1], _norm(self.head[0], t)).view(B, 1, J, 3)
