"""PoseFormer wrapper (reference modules/movements/pose_former/pose_former.py:9-138; Zheng et al., ICCV 2021).

The transformer itself is third-party code (``third_party/PoseFormer`` git submodule, empty in the reference snapshot
and un-pinned: parity of its arithmetic is UNPINNED). What the reference owns -- and what is restated
here -- is the wrapper: a sliding window of ``receptive_frames`` over the clip, each window's (B,1,J,3) centre-frame
prediction broadcast into frames [i+shift, i+shift+receptive) (later windows overwrite), ``eval_slice`` restricted to
the frames that have a full receptive field, output type absolute_loc, AdamW(4e-4, wd 0.1) + ExponentialLR(0.99).
``inner_model`` injects a transformer; otherwise the third-party package is used when it is importable and, when it is not
(the submodule is empty in the reference checkout), the build's own restatement of the published architecture,
``pose_transformer.PoseTransformer`` (parity-unpinned).
"""
import torch

from pedestrians_video_2_carla_amd.modules.flow.output_types import MovementsModelOutputType
from pedestrians_video_2_carla_amd.modules.movements.movements import MovementsModel
from pedestrians_video_2_carla_amd.utils.exceptions import NotAvailableException


def _third_party_model(**kw):
    try:
        from common.model_poseformer import PoseTransformer        # third_party/PoseFormer on sys.path
    except ImportError:
        from pedestrians_video_2_carla_amd.modules.movements.pose_former.pose_transformer import PoseTransformer
    return PoseTransformer(**kw)


class PoseFormer(MovementsModel):
    def __init__(self, clip_length: int = 30, receptive_frames: int = 9, single_joint_embeddings_size=32, depth=4,
                 num_heads=8, mlp_ratio=2, qkv_bias=True, qk_scale=None, drop_rate=0, attn_drop_rate=0,
                 drop_path_rate=0.2, input_features=2, output_features=3, inner_model: torch.nn.Module = None,
                 compute_dtype: torch.dtype = torch.float32, share_spatial=None, **kwargs):
        super().__init__(**kwargs)
        # share_spatial: run the per-frame (spatial) half of the build's PoseTransformer once per FRAME instead of once per
        # (window, frame) -- every frame sits in up to `receptive_frames` windows. None (default): only when that is exactly
        # the same function (eval mode, or no dropout / stochastic depth in the spatial half); True: always (in training the
        # stochastic-depth drops of a frame are then shared by its windows); False: never.
        self.share_spatial = share_spatial
        self.__n_out = len(self.output_nodes)
        self.__out_features = output_features
        self.__clip_length = clip_length
        self.__receptive = receptive_frames
        self.__shift = receptive_frames // 2
        assert len(self.input_nodes) == self.__n_out
        self.pose_former = inner_model if inner_model is not None else _third_party_model(
            num_frame=receptive_frames, num_joints=len(self.input_nodes), in_chans=input_features,
            embed_dim_ratio=single_joint_embeddings_size, depth=depth, num_heads=num_heads, mlp_ratio=mlp_ratio,
            qkv_bias=qkv_bias, qk_scale=qk_scale, drop_rate=drop_rate, attn_drop_rate=attn_drop_rate,
            drop_path_rate=drop_path_rate, norm_layer=None, **({'compute_dtype': compute_dtype} if compute_dtype != torch.float32 else {}))
        self._hparams.update({
            'receptive_frames': receptive_frames, 'single_joint_embeddings_size': single_joint_embeddings_size,
            'depth': depth, 'num_heads': num_heads, 'mlp_ratio': mlp_ratio, 'qkv_bias': qkv_bias, 'qk_scale': qk_scale,
            'drop_rate': drop_rate, 'attn_drop_rate': attn_drop_rate, 'drop_path_rate': drop_path_rate,
        })

    @staticmethod
    def add_model_specific_args(parent_parser):
        parent_parser = MovementsModel.add_model_specific_args(parent_parser)
        group = parent_parser.add_argument_group('PoseFormer Movements Module')
        group.add_argument('--single_joint_embeddings_size', default=32, type=int)
        group.add_argument('--receptive_frames', default=9, type=int)
        return parent_parser

    @property
    def output_type(self) -> MovementsModelOutputType:
        return MovementsModelOutputType.absolute_loc

    @property
    def eval_slice(self):
        return slice(self.__shift, self.__clip_length - self.__receptive + self.__shift + 1)

    def forward(self, x, *args, **kwargs):
        B, T = x.shape[:2]
        n_windows = self.__clip_length - self.__receptive + 1
        inner = self.pose_former
        if (hasattr(inner, 'forward_clip') and self.share_spatial is not False
                and (self.share_spatial is True or inner.spatial_is_deterministic())):
            centre = inner.forward_clip(x[:, :n_windows + self.__receptive - 1], n_windows)
        else:
            # all windows in one batched call: (B, W, R, J, C) -> (B*W, R, J, C)
            windows = x.unfold(1, self.__receptive, 1)[:, :n_windows].permute(0, 1, 4, 2, 3)
            centre = inner(windows.reshape(B * n_windows, self.__receptive, *x.shape[2:]))
        centre = centre.reshape(B, n_windows, self.__n_out, self.__out_features)
        # frame f receives the prediction of the LAST window i with i+shift <= f < i+shift+receptive (overwrite order): window
        # f - shift for the frames up to the last window's centre, the last window for the (receptive - 1) frames behind it, zero
        # before the first centre and beyond the last window's reach (reference pose_former.py:117-127). Written as a
        # concatenation of slices: the gather ``centre[:, index]`` of the first version has an index_put with a device sort and
        # scratch buffers in its backward, the kind of framework op a captured step cannot rely on here (ops.py, "broadcast
        # parameters"); slices and concatenation are element-wise both ways.
        shift, W = self.__shift, n_windows
        n_head = min(shift, T)
        n_body = max(0, min(W, T - shift))
        n_tail = max(0, min(T - shift - W, self.__receptive - 1))
        n_rest = T - n_head - n_body - n_tail
        parts = []
        if n_head:
            parts.append(centre.new_zeros(B, n_head, self.__n_out, self.__out_features))
        if n_body:
            parts.append(centre[:, :n_body])
        parts.extend([centre[:, W - 1:W]] * n_tail)
        if n_rest:
            parts.append(centre.new_zeros(B, n_rest, self.__n_out, self.__out_features))
        return torch.cat(parts, dim=1)

    def configure_optimizers(self):
        optimizer = torch.optim.AdamW(self.parameters(), lr=0.0004, weight_decay=0.1)
        return {'optimizer': optimizer, 'lr_scheduler': torch.optim.lr_scheduler.ExponentialLR(optimizer, gamma=0.99)}
