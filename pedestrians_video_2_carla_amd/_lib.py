"""ctypes binding of libp2c_hip.so (C ABI in include/p2c.h).

There is no CPU fallback: if the shared library is missing or an op is given host tensors, the call raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('P2C_LIB_PATH') or os.path.join(_HERE, 'csrc', 'libp2c_hip.so')    # (P2C_LIB_PATH: A/B builds, tools only)
P2C_JOINTS = 26

KIND = {'pose_changes_6d': 0, 'pose_changes': 1, 'relative_rot_6d': 2, 'relative_rot': 3, 'absolute_loc': 4}
TRANSFORM = {'none': 0, 'hips_neck': 1, 'bbox': 2, 'hips_neck_bbox': 3}
PRECISION = {'fp32': 0, 'bf16': 1, 'bf16x3': 2}

_f32p = ctypes.c_void_p  # device pointers travel as plain addresses
_i32 = ctypes.c_int32


class PoseHeadDesc(ctypes.Structure):
    """Mirror of ``p2c_pose_head_desc`` (include/p2c.h) -- keep field order in sync."""
    _fields_ = [
        ('B', _i32), ('T', _i32), ('kind', _i32), ('transform', _i32), ('t0', _i32), ('t1', _i32),
        ('mask_missing_joints', _i32), ('hips_lane', _i32), ('n_hips', _i32), ('n_neck', _i32),
        ('hips_idx', _i32 * 2), ('neck_idx', _i32 * 2),
        ('gt2d_joints', _i32), ('gt2d_channels', _i32), ('gt3d_joints', _i32),
        ('gmap2d', _i32 * P2C_JOINTS), ('gmap3d', _i32 * P2C_JOINTS), ('n_common2d', _i32), ('n_common3d', _i32), ('world_absolute', _i32),
        ('cam_f', ctypes.c_float), ('cam_cx', ctypes.c_float), ('cam_cy', ctypes.c_float),
        ('cam_dist', ctypes.c_float), ('cam_elev', ctypes.c_float), ('near_zero', ctypes.c_float),
        ('y', _f32p), ('skel_type', _f32p), ('ref_rel_loc', _f32p), ('ref_rel_rot', _f32p),
        ('ref_hn_shift', _f32p), ('ref_hn_scale', _f32p), ('dloc', _f32p), ('drot', _f32p),
        ('gt2d', _f32p), ('gt3d', _f32p),
        ('partials', _f32p), ('loss_sums', _f32p), ('losses', _f32p), ('final_rel_rot', _f32p),
        ('out_pose_changes', _f32p), ('out_projection_2d', _f32p), ('out_projection_2d_transformed', _f32p),
        ('out_shift', _f32p), ('out_scale', _f32p), ('out_relative_pose_loc', _f32p),
        ('out_relative_pose_rot', _f32p), ('out_absolute_pose_loc', _f32p), ('out_absolute_pose_rot', _f32p),
        ('out_world_loc', _f32p), ('out_world_rot', _f32p), ('defer_loss_finalize', _i32),
        ('gt_rot', _f32p), ('grad_loss_rot', _f32p),
    ]


P2C_MLP_MAX_LAYERS = 8


class MlpDesc(ctypes.Structure):
    """Mirror of ``p2c_mlp_desc`` (include/p2c.h)."""
    _fields_ = [
        ('n_layers', _i32), ('dims', _i32 * (P2C_MLP_MAX_LAYERS + 1)), ('N', ctypes.c_int64), ('x', _f32p),
        ('W', _f32p * P2C_MLP_MAX_LAYERS), ('b', _f32p * P2C_MLP_MAX_LAYERS), ('y', _f32p), ('gy', _f32p),
        ('gW', _f32p * P2C_MLP_MAX_LAYERS), ('gb', _f32p * P2C_MLP_MAX_LAYERS), ('partials', _f32p), ('w_image', _f32p),
        ('fused_adamw', _f32p), ('skip_pack', ctypes.c_int32), ('saved', _f32p), ('precision', ctypes.c_int32),
    ]


class TrainStepDesc(ctypes.Structure):
    """Mirror of ``p2c_train_step_desc`` (include/p2c.h)."""
    _fields_ = [('head', PoseHeadDesc), ('mlp', MlpDesc), ('pair_counts', _f32p)]


# every symbol include/p2c.h declares: (restype, argtypes)
_vp, _i64, _ip = ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int32)
class AdamWDesc(ctypes.Structure):
    """p2c_adamw_desc (include/p2c.h)."""
    _fields_ = [('n', ctypes.c_int64), ('param', _f32p), ('grad', _f32p), ('exp_avg', _f32p), ('exp_avg_sq', _f32p),
                ('step', _f32p), ('ticket', _f32p), ('hyper', _f32p), ('adamw', ctypes.c_int32),
                ('zero_grad', ctypes.c_int32), ('scatter_idx', _f32p), ('scatter_dst', _f32p)]


class GemmDesc(ctypes.Structure):
    """p2c_gemm_desc (include/p2c.h)."""
    _fields_ = [('M', ctypes.c_int32), ('N', ctypes.c_int32), ('K', ctypes.c_int32), ('trans_b', ctypes.c_int32),
                ('a', _f32p), ('lda', ctypes.c_int64), ('b', _f32p), ('ldb', ctypes.c_int64), ('c', _f32p), ('ldc', ctypes.c_int64),
                ('bias', _f32p), ('act', ctypes.c_int32), ('rows_per_scale', ctypes.c_int32), ('aux', _f32p), ('aux_out', _f32p),
                ('ldaux', ctypes.c_int64), ('row_scale', _f32p), ('residual', _f32p), ('ldr', ctypes.c_int64)]


class LstmDesc(ctypes.Structure):
    """p2c_lstm_desc (include/p2c.h)."""
    _fields_ = [('T', ctypes.c_int32), ('B', ctypes.c_int32), ('H', ctypes.c_int32), ('gx', _f32p), ('h0', _f32p),
                ('c0', _f32p), ('w_hh', _f32p), ('out', _f32p), ('hT', _f32p), ('cT', _f32p), ('acts', _f32p),
                ('cs', _f32p), ('g_out', _f32p), ('g_hT', _f32p), ('g_cT', _f32p), ('g_gx', _f32p), ('g_h0', _f32p),
                ('g_c0', _f32p), ('bias_a', _f32p), ('bias_b', _f32p), ('g_gx_bt', _f32p), ('gx_bt', ctypes.c_int32),
                ('out_drop', _f32p), ('drop_state', ctypes.c_void_p), ('drop_p', ctypes.c_float), ('drop_site', ctypes.c_int32)]


class DecoderDesc(ctypes.Structure):
    """p2c_decoder_desc (include/p2c.h)."""
    _fields_ = [('T', ctypes.c_int32), ('B', ctypes.c_int32), ('H', ctypes.c_int32), ('O', ctypes.c_int32)] + [
        (n, _f32p) for n in ('k0', 'c0', 'k1', 'c1', 'w_ih0', 'w_ih1', 'w_fc', 'b_fc', 'x0', 'drop', 'out', 'acts0', 'acts1',
                             'h0d', 'h1', 'g_out', 'g_gates0', 'g_gates1', 'g_outtot', 'g_c0', 'g_c1',
                             'hid0', 'hid1', 'w_hh0', 'w_hh1', 'b0a', 'b0b', 'b1a', 'b1b', 'kw0', 'kw1', 'out_bt',
                             'g_k0', 'g_k1', 'g_hid0', 'g_hid1')] + [('g_out_bt', ctypes.c_int32), ('force', _f32p), ('target', _f32p),
                                                     ('drop_state', ctypes.c_void_p), ('drop_p', ctypes.c_float), ('drop_site', ctypes.c_int32)]


class AtbProblem(ctypes.Structure):
    """p2c_atb_problem (include/p2c.h)."""
    _fields_ = [('a', _f32p), ('a_stride', _i64), ('b', _f32p), ('b_stride', _i64), ('K', _i64), ('M', _i32), ('N', _i32),
                ('out', _f32p), ('out_stride', _i64), ('bias_out', _f32p), ('bias_out2', _f32p), ('flags', _i32)]


class CollateDesc(ctypes.Structure):
    """p2c_collate_desc (include/p2c.h)."""
    _fields_ = [('N', _i64), ('T', _i32), ('Jd', _i32), ('C', _i32), ('raw', _f32p), ('is_flipped', _vp),
                ('flip_perm', _ip), ('rotation_deg', _f32p), ('bboxes', _f32p), ('clip_size', _f32p), ('noise', _f32p),
                ('miss_u', _f32p), ('miss_prob', ctypes.POINTER(ctypes.c_float)), ('transform', _i32), ('n_hips', _i32),
                ('hips_idx', _i32 * 2), ('n_neck', _i32), ('neck_idx', _i32 * 2), ('near_zero', ctypes.c_float),
                ('return_confidence', _i32), ('Ji', _i32), ('K', _i32), ('src_idx', _ip), ('dst_idx', _ip),
                ('frames', _f32p), ('t_projection_2d', _f32p), ('t_deformed', _f32p), ('t_transformed', _f32p),
                ('shift', _f32p), ('scale', _f32p), ('bboxes_out', _f32p)]


SYMBOLS = {
    'p2c_version': (ctypes.c_char_p, []),
    'p2c_pose_head_workspace_floats': (_i64, [_i32]),
    'p2c_pose_head_set_time_parallel_max_batch': (ctypes.c_int, [ctypes.c_int32]),
    'p2c_pose_head_set_packed_min_batch': (ctypes.c_int, [ctypes.c_int32]),
    'p2c_pose_head_set_chain_min_batch': (ctypes.c_int, [ctypes.c_int32]),
    'p2c_pose_head_fwd': (ctypes.c_int, [ctypes.POINTER(PoseHeadDesc), _vp]),
    'p2c_pose_head_fwd_launch': (ctypes.c_int, [ctypes.POINTER(PoseHeadDesc), ctypes.c_int32, _vp]),
    'p2c_gemm': (ctypes.c_int, [ctypes.POINTER(GemmDesc), _vp]),
    'p2c_debug_poison_lds': (ctypes.c_int, [_vp]),
    'p2c_frame_mean_fwd': (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    'p2c_gemm_tn_workspace_floats': (_i64, [_i32, _i32, _i32]),
    'p2c_gemm_reload_env': (None, []),
    'p2c_gemm_tn': (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp]),
    'p2c_pose_head_bwd': (ctypes.c_int, [ctypes.POINTER(PoseHeadDesc), ctypes.POINTER(_vp * 3), _vp, _vp, _vp, _vp, _vp]),
    'p2c_normalize_fwd': (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _ip, _i32, _ip,
                                         ctypes.c_float, _vp]),
    'p2c_normalize_bwd': (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _ip, _i32, _ip,
                                         ctypes.c_float, _vp]),
    'p2c_loss2d_workspace_floats': (_i64, [_i64]),
    'p2c_loss2d_fwd': (ctypes.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _ip, _ip, _i32, _i32, _vp, _vp,
                                      _vp, _vp]),
    'p2c_loss2d_bwd': (ctypes.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _ip, _ip, _i32, _i32, _vp, _vp,
                                      _vp, _vp]),
    'p2c_remap_nodes': (ctypes.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _ip, _ip, _vp]),
    'p2c_mlp_workspace_floats': (_i64, [ctypes.POINTER(MlpDesc)]),
    'p2c_mlp_saved_floats': (_i64, [ctypes.POINTER(MlpDesc)]),
    'p2c_mlp_image_floats': (_i64, [ctypes.POINTER(MlpDesc)]),
    'p2c_mlp_pack': (ctypes.c_int, [ctypes.POINTER(MlpDesc), _vp]),
    'p2c_mlp_image_index': (_i64, [ctypes.POINTER(MlpDesc), _ip, _i64]),
    'p2c_adamw_step': (ctypes.c_int, [ctypes.POINTER(AdamWDesc), _vp]),
    'p2c_collate_fwd': (ctypes.c_int, [ctypes.POINTER(CollateDesc), _vp]),
    'p2c_lstm_rec_fwd': (ctypes.c_int, [ctypes.POINTER(LstmDesc), _vp]),
    'p2c_lstm_rec_bwd': (ctypes.c_int, [ctypes.POINTER(LstmDesc), _vp]),
    'p2c_decoder_fwd': (ctypes.c_int, [ctypes.POINTER(DecoderDesc), _vp]),
    'p2c_decoder_bwd': (ctypes.c_int, [ctypes.POINTER(DecoderDesc), _vp]),
    'p2c_eval_workspace_floats': (_i64, [_i64]),
    'p2c_eval_pose3d': (ctypes.c_int, [_vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _ip, _ip, ctypes.c_int32, _ip,
                                       ctypes.c_int32, _vp, _vp, _vp, _vp, _vp]),
    'p2c_eval_pck': (ctypes.c_int, [_vp, _vp, _vp, _i64] + [ctypes.c_int32] * 4 + [_ip] + [ctypes.c_int32] * 3
                     + [_ip, ctypes.c_int32, _ip, ctypes.c_int32, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp]),
    'p2c_embed_workspace_floats': (_i64, [ctypes.c_int32] * 5),
    'p2c_embed_fwd': (ctypes.c_int, [_vp, _vp, _vp, _i64, _i64, _vp] + [ctypes.c_int32] * 6 + [_vp]),
    'p2c_embed_bwd': (ctypes.c_int, [_vp, _vp, _i64, _i64, _vp, _vp, _vp] + [ctypes.c_int32] * 6 + [_vp]),
    'p2c_fold_fwd': (ctypes.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp] + [ctypes.c_int32] * 4 + [_vp]),
    'p2c_fold_bwd': (ctypes.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, ctypes.c_int32, _vp, _vp, _vp, _vp] + [ctypes.c_int32] * 4 + [_vp]),
    'p2c_atb_workspace_floats': (_i64, [_i64, _i32, _i32, _i32]),
    'p2c_atb': (ctypes.c_int, [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _i64, _vp, _i32, _vp, _vp]),
    'p2c_atb_scaled': (ctypes.c_int, [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _i64, _vp, _i32, _vp, _i64, _vp, _vp]),
    'p2c_graph_node_counts': (ctypes.c_int, [_vp, ctypes.POINTER(_i32), ctypes.POINTER(_i32)]),
    'p2c_copy_group': (ctypes.c_int, [_vp, _vp, _vp, _i32, _vp]),
    'p2c_layernorm_supported': (ctypes.c_int, [_i32]),
    'p2c_layernorm_workspace_floats': (_i64, [_i64, _i32]),
    'p2c_layernorm_fwd': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, ctypes.c_float, _vp]),
    'p2c_layernorm_bwd': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i64, _i32, _vp]),
    'p2c_attn_small_supported': (ctypes.c_int, [_i32, _i32, _i32]),
    'p2c_attn_small_fwd': (ctypes.c_int, [_vp, _vp, ctypes.c_float, _i32, _i32, _i32, _i32, _vp]),
    'p2c_attn_small_bwd': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_float, _i32, _i32, _i32, _i32, _vp]),
    'p2c_atb_group_workspace_floats': (_i64, [ctypes.POINTER(AtbProblem), _i32]),
    'p2c_atb_group': (ctypes.c_int, [ctypes.POINTER(AtbProblem), _i32, _vp, _vp]),
    'p2c_mlp_fwd': (ctypes.c_int, [ctypes.POINTER(MlpDesc), _vp]),
    'p2c_mlp_bwd': (ctypes.c_int, [ctypes.POINTER(MlpDesc), _vp]),
    'p2c_train_step_supported': (ctypes.c_int, [ctypes.POINTER(TrainStepDesc)]),
    'p2c_train_step_workspace_floats': (_i64, [ctypes.POINTER(TrainStepDesc)]),
    'p2c_train_step': (ctypes.c_int, [ctypes.POINTER(TrainStepDesc), ctypes.POINTER(_vp * 3), _vp]),
    'p2c_train_step_launch': (ctypes.c_int, [ctypes.POINTER(TrainStepDesc), ctypes.POINTER(_vp * 3), _i32, _vp]),
    'p2c_count_target_pairs': (ctypes.c_int, [ctypes.POINTER(PoseHeadDesc), _vp, _vp]),
    'p2c_train_step_set_stream_min_batch': (ctypes.c_int, [ctypes.c_int32]),
    'p2c_train_step_set_wgrad_stream_min_batch': (ctypes.c_int, [ctypes.c_int32]),
}

_lib = None


class P2CError(RuntimeError):
    pass


def grad_loss_pointers(g0=None, g1=None, g2=None, vector=None):
    """The ``grad_losses`` argument of p2c_pose_head_bwd: three nullable device pointers (data_ptr ints or None);
    ``vector`` = data_ptr of a contiguous 3-float gradient."""
    if vector is not None:
        g0, g1, g2 = vector, vector + 4, vector + 8
    return ctypes.byref((_vp * 3)(g0, g1, g2))


def build(verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into csrc/libp2c_hip.so (hipcc cross-compiles without a GPU)."""
    res = subprocess.run(['make', '-j', str(min(os.cpu_count() or 1, 8)), '-C', os.path.join(_HERE, 'csrc')], capture_output=True, text=True)
    if verbose or res.returncode:
        print(res.stdout, res.stderr)
    if res.returncode:
        raise P2CError('building libp2c_hip.so failed:\n' + res.stderr)
    return LIB_PATH


def lib() -> ctypes.CDLL:
    """Load the library once (torch first, so its bundled libamdhip64.so.7 is the HIP runtime both sides use)."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (side effect: HIP runtime already mapped)
        if not os.path.exists(LIB_PATH):
            raise P2CError(f'{LIB_PATH} is missing: run `python __graft_entry__.py` (build()) or `make -C '
                           f'{os.path.join(_HERE, "csrc")}`. There is no CPU fallback for the HIP hot path.')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)  # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        if os.environ.get('P2C_POISON_LDS') == '1':
            handle = _PoisonedLib(handle)
        _lib = handle
    return _lib


class _PoisonedLib:
    """Test audit (P2C_POISON_LDS=1): every launching entry point -- int result, stream as its last argument -- is preceded by
    p2c_debug_poison_lds on that stream, so a kernel that reads LDS it never wrote returns NaN instead of whatever the previous
    workgroup left behind. Run the GPU suite once under it after touching a kernel's LDS layout."""

    def __init__(self, handle):
        self._h = handle
        self._wrap = {n for n, (res, args) in SYMBOLS.items()
                      if res is ctypes.c_int and args and args[-1] is _vp and not n.startswith('p2c_debug')
                      and n not in ('p2c_graph_node_counts',)}

    def __getattr__(self, name):
        fn = getattr(self._h, name)
        if name not in self._wrap:
            return fn
        poison = self._h.p2c_debug_poison_lds

        def call(*a):
            poison(a[-1])
            return fn(*a)
        return call


def check(rc: int, what: str):
    if rc != 0:
        kind = {-1: 'null pointer', -2: 'bad shape', -3: 'bad enum', -4: 'bad index'}.get(rc, f'hipError_t {rc}')
        raise P2CError(f'{what} failed: {kind} (rc={rc})')
