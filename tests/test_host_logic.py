"""Host-side logic and the C-ABI surface -- CPU only, no compute calls."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------------------------ C ABI
def test_library_loads_and_exports_every_declared_symbol():
    from pedestrians_video_2_carla_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = _lib.lib()
    header = open(os.path.join(ROOT, 'include', 'p2c.h')).read()
    declared = set(re.findall(r'P2C_API[^;(]*?\b(p2c_\w+)\s*\(', header))
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.p2c_version().startswith(b'p2c-hip')
    assert lib.p2c_pose_head_workspace_floats(256) >= 128 * 3


def test_descriptor_layout_matches_the_header(tmp_path):
    """ctypes mirrors vs the C structs: same size and same offset for every field."""
    from pedestrians_video_2_carla_amd._lib import (AdamWDesc, CollateDesc, DecoderDesc, LstmDesc, MlpDesc, PoseHeadDesc,
                                                    TrainStepDesc)
    for cname, ctype in (('p2c_pose_head_desc', PoseHeadDesc), ('p2c_mlp_desc', MlpDesc), ('p2c_adamw_desc', AdamWDesc),
                         ('p2c_lstm_desc', LstmDesc), ('p2c_decoder_desc', DecoderDesc), ('p2c_collate_desc', CollateDesc),
                         ('p2c_train_step_desc', TrainStepDesc)):
        fields = [f[0] for f in ctype._fields_]
        src = tmp_path / f'{cname}.c'
        body = '\n'.join(f'  printf("{f} %zu\\n", offsetof({cname}, {f}));' for f in fields)
        src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "p2c.h"\nint main(void) {\n'
                       f'  printf("sizeof %zu\\n", sizeof({cname}));\n' + body + '\n  return 0;\n}\n')
        exe = tmp_path / cname
        subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True)
        out = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True)
                   .stdout.strip().splitlines())
        assert int(out['sizeof']) == ctypes.sizeof(ctype), cname
        for f in fields:
            assert int(out[f]) == getattr(ctype, f).offset, (cname, f)


def test_ops_refuse_host_tensors():
    from pedestrians_video_2_carla_amd import ops, _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    with pytest.raises(_lib.P2CError):
        ops.pose_head(torch.zeros(1, 2, 26, 6), ops.PoseHeadSpec(), torch.zeros(1, dtype=torch.int32))
    with pytest.raises(_lib.P2CError):
        ops.normalize(torch.zeros(2, 26, 2), 'hips_neck')


# ----------------------------------------------------------------------------------------------------- skeleton registry
def test_common_indices_match_reference(golden):
    from pedestrians_video_2_carla_amd.data.base.skeleton import get_common_indices
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.data.openpose.skeleton import BODY_25_SKELETON, COCO_SKELETON
    g = golden('common_indices')
    sk = {'body25': BODY_25_SKELETON, 'coco': COCO_SKELETON, 'carla': CARLA_SKELETON}
    for a in sk:
        for b in sk:
            if a == b:
                continue
            o, i = get_common_indices(input_nodes=sk[a], output_nodes=sk[b])
            assert list(o) == g[f'in_{a}__out_{b}__out_idx'].tolist()
            assert list(i) == g[f'in_{a}__out_{b}__in_idx'].tolist()
    assert get_common_indices(CARLA_SKELETON, CARLA_SKELETON) == (slice(None), slice(None))
    assert CARLA_SKELETON.get_hips_point().value == 1 and CARLA_SKELETON.get_neck_point().value == 8
    assert len(CARLA_SKELETON) == 26 and len(BODY_25_SKELETON) == 25 and len(COCO_SKELETON) == 18


def test_reference_tables_match_reference(golden):
    from pedestrians_video_2_carla_amd.data.carla import reference as R
    g = golden('reference_tables')
    for got, key in ((R.get_relative_tensors()[0], 'rel_loc'), (R.get_relative_tensors()[1], 'rel_rot'),
                     (R.get_absolute_tensors()[0], 'abs_loc'), (R.get_absolute_tensors()[1], 'abs_rot')):
        assert torch.allclose(got, g[key], atol=1e-6), key
    assert torch.allclose(R.get_projections(), g['projections'], rtol=1e-6, atol=1e-4)
    st = R.skeleton_types_from_meta({'age': ['adult', 'child', 'senior'], 'gender': ['male', 'female', 'neutral']})
    assert st.tolist() == [1, 2, 0]
    with pytest.raises(KeyError):      # ControlledPedestrian knows only the four CARLA types
        R.skeleton_types_from_meta({'age': ['senior'], 'gender': ['male']}, strict=True)


# ---------------------------------------------------------------------------------------------------------- model plugins
def _load(golden, name, cls, **kw):
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    g = golden(name)
    model = cls(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, **kw).eval()
    model.load_state_dict({k[4:]: v for k, v in g.items() if k.startswith('sd__')})   # reference checkpoint keys
    assert sum(p.numel() for p in model.parameters()) == int(g['n_params'])
    return g, model


@pytest.mark.parametrize('otype', ['pose_changes', 'absolute_loc', 'pose_2d'])
def test_linear_ae_drops_in(golden, otype):
    from pedestrians_video_2_carla_amd.modules.movements.linear_ae import LinearAE
    from pedestrians_video_2_carla_amd.modules.flow.output_types import MovementsModelOutputType as MT
    g, model = _load(golden, 'model_linear_ae_' + otype, LinearAE, movements_output_type=MT[otype])
    assert torch.allclose(model(g['frames']), g['out'], atol=1e-6)
    if otype == 'pose_changes':         # fused format: raw 6-D, orthonormalised later by the pose head
        from pedestrians_video_2_carla_amd.transforms.rotation_conversions import rotation_6d_to_matrix
        model.rotation_output_format = 'rotation_6d'
        y6 = model(g['frames'])
        assert y6.shape == (4, 16, 26, 6) and torch.allclose(rotation_6d_to_matrix(y6), g['out'], atol=1e-6)


def test_seq2seq_embeddings_drops_in(golden):
    from pedestrians_video_2_carla_amd.modules.movements.seq2seq import Seq2SeqEmbeddings
    from pedestrians_video_2_carla_amd.modules.flow.output_types import MovementsModelOutputType as MT
    g, model = _load(golden, 'model_seq2seq_embeddings_pose_2d', Seq2SeqEmbeddings, movements_output_type=MT.pose_2d)
    assert torch.allclose(model(g['frames']), g['out'], atol=1e-5)      # pins the decoder-restarts-from-encoder quirk


def test_pose_former_wrapper_window_semantics():
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.modules.movements.pose_former import PoseFormer
    from pedestrians_video_2_carla_amd.utils.exceptions import NotAvailableException

    class Inner(torch.nn.Module):
        def forward(self, x):
            return x[:, 4:5, :, :1].repeat(1, 1, 1, 3) + x.mean(1, keepdim=True)[..., 1:].repeat(1, 1, 1, 3)

    T = 81
    pf = PoseFormer(clip_length=T, inner_model=Inner(), input_nodes=CARLA_SKELETON)
    x = torch.randn(2, T, 26, 2)
    ref = torch.zeros(2, T, 26, 3)
    for i in range(T - 9 + 1):                      # the reference's loop, pose_former.py:122-125
        ref[:, i + 4:i + 9 + 4] = Inner()(x[:, i:i + 9])
    assert torch.allclose(pf(x), ref, atol=1e-6)
    assert pf.eval_slice == slice(4, 77) and pf.output_type.name == 'absolute_loc'
    # without an injected model (and without the third-party package) the build's own restatement of the published
    # architecture is constructed (round 1 raised NotAvailableException here)
    own = PoseFormer(clip_length=T, input_nodes=CARLA_SKELETON)
    from pedestrians_video_2_carla_amd.modules.movements.pose_former.pose_transformer import PoseTransformer
    assert isinstance(own.pose_former, PoseTransformer) and NotAvailableException is not None
    assert own.pose_former(torch.randn(3, 9, 26, 2)).shape == (3, 1, 26, 3)


# ------------------------------------------------------------------------------------------------------------------ flows
def _flow(loss_modes, **kw):
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.modules.flow.pose_lifting import LitPoseLiftingFlow
    from pedestrians_video_2_carla_amd.modules.movements.linear_ae import LinearAE
    model = LinearAE(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON)
    return LitPoseLiftingFlow(movements_model=model, loss_modes=loss_modes, **kw)


def test_loss_mode_resolution_and_hparams():
    flow = _flow(['loc_2d_3d'])
    assert [m[0] for m in flow._losses_to_calculate] == ['loc_2d', 'loc_3d', 'loc_2d_3d']   # requirements first
    assert flow.outputs_key == 'projection_2d_transformed'
    assert flow.crucial_keys[0] == 'projection_2d_transformed' and 'absolute_pose_loc' in flow.crucial_keys
    assert flow.hparams['loss_modes'] == ['loc_2d_3d'] and flow.hparams['movements_model_name'] == 'LinearAE'
    assert flow.hparams['movements_lr'] == 1e-4 and flow.hparams['movements_weight_decay'] == 1e-8
    assert flow.movements_model.rotation_output_format == 'rotation_6d'
    opt = flow.configure_optimizers()
    assert len(opt) == 1 and isinstance(opt[0]['optimizer'], torch.optim.AdamW)              # ZeroTrajectory has none
    assert _flow(None)._loss_modes[0].name == 'loc_2d'
    assert _flow(['loc_2d'], transform='none').outputs_key == 'projection_2d'
    assert set(type(flow).get_available_models()['movements']) >= {'LinearAE', 'Seq2SeqEmbeddings', 'PoseFormer'}


def test_flow_without_datamodule_or_gpu_fails_loudly():
    flow = _flow(['loc_2d_3d'])
    frames = torch.zeros(2, 4, 26, 2)
    batch = (frames, {'projection_2d_transformed': frames, 'absolute_pose_loc': torch.zeros(2, 4, 26, 3)},
             {'age': ['adult', 'child'], 'gender': ['female', 'male']})
    flow.on_train_batch_start(batch, 0)
    with pytest.raises(RuntimeError, match='datamodule'):
        flow.training_step(batch, 0)

    class DM:
        transform_callable = None
    flow.attach_datamodule(DM())
    from pedestrians_video_2_carla_amd._lib import P2CError
    with pytest.raises(P2CError):                       # CPU tensors: no fallback
        flow.training_step(batch, 0)


def test_get_outputs_contract_and_nan_policy():
    flow = _flow(['loc_2d_3d'])
    sliced = {'targets': {}, 'pose_inputs': None}
    out = flow._get_outputs('train', 2, sliced, {'loc_2d': torch.tensor(1.0), 'loc_2d_3d': torch.tensor(3.0)})
    assert float(out['loss']) == 3.0 and set(out) == {'loss', 'preds', 'targets'}
    assert set(out['preds']) >= {'pose_changes', 'world_rot_changes', 'world_loc_changes', 'projection_2d_transformed'}
    with pytest.raises(RuntimeError, match="Couldn't calculate any loss"):
        flow._get_outputs('train', 2, sliced, {'loc_2d': torch.tensor(1.0)})
    strict = _flow(['loc_2d_3d'], strict_nan_check=True)
    assert not strict._loss_is_usable(torch.tensor(float('nan'))) and strict._loss_is_usable(torch.tensor(1.0))
    flow.log('train_loss/loc_2d', torch.tensor(float('nan')))
    with pytest.raises(RuntimeError):
        flow.check_finite('train')


def test_flat_parameters_keep_optimizer_semantics():
    """AdamW over the single flat tensor == AdamW over the parameter list (element-wise update)."""
    from pedestrians_video_2_carla_amd.parallel.flat import FlatParameters
    torch.manual_seed(0)
    a = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    b = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    b.load_state_dict(a.state_dict())
    oa = torch.optim.AdamW(a.parameters(), lr=1e-2, weight_decay=1e-2)
    flat = FlatParameters(b.parameters())
    ob = flat.rebuild_optimizer(torch.optim.AdamW(b.parameters(), lr=1e-2, weight_decay=1e-2))
    x = torch.randn(11, 5)
    for _ in range(5):
        oa.zero_grad()
        a(x).pow(2).sum().backward()
        oa.step()
        flat.zero_grad()
        b(x).pow(2).sum().backward()
        ob.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, atol=1e-6)
    assert flat.nbytes() == 4 * sum(p.numel() for p in a.parameters())


# ------------------------------------------------------------------------------ the other model plugins
def _load_sd(model, g, sd_from=None):
    sd = {k[4:]: v for k, v in (sd_from or g).items() if k.startswith('sd__')}
    model.load_state_dict(sd)                 # reference checkpoint keys load unchanged
    return model


@pytest.mark.parametrize('name,kw', [('linear_ae_residual', {}), ('linear_ae_residual_leaky', {'linear_size': 64})])
def test_linear_ae_residual_drops_in(golden, name, kw):
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.modules.movements import linear_ae
    cls = {'linear_ae_residual': linear_ae.LinearAEResidual, 'linear_ae_residual_leaky': linear_ae.LinearAEResidualLeaky}[name]
    g = golden('model_' + name)
    model = _load_sd(cls(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, **kw).eval(), g)
    assert sum(p.numel() for p in model.parameters()) == int(g['n_params'])
    loc, rot = model(g['frames'])
    assert model.output_type.name == 'absolute_loc_rot'
    assert torch.allclose(loc, g['out_loc'], atol=1e-5) and torch.allclose(rot, g['out_rot'], atol=1e-5)
    assert isinstance(model.configure_optimizers()['optimizer'], torch.optim.Adam)


def test_seq2seq_embeddings_at_the_reference_configs_hidden_size(golden):
    """hidden_size 128 + pose_changes (reference configs/compare/carla-recorded_autoencoder_tests.yaml:38, seq2seq.py:245-288): the
    reference's own state_dict loads and its own output comes back (CPU: the nn.LSTM path; the device side of the same fixture is
    tests/test_lstm_gpu.py::test_reference_run_of_the_hidden_128_model_on_the_hip_recurrence)."""
    from pedestrians_video_2_carla_amd.modules.flow.output_types import MovementsModelOutputType as MT
    from pedestrians_video_2_carla_amd.modules.movements.seq2seq import Seq2SeqEmbeddings
    g, model = _load(golden, 'model_seq2seq_embeddings_h128_pose_changes', Seq2SeqEmbeddings, movements_output_type=MT.pose_changes,
                     hidden_size=128, single_joint_embeddings_size=8)
    assert torch.allclose(model(g['frames']), g['out'], atol=1e-5)


@pytest.mark.parametrize('name', ['a', 'b', 'c'])
def test_seq2seq_residual_variants_drop_in(golden, name):
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.modules.flow.output_types import MovementsModelOutputType as MT
    from pedestrians_video_2_carla_amd.modules.movements import seq2seq
    g = golden('model_seq2seq_residual_' + name)
    cls = getattr(seq2seq, 'Seq2SeqResidual' + name.upper())
    if name == 'c':
        model = cls(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, movements_output_type=MT.pose_changes,
                    hidden_size=16, single_joint_embeddings_size=8).eval()
        _load_sd(model, g)
    else:       # same weights as the Seq2SeqEmbeddings golden (asserted when the vectors were generated)
        model = cls(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, movements_output_type=MT.pose_2d).eval()
        _load_sd(model, g, golden('model_seq2seq_embeddings_pose_2d'))
    assert torch.allclose(model(g['frames']), g['out'], atol=1e-5)


def test_bench_refuses_to_report_more_gpus_than_ranks(tmp_path):
    """`python bench.py --gpus N` outside torchrun starts its own N ranks (reference: Lightning starts one process per GPU,
    modeling.py:275-282); with fewer devices than N it must fail loudly instead of printing an n_gpus line from fewer ranks,
    and a torchrun-style environment whose WORLD_SIZE disagrees with --gpus is an error too."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('needs a host with fewer than two GPUs')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'n_gpus' not in r.stdout and '--gpus 2' in r.stderr, (r.returncode, r.stdout, r.stderr[-500:])
    env.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'n_gpus' not in r.stdout and 'WORLD_SIZE=1' in r.stderr, (r.returncode, r.stdout, r.stderr[-500:])


def test_bench_launcher_starts_n_ranks_relays_one_line_and_fails_with_a_rank(tmp_path):
    """bench.launch_ranks on a stub rank program (no GPU): N children, each with RANK / LOCAL_RANK / WORLD_SIZE and the same
    MASTER_ADDR / MASTER_PORT; only rank 0's stdout comes through (ONE JSON line); a failing rank makes the launcher return its
    exit code and stops the ranks that are still running."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stub = tmp_path / 'rank_stub.py'
    stub.write_text(
        'import json, os, sys, time\n'
        'r = int(os.environ["RANK"])\n'
        'rec = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")}\n'
        'rec["argv"] = sys.argv[1:]\n'
        'open(os.path.join(sys.argv[1], f"rank{r}.json"), "w").write(json.dumps(rec))\n'
        'if len(sys.argv) > 2 and sys.argv[2] == "fail" and r == 1:\n'
        '    sys.exit(7)\n'
        'if len(sys.argv) > 2 and sys.argv[2] == "fail" and r != 1:\n'
        '    time.sleep(60)          # must be stopped by the launcher, not run out\n'
        'print(json.dumps({"rank": r, "n_gpus": int(os.environ["WORLD_SIZE"])}), flush=True)\n')
    driver = ('import sys; sys.path.insert(0, %r); import bench; '
              'sys.exit(bench.launch_ranks(3, sys.argv[1:], script=%r, have=3))' % (root, str(stub)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    ok = tmp_path / 'ok'
    ok.mkdir()
    r = subprocess.run([sys.executable, '-c', driver, str(ok)], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {'rank': 0, 'n_gpus': 3}, r.stdout      # rank 0's line, nobody else's
    recs = [json.loads((ok / f'rank{i}.json').read_text()) for i in range(3)]
    assert [x['RANK'] for x in recs] == ['0', '1', '2'] and [x['LOCAL_RANK'] for x in recs] == ['0', '1', '2']
    assert all(x['WORLD_SIZE'] == '3' and x['MASTER_ADDR'] == '127.0.0.1' and x['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' for x in recs)
    assert len({x['MASTER_PORT'] for x in recs}) == 1 and recs[0]['MASTER_PORT'].isdigit()
    assert all(x['argv'] == [str(ok)] for x in recs)
    bad = tmp_path / 'bad'
    bad.mkdir()
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, '-c', driver, str(bad), 'fail'], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 7 and time.time() - t0 < 45, (r.returncode, time.time() - t0)       # rank 1's code; the sleepers were stopped
    assert 'n_gpus' not in r.stdout


def test_static_batch_owns_an_int32_skeleton_type_index():
    """Graph mode: the captured launches keep the address of meta['skel_type'], so staging must hand the flow ONE int32 tensor
    on the frames' device whatever the batch carried -- an int64 tensor, age / gender strings, or nothing (adult female,
    data/carla/reference.py) -- and an int32 tensor already there is kept as is."""
    from pedestrians_video_2_carla_amd.trainer import Trainer
    tr = Trainer()
    frames = torch.zeros(3, 4, 26, 2)
    for meta, want in (({'skel_type': torch.tensor([0, 1, 2])}, [0, 1, 2]),
                       ({'age': ['adult', 'child', 'adult'], 'gender': ['male', 'female', 'female']}, None),
                       ({}, [0, 0, 0])):
        _, _, m = tr._with_skel_type((frames, {}, meta))
        st = m['skel_type']
        assert st.dtype == torch.int32 and st.device == frames.device and st.shape == (3,)
        if want is not None:
            assert st.tolist() == want
    keep = torch.tensor([3, 1, 0], dtype=torch.int32)
    batch = (frames, {}, {'skel_type': keep})
    assert tr._with_skel_type(batch) is batch


def test_stochastic_depth_factors_of_a_stack_come_from_one_draw():
    """PoseTransformer draws the survivor factors of a whole stack of blocks at once (``_stack_factors``): block i's two rows
    are Bernoulli(keep_i) / keep_i like the per-block draw of the reference's DropPath (the third-party PoseTransformer bound at
    modules/movements/pose_former/pose_former.py:62-76), blocks that drop nothing get none, eval mode draws nothing."""
    import torch
    from pedestrians_video_2_carla_amd.modules.movements.pose_former import pose_transformer as PT
    m = PT.PoseTransformer(num_frame=9, num_joints=26, in_chans=2, embed_dim_ratio=8, depth=4, num_heads=2, mlp_ratio=2.,
                           drop_path_rate=0.3)
    x = torch.zeros(200000, 1)
    m.train()
    torch.manual_seed(3)
    keep = m._keep_of(m.blocks, x)
    factors = PT._stack_factors(m.blocks, keep, x)
    assert keep.shape == (6, 1) and factors[0] == (None, None)
    for blk, (f1, f2) in list(zip(m.blocks, factors))[1:]:
        k = 1.0 - blk.drop_path.p
        for f in (f1, f2):
            assert f.shape == (200000,)
            vals = torch.unique(f)
            assert vals.numel() == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 1.0 / k) < 1e-6
            assert abs(float((f != 0).float().mean()) - k) < 5e-3
        assert not torch.equal(f1, f2)
    m.eval()
    assert PT._stack_factors(m.blocks, keep, x) == [None] * 4
    m.train()
    y = m(torch.randn(3, 9, 26, 2))
    assert y.shape == (3, 1, 26, 3) and torch.isfinite(y).all()


def _drop_keep(state, site, p, n):
    """The mask hash of csrc/p2c_rec_dev.h (drop_keys / drop_value) restated with numpy integers: keep flags of n elements."""
    M = np.uint64(0xFFFFFFFF)

    def mix(x):
        x = x.astype(np.uint64)
        x ^= x >> np.uint64(16)
        x = (x * np.uint64(0x7feb352d)) & M
        x ^= x >> np.uint64(15)
        x = (x * np.uint64(0x846ca68b)) & M
        x ^= x >> np.uint64(16)
        return x
    s0, s1, step = (int(v) & 0xFFFFFFFF for v in state[:3])
    k0 = mix(np.array([s0 ^ ((step * 0x9E3779B9) & 0xFFFFFFFF) ^ (((site + 1) * 0x632BE59B) & 0xFFFFFFFF)], dtype=np.uint64))[0]
    k1 = mix(np.array([(s1 + step + 0x85EBCA6B * (site + 1)) & 0xFFFFFFFF], dtype=np.uint64))[0]
    e = np.arange(n, dtype=np.uint64)
    h = mix((e * np.uint64(0x9E3779B1) + k0) & M) ^ k1
    return h >= np.uint64(int(p * 4294967296.0))


def test_in_kernel_dropout_streams_are_seeded_reproducibly_and_draw_independent_masks():
    """ops.dropout_state (the state of the dropout masks the time-loop kernels draw themselves): one draw from the default CPU
    generator -- the same state again under the same seed, another one for the next module; and the documented hash behind it
    (restated here with numpy; the GPU tests hold the kernels to the same restatement bit for bit): keep rate 1 - p, masks of
    consecutive steps, of two sites and of two seeds uncorrelated, no structure along the element index."""
    from pedestrians_video_2_carla_amd import ops
    cpu = torch.device('cpu')
    torch.manual_seed(123)
    a, b = ops.dropout_state(cpu).tolist(), ops.dropout_state(cpu).tolist()
    torch.manual_seed(123)
    a2 = ops.dropout_state(cpu).tolist()
    assert a == a2 and a[:2] != b[:2] and a[2:] == [0, 0]
    n, p = 1 << 18, 0.2
    sigma = (p * (1 - p) / n) ** 0.5
    base = _drop_keep(a, 0, p, n)
    assert abs(base.mean() - (1 - p)) < 4 * sigma
    others = [_drop_keep(a[:2] + [1, 1], 0, p, n), _drop_keep(a, 1, p, n), _drop_keep(b, 0, p, n)]
    for other in others:                       # next step / other site / other module: independent of the first mask
        assert abs(other.mean() - (1 - p)) < 4 * sigma
        both = (base & other).mean()
        assert abs(both - (1 - p) ** 2) < 5 * sigma
    for lag in (1, 2, 64, 64 * 512):            # neighbours along the unit, batch and time axes of a (T, 512, 64) mask
        both = (base[:-lag] & base[lag:]).mean()
        assert abs(both - (1 - p) ** 2) < 5 * sigma, lag
    for p_edge, want in ((0.0, 1.0),):
        assert _drop_keep(a, 0, p_edge, 1024).mean() == want
