"""Build profiles/traffic.json (HBM bytes per launch group, from rocprofv3 PMC passes) for bench.py's roofline.traffic.

    python tools/make_traffic.py OUT.json  TAG=B:FETCH_DIR:WRITE_DIR ...

FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request, so it is doubled
(MI355X_MICROARCH.md, "HBM"). Each pass is a separate run (the two counters do not fit one pass). Launch groups are the
ones bench.py times together.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarise
from traffic_stamp import src_sha16

GROUPS = {
    # lean training kernels only (the materialising pose_head_rot_fwd<.., true> runs once at set-up, not in the step)
    'pose_head_rot_fwd<6D>(+loss_finalize)': ('pose_head_rot_fwd_tp<0>', 'pose_head_rot_fwd<0, false>', 'loss_finalize'),
    'pose_head_rot_bwd<6D>': ('pose_head_rot_bwd',),
    'mlp_fwd': ('mlp_fwd_kernel',),
    'mlp_bwd(+reduce)': ('mlp_bwd_kernel', 'mlp_wgrad_kernel', 'mlp_reduce_kernel', 'mlp_reduce_small_kernel'),
    'mlp_bwd(+reduce+adamw)': ('mlp_bwd_kernel', 'mlp_wgrad_kernel', 'mlp_reduce_kernel', 'mlp_reduce_small_kernel'),
    'adamw': ('adamw_kernel',),
    'train_clip_kernel': ('train_clip_kernel', 'train_stream_kernel'),      # (the first launch: latency / throughput form by batch)
    # the names bench.py's roofline_sweep uses (head_kernel_names): one kernel each, the forward without its finalize launch
    'pose_head_rot_fwd_tp<6D>': ('pose_head_rot_fwd_tp<0',),
    'pose_head_rot_bwd_tangent_tp<6D>': ('pose_head_rot_bwd_tangent_tp<0',),
    'pose_head_rot_fwd<6D>': ('pose_head_rot_fwd<0, false>',),
    'pose_head_rot_bwd_tangent<6D>': ('pose_head_rot_bwd_tangent<0>',),
    'pose_head_chain_fwd<6D>': ('pose_head_chain_fwd<0>',),
    'pose_head_chain_bwd<6D>': ('pose_head_chain_bwd<0>',),
    'train_wgrad_kernel(+adamw+loss)': ('train_wgrad_kernel', 'wgrad_stream_kernel', 'wgrad_reduce_kernel'),
}


def group_bytes(summary, counter, scale):
    out = {}
    for group, parts in GROUPS.items():
        total, found = 0.0, False
        for kernel, vals in summary.items():
            if any(p in kernel for p in parts) and counter in vals:
                total += vals[counter] * 1024.0 * scale
                found = True
        if found:
            out[group] = total
    return out


def main():
    out_path, specs = sys.argv[1], sys.argv[2:]
    traffic = {}
    for spec in specs:
        B, fetch_dir, write_dir = spec.split(':')
        rd = group_bytes(summarise(fetch_dir), 'FETCH_SIZE', 2.0)
        wr = group_bytes(summarise(write_dir), 'WRITE_SIZE', 1.0)
        for g in rd:
            traffic[f'{g}@B{B}'] = {'read_bytes': round(rd[g]), 'write_bytes': round(wr.get(g, 0.0)),
                                   'bytes': round(rd[g] + wr.get(g, 0.0)), 'src_sha16': src_sha16(g)}
    with open(out_path, 'w') as f:
        json.dump(traffic, f, indent=1, sort_keys=True)
    print(json.dumps(traffic, indent=1, sort_keys=True))


if __name__ == '__main__':
    main()
