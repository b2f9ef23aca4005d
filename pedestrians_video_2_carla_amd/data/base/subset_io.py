"""The reference's common on-disk subset format (data/base/base_datamodule.py:468-508 ``_save_subset``; read back by
``BaseDataset.__init__`` / ``_get_raw_projection_2d`` / ``_get_targets`` / ``_get_meta``, base_dataset.py:60-150):

    projection_2d            (N, T, J, 2|3) float
    targets/<name>           (N, ...)       one dataset per target tensor
    meta/<name>              (N, ...)       numeric meta as is; string meta as uint16 codes + attribute ``labels`` (or, past
                                            64 KB of labels, as a fixed-length ASCII dataset)

Written as HDF5 when ``h5py`` is importable (chunks of one clip, as the reference) and otherwise -- this image has no h5py --
as an ``.npz`` container with exactly those keys (``meta/<name>__labels`` holds the label table). ``load_subset`` reads
either back into host arrays; nothing here touches the GPU.
"""
import os
from typing import Dict, Iterable, Tuple

import numpy as np

try:                       # optional: absent from the build image
    import h5py
except ImportError:        # pragma: no cover - depends on the environment
    h5py = None


_ATTRIBUTE_LIMIT = 64 * 1024      # base_datamodule.py:494: a label table at or past 64 KB cannot be an HDF5 attribute


def _fixed_ascii(strings) -> np.ndarray:
    """latin-1 byte strings as one fixed-length 'S' array (an empty list gives an empty 'S1' array)."""
    enc = [str(s).encode('latin-1') for s in strings]
    return np.array(enc, dtype=f'S{max((len(s) for s in enc), default=1) or 1}')


def _encode_strings(values: Iterable):
    """string meta -> (uint16 codes, label table) while the table stays below 64 KB -- the reference's attribute form
    (base_datamodule.py:487-499) -- else (fixed-length ASCII values, None): its dataset form (500-506), which is also what
    more than 65 535 distinct strings (per-clip ids) need, since the codes are uint16."""
    values = [str(v) for v in values]
    unique = sorted(set(values))
    labels = _fixed_ascii(unique)
    if labels.nbytes < _ATTRIBUTE_LIMIT and len(unique) <= np.iinfo(np.uint16).max + 1:
        mapping = {s: i for i, s in enumerate(unique)}
        return np.array([mapping[s] for s in values], dtype=np.uint16), labels
    return _fixed_ascii(values), None


def save_subset(save_dir: str, name: str, projection_2d: np.ndarray, targets: Dict[str, np.ndarray],
                meta: Dict[str, Iterable], prefer_hdf5: bool = True) -> str:
    """Write one subset; returns the file path (``<name>.hdf5`` or ``<name>.npz``)."""
    os.makedirs(save_dir, exist_ok=True)
    if h5py is not None and prefer_hdf5:
        path = os.path.join(save_dir, f'{name}.hdf5')
        with h5py.File(path, 'w') as f:
            f.create_dataset('projection_2d', data=projection_2d, chunks=(1, *projection_2d.shape[1:]))
            for k, v in targets.items():
                f.create_dataset(f'targets/{k}', data=v, chunks=(1, *v.shape[1:]))
            for k, v in meta.items():
                if isinstance(v, np.ndarray) and v.dtype.kind not in 'USO':
                    f.create_dataset(f'meta/{k}', data=v, chunks=(1, *v.shape[1:]) if v.ndim > 1 else None)
                else:
                    codes, labels = _encode_strings(v)
                    if labels is None:             # dataset form: the encoded strings themselves
                        f.create_dataset(f'meta/{k}', data=codes.astype(h5py.string_dtype('ascii', codes.dtype.itemsize)))
                    else:
                        f.create_dataset(f'meta/{k}', data=codes)
                        f[f'meta/{k}'].attrs['labels'] = labels.astype(h5py.string_dtype('ascii', labels.dtype.itemsize))
        return path
    path = os.path.join(save_dir, f'{name}.npz')
    arrays = {'projection_2d': np.asarray(projection_2d)}
    for k, v in targets.items():
        arrays[f'targets/{k}'] = np.asarray(v)
    for k, v in meta.items():
        if isinstance(v, np.ndarray) and v.dtype.kind not in 'USO':
            arrays[f'meta/{k}'] = v
        else:
            codes, labels = _encode_strings(v)
            arrays[f'meta/{k}'] = codes
            if labels is not None:
                arrays[f'meta/{k}__labels'] = labels
    np.savez(path, **arrays)
    return path


def load_subset(path: str) -> Tuple[np.ndarray, Dict[str, np.ndarray], Dict[str, Iterable]]:
    """(projection_2d, targets, meta) as host arrays; string meta come back as lists of str."""
    targets, meta = {}, {}
    if path.endswith('.npz'):
        with np.load(path, allow_pickle=False) as d:
            projection_2d = d['projection_2d']
            for k in d.files:
                if k.startswith('targets/'):
                    targets[k[8:]] = d[k]
                elif k.startswith('meta/') and not k.endswith('__labels'):
                    name = k[5:]
                    if f'{k}__labels' in d.files:
                        labels = [s.decode('latin-1') for s in d[f'{k}__labels']]
                        meta[name] = [labels[i] for i in d[k]]
                    elif d[k].dtype.kind == 'S':
                        meta[name] = [s.decode('latin-1') for s in d[k]]
                    else:
                        meta[name] = d[k]
        return projection_2d, targets, meta
    if h5py is None:
        raise RuntimeError(f'{path}: reading HDF5 subsets needs h5py, which is not installed; .npz subsets always work')
    with h5py.File(path, 'r') as f:
        projection_2d = f['projection_2d'][()]
        for k, v in f.get('targets', {}).items():
            targets[k] = v[()]
        for k, v in f.get('meta', {}).items():
            if 'labels' in v.attrs:
                labels = [s.decode('latin-1') if isinstance(s, bytes) else str(s) for s in v.attrs['labels']]
                meta[k] = [labels[i] for i in v[()]]
            elif v.dtype.kind in 'SO':
                meta[k] = [s.decode('latin-1') if isinstance(s, bytes) else str(s) for s in v[()]]
            else:
                meta[k] = v[()]
    return projection_2d, targets, meta
