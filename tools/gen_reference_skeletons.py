#!/usr/bin/env python3
"""Re-enter the reference's UE4 skeleton dumps as the build's own data file.

The four CARLA reference skeletons (and the one absolute-pose golden dump) are *data*:
26 bones x (location cm, rotation deg), extracted from UE4 by the reference's authors
(/root/reference/src/pedestrians_video_2_carla/data/carla/files/sk_*.yaml, structure.yaml).
This script reads them where they lie and writes ONE compact json:

    pedestrians_video_2_carla_amd/data/carla/files/reference_skeletons.json

Only numbers and bone names travel; no reference code is read or executed.
Run in the build container only (the reference tree does not exist on the GPU box).
"""
import json
import os
import sys

import yaml

REF = '/root/reference/src/pedestrians_video_2_carla/data/carla/files'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   'pedestrians_video_2_carla_amd', 'data', 'carla', 'files', 'reference_skeletons.json')

FILES = {
    'adult_female': 'sk_female_relative.yaml',
    'adult_male': 'sk_male_relative.yaml',
    'child_female': 'sk_girl_relative.yaml',
    'child_male': 'sk_kid_relative.yaml',
    'adult_female_absolute': 'sk_female_absolute.yaml',
}


def flatten(structure, parent, names, parents):
    """DFS pre-order over the nested one-key dicts of structure.yaml."""
    for node in structure:
        (name, children), = node.items()
        names.append(name)
        parents.append(parent)
        if children:
            flatten(children, len(names) - 1, names, parents)


def main():
    if not os.path.isdir(REF):
        sys.exit('reference tree not present; the committed json is the artefact to use')
    with open(os.path.join(REF, 'structure.yaml')) as f:
        structure = yaml.safe_load(f)['structure']
    names, parents = [], []
    flatten(structure, -1, names, parents)

    out = {'bones': names, 'parents': parents, 'skeletons': {}}
    for key, fn in FILES.items():
        with open(os.path.join(REF, fn)) as f:
            tr = yaml.safe_load(f)['transforms']
        assert list(tr.keys()) == names, (fn, 'bone order differs from structure DFS order')
        out['skeletons'][key] = {
            # UE4 units: centimetres / degrees, [x, y, z] and [pitch, yaw, roll]
            'location_cm': [[tr[n]['location'][a] for a in 'xyz'] for n in names],
            'rotation_deg': [[tr[n]['rotation'][a] for a in ('pitch', 'yaw', 'roll')] for n in names],
        }
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, 'w') as f:
        json.dump(out, f, indent=1)
    print('wrote', OUT, len(names), 'bones', parents)


if __name__ == '__main__':
    main()
