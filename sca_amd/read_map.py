"""Obstacle maps for the obstacle-dense scenario: the `.binvox` voxel file of exp3 "low altitude search"
and its conversion to obstacle spheres, with the semantics of mamp/read_map.py:15-85.

binvox: ASCII header (`#binvox 1`, `dim dx dy dz`, `translate tx ty tz`, `scale s`, `data`), then (value, count) byte pairs
(run-length coding) over a dx*dy*dz grid stored x-major, z next, y fastest; the reference swaps to x, y, z (read_map.py:15-28).
`read_obstacle` walks occupied voxels in the reference's loop order (`x`, then the index it calls `y` over dims[2], then `z`
over dims[1], read_map.py:58-62) and keeps every 11th voxel above z = -1 and every 1001st below it (the two counters of
read_map.py:67-84), as spheres of radius 0.2.  The walk is done with array operations, not three Python loops.
"""
import numpy as np

from .env import Obstacle


class Voxels:
    def __init__(self, data, dims, translate, scale, axis_order):
        if axis_order not in ('xzy', 'xyz'):
            raise ValueError('axis_order must be xzy or xyz')
        self.data, self.dims, self.translate, self.scale, self.axis_order = data, dims, translate, scale, axis_order


def read_header(fp):
    line = fp.readline().strip()
    if not line.startswith(b'#binvox'):
        raise IOError('Not a binvox file')
    dims = [int(v) for v in fp.readline().strip().split(b' ')[1:]]
    translate = [float(v) for v in fp.readline().strip().split(b' ')[1:]]
    scale = [float(v) for v in fp.readline().strip().split(b' ')[1:]][0]
    fp.readline()                                   # the `data` line
    return dims, translate, scale


def read_as_3d_array(fp, fix_coords=True):
    dims, translate, scale = read_header(fp)
    raw = np.frombuffer(fp.read(), dtype=np.uint8)
    if raw.size % 2:
        raise IOError('binvox: odd number of run-length bytes')
    data = np.repeat(raw[::2], raw[1::2]).astype(bool)
    if data.size != dims[0] * dims[1] * dims[2]:
        raise IOError('binvox: run lengths do not add up to the grid size')
    data = data.reshape(dims)
    if fix_coords:
        return Voxels(np.transpose(data, (0, 2, 1)), dims, translate, scale, 'xyz')
    return Voxels(data, dims, translate, scale, 'xzy')


def obstacle_positions(model, center, resolution=0.1, bias=(-13.5, -13.5, -1.4), tree_every=10, floor_every=1000):
    """[k, 3] sphere centres in the order the reference appends them (= obstacle ids)."""
    d = model.data[:model.dims[0], :model.dims[2], :model.dims[1]]            # the index ranges of read_map.py:58-60
    x, y, z = np.nonzero(d)                                                     # C order == x outer, y, z inner
    pos = np.empty((len(x), 3))
    pos[:, 0] = (y + model.translate[1]) * resolution + bias[0] + center[0]    # read_map.py:62-64 (x and y swapped there)
    pos[:, 1] = (x + model.translate[0]) * resolution + bias[1] + center[1]
    pos[:, 2] = z * resolution + bias[2]
    above = pos[:, 2] > -1
    keep = np.zeros(len(x), bool)
    ia = np.flatnonzero(above)
    keep[ia[tree_every::tree_every + 1]] = True          # counter reaches 10 on the 11th, 22nd, ... above-ground voxel
    ib = np.flatnonzero(~above)
    keep[ib[floor_every::floor_every + 1]] = True        # counter reaches 1000 on the 1001st, 2002nd, ... ground voxel
    return pos[keep]


def read_obstacle(center, environ, obs_path):
    """mamp/read_map.py:42-85: list of Obstacle (sphere r = 0.2) for `environ == "exp3"`, [] otherwise."""
    if environ != 'exp3':
        return []
    with open(obs_path, 'rb') as f:
        model = read_as_3d_array(f)
    pos = obstacle_positions(model, center)
    return [Obstacle(pos=[float(p[0]), float(p[1]), float(p[2])], shape_dict={'shape': 'sphere', 'feature': 0.2}, id=i)
            for i, p in enumerate(pos)]
