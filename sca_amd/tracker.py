"""DubinsTracker: the native (C++, host-side) replacement of the reference's v_pref tracker for SCAPolicy /
RVO3dDubinsPolicy (mamp/policies/sca/scaPolicy.py:264-338 over dubinsmaneuver2d/3d.py).  Usable as MACAEnv(v_pref_fn=...).
"""
import ctypes as C
import math
import os

from . import hostinfo

import numpy as np

from . import _lib


class DubinsTracker:
    def __init__(self, goal, goal_heading, pref_speed, zaxis=None, turning_radius=1.5,
                 pitchlims=(-math.pi / 4, math.pi / 4), neighbor_dist=10.0, nthreads=None):
        self.L = _lib.lib()
        goal = _lib.as_d(goal).reshape(-1, 3)
        self.n = n = len(goal)
        goal_heading = _lib.as_d(goal_heading).reshape(n, 3)
        pref_speed = _lib.as_d(np.broadcast_to(pref_speed, (n,)))
        zaxis = np.zeros(n, np.uint8) if zaxis is None else np.ascontiguousarray(zaxis, np.uint8)
        self.h = self.L.sca_tracker_create(n, _lib.ptr(goal, C.c_double), _lib.ptr(goal_heading, C.c_double),
                                           _lib.ptr(pref_speed, C.c_double), _lib.ptr(zaxis, C.c_uint8),
                                           float(turning_radius), float(pitchlims[0]), float(pitchlims[1]), float(neighbor_dist))
        if not self.h:
            raise RuntimeError('sca_tracker_create failed')
        self.nthreads = nthreads or min(hostinfo.usable_cores(), 256)
        self._nbr0 = np.full(n, -1.0)          # agent.neighbors[0][1] as the last computeNeighbors left it

    def close(self):
        if getattr(self, 'h', None):
            self.L.sca_tracker_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_neighbor_dist(self, neighbor_dist=None):
        """agent.neighborDist per agent (None: the one value given to the constructor)"""
        if neighbor_dist is None:
            rc = self.L.sca_tracker_set_neighbor_dist(self.h, None)
        else:
            nd = _lib.as_d(np.broadcast_to(neighbor_dist, (self.n,)))
            rc = self.L.sca_tracker_set_neighbor_dist(self.h, _lib.ptr(nd, C.c_double))
        if rc != 0:
            raise RuntimeError(f'sca_tracker_set_neighbor_dist rc={rc}')

    def set_agent_params(self, turning_radius=None, pitch_lo=None, pitch_hi=None):
        """agent.turning_radius / agent.pitchlims per agent (arrays of n; None = the constructor's value)"""
        keep = []

        def arr(a):
            if a is None:
                return None
            b = _lib.as_d(np.broadcast_to(a, (self.n,)))
            keep.append(b)
            return _lib.ptr(b, C.c_double)
        rc = self.L.sca_tracker_set_agent_params(self.h, arr(turning_radius), arr(pitch_lo), arr(pitch_hi))
        if rc != 0:
            raise RuntimeError(f'sca_tracker_set_agent_params rc={rc}')

    def note_neighbors(self, nbr_valid, nbr_n, nbr_dsq):
        """Remember agent.neighbors[0] of the policy pass that just ran (read by the next compute_v_pref, scaPolicy.py:299)."""
        v = np.asarray(nbr_valid).astype(bool)
        first = np.where(np.asarray(nbr_n) > 0, np.asarray(nbr_dsq)[:, 0], -1.0)
        self._nbr0[v] = first[v]

    def note_nbr0(self, nbr0):
        """Same from the compact form of sca_get_nbr0 (-2 = list untouched by the pass)."""
        nbr0 = np.asarray(nbr0)
        m = nbr0 > -2.0
        self._nbr0[m] = nbr0[m]

    def vpref(self, pos, vel, heading, active):
        n = self.n
        pos = _lib.as_d(pos).reshape(n, 3)
        vel = np.ascontiguousarray(vel, np.float32).reshape(n, 3)
        heading = _lib.as_d(heading).reshape(n, 3)
        active = np.ascontiguousarray(active, np.uint8).reshape(n)
        out = np.zeros((n, 3))
        rc = self.L.sca_tracker_vpref(self.h, _lib.ptr(pos, C.c_double), _lib.ptr(vel, C.c_float), _lib.ptr(heading, C.c_double),
                                      _lib.ptr(active, C.c_uint8), _lib.ptr(self._nbr0, C.c_double), _lib.ptr(out, C.c_double),
                                      int(self.nthreads))
        if rc != 0:
            raise RuntimeError(f'sca_tracker_vpref rc={rc}')
        return out

    def replans(self):
        r = np.zeros(self.n, np.int32)
        self.L.sca_tracker_replans(self.h, _lib.ptr(r, C.c_int32))
        return r

    # MACAEnv(v_pref_fn=tracker): called once per step before the policy pass
    def __call__(self, env):
        if env._last_neighbors is not None:
            nb = env._last_neighbors
            self.note_neighbors(nb['nbr_valid'], nb['nbr_n'], nb['nbr_dsq'])
        active = ((env.flags & 7) == 0) & env._ext
        return self.vpref(env.pos, env.vel, env.heading, active.astype(np.uint8))


def dubins_plan(qi, qf, rmin=1.5, pitchlims=(-math.pi / 4, math.pi / 4), max_samples=0):
    """dubinsmaneuver3d.dubinsmaneuver3d(qi, qf, Rmin, pitchlims): returns (length, mode, samples[k,5])."""
    L = _lib.lib()
    qi = _lib.as_d(qi)[:5].copy()
    qf = _lib.as_d(qf)[:5].copy()
    length = C.c_double(0)
    mode = C.create_string_buffer(8)
    ns = C.c_int32(0)
    samples = np.zeros((max(max_samples, 1), 5))
    rc = L.sca_dubins_plan(_lib.ptr(qi, C.c_double), _lib.ptr(qf, C.c_double), float(rmin), float(pitchlims[0]),
                           float(pitchlims[1]), C.byref(length), mode, C.byref(ns),
                           _lib.ptr(samples, C.c_double) if max_samples else None, int(max_samples))
    if rc != 0:
        raise RuntimeError(f'sca_dubins_plan rc={rc}')
    return length.value, mode.value.decode(), samples[:min(ns.value, max_samples)], ns.value


def libm_check():
    """(matches, mismatches[5]) -- does this host's libm give the bits of the restated glibc 2.35 x86-64 FMA build the tracker
    computes with (sca_libm_check)?  False means: a Python reference run on THIS host would differ from the golden vectors, and
    from this library, in the last bit of some path lengths."""
    bad = (C.c_int64 * 5)()
    r = _lib.lib().sca_libm_check(bad)
    return r == 0, list(bad)
