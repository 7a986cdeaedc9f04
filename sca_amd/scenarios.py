"""Synthetic swarm scenarios: deterministic generalisations of the reference's generators
(run_example/run_sca.py:17-81, run_example/run_orca.py:16-71) to arbitrary N.

Each generator returns a dict: start[N,6] (x,y,z,yaw,pitch,roll), goal[N,6], obs_pos[M,3], obs_radius[M].
"""
import math

import numpy as np


def _mod2pi(theta):                      # mamp/util.py:113
    return theta - 2.0 * math.pi * math.floor(theta / 2.0 / math.pi)


def circle(n, rad=None, center=(0.0, 0.0), z=10.0):
    """run_sca.set_circle_pos (run_sca.py:17-30): agents on a circle, goal = antipode.
    rad defaults to 1.25 * n / (2 pi): arc spacing 1.25 m, i.e. ~16 neighbours within neighborDist = 10 m."""
    if rad is None:
        rad = 1.25 * n / (2.0 * math.pi)
    start = np.zeros((n, 6))
    for j in range(n):
        ang = 2 * j * np.pi / n
        start[j] = [center[0] + round(rad * np.cos(ang), 2), center[1] + round(rad * np.sin(ang), 2), z,
                    round(_mod2pi(ang + np.pi), 5), 0.0, 0.0]
    goal = np.array([start[(j + int(n / 2)) % n] for j in range(n)])
    return dict(start=start, goal=goal, obs_pos=np.zeros((0, 3)), obs_radius=np.zeros(0))


def random_cube(n, seed=0, min_sep=1.5, z_offset=None):
    """run_sca.set_random_pos (run_sca.py:33-50) at constant density: cube half-side 25 * (n/100)^(1/3),
    seeded numpy Generator, rejection sampling for a minimum pairwise start distance.  The cube is lifted by its half-side
    + 5 m, as the reference's is (half-side 25, offset 30): everything stays above the ground plane that the posture
    constraint enforces (util.py:16, `pos.z + dt * v.z >= 0`); an agent below it has no admissible candidate at all."""
    rng = np.random.default_rng(seed)
    r = 25.0 * (n / 100.0) ** (1.0 / 3.0)
    if z_offset is None:
        z_offset = r + 5.0
    cell = max(min_sep, 1e-9)
    grid = {}
    pts = []

    def ok(p):
        c = tuple(np.floor(p / cell).astype(int))
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (-1, 0, 1):
                    for q in grid.get((c[0] + dx, c[1] + dy, c[2] + dz), ()):
                        if np.linalg.norm(p - q) < min_sep:
                            return False
        grid.setdefault(c, []).append(p)
        return True

    while len(pts) < n:
        p = rng.uniform(-r, r, 3)
        if ok(p):
            pts.append(p)
    start = np.zeros((n, 6))
    start[:, :3] = np.array(pts)
    start[:, 3] = rng.uniform(0.0, 2 * np.pi, n)
    goal = np.zeros((n, 6))
    goal[:, :3] = rng.uniform(-r, r, (n, 3))
    goal[:, 3] = rng.uniform(0.0, 2 * np.pi, n)
    start[:, 2] += z_offset
    goal[:, 2] += z_offset
    return dict(start=start, goal=goal, obs_pos=np.zeros((0, 3)), obs_radius=np.zeros(0))


def takeoff_landing(n, pitch=30.0):
    """run_sca.set_takeoff_landing_pos (run_sca.py:53-81) + the 8 sphere obstacles of build_obstacles
    (run_sca.py:139-150): the 16-agent cell tiled on a square lattice."""
    cells = (n + 15) // 16
    side = int(math.ceil(math.sqrt(cells)))
    start, goal, obs = [], [], []
    rad = 4.0
    for c in range(cells):
        cx, cy = (c % side) * pitch, (c // side) * pitch
        k = min(16, n - 16 * c)
        landing = k - int(k / 2)
        takeoff = int(k / 2)
        pos = []
        for j in range(landing):
            pos.append([cx + round(rad * np.cos(2 * j * np.pi / landing), 2), cy + round(rad * np.sin(2 * j * np.pi / landing), 2),
                        10.0, round(np.pi / 2, 5), 0, 0])
        for j in range(landing, k):
            pos.append([cx + round(rad * np.cos(2 * j * np.pi / takeoff), 2), cy + round(rad * np.sin(2 * j * np.pi / takeoff), 2),
                        0.0, round(-np.pi / 2, 5), 0, 0])
        g = []
        for j in range(landing):
            g.append(pos[j + landing] if j + landing < k else pos[j])
        for j in range(landing, k):
            g.append(pos[j - takeoff])
        start += pos
        goal += g
        for j in range(8):
            obs.append([cx + round(rad * np.cos(2 * j * np.pi / 8), 2), cy + round(rad * np.sin(2 * j * np.pi / 8), 2), 5.0])
    return dict(start=np.array(start, float), goal=np.array(goal, float), obs_pos=np.array(obs, float),
                obs_radius=np.full(len(obs), 1.0))


def max_run_dist(start, goal):
    """agent.py:74: 3.0 * l3norm(start, goal) with the rounded l3norm of util.py:104."""
    d = np.sqrt(((np.asarray(start)[:, :3] - np.asarray(goal)[:, :3]) ** 2).sum(1))
    return 3.0 * np.array([round(float(x), 5) for x in d])


def spawn_n_drones(n, center=(35.0, 30.0), rad=10.0, environment='exp3'):
    """run_example/run_sca.py:84-103: ring of n drones facing inward-tangent, goals at the antipodes; exp3 flies at z = 2."""
    height = 2 if environment == 'exp3' else 10
    start, goal = [], []
    for i in range(n):
        c, s = math.cos(2 * i * np.pi / n), math.sin(2 * i * np.pi / n)
        start.append([center[0] + rad * c, center[1] + rad * s, height, np.deg2rad(-90 - i * 360 / n), 0, 0])
        goal.append([center[0] - rad * c, center[1] - rad * s, height, np.deg2rad(90 - i * 360 / n), 0, 0])
    return dict(start=np.array(start, float), goal=np.array(goal, float), obs_pos=np.zeros((0, 3)), obs_radius=np.zeros(0))


def sphere(n, rad=25.0, z_value=30.0):
    """run_example/run_orca.py:36-54 set_sphere (also run_rvo.py, run_srvo.py): n starts on a Fibonacci sphere of radius 25
    lifted by 30 m, goals at the antipodes, all headings 0."""
    phi = (math.sqrt(5.0) - 1.0) / 2.0
    start, goal = [], []
    for k in range(1, n + 1):
        zn = (2 * k - 1) / n - 1
        xn = math.sqrt(1 - zn ** 2) * math.cos(2 * math.pi * k * phi)
        yn = math.sqrt(1 - zn ** 2) * math.sin(2 * math.pi * k * phi)
        p = np.array([rad * xn, rad * yn, rad * zn, 0.0, 0.0, 0.0])
        g = -p
        p[2] += z_value
        g[2] += z_value
        start.append(p)
        goal.append(g)
    return dict(start=np.array(start, float), goal=np.array(goal, float), obs_pos=np.zeros((0, 3)), obs_radius=np.zeros(0))
