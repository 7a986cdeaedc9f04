"""The speculation trees of the many-lanes-per-plan re-plan kernels (sca_tracker.hip.h, plan3d_spec) -> sca_amd/csrc/sca_spec_trees.h.

The planner's local search (dubinsmaneuver3d.py:86-100) is a chain: a candidate radius that is feasible and shorter moves b and doubles
the step (S), anything else turns the step round and divides it by ten (F).  k_replan_group<16 / 32 / 64> evaluate 3 / 7 / 15 candidates of
the chain's possible continuations at once and then apply the verdicts in the sequential order -- WHICH continuations only sets how many
steps a round advances, never the result.  A balanced binary tree advances 2 / 3 / 4 steps.  But the chain is far from a coin toss: every
search of the reference starts  F SSSSSS FF SSSSS FF SSSSSS FF ...  (overshoot, one step back fails too, then five or six doublings until
the next overshoot), so the trees here follow the likely continuations deep and the unlikely ones not at all: 6.3-7.3 steps per round
with 15 candidates on held-out searches (4.2-5.0 with 7, 2.5-2.6 with 3).

  python tools/gen_spec_trees.py --record     (build container only: imports /root/reference) runs the reference's own planner on poses of
                                              BASELINE configs 2 / 4 / 5 (perturbed mid-flight poses, and poses out of the episodes themselves:
                                              tools/dump_episode_poses.py), and of a +-20-m cube and records each search's verdicts as a string
                                              of S / F -> tools/data/radius_search_outcomes.json (data; committed)
  python tools/gen_spec_trees.py              fits P(S | kind of the current run, its length, the two previous runs' lengths) to the first half of
                                              every family, builds per context the tree of the 15 / 7 / 3 likeliest continuations, and writes the
                                              header; prints the steps per round on the other half
  python tools/gen_spec_trees.py --check      the same without writing: exit status 1 when the committed header differs (tests/test_spec_trees.py)"""
import collections
import heapq
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, 'tools', 'data', 'radius_search_outcomes.json')
HEADER = os.path.join(ROOT, 'sca_amd', 'csrc', 'sca_spec_trees.h')
RUN_CAP, PREV_CAP, PREV2_CAP = 8, 4, 8      # context: run length 0 .. 8, previous run's length 0 .. 4, the run before that 0 .. 8
MAXD = {15: 12, 7: 6, 3: 3}                 # longest path of a tree (bounds the device's walk and its path loop)
R, PL = 1.5, [-math.pi / 4, math.pi / 4]


def trace(case):
    """the reference's search (dubinsmaneuver3d.py:70-100) on its own try_to_construct, recording the local search's verdicts"""
    import numpy as np
    from mamp.policies.sca import dubinsmaneuver3d as d3
    qi, qf = case
    m = d3.DubinsManeuver3D(np.array(qi, dtype=np.float64), np.array(qf, dtype=np.float64), R, PL)
    b = 1.0
    fb = d3.try_to_construct(m, R * b)
    doublings = 0
    while len(fb) < 2:
        b *= 2.0
        fb = d3.try_to_construct(m, R * b)
        doublings += 1
        if doublings > 200:
            return None
    seq = []
    step = 0.1
    while abs(step) > 1e-10:
        c = b + step
        if c < 1.0:
            c = 1.0
        fc = d3.try_to_construct(m, R * c)
        if len(fc) > 0 and fc[1].length < fb[1].length:
            b = c
            fb = fc
            step *= 2.
            seq.append('S')
            continue
        step *= -0.1
        seq.append('F')
    return ''.join(seq)


def record():
    sys.path.insert(0, '/root/reference')
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import matplotlib
    matplotlib.use('Agg')
    import numpy as np
    from multiprocessing import Pool
    from gen_dubins_kat_long import midflight
    from sca_amd import scenarios
    rng = np.random.default_rng(7)
    out = {}
    episode = json.load(open(os.path.join(ROOT, 'tools', 'data', 'episode_poses.json')))    # tools/dump_episode_poses.py
    for fam, cnt in (('c2', 80), ('c5', 80), ('c4', 16), ('cube', 80), ('c2ep', 0), ('c5ep', 0)):
        cases = []
        if fam.endswith('ep'):                               # poses of the benchmark episodes themselves (pos | heading, goal pose)
            cases = [(np.array(q[:3] + q[3:5] + [0.0]), np.array(g)) for q, g in episode[fam[:2]]]
        elif fam == 'cube':                                    # the family of tools/gen_dubins_kat.py: poses in a +-20-m cube
            for _ in range(cnt):
                qi = np.concatenate([rng.uniform(-20, 20, 3), [rng.uniform(0, 2 * math.pi), rng.uniform(-0.6, 0.6), 0.0]])
                qf = np.concatenate([rng.uniform(-20, 20, 3), [rng.uniform(0, 2 * math.pi), rng.uniform(-0.3, 0.3), 0.0]])
                cases.append((qi, qf))
        else:
            sc = scenarios.circle(1024) if fam == 'c2' else (scenarios.circle(100000) if fam == 'c4' else scenarios.takeoff_landing(16384))
            n = len(sc['start'])
            for i in rng.choice(n, cnt, replace=False):
                if fam == 'c5':
                    s, g = sc['start'][i], sc['goal'][i]
                    q = s.copy()
                    q[:3] = s[:3] + rng.uniform(0.05, 0.9) * (g[:3] - s[:3]) + np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.5, 1.5), 0.0])
                    q[3] = rng.uniform(0, 2 * math.pi)
                    q[4] = rng.uniform(-0.7, 0.7)
                else:
                    q = midflight(rng, sc['start'][i], sc['goal'][i], rng.uniform(0.01, 0.9), 2.0, 0.8, 0.4)
                cases.append((q, sc['goal'][i]))
        cases = [([float(x) for x in a], [float(x) for x in b]) for a, b in cases]
        with Pool(int(os.environ.get('KAT_WORKERS', '7'))) as pool:
            res = pool.map(trace, cases, chunksize=1)
        out[fam] = [s for s in res if s]
        print(fam, len(out[fam]), 'searches, mean length', sum(len(s) for s in out[fam]) / len(out[fam]), flush=True)
    os.makedirs(os.path.dirname(DATA), exist_ok=True)
    with open(DATA, 'w') as f:
        json.dump(out, f, indent=0)


def advance(ctx, ch):
    """the context after one more verdict (the device's update: integers only); ctx = (kind 0 none / 1 S / 2 F, the current run's length,
    the previous run's, the one before that), lengths uncapped"""
    kind = 1 if ch == 'S' else 2
    if ctx[0] == kind:
        return (kind, ctx[1] + 1, ctx[2], ctx[3])
    return (kind, 1, ctx[1], ctx[2])


def capped(ctx):
    return (ctx[0], min(ctx[1], RUN_CAP), min(ctx[2], PREV_CAP), min(ctx[3], PREV2_CAP))


def train(seqs):
    """P(S | context) and, for contexts the searches never showed, P(S | context without its oldest run)"""
    cnt, cnt3 = collections.defaultdict(lambda: [0.5, 0.5]), collections.defaultdict(lambda: [1.0, 1.0])
    for s in seqs:
        ctx = (0, 0, 0, 0)
        for ch in s:
            cnt[capped(ctx)][0 if ch == 'S' else 1] += 1
            cnt3[capped(ctx)[:3]][0 if ch == 'S' else 1] += 1
            ctx = advance(ctx, ch)
    return ({c: v[0] / (v[0] + v[1]) for c, v in cnt.items()}, {c: v[0] / (v[0] + v[1]) for c, v in cnt3.items()})


def p_success(model, ctx):
    c = capped(ctx)
    if c in model[0]:
        return model[0][c]
    return model[1].get(c[:3], 0.66)


def build_tree(model, ctx, nodes):
    """the `nodes` verdict paths with the highest probability of being reached from `ctx` (greedy = optimal for a product measure);
    returned in the order they were taken: index 0 is the empty path"""
    heap = [(-1.0, '', ctx)]
    chosen = []
    while heap and len(chosen) < nodes:
        negp, path, c = heapq.heappop(heap)
        chosen.append(path)
        if len(path) >= MAXD[nodes]:
            continue
        p = p_success(model, c)
        heapq.heappush(heap, (negp * p, path + 'S', advance(c, 'S')))
        heapq.heappush(heap, (negp * (1 - p), path + 'F', advance(c, 'F')))
    return tuple(chosen)


def balanced(nodes):
    depth = {3: 2, 7: 3, 15: 4}[nodes]
    paths = ['']
    for d in range(1, depth):
        paths += [''.join(t) for t in __import__('itertools').product('SF', repeat=d)]
    return tuple(paths)


def pack(tree):
    """per node two words.  First: path bits (bit i = 1: verdict i on the way is S; 12 bits) | length << 12 | (index of the S child + 1) << 16
    | (F child + 1) << 21 | (index of the node whose success the path assumes last, + 1; 0: none) << 26.  Second: the quads of the node's
    ancestors | the verdicts the path assumes of them << 16."""
    words = []
    index = {p: i for i, p in enumerate(tree)}
    for p in tree:
        assert len(p) <= 12 and (p == '' or p[:-1] in index), 'a node without its parent'
        bits = sum(1 << i for i, ch in enumerate(p) if ch == 'S')
        cs, cf = index.get(p + 'S', -1) + 1, index.get(p + 'F', -1) + 1
        last_s = p.rfind('S')
        la = index[p[:last_s]] + 1 if last_s >= 0 else 0
        anc_mask = sum(1 << index[p[:j]] for j in range(len(p)))
        anc_bits = sum(1 << index[p[:j]] for j in range(len(p)) if p[j] == 'S')
        words.append((bits | (len(p) << 12) | (cs << 16) | (cf << 21) | (la << 26), anc_mask | (anc_bits << 16)))
    return words


def all_contexts():
    return [(k, r, p, q) for k in range(3) for r in range(RUN_CAP + 1) for p in range(PREV_CAP + 1) for q in range(PREV2_CAP + 1)]


def simulate(trees, of_ctx, seqs):
    rounds = steps = 0
    for s in seqs:
        i = 0
        ctx = (0, 0, 0, 0)
        while i < len(s):
            tree = set(trees[of_ctx[capped(ctx)]])
            path = ''
            while i < len(s) and path in tree:
                path += s[i]
                ctx = advance(ctx, s[i])
                i += 1
            rounds += 1
            steps += len(path)
    return steps / rounds


def generate():
    data = json.load(open(DATA))
    fams = sorted(data)
    fit = [s for f in fams for s in data[f][: len(data[f]) // 2]]
    model = train(fit)
    text = ['// sca_spec_trees.h -- GENERATED by tools/gen_spec_trees.py from tools/data/radius_search_outcomes.json: do not edit.',
            '// Which continuations of the radius search a round of plan3d_spec evaluates, per context (kind of the current run of verdicts,',
            '// its length, the lengths of the two runs before it).  The trees only set how far a round gets; tree 0 is the balanced one.',
            '#pragma once', '#include <cstdint>',
            '#if defined(__HIPCC__)', '#define SCA_SPEC_TAB __device__ const', '#else', '#define SCA_SPEC_TAB static const', '#endif',
            'namespace sca_spec {',
            f'constexpr int RUN_CAP = {RUN_CAP}, PREV_CAP = {PREV_CAP}, PREV2_CAP = {PREV2_CAP}, CONTEXTS = 3 * (RUN_CAP + 1) * (PREV_CAP + 1) * (PREV2_CAP + 1);',
            '// index of a context: ((kind * (RUN_CAP + 1) + run) * (PREV_CAP + 1) + prev) * (PREV2_CAP + 1) + prev2   (lengths capped)',
            '// a node, two words: path bits (bit i = 1: the i-th verdict on the way to it is a success) | length << 12 | (S child + 1) << 16 | (F child + 1) << 21 |',
            '// (the node whose success the path assumes last + 1) << 26;   ancestors\' quads | the verdicts the path assumes of them << 16']
    report = []
    for nodes in (15, 7, 3):
        trees = [balanced(nodes)]
        of_ctx = {}
        for c in all_contexts():
            reachable = c == (0, 0, 0, 0) or (c[0] != 0 and c[1] >= 1 and (c[2] >= 1 or c[3] == 0))
            t = build_tree(model, c, nodes) if reachable else trees[0]
            if t not in trees:
                trees.append(t)
            of_ctx[c] = trees.index(t)
        for f in fams:
            held = data[f][len(data[f]) // 2:]
            report.append(f'{f}: {nodes} candidates per round: {simulate(trees, of_ctx, held):.2f} steps per round on the held-out half '
                          f'(balanced tree: {simulate([balanced(nodes)], {c: 0 for c in all_contexts()}, held):.2f})')
        slots = 16 if nodes == 15 else (8 if nodes == 7 else 4)
        text.append(f'constexpr int TREES{nodes} = {len(trees)}, MAXD{nodes} = {max(len(p) for t in trees for p in t)};')
        text.append(f'SCA_SPEC_TAB uint8_t TREE{nodes}_OF_CONTEXT[CONTEXTS] = {{' + ', '.join(str(of_ctx[c]) for c in all_contexts()) + '};')
        text.append(f'SCA_SPEC_TAB uint32_t TREE{nodes}_NODES[TREES{nodes} * {slots} * 2] = {{')
        for t in trees:
            w = pack(t)
            w += [(0, 0)] * (slots - len(w))              # the spare quad evaluates the root again
            text.append('    ' + ', '.join(f'0x{a:08x}u, 0x{b:08x}u' for a, b in w) + ',    // ' + ' '.join(p or '.' for p in t))
        text.append('};')
    text.append('}  // namespace sca_spec')
    return '\n'.join(text) + '\n', report


if __name__ == '__main__':
    if '--record' in sys.argv:
        record()
        sys.exit(0)
    header, report = generate()
    if '--check' in sys.argv:
        same = os.path.exists(HEADER) and open(HEADER).read() == header
        print('sca_spec_trees.h', 'is up to date' if same else 'DIFFERS from what the data gives')
        sys.exit(0 if same else 1)
    with open(HEADER, 'w') as f:
        f.write(header)
    print('\n'.join(report))
