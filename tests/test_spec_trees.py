"""sca_spec_trees.h (generated): which continuations of the radius search a round of the many-lanes-per-plan re-plan kernels evaluates.
The trees only set how many steps a round advances -- the verdicts are applied in the sequential order whatever the tree -- but a table
that contradicts itself (a node whose path is not its parent's path plus one verdict, a child index that points elsewhere) would make a
quad evaluate another candidate than the walk assumes.  So: the committed header is what the committed data gives, and every tree in it
is a prefix-closed set of verdict paths whose child links say what the paths say."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'sca_amd', 'csrc', 'sca_spec_trees.h')


def _tables():
    text = open(HEADER).read()
    consts = {k: int(v) for k, v in re.findall(r'(\w+) = (\d+)[,;]', text)}
    out = {}
    for nodes, slots in ((15, 16), (7, 8), (3, 4)):
        of_ctx = [int(x) for x in re.search(rf'TREE{nodes}_OF_CONTEXT\[CONTEXTS\] = \{{([^}}]*)\}}', text).group(1).split(',')]
        body = re.search(rf'TREE{nodes}_NODES\[[^\]]*\] = \{{(.*?)\n\}};', text, re.S).group(1)
        words = [int(x, 16) for x in re.findall(r'0x([0-9a-f]{8})u', body)]
        out[nodes] = dict(slots=slots, trees=consts[f'TREES{nodes}'], maxd=consts[f'MAXD{nodes}'], of_ctx=of_ctx, words=words)
    return consts, out


def test_header_is_what_the_recorded_searches_give():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'gen_spec_trees.py'), '--check'], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_every_tree_is_consistent():
    consts, tabs = _tables()
    contexts = 3 * (consts['RUN_CAP'] + 1) * (consts['PREV_CAP'] + 1) * (consts['PREV2_CAP'] + 1)
    for nodes, t in tabs.items():
        assert len(t['of_ctx']) == contexts and max(t['of_ctx']) < t['trees'] and t['of_ctx'][0] != 0      # the opening context has a tree of its own
        assert len(t['words']) == t['trees'] * t['slots'] * 2
        assert t['maxd'] <= 12                                       # the path's field is twelve bits wide
        for k in range(t['trees']):
            w = t['words'][k * t['slots'] * 2:(k + 1) * t['slots'] * 2]
            first, second = w[0::2], w[1::2]
            assert first[nodes] == 0 and second[nodes] == 0          # the spare quad: the root again, counted out by the kernel
            paths = []
            for x in first[:nodes]:
                n = (x >> 12) & 15
                assert n <= t['maxd'] and (x & 0xfff) >> n == 0
                paths.append(''.join('S' if (x >> i) & 1 else 'F' for i in range(n)))
            assert paths[0] == '' and len(set(paths)) == nodes
            index = {p: i for i, p in enumerate(paths)}
            for i, (p, x, y) in enumerate(zip(paths, first, second)):
                assert p == '' or p[:-1] in index                    # prefix-closed: the walk reaches every node through its parent
                for ch, shift in (('S', 16), ('F', 21)):
                    assert (x >> shift) & 31 == index.get(p + ch, -1) + 1
                j = p.rfind('S')                                     # the node whose success the path assumes last: its length is the best one here
                assert (x >> 26) & 31 == (index[p[:j]] + 1 if j >= 0 else 0)
                assert y & 0xffff == sum(1 << index[p[:q]] for q in range(len(p)))
                assert y >> 16 == sum(1 << index[p[:q]] for q in range(len(p)) if p[q] == 'S')
        # tree 0: the balanced one (the fallback of contexts no search showed)
        depth = {15: 4, 7: 3, 3: 2}[nodes]
        w0 = t['words'][:2 * nodes:2]
        assert sorted((x >> 12) & 15 for x in w0) == sorted(d for d in range(depth) for _ in range(2 ** d))


def test_trees_beat_the_balanced_tree_on_the_recorded_searches():
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import gen_spec_trees as G
    _, report = G.generate()
    for line in report:
        m = re.search(r'(\d+) candidates per round: ([\d.]+) steps .*balanced tree: ([\d.]+)', line)
        nodes, got, bal = int(m.group(1)), float(m.group(2)), float(m.group(3))
        assert got > bal * {15: 1.5, 7: 1.4, 3: 1.2}[nodes], line


def test_all_nodes_walk_on_the_recorded_searches():
    """The device's walk (sca_tracker.hip.h, plan3d_spec) restated on the tables: every node forms the verdict it would get on the
    path, the path is the set of nodes whose ancestors' verdicts are the ones their paths assume, it ends at the one node whose
    verdict leads out of the tree -- and that must advance every recorded search exactly as the sequential loop does.  Nodes off
    the path get random verdicts and random validity (they hold candidates the loop never tries)."""
    import json
    import random
    consts, tabs = _tables()
    run, prev, prev2 = consts['RUN_CAP'], consts['PREV_CAP'], consts['PREV2_CAP']
    data = json.load(open(os.path.join(ROOT, 'tools', 'data', 'radius_search_outcomes.json')))
    rnd = random.Random(5)
    for nodes, t in tabs.items():
        slots = t['slots']
        rounds = 0
        for fam in sorted(data):
            for s in data[fam][::3]:
                i = 0
                ck = cr = cp = cq = 0
                while i < len(s):
                    tree = t['of_ctx'][((ck * (run + 1) + min(cr, run)) * (prev + 1) + min(cp, prev)) * (prev2 + 1) + min(cq, prev2)]
                    w = t['words'][tree * slots * 2:(tree + 1) * slots * 2]
                    first, second = w[0::2], w[1::2]
                    acc, valid = [], []
                    for q in range(slots):
                        n = (first[q] >> 12) & 15
                        path = ''.join('S' if (first[q] >> k) & 1 else 'F' for k in range(n))
                        on_path = s[i:i + n] == path
                        if on_path and i + n < len(s):
                            acc.append(s[i + n] == 'S'); valid.append(True)
                        elif on_path:
                            acc.append(rnd.random() < 0.5); valid.append(False)          # behind the loop's end
                        else:
                            acc.append(rnd.random() < 0.5); valid.append(rnd.random() < 0.9)
                    ga = sum(1 << q for q in range(slots) if acc[q])
                    gv = sum(1 << q for q in range(slots) if valid[q])
                    last = []
                    for q in range(slots - 1):
                        am, ab = second[q] & 0xffff, second[q] >> 16
                        on = valid[q] and (ga & am) == ab and (gv & am) == am
                        child = (first[q] >> (16 if acc[q] else 21)) & 31
                        if on and not (child > 0 and (gv >> (child - 1)) & 1):
                            last.append(q)
                    assert len(last) == 1, (nodes, fam, i, last)
                    n = (first[last[0]] >> 12) & 15
                    got = ''.join('S' if (first[last[0]] >> k) & 1 else 'F' for k in range(n)) + ('S' if acc[last[0]] else 'F')
                    assert got == s[i:i + n + 1], (nodes, fam, i, got)
                    for ch in got:
                        kind = 1 if ch == 'S' else 2
                        if kind == ck:
                            cr += 1
                        else:
                            cq, cp, cr, ck = cp, cr, 1, kind
                    i += n + 1
                    rounds += 1
        assert rounds > 500
