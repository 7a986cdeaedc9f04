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
