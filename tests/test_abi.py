"""The C-ABI library loads without a GPU and exports every symbol include/sca_hip.h declares; the ctypes table in
sca_amd/_lib.py covers exactly that set.  No compute calls here."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, 'include', 'sca_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return set(re.findall(r'\b(sca_[a-z0-9_]+)\s*\(', txt))


def test_library_exports_header_symbols():
    from sca_amd import _lib
    L = _lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), s
    assert syms == set(_lib.SIGNATURES), syms ^ set(_lib.SIGNATURES)
    assert L.sca_version() >= 100


def test_default_params_are_the_reference_constants():
    import ctypes as C
    import math
    from sca_amd import _lib
    p = _lib.Params()
    _lib.lib().sca_default_params(C.byref(p))
    # agent.py:27-36, config.py:2-3
    assert (p.neighbor_dist, p.max_neighbors, p.time_step, p.time_horizon, p.max_speed) == (10.0, 16, 0.1, 10.0, 1.0)
    assert p.max_heading_change == math.pi / 4 and p.near_goal_threshold == 0.5


def test_no_cpu_path_without_gpu():
    """On a machine without a GPU the product must fail loudly instead of computing on the CPU."""
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from sca_amd import solver as S
    with pytest.raises(S.ScaError):
        S.BatchedSolver(max_agents=8)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'sca_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dirpath, f)).read()
                for pat in ('liboracle', 'from oracle', 'import oracle', 'orc_', 'oracle.py'):
                    assert pat not in txt, (f, pat)
