"""The C-ABI library loads without a GPU and exports every symbol include/sca_hip.h declares; the ctypes table in
sca_amd/_lib.py covers exactly that set.  No compute calls here."""
import os

import pytest
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
    _lib.lib().sca_default_params_v2(C.byref(p), C.sizeof(p))
    # agent.py:27-36, config.py:2-3
    assert (p.neighbor_dist, p.max_neighbors, p.time_step, p.time_horizon, p.max_speed) == (10.0, 16, 0.1, 10.0, 1.0)
    assert p.max_heading_change == math.pi / 4 and p.near_goal_threshold == 0.5
    assert p.dt_nominal == 0.1                                    # agent.py:41
    assert C.sizeof(_lib.Params) == 64 and _lib.lib().sca_version() >= 101 and p.struct_bytes == 64


def test_version_100_callers_keep_their_56_byte_struct():
    """sca_params grew a trailing dt_nominal in version 101.  The version-100 entry point must not write beyond the 56 bytes
    such a caller allocated, and marks the struct (struct_bytes = 0, where `reserved` sat) so that sca_create does not read beyond them
    either (it integrates with time_step then, as version 100 did)."""
    import ctypes as C
    from sca_amd import _lib
    buf = (C.c_ubyte * 72)(*([0xA5] * 72))
    _lib.lib().sca_default_params(C.cast(buf, C.c_void_p))
    assert bytes(buf[56:]) == bytes([0xA5] * 16)                   # nothing written behind the old struct
    p = _lib.Params.from_buffer_copy(bytes(buf[:64]))
    assert (p.neighbor_dist, p.max_neighbors, p.time_step, p.struct_bytes) == (10.0, 16, 0.1, 0)
    # a version-100 struct with garbage where dt_nominal would be: the range check must not see it
    ctx = C.c_void_p()
    p.time_step = 0.2
    rc = _lib.lib().sca_create(C.byref(p), 0, 8, 1, C.byref(ctx))
    assert 'dt_nominal' not in _lib.lib().sca_last_error(ctx).decode(), rc
    _lib.lib().sca_destroy(ctx)


@pytest.mark.parametrize('field,value', [('neighbor_dist', 0.0), ('neighbor_dist', float('nan')), ('time_step', -0.1), ('time_horizon', 0.0),
                                         ('max_speed', float('inf')), ('dt_nominal', 0.0), ('max_neighbors', 0), ('max_neighbors', 17),
                                         ('max_heading_change', -0.1), ('max_heading_change', 3.2), ('near_goal_threshold', -1.0)])
def test_create_refuses_parameters_the_kernels_were_not_built_for(field, value):
    """sca_create checks sca_params before it touches the device: SCA_ERR_ARG and a message that names the field."""
    import ctypes as C
    from sca_amd import _lib
    L = _lib.lib()
    p = _lib.Params()
    L.sca_default_params_v2(C.byref(p), C.sizeof(p))
    setattr(p, field, value)
    ctx = C.c_void_p()
    assert L.sca_create(C.byref(p), 0, 8, 1, C.byref(ctx)) == -1          # SCA_ERR_ARG
    assert field in L.sca_last_error(ctx).decode()
    L.sca_destroy(ctx)


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


def test_python_constants_mirror_the_header():
    """solver.py's NBR_* / POL_* / FORM_* are the header's SCA_NBR_* / SCA_POLICY_* / SCA_FORM_* values (a drop-in host reads
    either)."""
    from sca_amd import solver as S
    txt = open(os.path.join(ROOT, 'include', 'sca_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    hdr = {k: int(v) for k, v in re.findall(r'\b(SCA_[A-Z0-9_]+)\s*=\s*(-?\d+)', txt)}
    hdr.update({k: int(v) for k, v in re.findall(r'#define\s+(SCA_[A-Z0-9_]+)\s+(-?\d+)\b', txt)})
    pairs = {'SCA_NBR_KDTREE': S.NBR_KDTREE, 'SCA_NBR_GRID': S.NBR_GRID, 'SCA_NBR_KDTREE_HOSTBUILD': S.NBR_KDTREE_HOSTBUILD, 'SCA_NBR_AUTO': S.NBR_AUTO,
             'SCA_POLICY_SCA': S.POL_SCA, 'SCA_POLICY_RVO3D': S.POL_RVO3D, 'SCA_POLICY_SRVO3D': S.POL_SRVO3D,
             'SCA_POLICY_ORCA3D': S.POL_ORCA3D, 'SCA_POLICY_ORCA3D_LP': S.POL_ORCA3D_LP, 'SCA_POLICY_RVO3D_DUBINS': S.POL_RVO3D_DUBINS,
             'SCA_FORM_SOLVE_SPLIT': S.FORM_SOLVE_SPLIT, 'SCA_FORM_TRACK_FUSED': S.FORM_TRACK_FUSED,
             'SCA_FORM_REPLAN_LANE': S.FORM_REPLAN_LANE, 'SCA_FORM_REPLAN_FEW': S.FORM_REPLAN_FEW, 'SCA_FORM_LP_LANE': S.FORM_LP_LANE, 'SCA_FORM_SOLVE_FB': S.FORM_SOLVE_FB, 'SCA_FORM_ACTION_FB': S.FORM_ACTION_FB, 'SCA_FORM_AUTO_TAIL': S.FORM_AUTO_TAIL,
             'SCA_MAX_NEIGHBORS': S.K}
    for name, val in pairs.items():
        assert hdr.get(name) == val, (name, hdr.get(name), val)
