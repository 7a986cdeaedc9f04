"""sca_amd -- MI355X-native batched SCA / RVO3D / S-RVO3D / ORCA3D velocity solver.

Drop-in for the per-agent hot path of wuuya1/SCA (mamp.policies + neighbour search + MACAEnv step loop):
hand-written HIP kernels for gfx950 behind the C-ABI in include/sca_hip.h.
"""
__version__ = '0.1.0'
