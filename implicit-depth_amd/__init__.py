"""MI355X-native cost-volume hot path of nianticlabs/implicit-depth.

The heavy lifting lives in ``csrc/`` (hand-written gfx950 HIP kernels behind the C ABI of
``include/idh.h``); the Python modules here mirror the reference's ``nn.Module`` interface
for this path so they can be swapped into ``BDModel`` / ``DepthModel`` like the reference's
own ``to_fast()`` precedent (reference ``test_bd.py:80-81``).
"""
__version__ = "0.1.0"
