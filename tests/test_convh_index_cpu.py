"""The slab convolution kernel's index scheme, replayed lane by lane on the CPU (tools/emulate_convh.py) against a direct
convolution: padded position grid, 2-D tiles with the fused 2x2 max-pool, persistent workgroups walking over tiles.  No GPU: this
pins the integer bookkeeping of csrc/ssdhip_convh.hip (slab <-> pixel map, chunk permutation, tap displacements, weight ring,
fragment <-> output map); the kernel itself is compared bit for bit with the implicit-GEMM kernel in tests/test_conv_gpu.py."""
import pytest

from tools.emulate_convh import emulate


@pytest.mark.parametrize("case", [
    (2, 5, 6, 128, 128, 4, 5, None, 0, False),        # position grid, one workgroup per tile
    (40, 6, 7, 128, 128, 4, 5, 8, 0, False),          # position grid, persistent workgroups: nine tiles on eight workgroups
    (1, 9, 13, 128, 128, 4, 6, None, 4, True),        # 16 x 16 tiles, fused pooling, odd map
])
def test_slab_index_scheme_reproduces_the_convolution(case):
    B, H, W, Cin, Cout, NW, SPW, G, csh, pool = case
    good, y, ref = emulate(B, H, W, Cin, Cout, NW, SPW, G=G, csh=csh, pool=pool)
    assert good, "%d outputs wrong" % int((y != ref).sum())
