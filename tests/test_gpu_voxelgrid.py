"""GPU parity of the VoxelGrid downsample (voxelgrid.hip through the C-ABI) against the oracle: float
centroids bit-exact (same float operations in the same order), output order and count identical; and
the fused downsample + ScanContext build against oracle downsample + oracle build."""
import numpy as np
import pytest

from test_oracle_voxelgrid import _cloud

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vgmod():
    from navtech_radar_slam_amd import _rsx, voxelgrid
    assert _rsx.device_count() >= 1
    return voxelgrid


@pytest.mark.parametrize("seed,n,leaf", [(1, 700, 0.4), (2, 4000, 0.4), (3, 120000, 0.4), (4, 5000, 0.2), (5, 5000, 2.5)])
def test_bit_exact(vgmod, oracle, seed, n, leaf):
    p = _cloud(seed, n)
    p[n // 2, 1] = np.nan
    vg = vgmod.VoxelGrid(leaf=leaf)
    got = vg.filter(p)
    want, ov = oracle.voxelgrid_filter(p, leaf)
    assert not ov and got.shape == want.shape and np.array_equal(got, want)
    assert np.array_equal(vg.filter(p), got)                        # same handle, same bytes
    got3 = vg.filter(p[:, :3])
    assert np.array_equal(got3[:, :3], want[:, :3]) and not got3[:, 3].any()


def test_edge_cases(vgmod, oracle):
    vg = vgmod.VoxelGrid()
    assert len(vg.filter(np.zeros((0, 4), np.float32))) == 0
    one = np.array([[1.0, 2.0, 3.0, 9.0]], dtype=np.float32)
    assert np.array_equal(vg.filter(one), one)
    assert np.array_equal(vg.filter(np.tile(one, (70, 1))), oracle.voxelgrid_filter(np.tile(one, (70, 1)))[0])
    assert len(vg.filter(np.full((4, 4), np.nan, dtype=np.float32))) == 0
    far = np.array([[0, 0, 0, 1], [3e4, 3e4, 3e3, 2], [1, 1, 1, 3]], dtype=np.float32)
    tiny = vgmod.VoxelGrid(leaf=0.01)
    want, ov = oracle.voxelgrid_filter(far, 0.01)
    assert ov and np.array_equal(tiny.filter(far), want)               # overflow guard: input unchanged
    from navtech_radar_slam_amd._rsx import RsxError
    with pytest.raises(RsxError):
        vgmod.VoxelGrid(leaf=0.0).filter(one)


def test_downsample_then_build_matches_oracle(vgmod, oracle):
    from navtech_radar_slam_amd import scancontext, synth
    vg = vgmod.VoxelGrid(leaf=0.4)
    g = scancontext.SCManager()
    rng = np.random.default_rng(3)
    for i in range(12):
        c = synth.radar_cloud(rng, n_points=int(rng.integers(500, 4000)), binary_z=(i % 2 == 0))
        dense = np.concatenate([c, c + rng.normal(0, 0.05, c.shape).astype(np.float32)]).astype(np.float32)  # crowded voxels
        assert g.makeAndSaveScancontextAndKeysDownsampled(dense, vg) == i
        ds, _ = oracle.voxelgrid_filter(dense, 0.4)
        want = oracle.make_scancontext(ds)
        assert np.array_equal(g.descriptor(i), want), f"keyframe {i}"
    assert len(g) == 12
