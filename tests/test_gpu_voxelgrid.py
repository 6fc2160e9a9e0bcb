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


def test_large_clouds_edge_cases(vgmod, oracle):
    """the cooperative kernel (more than 4096 points): the overflow guard, non-finite points scattered through the cloud, one
    crowded voxel that spans workgroup tiles, every point in ONE voxel, sizes around the tile boundaries"""
    rng = np.random.default_rng(11)
    far = _cloud(6, 9000)
    far[17, :3] = [3e4, 3e4, 3e3]
    tiny = vgmod.VoxelGrid(leaf=0.01)
    want, ov = oracle.voxelgrid_filter(far, 0.01)
    assert ov and np.array_equal(tiny.filter(far), want)               # overflow guard: input unchanged
    vg = vgmod.VoxelGrid(leaf=0.4)
    p = _cloud(7, 30000)
    p[rng.choice(30000, 3000, replace=False), rng.integers(0, 3, 3000)] = np.float32(np.nan)
    p[5, 0] = np.float32(np.inf)
    want, ov = oracle.voxelgrid_filter(p, 0.4)
    assert not ov and np.array_equal(vg.filter(p), want)
    crowd = _cloud(8, 20000)
    crowd[2000:9000, :3] = np.float32([10.05, -3.02, 0.1]) + rng.uniform(0, 0.3, (7000, 3)).astype(np.float32)  # 7000 points, one voxel
    want, _ = oracle.voxelgrid_filter(crowd, 0.4)
    assert np.array_equal(vg.filter(crowd), want)
    one = np.tile(np.float32([[1.0, 2.0, 3.0, 0.5]]), (5000, 1)) + rng.uniform(0, 0.01, (5000, 4)).astype(np.float32)
    want, _ = oracle.voxelgrid_filter(one, 0.4)
    got = vg.filter(one)
    assert len(want) == 1 and np.array_equal(got, want)
    assert len(vg.filter(np.full((5000, 4), np.nan, dtype=np.float32))) == 0
    for n in (4097, 5120, 5121, 131072, 131073):
        q = _cloud(n, n)
        want, _ = oracle.voxelgrid_filter(q, 0.4)
        assert np.array_equal(vg.filter(q), want), n


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
