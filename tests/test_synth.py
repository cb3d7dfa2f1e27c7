import numpy as np


def test_sequences_are_seeded_and_quantised(pkg):
    a = pkg.synth.Sequence("spheres", 64, 48, n_frames=2, seed=5)
    b = pkg.synth.Sequence("spheres", 64, 48, n_frames=2, seed=5)
    d0, R0, t0 = a.frame(1)
    d1, R1, t1 = b.frame(1)
    assert np.array_equal(d0, d1) and np.array_equal(R0, R1) and np.array_equal(t0, t1)
    u = a.depth_u16(1)
    assert u.dtype == np.uint16 and np.array_equal(d0, u.astype(np.float32) * np.float32(0.001))
    assert (d0 == 0).any() and (d0 > 0).any()             # spheres + empty background


def test_tum_stream_is_all_valid_and_fr1xyz_like(pkg):
    s = pkg.synth.Sequence("tum", 160, 120, n_frames=40, seed=0)
    d, R, t = s.frame(7)
    assert (d > 0.5).all() and (d < 3.5).all()
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-5
    steps = [np.linalg.norm(s.pose(i + 1)[1] - s.pose(i)[1]) for i in range(39)]
    assert max(steps) < 0.02                                # <= ~1.4 cm per frame
    assert s.unit == np.float32(1.0 / 5000)


def test_kinect_noise_quantises_disparity(pkg):
    rng = np.random.default_rng(0)
    z = np.full((8, 8), 2.0)
    zn = pkg.synth.kinect_noise(z, rng)
    d = (3.0 - 1.0 / zn) / 2.85e-3
    assert np.abs(d - np.round(d)).max() < 1e-6


def test_look_at_is_a_rotation(pkg):
    R = pkg.synth.look_at([2.0, -1.0, 0.3])
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12 and abs(np.linalg.det(R) - 1) < 1e-12
    f = R[:, 2]
    assert np.allclose(f, -np.array([2.0, -1.0, 0.3]) / np.linalg.norm([2.0, -1.0, 0.3]))
