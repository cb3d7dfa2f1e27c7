"""PhotoBA (config C5): PhotometricOptimizer restated in the oracle and run on the GPU.
not gpu: known-answer behaviour of the oracle (energy minimal at the true poses, perturbation raises it,
         optimisation brings energy and poses back).
gpu:     energy / one pose step / one distance step from identical state at the 1e-4 bar (6 and 50 keyframes), and the
         full optimize() trajectory, against the oracle."""
import numpy as np
import pytest


def _scene(pkg, O, W=160, H=120, n=6, vs=0.02, perturb=True, trunc=5):
    seq = pkg.synth.Sequence("tum", W, H, n_frames=n, seed=0, noise=False)
    vs = np.float32(vs)
    T = np.float32(trunc) * vs
    frames = [seq.frame(i) for i in range(n)]
    imgs = np.stack([pkg.synth.render_color_bgr(seq, i) for i in range(n)])
    P = np.stack([pkg.synth.pose16(*seq.pose(i)) for i in range(n)])
    Pp = P.copy()
    if perturb:
        rng = np.random.default_rng(0)
        for i in range(1, n):
            xi = np.concatenate([0.004 * rng.standard_normal(3), 0.003 * rng.standard_normal(3)]).astype(np.float32)
            p7 = O.se3_exp_mul(xi, np.concatenate([P[i][:3, 3], O.R_to_quat(P[i][:3, :3])]))
            Pp[i][:3, :3] = O.quat_to_R(p7[3:])
            Pp[i][:3, 3] = p7[:3]
    return seq, vs, T, frames, imgs, P, Pp


def _oracle_map(O, seq, vs, T, frames):
    o = O.Oracle(vs, T, seq.W, seq.H, seq.K)
    for d, R, t in frames:
        o.update(d, R, t)
    return o


def test_oracle_photoba_known_answers(pkg, O):
    seq, vs, T, frames, imgs, P, Pp = _scene(pkg, O)
    n = len(frames)
    ba_gt = O.PhotoBA(_oracle_map(O, seq, vs, T, frames), imgs, P, np.arange(n))
    E_gt = ba_gt.energy()
    o = _oracle_map(O, seq, vs, T, frames)
    ba = O.PhotoBA(o, imgs, Pp, np.arange(n))
    E_pert = ba.energy()
    assert E_pert > 20 * E_gt                                   # wrong poses break photo-consistency
    conv, en = ba.optimize(6)
    assert en[0] == pytest.approx(E_pert) and en[-1] < 0.05 * E_pert and en[-1] < 2 * E_gt
    assert np.all(np.diff(en)[:4] < 0)                          # monotone at the start
    Pn = ba.poses()
    assert np.abs(Pn - P).max() < 0.3 * np.abs(Pp - P).max()    # poses move back towards the truth
    assert np.array_equal(Pn[:, 3], np.tile([0, 0, 0, 1], (n, 1)))
    for R in Pn[:, :3, :3]:
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-5


def test_oracle_photoba_ignores_invisible_keyframes(pkg, O):
    seq, vs, T, frames, imgs, P, Pp = _scene(pkg, O, n=4, perturb=False)
    o = _oracle_map(O, seq, vs, T, frames)
    e_all = O.PhotoBA(o, imgs, P, np.arange(4)).energy()
    e_none = O.PhotoBA(o, imgs, P, np.array([40, 41, 42, 43])).energy()     # bits never set in vis_
    assert e_all > 0 and e_none == 0.0


def test_oracle_photoba_trunc_l2_known_answers(pkg, O):
    """LossFunction::TRUNC_L2 (PhotometricOptimizer.cpp:364,:542): a keyframe whose intensity exceeds lambda in any channel is
    left out of a voxel's sums.  lambda above every intensity (images are in [0, 1]) = the default loss, bit for bit;
    lambda = 0 leaves out every keyframe that sees anything: the steps do nothing; in between the step differs."""
    seq, vs, T, frames, imgs, P, Pp = _scene(pkg, O)
    n = len(frames)

    def step(loss, lam):
        o = _oracle_map(O, seq, vs, T, frames)
        ba = O.PhotoBA(o, imgs, Pp, np.arange(n))
        if loss is not None:
            ba.set_loss(loss, lam)
        ba.solve_pose()
        ba.solve_dist()
        return ba.poses(), o.export()[1][:, 0].copy()

    p_def, d_def = step(None, 0)
    p_big, d_big = step(4, 2.0)
    assert np.array_equal(p_def, p_big) and np.array_equal(d_def, d_big)
    p_cau, d_cau = step(1, 0.01)                                  # any loss but TRUNC_L2 ignores lambda, as in the reference
    assert np.array_equal(p_def, p_cau) and np.array_equal(d_def, d_cau)
    p_zero, d_zero = step(4, 0.0)
    o0 = _oracle_map(O, seq, vs, T, frames)
    assert np.array_equal(p_zero, Pp) and np.array_equal(d_zero, o0.export()[1][:, 0])
    p_mid, d_mid = step(4, 0.5)
    assert np.abs(p_mid - p_def).max() > 1e-5 and np.abs(p_mid - Pp).max() > 1e-4


def _gpu_and_oracle_on_the_same_map(pkg, O, n=6, W=160, H=120, vs=0.02, trunc=5, cap=20):
    """Fuse on the GPU (with vis_), fuse in the oracle, then give the oracle the GPU's voxel values: both BA implementations
    start from identical maps (key sets and vis_ are bit-exact anyway, tests/test_gpu_parity.py)."""
    seq, vs, T, frames, imgs, P, Pp = _scene(pkg, O, n=n, W=W, H=H, vs=vs, trunc=trunc)
    o = _oracle_map(O, seq, vs, T, frames)
    g = pkg.GradSdf(vs, T, seq.W, seq.H, seq.K, capacity_log2=cap)
    g.enable_vis(64)
    for d, R, t in frames:
        g.update(d, R, t)
    keys, pay = g.export(sorted=True)
    assert o.set_payload(keys, pay) == 0 and o.count() == len(keys)
    return seq, g, o, imgs, P, Pp


@pytest.mark.gpu
@pytest.mark.parametrize("n,W,H,vs,trunc,cap", [(6, 160, 120, 0.02, 5, 20), (50, 160, 120, 0.02, 5, 20), (50, 640, 480, 0.01, 10, 22)],
                         ids=["6kf-160x120", "50kf-160x120", "C5-50kf-640x480-1cm"])
def test_gpu_photoba_steps_match_oracle(pkg, O, n, W, H, vs, trunc, cap):
    """Every sweep of PhotometricOptimizer against the oracle FROM IDENTICAL STATE (same voxel values, same poses), at the
    bar of BASELINE.json's north_star (1e-4 on distance and pose; energy to 1e-4 relative): getEnergy, one solvePose, one
    solveDist.  n = 50 keyframes is the size of BASELINE config C5; the last case is C5 as configured (640x480 frames on the
    1 cm / trunc 10 map of the bench stream, ~1.6 x 10^6 voxels; the oracle's part takes about a minute)."""
    seq, g, o, imgs, P, Pp = _gpu_and_oracle_on_the_same_map(pkg, O, n=n, W=W, H=H, vs=vs, trunc=trunc, cap=cap)
    idx = np.arange(n)
    ba = O.PhotoBA(o, imgs, Pp, idx)
    g.ba_setup(imgs, Pp, idx)
    # the reference adds ~10^5 .. 10^6 float terms into one float in hash-map order: its own value is uncertain at the 1e-3
    # level; the GPU sweep (double partial sums) is held to the order-independent sum of the same terms
    e_o, e_g = ba.energy_f64(), g.ba_energy()
    assert e_o > 0 and e_g == pytest.approx(e_o, rel=1e-4)
    assert ba.energy() == pytest.approx(e_o, rel=3e-3)
    ba.solve_pose()
    g.ba_solve_pose()
    step = np.abs(ba.poses() - Pp).max()
    assert step > 1e-3                                           # the step is real ...
    assert np.abs(g.ba_poses() - ba.poses()).max() < 1e-4         # ... and the same (measured: 4e-6)
    # the distance step from identical state again: the oracle takes the GPU's poses
    Pg = g.ba_poses()
    ba2 = O.PhotoBA(o, imgs, Pg, idx)
    assert g.ba_energy() == pytest.approx(ba2.energy_f64(), rel=1e-4)
    before = g.export(sorted=True)[1][:, 0].copy()
    ba2.solve_dist()
    g.ba_solve_dist()
    ko, po = o.export()
    kg, pg = g.export(sorted=True)
    assert np.array_equal(kg, ko)
    assert np.abs(pg[:, 0] - before).max() > 1e-4                # distances did move ...
    assert np.abs(pg[:, 0] - po[:, 0]).max() < 1e-6              # ... identically (measured: 7e-9)
    assert g.ba_energy() == pytest.approx(ba2.energy_f64(), rel=1e-4)
    g.close()


@pytest.mark.gpu
def test_gpu_photoba_trunc_l2_matches_oracle(pkg, O):
    """The TRUNC_L2 gates of solvePose / solveDist on the GPU against the oracle, from identical state."""
    n = 6
    seq, g, o, imgs, P, Pp = _gpu_and_oracle_on_the_same_map(pkg, O, n=n)
    idx = np.arange(n)
    ba = O.PhotoBA(o, imgs, Pp, idx)
    ba.set_loss(4, 0.5)
    g.ba_setup(imgs, Pp, idx)
    g.ba_set_loss(4, 0.5)
    ba.solve_pose()
    g.ba_solve_pose()
    assert np.abs(ba.poses() - Pp).max() > 1e-4
    assert np.abs(g.ba_poses() - ba.poses()).max() < 1e-4
    # the gate really was active: the untruncated step is another one
    ref = O.PhotoBA(o, imgs, Pp, idx)
    ref.solve_pose()
    assert np.abs(ref.poses() - ba.poses()).max() > 1e-5
    ba2 = O.PhotoBA(o, imgs, g.ba_poses(), idx)
    ba2.set_loss(4, 0.5)
    ba2.solve_dist()
    g.ba_solve_dist()
    ko, po = o.export()
    kg, pg = g.export(sorted=True)
    assert np.array_equal(kg, ko) and np.abs(pg[:, 0] - po[:, 0]).max() < 1e-6
    with pytest.raises(Exception):
        g.ba_set_loss(7, 0.5)
    g.close()


@pytest.mark.gpu
def test_gpu_photoba_optimize_matches_oracle(pkg, O):
    """The whole optimize() (:611-662) from identical state: same number of steps, the energy after every pose / distance
    step within 3e-3 (the oracle's optimize() uses the reference's single-float energy sum, uncertain at that level -- see
    the step test; single sweeps agree to 1e-5), final poses within 1e-4, and the optimisation does its job (energy
    down, poses back towards the truth)."""
    seq, g, o, imgs, P, Pp = _gpu_and_oracle_on_the_same_map(pkg, O, n=6)
    n = 6
    ba = O.PhotoBA(o, imgs, Pp, np.arange(n))
    g.ba_setup(imgs, Pp, np.arange(n))
    conv_g, en_g = g.ba_optimize(5)
    conv_o, en_o = ba.optimize(5)
    assert conv_g == conv_o and len(en_g) == len(en_o) >= 5
    assert np.allclose(en_g, en_o, rtol=3e-3)
    assert np.abs(g.ba_poses() - ba.poses()).max() < 1e-4
    assert en_g[-1] < 0.3 * en_g[0]
    assert np.abs(g.ba_poses() - P).max() < 0.5 * np.abs(Pp - P).max()
    g.close()


@pytest.mark.gpu
def test_gpu_photoba_pose_sweep_with_the_energy_sweeps_means_is_bit_identical(pkg, O, monkeypatch):
    """Round 6: inside optimize() the pose sweep reads every gated voxel's mean intensity / keyframe set from what the energy sweep
    in front of it left behind (k_ba_energy<true> -> k_ba_pose<true, 4>) instead of repeating that loop.  Same arithmetic in the
    same order: the pose step must be the one of GSDF_BA_MEAN_CACHE=0 bit for bit.  Compared on ONE table (another copy of the
    same map numbers its slots differently, and the sweeps add in slot order): energy + pose step with the cache (mode 2 lets the
    stand-alone gsdf_ba_solve_pose use what the energy sweep just wrote), poses set back by a second gsdf_ba_setup -- a pose step
    does not touch the map --, energy + pose step without it."""
    seq, g, o, imgs, P, Pp = _gpu_and_oracle_on_the_same_map(pkg, O, n=6)
    out = []
    for mode in ("2", "0", "2"):
        monkeypatch.setenv("GSDF_BA_MEAN_CACHE", mode)
        g.ba_setup(imgs, Pp, np.arange(6))
        e0 = g.ba_energy()
        g.ba_solve_pose()
        out.append((np.float32(e0), g.ba_poses().copy(), np.float32(g.ba_energy())))
    g.close()
    assert np.abs(out[0][1] - Pp).max() > 1e-4                                  # the step is real
    for a, b in ((out[0], out[1]), (out[0], out[2])):
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2] == b[2]


@pytest.mark.gpu
def test_photometric_optimizer_cpp_facade_executed(pkg, O, tmp_path):
    """host/PhotometricOptimizer.h (the facade for ps_optimizer/PhotometricOptimizer.h:68-186) EXECUTED from C++ --
    host/photoba_selftest fuses the frames through MapGradPixelSdf::update (vis_ on), then setImages / setPoses / setKeyframes,
    getEnergy, solvePose, solveDist, optimize -- against the same steps driven through the C-ABI from here: same energies,
    same poses (both sides are the GPU engine; sums of a sweep are order-free up to float noise)."""
    import os
    import subprocess
    from conftest import ROOT
    host = os.path.join(ROOT, "gradient-sdf_amd", "host")
    subprocess.check_call(["make", "-C", host, "-s"])
    n, W, H, vsf, trunc = 6, 160, 120, 0.02, 5
    seq, vs, T, frames, imgs, P, Pp = _scene(pkg, O, W=W, H=H, n=n, vs=vsf, trunc=trunc)
    d = tmp_path
    np.asarray(seq.K, np.float32).reshape(9).tofile(d / "K.bin")
    np.stack([f[0] for f in frames]).astype(np.float32).tofile(d / "depth.bin")
    imgs.astype(np.float32).tofile(d / "images.bin")
    # the facade takes its fusion poses as Matrix4f -> SE3 (quaternion) -> R, like main_photo_ba.cpp; feed the C-ABI side the same R
    P.astype(np.float32).tofile(d / "poses_true.bin")
    Pp.astype(np.float32).tofile(d / "poses_start.bin")
    out = subprocess.run([os.path.join(host, "photoba_selftest"), str(d), str(W), str(H), str(n), repr(vsf), str(trunc)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "photoba_selftest: OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    lines = {}
    for ln in out.stdout.splitlines():
        k, _, rest = ln.partition(" ")
        lines.setdefault(k, []).append(rest.split())
    g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=20)
    g.enable_vis(64)
    for (dep, R, t), Pi in zip(frames, P):
        Rq = O.quat_to_R(O.R_to_quat(Pi[:3, :3].astype(np.float32)))          # SE3(Matrix4f).rotationMatrix()
        g.update(dep, Rq, Pi[:3, 3].astype(np.float32))
    assert int(lines["voxels"][0][0]) == g.count() and int(lines["voxels"][0][2]) == n
    idx = np.arange(n)
    g.ba_setup(imgs, Pp, idx)
    e0 = g.ba_energy()
    g.ba_solve_pose()
    e1 = g.ba_energy()
    p1 = g.ba_poses()
    g.ba_solve_dist()
    e2 = g.ba_energy()
    got = [float(v) for v in lines["steps"][0]]
    assert got == pytest.approx([e0, e1, e2], rel=1e-5) and e1 < 0.5 * e0
    pf = np.array([[float(v) for v in row[1:]] for row in lines["pose_after_step"]]).reshape(n, 4, 4)
    assert np.abs(pf - p1).max() < 1e-6 and np.abs(pf - Pp).max() > 1e-3       # solvePose() moved the poses, and download() brought them back
    conv, en = g.ba_optimize(4)
    row = lines["optimize"][0]
    assert int(row[0]) == int(conv) and [float(v) for v in row[1:]] == pytest.approx(list(en), rel=1e-4)
    pe = np.array([[float(v) for v in r[1:]] for r in lines["pose_final"]]).reshape(n, 4, 4)
    assert np.abs(pe - g.ba_poses()).max() < 1e-5
    g.close()
