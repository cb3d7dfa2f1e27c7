"""PhotoBA (config C5): PhotometricOptimizer restated in the oracle and run on the GPU.
not gpu: known-answer behaviour of the oracle (energy minimal at the true poses, perturbation raises it,
         optimisation brings energy and poses back).
gpu:     energy / one pose step / one distance step / full optimize against the oracle."""
import numpy as np
import pytest


def _scene(pkg, O, W=160, H=120, n=6, vs=0.02, perturb=True):
    seq = pkg.synth.Sequence("tum", W, H, n_frames=n, seed=0, noise=False)
    vs = np.float32(vs)
    T = np.float32(5) * vs
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


@pytest.mark.gpu
def test_gpu_photoba_matches_oracle(pkg, O):
    seq, vs, T, frames, imgs, P, Pp = _scene(pkg, O)
    n = len(frames)
    o = _oracle_map(O, seq, vs, T, frames)
    g = pkg.GradSdf(vs, T, seq.W, seq.H, seq.K, capacity_log2=19)
    g.enable_vis(32)
    for d, R, t in frames:
        g.update(d, R, t)
    ba = O.PhotoBA(o, imgs, Pp, np.arange(n))
    g.ba_setup(imgs, Pp, np.arange(n))
    e_o, e_g = ba.energy(), g.ba_energy()
    assert e_g == pytest.approx(e_o, rel=2e-3)
    ba.solve_pose()
    g.ba_solve_pose()
    assert np.abs(g.ba_poses() - ba.poses()).max() < 2e-4
    e_o, e_g = ba.energy(), g.ba_energy()
    assert e_g == pytest.approx(e_o, rel=5e-3)
    ba.solve_dist()
    g.ba_solve_dist()
    ko, po = o.export()
    kg, pg = g.export()
    assert np.array_equal(kg, ko)
    assert np.abs(pg[:, 0] - po[:, 0]).max() < 2e-4             # distances after the Gauss-Newton step
    e_o, e_g = ba.energy(), g.ba_energy()
    assert e_g == pytest.approx(e_o, rel=1e-2)
    conv_g, en_g = g.ba_optimize(5)
    conv_o, en_o = ba.optimize(5)
    m = min(len(en_g), len(en_o))
    assert m >= 5 and abs(len(en_g) - len(en_o)) <= 2
    assert np.allclose(en_g[:m], en_o[:m], rtol=2e-2)           # the whole energy trajectory of optimize()
    assert en_g[-1] < 0.3 * en_g[0]
    assert np.abs(g.ba_poses() - P).max() < 0.5 * np.abs(Pp - P).max()
    g.close()
