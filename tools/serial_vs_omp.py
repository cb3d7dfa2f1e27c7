#!/usr/bin/env python3
"""What would the reference's OWN two builds do against each other?  

The lockstep tests hold the GPU engine to the serial oracle frame by frame and tolerate a few "flips" -- frames on which the
two sides' stop tests fall differently -- with the argument that a chaotic frame's decision is made by last-bit noise, "as it
would be between the reference's own serial and OMP builds" (RigidPointOptimizerOmp.cpp:68-69: four per-thread partial sums
instead of one sequential sum; MapGradPixelSdfOmp.cpp:112: fusion order left to the scheduler).  This tool MEASURES that: the
serial oracle against its OMP-structured variant in the same harness (tests/lockstep.py), same streams, same rules.

    python tools/serial_vs_omp.py [bench|c1] [frames] | free    -> profiles/r05_serial_vs_omp.txt (appended to stdout)

CPU only; the bench stretch (48 frames of 640x480) takes a few minutes on 4 cores."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402
from lockstep import lockstep, flip_classes, OmpOracle  # noqa: E402


def run(which, n, threads=4):
    pkg, O = graft.package(), graft.oracle_module()
    W, H = 640, 480
    vs = np.float32(0.01)
    T = np.float32(10) * vs
    if which == "bench":
        seq = pkg.synth.Sequence("tum", W, H, n_frames=n, seed=0)
        d0, R0, t0 = seq.frame(0)
        pose = np.concatenate([t0, O.R_to_quat(R0)]).astype(np.float32)
        R0 = O.quat_to_R(pose[3:])
        frames = ((i, seq.frame(i)[0]) for i in range(1, n))
    else:
        seq = pkg.synth.Sequence("spheres", W, H, n_frames=n, seed=0, step_deg=0.5)
        depth = lambda i: seq.depth_u16(i).astype(np.float32) * np.float32(0.001)
        d0, R0, t0 = depth(0), np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
        pose = np.array([0, 0, 0, 0, 0, 0, 1], np.float32)
        frames = ((i, depth(i)) for i in range(1, n))
    o = O.Oracle(vs, T, W, H, seq.K, threads=1)
    m = O.Oracle(vs, T, W, H, seq.K, threads=threads)
    g = OmpOracle(m)
    o.update(d0, R0, t0)
    g.update(d0, R0, t0)
    t = time.time()
    n_conv, n_long, flips = lockstep(O, g, o, frames, pose)
    short, long_ = flip_classes(flips)
    ko, po = o.export()
    km, pm = m.export()
    same_keys = ko.shape == km.shape and np.array_equal(ko, km)
    print("%s stream, %d frames, serial oracle vs OMP-structured oracle (%d threads), lockstep from identical state:" % (which, n, threads))
    print("  converged %d, long frames (> 6 passes) %d, flips %d = %d on short frames + %d on long frames   [%.0f s]" % (
        n_conv, n_long, len(flips), short, long_, time.time() - t))
    for f in flips:
        print("    frame %d: serial %s after %d passes, OMP %s after %d passes, serial |xi|^2 at the earlier end %.3e" % (
            f[0], "converged" if f[1] else "not converged", f[2], "converged" if f[3] else "not converged", f[4], f[5]))
    print("  maps: key sets %s, %d voxels; max |dist| difference %.2e" % (
        "equal" if same_keys else "DIFFER", ko.shape[0], float(np.abs(po[:, 0] - pm[:, 0]).max()) if same_keys else float("nan")))
    return n_conv, n_long, flips


def free_running(n=6, threads=4):
    """The free-running comparison of test_track_and_fuse_stream_matches_oracle_loop (6 sphere frames, 640x480): each build
    tracks and fuses on its OWN poses and map -- how far do the reference's two builds drift apart per frame?"""
    pkg, O = graft.package(), graft.oracle_module()
    W, H = 640, 480
    seq = pkg.synth.Sequence("spheres", W, H, n_frames=n, seed=0, step_deg=0.5)
    vs = np.float32(0.01)
    T = np.float32(10) * vs
    fr = [seq.frame(i) for i in range(n)]
    a, b = O.Oracle(vs, T, W, H, seq.K, threads=1), O.Oracle(vs, T, W, H, seq.K, threads=threads)
    d0, R0, t0 = fr[0]
    p = np.concatenate([t0, O.R_to_quat(R0)]).astype(np.float32)
    R0q = O.quat_to_R(p[3:])
    a.update(d0, R0q, t0)
    b.update(d0, R0q, t0, omp=True)
    pa, pb = p.copy(), p.copy()
    print("free-running spheres stream, %d frames, serial vs OMP-structured oracle, each on its own poses and map:" % n)
    for i in range(1, n):
        ca, pa, ua, _, _ = a.track(fr[i][0], pa)
        cb, pb, ub, _, _ = b.track(fr[i][0], pb, omp=True)
        if ca:
            a.update(fr[i][0], O.quat_to_R(pa[3:]), pa[:3])
        if cb:
            b.update(fr[i][0], O.quat_to_R(pb[3:]), pb[:3], omp=True)
        print("  frame %d: serial %s / %d passes, OMP %s / %d passes, max |pose difference| %.2e" % (i, ca, ua, cb, ub, float(np.abs(pa - pb).max())))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "free":
        free_running()
        sys.exit(0)
    which = sys.argv[1] if len(sys.argv) > 1 else "bench"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else (48 if which == "bench" else 30)
    run(which, n)
