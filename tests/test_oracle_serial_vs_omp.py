"""What the reference's OWN two builds would do against each other -- the measurement behind the flip limits of the lockstep
tests (tests/test_gpu_parity.py: test_c1_every_frame_from_identical_state, test_bench_stream_every_frame_from_identical_state).

The reference ships a serial and an OMP build of the same algorithm (RigidPointOptimizer.cpp vs RigidPointOptimizerOmp.cpp:68-69:
four per-thread partial sums added at the end instead of one sequential sum; MapGradPixelSdf.cpp vs MapGradPixelSdfOmp.cpp:112:
fusion order left to the scheduler, so the running mean of :111 rounds differently).  Their normal equations differ in the last
bits exactly as the GPU's (pairwise float / f64 group sums) differ from the serial oracle's.  Here the serial oracle and its
OMP-structured variant go through the SAME lockstep harness as the GPU engine (tests/lockstep.py, same rules, same streams):
the number of frames whose stop test falls differently ("flips") between the reference's own two builds is the yardstick the
GPU is held to -- GPU-vs-serial flips <= serial-vs-OMP flips + 1, per stream.

Measured (`python tools/serial_vs_omp.py bench|c1`):
  round 5 (profiles/r05_serial_vs_omp.txt):
    C1 (30 sphere frames, 640x480):   1 flip  (0 on short frames, 1 on a frame that runs > 6 passes), 1 long frame
    bench stream (48 frames, 640x480): 2 flips (1 short: |xi|^2 = 1.012e-6 at the threshold; 1 long: 25 passes vs 12), 12 long frames
  round 6, after tsdf() and n_sq were restated (profiles/r06_serial_vs_omp.txt):
    C1: 2 flips (1 short: |xi|^2 = 9.18e-7; 1 long: 11 passes vs 9), 2 long frames
    bench stream: 2 flips (1 short: 1.018e-6; 1 long: 25 passes vs 24), 12 long frames
  The limits below keep the SMALLER of the two measurements per stream (C1: 1), so the GPU is held to <= 2 / <= 3 flips.
The fused maps of the two builds keep identical key sets; dist differs by <= 4e-8 (the running mean's order)."""
import os

import numpy as np
import pytest

from lockstep import lockstep, flip_classes, OmpOracle

# what the lockstep GPU tests import as their limits: flips measured between the reference's two builds (+ 1 allowed on top)
SERIAL_VS_OMP_FLIPS = {"c1": 1, "bench": 2}


def _run(pkg, O, which, n):
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
    m = O.Oracle(vs, T, W, H, seq.K, threads=4)
    g = OmpOracle(m)
    o.update(d0, R0, t0)
    g.update(d0, R0, t0)
    n_conv, n_long, flips = lockstep(O, g, o, frames, pose)
    ko, po = o.export()
    km, pm = m.export()
    assert np.array_equal(ko, km)                             # the two builds' maps: same voxels ...
    assert np.abs(po[:, 0] - pm[:, 0]).max() < 1e-6           # ... dist equal up to the running mean's summation order
    return n_conv, n_long, flips


def test_serial_vs_omp_c1(pkg, O):
    """C1 at its length: every frame from identical state, serial build against OMP build (21 s on 4 cores)."""
    n_conv, n_long, flips = _run(pkg, O, "c1", 30)
    short, long_ = flip_classes(flips)
    assert n_conv >= 20
    # the number the GPU test's limit is derived from (the OMP build's fusion order is up to the scheduler, so the count may
    # move by one between runs; a short-frame flip has to sit at the threshold, which the harness checks)
    assert len(flips) <= SERIAL_VS_OMP_FLIPS["c1"] + 1, flips


@pytest.mark.slow
@pytest.mark.skipif(os.environ.get("GSDF_SKIP_SLOW") == "1", reason="GSDF_SKIP_SLOW=1")
def test_serial_vs_omp_bench_stream(pkg, O):
    """The 48-frame bench stretch of test_bench_stream_every_frame_from_identical_state, serial build against OMP build
    (~3 minutes on 4 cores: a quarter of the frames run all 25 passes, three times each)."""
    n_conv, n_long, flips = _run(pkg, O, "bench", 48)
    short, long_ = flip_classes(flips)
    assert n_conv >= 25 and n_long >= 3, (n_conv, n_long)
    assert 1 <= len(flips) <= SERIAL_VS_OMP_FLIPS["bench"] + 1, flips   # the reference's own builds DO decide frames differently
