"""The lockstep harness: two implementations of the frame loop (main_scan_3d.cpp:255-270) kept on IDENTICAL state, frame by
frame.  Used by the GPU parity tests (engine against the serial oracle, tests/test_gpu_parity.py) and by the CPU measurement
of what the reference's own two builds would do (serial oracle against its OMP-structured variant,
tests/test_oracle_serial_vs_omp.py): the same rules judge both pairs."""
import numpy as np

TOL = 1e-4


class OmpOracle:
    """The oracle's OMP-structured variants (MapGradPixelSdfOmp.cpp:82,112: parallel-for fusion inside `critical`;
    RigidPointOptimizerOmp.cpp:68-69: 4-thread tracker with per-thread partial sums) behind the engine's call signatures."""

    def __init__(self, oracle):
        self.o = oracle

    def track(self, depth, pose7, iters=25):
        conv, pose, used, _, _ = self.o.track(depth, pose7, iters=iters, omp=True)
        return conv, pose, used

    def update(self, depth, R, t):
        return self.o.update(depth, R, t, omp=True)

    def export(self, sorted=True):
        return self.o.export()


def lockstep(O, g, o, frames, pose):
    """`g` (the implementation under test: .track(depth, pose7, iters=) -> (converged, pose7, passes), .update(depth, R, t)) and
    the serial oracle `o` in LOCKSTEP over `frames` ((index, depth image) pairs): every optimize() starts from the same pose on
    maps fused from the same poses (the oracle's), i.e. every frame is a "first frame after identical state", held to the
    north_star bar as it stands (1e-4 on the pose):
      * after k Gauss-Newton passes -- k two short of the oracle's own count, so that neither side's stop test is in play; k = 4
        on a frame that takes the oracle more than 6 passes (there Gauss-Newton cycles or wanders between voxel borders and
        amplifies the last bits in which the two sides' sums differ, pass after pass) --: same pass count, pose within 1e-4;
      * run to the end, `g` makes the oracle's decision with the oracle's pass count and ends within 1e-4 -- or, if the
        two stop tests fell differently, the oracle's |xi|^2 at the pass in question lies within 25 % of the 1e-6 threshold
        (the two sides' sums differ in their last bits; such frames are returned).
    Returns (frames that converged, frames that took the oracle more than 6 passes, frames decided differently as tuples
    (index, oracle converged, oracle passes, g converged, g passes, oracle's |xi|^2 at the earlier of the two ends))."""
    n_conv = n_long = 0
    flips = []
    for i, d in frames:
        conv_o, pose_o, used, trace, _ = o.track(d, pose)
        k = max(1, used - 2) if used <= 6 else 4              # (a frame that needs more passes cycles or wanders: its first passes are compared)
        ck_o, pose_k, used_k, _, _ = o.track(d, pose, iters=k)
        ck_g, pose_gk, passes_k = g.track(d, pose, iters=k)
        assert passes_k == used_k and bool(ck_g) == bool(ck_o), (i, k, passes_k, used_k, ck_g, ck_o)
        assert np.abs(pose_gk[:3] - pose_k[:3]).max() < TOL and np.abs(np.abs(pose_gk[3:]) - np.abs(pose_k[3:])).max() < TOL, (i, k, pose_gk, pose_k)
        cg, pose_g, passes = g.track(d, pose)
        if bool(cg) == bool(conv_o) and passes == used:
            if used <= 6:                                         # (longer runs amplify last bits: compared after k passes above)
                assert np.abs(pose_g[:3] - pose_o[:3]).max() < TOL and np.abs(np.abs(pose_g[3:]) - np.abs(pose_o[3:])).max() < TOL, (i, pose_g, pose_o)
        else:
            xi2 = trace[:used, 35]
            j = min(passes, used) - 1
            flips.append((i, bool(conv_o), used, bool(cg), passes, float(xi2[j])))
            # Two kinds.  A frame both sides end within a few passes: the stop tests fell differently, so the oracle's |xi|^2 at
            # that pass must sit at the threshold.  A frame on which one side runs long: Gauss-Newton cycles or wanders
            # (test_tracked_bench_stream_matches_oracle_frame_by_frame), the last bits in which the two sides' sums differ are
            # amplified pass after pass, and whether some iterate dips below the threshold is not determined by the state the
            # frame started from -- its first passes were compared above, the rest is counted by the caller.
            if max(used, passes) <= 6:
                assert abs(xi2[j] / 1e-6 - 1.0) < 0.25, flips[-1]
        pose = pose_o                                             # main_scan_3d.cpp:270: the last iterate is the next start, converged or not
        if conv_o:                                                # both maps take the frame at the oracle's pose
            n_conv += 1
            R, t = O.quat_to_R(pose[3:]), pose[:3]
            g.update(d, R, t)
            o.update(d, R, t)
        if used > 6:
            n_long += 1
    return n_conv, n_long, flips


def flip_classes(flips):
    """(flips on frames both sides end within 6 passes, flips on frames where one side runs longer)."""
    short = sum(1 for f in flips if max(f[2], f[4]) <= 6)
    return short, len(flips) - short
