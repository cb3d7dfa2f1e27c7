"""Frames per second of the Scan3D CLI (file decode + host->HBM staging + tracking + fusion, exports excluded):
the device-resident loop against --sync (blocking facade calls) and against the CPU oracle on the same files.
  C1: 30-frame RenderSpheres-style sequence, 640x480, 1 cm voxels, trunc 10 (BASELINE configs[0]), GT poses and tracked;
  C2: the S-tum bench stream as files (TUM layout), tracked.
usage: python tools/cli_fps.py [--frames-tum 120] [--oracle-frames 8]"""
import argparse, os, re, shutil, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.package()
O = G.oracle_module()
ap = argparse.ArgumentParser()
ap.add_argument("--frames-tum", type=int, default=120)
ap.add_argument("--oracle-frames", type=int, default=8)
a = ap.parse_args()
scan3d = os.path.join(ROOT, "gradient-sdf_amd", "host", "Scan3D")
tmp = tempfile.mkdtemp(prefix="gsdf_cli_")


def run(ds, dtype, extra, label):
    res = os.path.join(tmp, "out_" + re.sub(r"\W", "_", label)) + "/"
    os.makedirs(res, exist_ok=True)
    t0 = time.perf_counter()
    out = subprocess.run([scan3d, "--input", ds, "--results", res, "--scan-type", "grad-sdf", "--data-type", dtype,
                          "--voxel-size", "0.01", "--trunc", "10"] + extra, capture_output=True, text=True)
    wall = time.perf_counter() - t0
    if out.returncode != 0:
        print(label, "FAILED", out.stderr[-500:])
        return
    m = re.findall(r"([0-9.e+-]+) frames per second", out.stdout)
    if m:       # two clocks: from before the staging buffers / decoder threads exist (every frame's load inside), and behind that set-up
        print("%-46s %9.1f frames/s incl. the set-up of the staging pipeline, %9.1f behind it   (whole process incl. exports %.2f s)" % (
            label, float(m[0]), float(m[-1]), wall), flush=True)
    else:   # --sync prints per-call timers only: sum them
        ms = [float(v) for v in re.findall(r"(?:Load data|Point optimization|Integrate depth data into Sdf): ([0-9.e+-]+)ms", out.stdout)]
        s = [float(v) for v in re.findall(r"(?:Load data|Point optimization|Integrate depth data into Sdf): ([0-9.e+-]+)s\.", out.stdout)]
        n = len(re.findall(r"Working on frame", out.stdout)) - 1
        tot = sum(ms) / 1e3 + sum(s)
        print("%-46s %9.1f frames/s in the loop   (whole process incl. exports %.2f s)" % (label, n / tot, wall), flush=True)


# ---- C1: spheres, 30 frames ------------------------------------------------------------------------------------
seq = pkg.synth.Sequence("spheres", 640, 480, n_frames=30, seed=0, step_deg=0.5)
ds_gt = pkg.synth.write_dataset(seq, os.path.join(tmp, "c1_gt"), layout="synth")
ds_tr = pkg.synth.write_dataset(seq, os.path.join(tmp, "c1_tr"), layout="synth", with_poses=False)
run(ds_gt, "synth", [], "C1 spheres x30, GT poses, device-resident loop")
run(ds_gt, "synth", ["--sync"], "C1 spheres x30, GT poses, --sync")
run(ds_tr, "synth", [], "C1 spheres x30, tracked, device-resident loop")
run(ds_tr, "synth", ["--sync"], "C1 spheres x30, tracked, --sync")
# the oracle on the same content
vs = np.float32(0.01)
o = O.Oracle(vs, np.float32(10) * vs, 640, 480, seq.K, threads=1)
n = min(a.oracle_frames, 30)
fr = [seq.frame(i) for i in range(n)]
t0 = time.perf_counter()
for d, R, t in fr:
    o.update(d, R, t)
print("%-46s %9.2f frames/s   (CPU oracle, serial, %d frames)" % ("C1 spheres, GT poses", n / (time.perf_counter() - t0), n), flush=True)
o = O.Oracle(vs, np.float32(10) * vs, 640, 480, seq.K, threads=1)
pose = np.array([0, 0, 0, 0, 0, 0, 1], np.float32)
t0 = time.perf_counter()
for i, (d, R, t) in enumerate(fr):
    if i == 0:
        o.update(d, np.eye(3), np.zeros(3))
    else:
        conv, pose, _, _, _ = o.track(d, pose)
        if conv:
            o.update(d, O.quat_to_R(pose[3:]), pose[:3])
print("%-46s %9.2f frames/s   (CPU oracle, serial, %d frames)" % ("C1 spheres, tracked", n / (time.perf_counter() - t0), n), flush=True)

# ---- C2: the bench stream as TUM files ------------------------------------------------------------------------------
seq = pkg.synth.Sequence("tum", 640, 480, n_frames=a.frames_tum, seed=0)
ds = pkg.synth.write_dataset(seq, os.path.join(tmp, "c2"), layout="tum", with_poses=False)
for th in (2, 8, 16, 32):
    run(ds, "tum", ["--decode-threads", str(th)], "C2 S-tum x%d, tracked, %d decode threads" % (a.frames_tum, th))
run(ds, "tum", ["--sync"], "C2 S-tum x%d, tracked, --sync" % a.frames_tum)
shutil.rmtree(tmp, ignore_errors=True)
