"""Per-frame timers of the Scan3D CLI on the S-tum stream as files (its own "Load data" / "Point optimization + integration" lines):
where the loop waits -- for the decoders and the staging copies, or for the device.   usage: python tools/cli_frame_timers.py"""
import os, re, subprocess, sys, tempfile, shutil, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.package()
tmp = tempfile.mkdtemp(prefix="gsdf_cli_")
seq = pkg.synth.Sequence("tum", 640, 480, n_frames=400, seed=0)
ds = pkg.synth.write_dataset(seq, os.path.join(tmp, "c2"), layout="tum", with_poses=False)
scan3d = os.path.join(ROOT, "gradient-sdf_amd", "host", "Scan3D")
res = os.path.join(tmp, "out") + "/"; os.makedirs(res)
for th in (8, 16):
    out = subprocess.run([scan3d, "--input", ds, "--results", res, "--scan-type", "grad-sdf", "--data-type", "tum", "--voxel-size", "0.01", "--trunc", "10",
                          "--decode-threads", str(th)], capture_output=True, text=True)
    load = [float(v) for v in re.findall(r"Load data: ([0-9.e+-]+)ms", out.stdout)]
    enq = [float(v) for v in re.findall(r"Point optimization \+ integration \(enqueued\): ([0-9.e+-]+)ms", out.stdout)]
    fps = re.findall(r"([0-9.e+-]+) frames per second", out.stdout)
    big = sorted(range(len(load)), key=lambda i: -load[i])[:4]
    print("   largest Load data: %s" % [(i, round(load[i], 2)) for i in big], "largest enqueue:", [(i, round(enq[i], 2)) for i in sorted(range(len(enq)), key=lambda i: -enq[i])[:3]])
    print("threads %d: fps %s | Load data mean %.1f us (median %.1f, p90 %.1f, max %.1f) | track+fuse call mean %.1f us (median %.1f)" % (
        th, fps, 1e3 * np.mean(load), 1e3 * np.median(load), 1e3 * np.percentile(load, 90), 1e3 * np.max(load), 1e3 * np.mean(enq), 1e3 * np.median(enq)))
shutil.rmtree(tmp, ignore_errors=True)
