"""Seeded synthetic depth streams for the Gradient-SDF hot path (host side, numpy).

The reference ships its synthetic input only as MATLAB (matlab/RenderSpheres.m,
matlab/add_kinect_noise.m, matlab/poses.txt) and MATLAB is absent, so the same
recipe is restated here:

  * S-spheres: 5 random non-intersecting spheres, centres U[-0.5,0.5]^3, radii
    U[0.0625,0.5] (RenderSpheres.m:46-53), analytic ray-sphere depth
    z = (-B - sqrt(B^2-4AC))/2A (RenderSpheres.m:99-110), background 0,
    look-at-origin orbit at radius ~2 m (matlab/poses.txt).
  * S-tum: all-valid room (inward box + spheres), fr1/xyz-like translations
    +-0.2 m at ~1 cm/frame, TUM depth unit 1/5000 (TumrgbdLoader.h:62).
  * Kinect disparity-quantisation noise (add_kinect_noise.m:61-74).
  * depth quantised to uint16 and turned into float metres exactly like the
    loaders do: float(u16) * float(unit)  (ImageLoader.h:159-175).

Poses are camera->world (p_w = R p_c + t), like the reference (MapGradPixelSdf.cpp:103).
"""
import numpy as np

UNIT_SYNTH = np.float32(1.0 / 1000)     # SynthLoader.h:53,58
UNIT_TUM = np.float32(1.0 / 5000)       # TumrgbdLoader.h:62,68


def intrinsics(W=640, H=480):
    """Kinect-like K = [525 0 319.5; 0 525 239.5] (RenderSpheres.m:39) scaled with resolution."""
    s = W / 640.0
    return np.array([[525.0 * s, 0, (W - 1) / 2.0], [0, 525.0 * s, (H - 1) / 2.0], [0, 0, 1]], np.float32)


def make_spheres(seed=0, n=5):
    """RenderSpheres.m:46-53 with a seeded generator."""
    rng = np.random.default_rng(seed)
    sph = [np.concatenate([rng.random(3) - 0.5, [0.0625 + 0.4375 * rng.random()]])]
    while len(sph) < n:
        c = rng.random(3) - 0.5
        r = 0.0625 + 0.4375 * rng.random()
        S = np.array(sph)
        if np.all(np.sqrt(((S[:, :3] - c) ** 2).sum(1)) > S[:, 3] + r):
            sph.append(np.concatenate([c, [r]]))
    return np.array(sph)


def look_at(pos, target=(0, 0, 0), up=(0, 0, 1)):
    """camera->world rotation: columns = right, down, forward (z forward, y down)."""
    pos = np.asarray(pos, np.float64)
    f = np.asarray(target, np.float64) - pos
    f /= np.linalg.norm(f)
    r = np.cross(f, np.asarray(up, np.float64))
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    return np.stack([r, d, f], axis=1)


def rot_xyz(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def render_depth(K, W, H, R, t, spheres=None, box=None, plane_z=None):
    """Analytic depth (float64 metres, 0 = no hit) of spheres / inward box / fronto plane."""
    K = np.asarray(K, np.float64)
    u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    x0 = (u - K[0, 2]) / K[0, 0]
    y0 = (v - K[1, 2]) / K[1, 1]
    R = np.asarray(R, np.float64)
    t = np.asarray(t, np.float64)
    depth = np.full((H, W), np.inf)
    if spheres is not None and len(spheres):
        A = x0 ** 2 + y0 ** 2 + 1.0
        c = (np.asarray(spheres)[:, :3] - t) @ R            # camera frame (RenderSpheres.m:101)
        for s in range(len(spheres)):
            B = -2.0 * (x0 * c[s, 0] + y0 * c[s, 1] + c[s, 2])
            Cc = (c[s] ** 2).sum() - spheres[s][3] ** 2
            disc = B * B - 4 * A * Cc
            ok = disc >= 0
            z = np.where(ok, 0.5 * (-B - np.sqrt(np.where(ok, disc, 0.0))) / A, np.inf)
            z = np.where(z > 0, z, np.inf)
            depth = np.minimum(depth, z)
    if box is not None:
        lo, hi = np.asarray(box[0], np.float64), np.asarray(box[1], np.float64)
        d = np.stack([R[a, 0] * x0 + R[a, 1] * y0 + R[a, 2] for a in range(3)], 0)   # world ray dir (z-param)
        zb = np.full((H, W), np.inf)
        for a in range(3):
            with np.errstate(divide="ignore", invalid="ignore"):
                zl = np.where(d[a] < 0, (lo[a] - t[a]) / d[a], np.inf)
                zh = np.where(d[a] > 0, (hi[a] - t[a]) / d[a], np.inf)
            zb = np.minimum(zb, np.minimum(zl, zh))
        depth = np.minimum(depth, zb)
    if plane_z is not None:
        depth = np.minimum(depth, np.full((H, W), float(plane_z)))
    depth[~np.isfinite(depth)] = 0.0
    return depth


def kinect_noise(z, rng):
    """add_kinect_noise.m:61-74: disparity d = (3 - 1/z)/2.85e-3 + N(0, 0.5^2), rounded, inverted."""
    mask = z > 0
    d = np.zeros_like(z)
    d[mask] = (3.0 - 1.0 / z[mask]) / 2.85e-3
    d[mask] += 0.5 * rng.standard_normal(z.shape)[mask]
    d = np.round(d)
    out = z.copy()
    zi = -2.85e-3 * d + 3.0
    out[mask] = 1.0 / zi[mask]
    return out


def quantize_u16(z, unit):
    """imwrite(uint16(1000*I)) (RenderSpheres.m:135): round to nearest, saturate."""
    return np.clip(np.round(z / float(unit)), 0, 65535).astype(np.uint16)


def u16_to_metres(d16, unit):
    """cv::Mat::convertTo(CV_32FC1, unit_) -- ImageLoader.h:172."""
    return d16.astype(np.float32) * np.float32(unit)


class Sequence:
    """A seeded synthetic depth stream: frame(i) -> (depth float32 HxW metres, R 3x3 f32, t 3 f32)."""

    def __init__(self, kind="spheres", W=640, H=480, n_frames=30, seed=0, noise=True,
                 step_deg=0.5, unit=None, motion=1.0, pose_rows=None):
        self.kind, self.W, self.H, self.n, self.seed, self.noise = kind, W, H, n_frames, seed, noise
        # pose_rows: rows "ts tx ty tz qx qy qz qw" of a TUM-format pose file (e.g. the reference's matlab/poses.txt, kept as
        # data in tests/golden/ref_poses.txt): the frames are rendered from THOSE camera poses instead of the built-in orbit
        self.pose_rows = None if pose_rows is None else np.asarray(pose_rows, np.float64).reshape(-1, 8)
        self.motion = float(motion)      # "tum": scale of the trajectory's time (0.5 = half the motion between two frames)
        self.K = intrinsics(W, H)
        self.step_deg = step_deg
        if kind == "spheres":
            self.spheres = make_spheres(seed)
            self.box = None
            self.unit = UNIT_SYNTH if unit is None else np.float32(unit)
        elif kind == "tum":
            # all-valid room: inward box + 3 spheres in front of the back wall
            self.box = (np.array([-1.15, -2.0, -0.85]), np.array([1.25, 1.65, 1.05]))
            self.spheres = np.array([[-0.45, 0.95, -0.35, 0.30], [0.50, 1.10, 0.15, 0.35], [0.05, 0.75, -0.60, 0.18]])
            self.unit = UNIT_TUM if unit is None else np.float32(unit)
        elif kind == "plane":
            self.spheres, self.box = None, None
            self.unit = UNIT_SYNTH if unit is None else np.float32(unit)
        else:
            raise ValueError(kind)

    def pose(self, i):
        if self.pose_rows is not None:
            row = self.pose_rows[i]
            x, y, z, w = row[4:8] / np.linalg.norm(row[4:8])
            R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                          [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                          [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
            return R.astype(np.float32), row[1:4].astype(np.float32)
        if self.kind == "spheres":
            # look-at-origin orbit, radius 2 m, height -0.125 .. 0.25 (matlab/poses.txt shape),
            # step_deg per frame (poses.txt is ~4 deg/frame: too coarse for tracking)
            a = np.deg2rad(self.step_deg * i)
            hgt = -0.125 + 0.375 * (i / max(self.n - 1, 1))
            pos = np.array([2.0 * np.sin(a), -2.0 * np.cos(a), hgt])
            R = look_at(pos)
            return R.astype(np.float32), pos.astype(np.float32)
        if self.kind == "tum":
            # fr1/xyz-like: +-0.2 m translations, <= ~1.4 cm/frame, < 1 deg rotation
            i = i * self.motion
            pos = np.array([0.0, -0.65, 0.0]) + 0.2 * np.array([
                np.sin(2 * np.pi * i / 120.0), np.sin(2 * np.pi * i / 90.0 + 1.0) - np.sin(1.0),
                np.sin(2 * np.pi * i / 150.0 + 2.0) - np.sin(2.0)])
            R0 = look_at(np.array([0.0, -0.65, 0.0]), target=(0.0, 1.0, 0.0))
            wob = np.deg2rad(0.5)
            Rw = rot_xyz(wob * np.sin(2 * np.pi * i / 70.0), wob * np.sin(2 * np.pi * i / 110.0 + 0.5),
                         wob * np.sin(2 * np.pi * i / 95.0 + 1.5))
            return (R0 @ Rw).astype(np.float32), pos.astype(np.float32)
        return np.eye(3, dtype=np.float32), np.zeros(3, np.float32)

    def depth_u16(self, i, plane_z=1.5):
        R, t = self.pose(i)
        z = render_depth(self.K, self.W, self.H, R.astype(np.float64), t.astype(np.float64),
                         self.spheres, self.box, plane_z if self.kind == "plane" else None)
        if self.noise:
            z = kinect_noise(z, np.random.default_rng([self.seed, i]))
        return quantize_u16(z, self.unit)

    def frame(self, i):
        R, t = self.pose(i)
        return u16_to_metres(self.depth_u16(i), self.unit), R, t

    def frames(self):
        for i in range(self.n):
            yield self.frame(i)


def R_to_quat_np(R):
    """numpy float64 rotation->quaternion (x,y,z,w) for pose files (main_scan_3d.cpp:274-280 layout)."""
    R = np.asarray(R, np.float64)
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0)
        w = 0.5 * s
        s = 0.5 / s
        return np.array([(R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s, w])
    i = int(np.argmax(np.diag(R)))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    q = np.zeros(4)
    q[i] = 0.5 * s
    s = 0.5 / s
    q[3] = (R[k, j] - R[j, k]) * s
    q[j] = (R[j, i] + R[i, j]) * s
    q[k] = (R[k, i] + R[i, k]) * s
    return q


# ---- datasets on disk, in the layouts the reference's loaders read ---------------------------------

def _png_gray16(path, img16):
    """16-bit grayscale PNG via zlib (no imaging library in this image)."""
    import struct
    import zlib
    h, w = img16.shape
    raw = b"".join(b"\x00" + img16[y].astype(">u2").tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 0, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def layout_unit(layout):
    """Depth unit of a dataset layout's loader: SynthLoader 1/1000 (SynthLoader.h:58), TumrgbdLoader 1/5000 (TumrgbdLoader.h:62)."""
    return UNIT_TUM if layout == "tum" else UNIT_SYNTH


def layout_depth_u16(seq, i, layout):
    """The uint16 depth image write_dataset stores for frame i: the sequence's own quantisation when its unit is the
    layout's, else the frame re-quantised to the layout's unit (so that the loader reads metres back)."""
    d16 = seq.depth_u16(i)
    if np.float32(seq.unit) == np.float32(layout_unit(layout)):
        return d16
    return quantize_u16(u16_to_metres(d16, seq.unit).astype(np.float64), layout_unit(layout))


def write_dataset(seq, out_dir, layout="synth", with_poses=True, pose_file="pose.txt"):
    """Write `seq` as a dataset directory: intrinsics.txt, depth PNGs (u16), optional TUM pose file.
    layout "synth": depth/%03d.png 1-based (SynthLoader.h:64-83); "tum": associated.txt + depth/<ts>.png
    (TumrgbdLoader.h:80-104).  Returns the directory (with trailing slash, as --input expects)."""
    import os
    out_dir = os.path.join(out_dir, "")
    os.makedirs(os.path.join(out_dir, "depth"), exist_ok=True)
    with open(out_dir + "intrinsics.txt", "w") as f:
        for r in range(3):
            f.write(" ".join(repr(float(seq.K[r, c])) for c in range(3)) + "\n")
    assoc, poses = [], []
    for i in range(seq.n):
        ts = "%03d" % (i + 1) if layout == "synth" else "%.6f" % (1305031102.0 + i / 30.0)
        name = "depth/%s.png" % ts
        _png_gray16(out_dir + name, layout_depth_u16(seq, i, layout))
        assoc.append("%s rgb/%s.png %s %s" % (ts, ts, ts, name))
        R, t = seq.pose(i)
        q = R_to_quat_np(R)
        poses.append("%s %.9g %.9g %.9g %.9g %.9g %.9g %.9g" % (ts, t[0], t[1], t[2], q[0], q[1], q[2], q[3]))
    if layout == "tum":
        with open(out_dir + "associated.txt", "w") as f:
            f.write("# rgb_timestamp rgb_file depth_timestamp depth_file\n" + "\n".join(assoc) + "\n")
    if with_poses:
        with open(out_dir + pose_file, "w") as f:
            f.write("\n".join(poses) + "\n")
    return out_dir


# ---- colour keyframes for the photometric bundle adjustment (PhotoBA, config C5) ---------------------------

def albedo(p):
    """Smooth procedural RGB texture of a world point (so that every view of a surface point agrees)."""
    p = np.asarray(p, np.float64)
    r = 0.5 + 0.25 * np.sin(9.0 * p[..., 0] + 1.0) + 0.25 * np.sin(7.0 * p[..., 1] * p[..., 2] + 0.3)
    g = 0.5 + 0.25 * np.sin(8.0 * p[..., 1] - 0.5) + 0.25 * np.cos(6.0 * p[..., 0] + 5.0 * p[..., 2])
    b = 0.5 + 0.25 * np.sin(10.0 * p[..., 2] + 2.0) + 0.25 * np.sin(5.0 * (p[..., 0] - p[..., 1]))
    return np.stack([r, g, b], -1)


def render_color_bgr(seq, i, R=None, t=None):
    """float32 BGR image in [0,1] (cv::imread order, ImageLoader.h:198-216) of frame i of `seq`, textured
    with albedo() at the analytic hit point; pixels without a hit are black."""
    Rs, ts = seq.pose(i)
    R = Rs if R is None else R
    t = ts if t is None else t
    R64, t64 = np.asarray(R, np.float64), np.asarray(t, np.float64)
    z = render_depth(seq.K, seq.W, seq.H, R64, t64, seq.spheres, seq.box)
    K = np.asarray(seq.K, np.float64)
    u, v = np.meshgrid(np.arange(seq.W, dtype=np.float64), np.arange(seq.H, dtype=np.float64))
    d = np.stack([(u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], np.ones_like(u)], -1)
    pw = t64 + z[..., None] * (d @ R64.T)
    rgb = albedo(pw)
    rgb[z <= 0] = 0.0
    return np.ascontiguousarray(rgb[..., ::-1], dtype=np.float32)


def pose16(R, t):
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = R
    T[:3, 3] = t
    return T
