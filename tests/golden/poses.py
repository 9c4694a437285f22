"""Seeded synthetic pose dataset with the accessor contract ``HandDataset`` expects from its ``pose_dataset``
(reference ``handataset.py:103-371``: get_image / get_center_scale / get_sides / get_joints2d / get_joints3d /
get_verts3d / get_objpoints3d | get_obj_verts_faces / get_camintr, attribute ``all_queries``).

Shared by the golden generator (which feeds it to the reference's own ``HandDataset``), the oracle tests and the
product tests, so all three see bit-identical inputs.  Nothing here touches /root/reference.
"""
import numpy as np


def _octahedron(radius, centre):
    v = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32) * radius + centre
    f = np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]], np.int16)
    return v.astype(np.float32), f


class SeededPoses:
    """``key(name)`` maps a query name to whatever key type the consumer uses (the reference's Enum members, or plain
    strings for the oracle / product mirror)."""

    def __init__(self, n=4, src_hw=(270, 480), seed=0, mesh_objects=False, as_pil=False, point_nb=600,
                 base_key=lambda name: name, trans_key=lambda name: name):
        self.n, self.src_hw, self.seed, self.mesh_objects, self.as_pil = n, tuple(src_hw), seed, mesh_objects, as_pil
        self.point_nb = point_nb
        base = ["images", "joints2d", "joints3d", "verts3d", "sides", "camintrs", "objverts3d" if mesh_objects else "objpoints3d"]
        if mesh_objects:
            base.append("objfaces")
        trans = ["images", "joints2d", "joints3d", "verts3d", "camintrs", "objpoints3d", "affinetrans", "center3d"]
        self.all_queries = [base_key(b) for b in base] + [trans_key(t) for t in trans]
        self.base_names, self.trans_names = base, trans  # for string-keyed consumers (the oracle)
        self.image_names = ["synthetic_%04d" % i for i in range(n)]
        self.links = None

    def __len__(self):
        return self.n

    def _rng(self, idx, salt):
        return np.random.RandomState(self.seed * 1000003 + idx * 101 + salt)

    def _image_array(self, idx):
        H, W = self.src_hw
        rng = self._rng(idx, 1)
        coarse = rng.randint(0, 256, size=((H + 15) // 16, (W + 15) // 16, 3)).astype(np.float32)
        smooth = np.kron(coarse, np.ones((16, 16, 1), np.float32))[:H, :W]
        noise = rng.randint(-40, 41, size=(H, W, 3)).astype(np.float32)
        return np.clip(smooth + noise, 0, 255).astype(np.uint8)

    def get_image(self, idx):
        arr = self._image_array(idx)
        if self.as_pil:
            from PIL import Image

            return Image.fromarray(arr, "RGB")
        return arr

    def get_joints2d(self, idx):
        H, W = self.src_hw
        rng = self._rng(idx, 2)
        c = np.array([rng.uniform(0.3 * W, 0.7 * W), rng.uniform(0.3 * H, 0.7 * H)])
        return (c + rng.normal(0, 0.08 * min(H, W), size=(21, 2))).astype(np.float32)

    def get_center_scale(self, idx):  # as the FHB / ObMan readers: bounding box of the 2-D joints (handutils.py:8-37)
        j = self.get_joints2d(idx)
        mn, mx = j.min(0), j.max(0)
        center = np.asarray([int((mx[0] + mn[0]) / 2), int((mx[1] + mn[1]) / 2)])
        scale = max(mx[0] - mn[0], mx[1] - mn[1]) * 2.2
        return center, scale

    def get_sides(self, idx):
        return "right" if (idx + self.seed) % 2 else "left"

    def get_joints3d(self, idx):
        return self._rng(idx, 3).normal(0, 40, size=(21, 3)).astype(np.float32)

    def get_verts3d(self, idx):
        return self._rng(idx, 4).normal(0, 45, size=(778, 3)).astype(np.float32)

    def get_objpoints3d(self, idx, point_nb=600):
        rng = self._rng(idx, 5)
        d = rng.normal(size=(point_nb, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        return (d * rng.uniform(20, 80, size=3) + rng.normal(0, 30, size=3)).astype(np.float32)

    def get_obj_verts_faces(self, idx):
        rng = self._rng(idx, 6)
        return _octahedron(rng.uniform(30, 70), rng.normal(0, 30, size=3).astype(np.float32))

    def get_camintr(self, idx):
        H, W = self.src_hw
        return np.array([[480.0, 0, W / 2], [0, 480.0, H / 2], [0, 0, 1]], np.float32)

    def get_meta(self, idx):
        return {"objname": "octahedron"}


CASES = {
    # name: (pose kwargs, HandDataset kwargs, sample indices, seed)
    "fhb_like_train": (dict(n=4, src_hw=(270, 480), seed=1), dict(inp_res=64, sides="left"), [0, 1, 2, 3], 11),
    "obman_like_mesh_pad": (dict(n=3, src_hw=(256, 256), seed=2, mesh_objects=True),
                            dict(inp_res=96, sides="both", black_padding=True, point_nb=64), [0, 1, 2], 12),
    "eval_no_aug": (dict(n=2, src_hw=(120, 200), seed=3), dict(inp_res=64, train=False, sides="right", center_idx=-1), [0, 1], 13),
    "strong_jitter": (dict(n=3, src_hw=(90, 130), seed=4),
                      dict(inp_res=48, blur_radius=3.0, hue=0.5, brightness=1.2, contrast=0.9, saturation=1.5, max_rot=0.5,
                           scale_jittering=0.1, center_jittering=0.4, sides="left"), [0, 1, 2], 14),
    "block_rot_no_color": (dict(n=2, src_hw=(100, 100), seed=5),
                           dict(inp_res=32, hue=0, brightness=0, contrast=0, saturation=0, blur_radius=0.0, block_rot=True, max_rot=0.7),
                           [0, 1], 15),
    "full_res": (dict(n=1, src_hw=(270, 480), seed=6), dict(inp_res=256, sides="left"), [0], 16),
}
QUERIES = ["affinetrans", "images", "verts3d", "center3d", "joints3d", "objpoints3d", "camintrs", "sides", "joints2d"]
