"""Seeded stand-in for an ObMan / FHB reader (no dataset files in this environment): random RGB frames of a given size and
hand / object annotations with the statistics ``obman_train_amd.synthetic.make_batch`` uses, behind the pose-dataset
accessor protocol ``HandDataset`` consumes.  FHB frames are 1920x1080 stored at 1/4 scale = 480x270
(reference ``fhbhands.py:505-508``), ObMan renders are 256x256 (``obman.py``)."""
import numpy as np

from ..contactzones import hand_template
from ..queries import BaseQueries, TransQueries


class SyntheticPoses:
    name = "synthetic"

    def __init__(self, n=256, src_hw=(270, 480), seed=0, n_obj=600, frame_pool=8, split="train"):
        self.n, self.src_hw, self.seed, self.n_obj = int(n), tuple(src_hw), int(seed), int(n_obj)
        self.split = split
        self._frames = {}  # a real reader decodes a JPEG here; the stand-in cycles through a few cached random frames
        self._frame_pool = int(frame_pool)
        self.all_queries = [BaseQueries.images, BaseQueries.joints2d, BaseQueries.joints3d, BaseQueries.verts3d, BaseQueries.sides,
                            BaseQueries.camintrs, BaseQueries.objpoints3d, TransQueries.images, TransQueries.joints2d,
                            TransQueries.joints3d, TransQueries.verts3d, TransQueries.camintrs, TransQueries.objpoints3d,
                            TransQueries.affinetrans, TransQueries.center3d]
        self.image_names = ["synthetic_%06d" % i for i in range(self.n)]
        tmpl = hand_template()[0].astype(np.float32) * 1000.0
        self._template = tmpl - tmpl.mean(0, keepdims=True)

    def __len__(self):
        return self.n

    def _rng(self, idx, salt):
        return np.random.RandomState((self.seed * 1000003 + idx * 101 + salt) % (2 ** 31))

    def get_image(self, idx):
        H, W = self.src_hw
        k = idx % self._frame_pool if self._frame_pool else idx
        if k not in self._frames:
            self._frames[k] = self._rng(k, 1).randint(0, 256, size=(H, W, 3)).astype(np.uint8)
        return self._frames[k]

    def get_joints2d(self, idx):
        H, W = self.src_hw
        rng = self._rng(idx, 2)
        c = np.array([rng.uniform(0.35 * W, 0.65 * W), rng.uniform(0.35 * H, 0.65 * H)])
        return (c + rng.normal(0, 0.1 * min(H, W), size=(21, 2))).astype(np.float32)

    def get_center_scale(self, idx):
        from . import handutils

        j = self.get_joints2d(idx)
        return handutils.get_annot_center(j), handutils.get_annot_scale(j)

    def get_sides(self, idx):
        return "left"

    def get_joints3d(self, idx):
        return self._rng(idx, 3).normal(0, 40, size=(21, 3)).astype(np.float32)

    def get_verts3d(self, idx):
        return (self._template + self._rng(idx, 4).normal(0, 5, size=(778, 3))).astype(np.float32)

    def get_objpoints3d(self, idx, point_nb=600):
        rng = self._rng(idx, 5)
        d = rng.normal(size=(point_nb, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        return (d * rng.uniform(20, 80, size=3) + rng.normal(0, 30, size=3) + np.array([0.0, -60.0, 0.0])).astype(np.float32)

    def get_camintr(self, idx):
        H, W = self.src_hw
        return np.array([[480.0, 0, W / 2], [0, 480.0, H / 2], [0, 0, 1]], np.float32)

    def get_meta(self, idx):
        return {"objname": "ellipsoid"}
