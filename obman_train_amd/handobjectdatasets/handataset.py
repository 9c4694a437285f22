"""``HandDataset``: the reference's sample contract (``handobjectdatasets/handataset.py:27-420``) with the image work
deferred to the GPU.

Same constructor, same queries, same RNG consumption (global ``np.random`` / ``random``, in the reference's order:
centre jitter, scale jitter, rotation, [surface samples of a mesh object], blur radius, colour factors, op shuffle), same
annotation arithmetic.  The one difference: ``sample[TransQueries.images]`` is an ``ImagePlan`` - the source pixels plus
the drawn parameters - and ``collate`` turns the plans of a batch into the ``[B,3,inp_res,inp_res]`` device tensor with
``DeviceImageStage`` (bit-identical to the PIL pipeline of ``handataset.py:373-405``).

``pose_dataset`` is any object with the reference's accessor protocol (``get_image`` returning a PIL image or a uint8
``[H,W,3]`` array, ``get_center_scale``, ``get_sides``, ``get_joints2d``, ``get_joints3d``, ``get_verts3d``,
``get_objpoints3d`` or ``get_obj_verts_faces``, ``get_camintr``, ``get_meta``, ``get_manoidxs``, ``all_queries``).
"""
import random
import traceback

import numpy as np
import torch

from ..queries import BaseQueries, TransQueries
from . import handutils, imgtrans, vertexsample
from .imagestage import DeviceImageStage, ImagePlan


class _Wants:
    """Membership tests over the queries of one ``get_sample`` call."""

    def __init__(self, queries):
        self._q = set(queries)

    def __call__(self, key):
        return key in self._q

    def any(self, *keys):
        return any(k in self._q for k in keys)


class _View:
    """What was drawn for the image of one sample: mirror flag, crop centre / scale, in-plane rotation, and the affine maps
    derived from them (``affine`` source -> crop pixels, ``post_rot`` = the part that acts on the intrinsics)."""

    __slots__ = ("flip", "pixels", "width", "center", "scale", "rot", "affine", "post_rot")

    def __init__(self):
        self.flip, self.pixels, self.width = False, None, None
        self.center = self.scale = None
        self.rot, self.affine, self.post_rot = 0, None, None

    def mirror_x(self, pts2d):
        """x -> width - x on a copy (pixel coordinates of the un-mirrored source image)."""
        if not self.flip:
            return pts2d
        pts2d = pts2d.copy()
        pts2d[:, 0] = self.width - pts2d[:, 0]
        return pts2d

    def to_crop(self, pts2d):
        return torch.from_numpy(np.array(handutils.transform_coords(pts2d, self.affine)))


class _Rigid3D:
    """The rigid part of the augmentation applied to every 3-D annotation: mirror about the camera's x axis (when the hand
    side was switched) and the image's in-plane rotation about the optical axis (float32, like the reference's matrix)."""

    def __init__(self, rot, flip):
        c, s_ = np.cos(rot), np.sin(rot)
        self.R = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]]).astype(np.float32)
        self.flip = flip

    def mirrored(self, pts):
        if self.flip:  # in place, as the reference does on the arrays its accessors return
            pts[:, 0] = -pts[:, 0]
        return pts

    def spun(self, pts):
        return self.R.dot(pts.transpose(1, 0)).transpose()


_HAND_ROOT_SOURCES = (TransQueries.joints3d, BaseQueries.joints3d, TransQueries.verts3d)
_OBJECT_TARGETS = (TransQueries.objpoints3d, TransQueries.objverts3d)


class HandDataset(torch.utils.data.Dataset):
    def __init__(self, pose_dataset, center_idx=9, point_nb=600, inp_res=256, max_rot=np.pi, normalize_img=False,
                 split="train", scale_jittering=0.3, center_jittering=0.2, train=True, hue=0.15, saturation=0.5,
                 contrast=0.5, brightness=0.5, blur_radius=0.5,
                 queries=(BaseQueries.images, TransQueries.joints2d, TransQueries.verts3d, TransQueries.joints3d),
                 sides="both", block_rot=False, black_padding=False, as_obj_only=False):
        if normalize_img:
            # handataset.py:399-400 reads self.mean / self.std, which the reference never defines (AttributeError there)
            raise NotImplementedError("normalize_img=True has no defined statistics in the reference; use the default")
        self.pose_dataset = pose_dataset
        self.as_obj_only = as_obj_only
        self.inp_res = inp_res
        self.point_nb = point_nb
        self.normalize_img = normalize_img
        self.center_idx = center_idx
        self.sides = sides
        self.black_padding = black_padding
        self.hue, self.contrast, self.brightness, self.saturation = hue, contrast, brightness, saturation
        self.blur_radius = blur_radius
        self.max_rot = max_rot
        self.block_rot = block_rot
        self.train = train
        self.scale_jittering = scale_jittering
        self.center_jittering = center_jittering
        self.queries = list(queries)
        self.split = split

    def __len__(self):
        return len(self.pose_dataset)

    # -------------------------------------------------------------------------------------------------- one sample
    # get_sample is assembled from five steps; the order of the RNG draws is the reference's (handataset.py:104-372):
    # centre jitter, scale jitter, rotation [_draw_view]; surface samples of a mesh object [_object_annotations]; blur radius,
    # colour factors, op shuffle [_image_plan].
    def get_sample(self, idx, query=None):
        want = _Wants(self.queries if query is None else query)
        out = {}
        view = self._draw_view(idx, want, out)
        self._pixel_annotations(idx, want, view, out)
        if want.any(BaseQueries.segms, TransQueries.segms):
            raise NotImplementedError("segmentation maps are not on the training path (traineval.py:77-88 never requests them)")
        rigid = _Rigid3D(view.rot, view.flip)
        root, object_only = self._hand_annotations(idx, want, rigid, out)
        root = self._object_annotations(idx, want, rigid, root, object_only, out)
        if want(TransQueries.center3d):
            out[TransQueries.center3d] = root
        if want(BaseQueries.manoidxs):
            out[BaseQueries.manoidxs] = self.pose_dataset.get_manoidxs(idx)
        if want(TransQueries.images):
            out[TransQueries.images] = self._image_plan(view)
        if want(BaseQueries.meta):
            out[BaseQueries.meta] = self.pose_dataset.get_meta(idx)
        return out

    def _draw_view(self, idx, want, out):
        """Hand side (and whether the sample is mirrored onto ``self.sides``), source pixels, the drawn crop and rotation."""
        src, view = self.pose_dataset, _View()
        with_image = want.any(BaseQueries.images, TransQueries.images)
        if with_image:
            view.center, view.scale = src.get_center_scale(idx)
        if want(BaseQueries.sides):
            side = src.get_sides(idx)
            if self.sides in ("right", "left") and side != self.sides:
                view.flip, side = True, self.sides
            out[BaseQueries.sides] = side
        if with_image:
            view.pixels = np.asarray(src.get_image(idx))  # PIL image or array; the mirror flip itself happens on the GPU
            view.width = view.pixels.shape[1]
            if want(BaseQueries.images):
                out[BaseQueries.images] = view.pixels[:, ::-1] if view.flip else view.pixels
        if view.flip:
            view.center[0] = view.width - view.center[0]  # no image requested: fails like the reference (nothing to mirror about)
        if self.train and with_image:
            shift = self.center_jittering * view.scale * np.random.uniform(low=-1, high=1, size=2)
            view.center = view.center + shift.astype(int)
            zoom = np.clip(self.scale_jittering * np.random.randn() + 1, 1 - self.scale_jittering, 1 + self.scale_jittering)
            view.scale = view.scale * zoom
            view.rot = np.random.uniform(low=-self.max_rot, high=self.max_rot)
        if self.block_rot:
            view.rot = self.max_rot
        if want.any(TransQueries.joints2d, TransQueries.images):
            view.affine, view.post_rot = handutils.get_affine_transform(view.center, view.scale, [self.inp_res, self.inp_res],
                                                                        rot=view.rot)
            if want(TransQueries.affinetrans):
                out[TransQueries.affinetrans] = torch.from_numpy(view.affine)
        return view

    def _pixel_annotations(self, idx, want, view, out):
        """2-D joints / object points in source and crop pixels, camera intrinsics before and after the rotation."""
        src = self.pose_dataset
        for base_key, crop_key, getter in ((BaseQueries.joints2d, TransQueries.joints2d, src.get_joints2d),
                                           (BaseQueries.objpoints2d, TransQueries.objpoints2d,
                                            getattr(src, "get_objpoints2d", None))):
            if not want.any(base_key, crop_key):
                continue
            pts = view.mirror_x(getter(idx))
            if want(base_key):
                out[base_key] = torch.from_numpy(pts)
            if want(crop_key):
                out[crop_key] = view.to_crop(pts)
        if want.any(BaseQueries.camintrs, TransQueries.camintrs):
            intr = src.get_camintr(idx)
            if want(BaseQueries.camintrs):
                out[BaseQueries.camintrs] = intr
            if want(TransQueries.camintrs):
                out[TransQueries.camintrs] = view.post_rot.dot(intr)  # the rotation acts as an extrinsic

    def _hand_annotations(self, idx, want, rigid, out):
        """3-D joints and MANO vertices, mirrored / rotated with the image and expressed relative to the root joint.
        -> (root [3] or None, object_only): ``object_only`` = the sample has no hand to take a root from."""
        src = self.pose_dataset
        root, object_only = None, False
        if want.any(BaseQueries.joints3d, TransQueries.joints3d, TransQueries.verts3d, *_OBJECT_TARGETS):
            handless = not any(q in src.all_queries for q in _HAND_ROOT_SOURCES)
            object_only = (want.any(*_OBJECT_TARGETS) and handless) or self.as_obj_only
            if not object_only and want.any(*(_OBJECT_TARGETS + _HAND_ROOT_SOURCES)):
                joints = rigid.mirrored(src.get_joints3d(idx))
                if want(BaseQueries.joints3d):
                    out[BaseQueries.joints3d] = joints
                if self.train:
                    joints = rigid.spun(joints)
                if self.center_idx is not None:
                    root = (joints[9] + joints[0]) / 2 if self.center_idx == -1 else joints[self.center_idx]
                if want(TransQueries.joints3d):
                    out[TransQueries.joints3d] = torch.from_numpy(joints - root if self.center_idx is not None else joints)
        if want(TransQueries.verts3d):
            verts = rigid.spun(rigid.mirrored(src.get_verts3d(idx)))
            out[TransQueries.verts3d] = verts - root if self.center_idx is not None else verts
        return root, object_only

    def _object_annotations(self, idx, want, rigid, root, object_only, out):
        """Object point cloud (given, or sampled from the object's mesh) in the hand's frame; for an object-only sample the
        cloud is centred on its bounding box and inscribed in the unit sphere.  -> the root used (returned for center3d)."""
        src = self.pose_dataset
        cloud = None
        if want(TransQueries.objpoints3d) and BaseQueries.objpoints3d in src.all_queries:
            cloud = rigid.spun(rigid.mirrored(src.get_objpoints3d(idx, point_nb=self.point_nb)))
        elif want.any(TransQueries.objpoints3d, BaseQueries.objverts3d, TransQueries.objverts3d) and (
                BaseQueries.objverts3d in src.all_queries):
            mesh_verts, mesh_faces = src.get_obj_verts_faces(idx)
            mesh_verts = rigid.mirrored(mesh_verts)
            if want(BaseQueries.objverts3d):
                out[BaseQueries.objverts3d] = mesh_verts
            if want(TransQueries.objverts3d):
                posed = rigid.spun(mesh_verts)
                out[TransQueries.objverts3d] = posed - root if self.center_idx is not None else posed
            if want(BaseQueries.objfaces):
                out[BaseQueries.objfaces] = mesh_faces
            surface = vertexsample.points_from_mesh(mesh_faces, mesh_verts, vertex_nb=self.point_nb).astype(np.float32)
            cloud = rigid.spun(surface)
        elif want(TransQueries.objpoints3d):
            raise ValueError("Requested TransQueries.objpoints3d for dataset without BaseQueries.objpoints3d and "
                             "BaseQueries.objverts3d")
        if want(TransQueries.objpoints3d):
            if object_only:
                root = (cloud.max(0) + cloud.min(0)) / 2
            if self.center_idx is not None or object_only:
                cloud = cloud - root
            if cloud.max() > 5000:
                names = getattr(src, "image_names", None)
                print("object points beyond 5 m in sample {}".format(names[idx] if names is not None else idx))
            if object_only:
                cloud = cloud / np.linalg.norm(cloud, 2, 1).max()
            out[TransQueries.objpoints3d] = torch.from_numpy(cloud)
        return root

    def _image_plan(self, view):
        """What PIL would have been asked to do, recorded instead of done: the pixels are rendered on the GPU per batch."""
        blur, color_ops = (-1, 0, 0), []
        if self.train:
            blur = imgtrans.box_blur_weights(random.random() * self.blur_radius)
            color_ops = imgtrans.color_jitter_plan(brightness=self.brightness, saturation=self.saturation, hue=self.hue,
                                                   contrast=self.contrast)
        fixed = handutils.fixed_point_affine(view.affine, [self.inp_res, self.inp_res])
        return ImagePlan(view.pixels, view.flip, fixed, blur=blur, ops=color_ops)

    def __getitem__(self, idx):
        try:
            sample = self.get_sample(idx, self.queries)
        except Exception:  # same recovery as the reference (handataset.py:413-420): log, draw another sample
            traceback.print_exc()
            print("Encountered error processing sample {}".format(idx))
            sample = self.get_sample(random.randint(0, len(self) - 1), self.queries)
        return sample

    # ---------------------------------------------------------------------------------------------------- batches
    def image_stage(self, device="cuda", channels_last=False):
        return DeviceImageStage(inp_res=self.inp_res, black_padding=self.black_padding, channels_last=channels_last, device=device)

    @staticmethod
    def collate(samples, stage):
        """list of ``get_sample`` dicts -> batch dict as ``torch.utils.data.default_collate`` would build it, except that
        the ``ImagePlan`` entries become one device tensor rendered by ``stage``."""
        from torch.utils.data import default_collate

        plans = [s[TransQueries.images] for s in samples] if TransQueries.images in samples[0] else None
        rest = [{k: v for k, v in s.items() if k is not TransQueries.images} for s in samples]
        batch = default_collate(rest) if rest[0] else {}
        if plans is not None:
            batch[TransQueries.images] = stage(plans)
        return batch
