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


def one_query_in(candidates, pool):
    return any(c in pool for c in candidates)


def no_query_in(candidates, pool):
    return not one_query_in(candidates, pool)


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
    def get_sample(self, idx, query=None):
        query = self.queries if query is None else query
        pose = self.pose_dataset
        sample = {}
        wants_image = BaseQueries.images in query or TransQueries.images in query
        if wants_image:
            center, scale = pose.get_center_scale(idx)

        flip = False
        if BaseQueries.sides in query:
            hand_side = pose.get_sides(idx)
            if self.sides in ("right", "left") and hand_side != self.sides:
                flip, hand_side = True, self.sides  # mirror every hand onto the requested side
            sample[BaseQueries.sides] = hand_side

        if wants_image:
            img = np.asarray(pose.get_image(idx))  # PIL image or array; the mirror flip itself happens on the GPU
            src_w = img.shape[1]
            if BaseQueries.images in query:
                sample[BaseQueries.images] = img[:, ::-1] if flip else img
        if flip:
            center[0] = src_w - center[0]

        if self.train and wants_image:
            offsets = self.center_jittering * scale * np.random.uniform(low=-1, high=1, size=2)
            center = center + offsets.astype(int)
            jitter = self.scale_jittering * np.random.randn() + 1
            scale = scale * np.clip(jitter, 1 - self.scale_jittering, 1 + self.scale_jittering)
            rot = np.random.uniform(low=-self.max_rot, high=self.max_rot)
        else:
            rot = 0
        if self.block_rot:
            rot = self.max_rot
        rot_mat = np.array([[np.cos(rot), -np.sin(rot), 0], [np.sin(rot), np.cos(rot), 0], [0, 0, 1]]).astype(np.float32)

        if TransQueries.joints2d in query or TransQueries.images in query:
            affinetrans, post_rot_trans = handutils.get_affine_transform(center, scale, [self.inp_res, self.inp_res], rot=rot)
            if TransQueries.affinetrans in query:
                sample[TransQueries.affinetrans] = torch.from_numpy(affinetrans)
        if BaseQueries.joints2d in query or TransQueries.joints2d in query:
            joints2d = pose.get_joints2d(idx)
            if flip:
                joints2d = joints2d.copy()
                joints2d[:, 0] = src_w - joints2d[:, 0]
            if BaseQueries.joints2d in query:
                sample[BaseQueries.joints2d] = torch.from_numpy(joints2d)
        if TransQueries.joints2d in query:
            sample[TransQueries.joints2d] = torch.from_numpy(np.array(handutils.transform_coords(joints2d, affinetrans)))

        if BaseQueries.camintrs in query or TransQueries.camintrs in query:
            camintr = pose.get_camintr(idx)
            if BaseQueries.camintrs in query:
                sample[BaseQueries.camintrs] = camintr
            if TransQueries.camintrs in query:
                sample[TransQueries.camintrs] = post_rot_trans.dot(camintr)  # the rotation acts as an extrinsic

        if BaseQueries.objpoints2d in query or TransQueries.objpoints2d in query:
            objpoints2d = pose.get_objpoints2d(idx)
            if flip:
                objpoints2d = objpoints2d.copy()
                objpoints2d[:, 0] = src_w - objpoints2d[:, 0]
            if BaseQueries.objpoints2d in query:
                sample[BaseQueries.objpoints2d] = torch.from_numpy(objpoints2d)
            if TransQueries.objpoints2d in query:
                sample[TransQueries.objpoints2d] = torch.from_numpy(np.array(handutils.transform_coords(objpoints2d, affinetrans)))

        if BaseQueries.segms in query or TransQueries.segms in query:
            raise NotImplementedError("segmentation maps are not on the training path (traineval.py:77-88 never requests them)")

        # ---- 3-D annotations: flip x, rotate with the image, centre on the root joint
        center3d = None
        obj_only = False
        if one_query_in([BaseQueries.joints3d, TransQueries.joints3d, TransQueries.verts3d, TransQueries.objverts3d,
                         TransQueries.objpoints3d], query):
            center3d_queries = [TransQueries.joints3d, BaseQueries.joints3d, TransQueries.verts3d]
            obj_only = ((TransQueries.objverts3d in query or TransQueries.objpoints3d in query)
                        and no_query_in(center3d_queries, pose.all_queries)) or self.as_obj_only
            if not obj_only and one_query_in([TransQueries.objpoints3d, TransQueries.objverts3d] + center3d_queries, query):
                joints3d = pose.get_joints3d(idx)
                if flip:
                    joints3d[:, 0] = -joints3d[:, 0]
                if BaseQueries.joints3d in query:
                    sample[BaseQueries.joints3d] = joints3d
                if self.train:
                    joints3d = rot_mat.dot(joints3d.transpose(1, 0)).transpose()
                if self.center_idx is not None:
                    center3d = (joints3d[9] + joints3d[0]) / 2 if self.center_idx == -1 else joints3d[self.center_idx]
                if TransQueries.joints3d in query:
                    if self.center_idx is not None:
                        joints3d = joints3d - center3d
                    sample[TransQueries.joints3d] = torch.from_numpy(joints3d)

        if TransQueries.verts3d in query:
            verts = pose.get_verts3d(idx)
            if flip:
                verts[:, 0] = -verts[:, 0]
            verts = rot_mat.dot(verts.transpose(1, 0)).transpose()
            if self.center_idx is not None:
                verts = verts - center3d
            sample[TransQueries.verts3d] = verts

        obj_verts3d = None
        if TransQueries.objpoints3d in query and BaseQueries.objpoints3d in pose.all_queries:
            points3d = pose.get_objpoints3d(idx, point_nb=self.point_nb)
            if flip:
                points3d[:, 0] = -points3d[:, 0]
            obj_verts3d = rot_mat.dot(points3d.transpose(1, 0)).transpose()
        elif (TransQueries.objpoints3d in query or BaseQueries.objverts3d in query or TransQueries.objverts3d in query) and (
                BaseQueries.objverts3d in pose.all_queries):
            obj_verts3d, obj_faces = pose.get_obj_verts_faces(idx)
            if flip:
                obj_verts3d[:, 0] = -obj_verts3d[:, 0]
            if BaseQueries.objverts3d in query:
                sample[BaseQueries.objverts3d] = obj_verts3d
            if TransQueries.objverts3d in query:
                mesh = rot_mat.dot(obj_verts3d.transpose(1, 0)).transpose()
                sample[TransQueries.objverts3d] = mesh - center3d if self.center_idx is not None else mesh
            if BaseQueries.objfaces in query:
                sample[BaseQueries.objfaces] = obj_faces
            obj_verts3d = vertexsample.points_from_mesh(obj_faces, obj_verts3d, vertex_nb=self.point_nb).astype(np.float32)
            obj_verts3d = rot_mat.dot(obj_verts3d.transpose(1, 0)).transpose()
        elif TransQueries.objpoints3d in query:
            raise ValueError("Requested TransQueries.objpoints3d for dataset without BaseQueries.objpoints3d and "
                             "BaseQueries.objverts3d")
        if TransQueries.objpoints3d in query:
            if obj_only:
                center3d = (obj_verts3d.max(0) + obj_verts3d.min(0)) / 2
            if self.center_idx is not None or obj_only:
                obj_verts3d = obj_verts3d - center3d
            if obj_verts3d.max() > 5000:
                print("object points beyond 5 m in sample {}".format(getattr(pose, "image_names", [idx] * (idx + 1))[idx]))
            if obj_only:
                obj_verts3d = obj_verts3d / np.linalg.norm(obj_verts3d, 2, 1).max()  # inscribe in the unit sphere
            sample[TransQueries.objpoints3d] = torch.from_numpy(obj_verts3d)

        if TransQueries.center3d in query:
            sample[TransQueries.center3d] = center3d
        if BaseQueries.manoidxs in query:
            sample[BaseQueries.manoidxs] = pose.get_manoidxs(idx)

        # ---- the image: draw what PIL would have been asked to do, leave the pixels to the GPU
        if TransQueries.images in query:
            blur, color_ops = (-1, 0, 0), []
            if self.train:
                blur = imgtrans.box_blur_weights(random.random() * self.blur_radius)
                color_ops = imgtrans.color_jitter_plan(brightness=self.brightness, saturation=self.saturation, hue=self.hue,
                                                       contrast=self.contrast)
            fixed = handutils.fixed_point_affine(affinetrans, [self.inp_res, self.inp_res])
            sample[TransQueries.images] = ImagePlan(img, flip, fixed, blur=blur, ops=color_ops)

        if BaseQueries.meta in query:
            sample[BaseQueries.meta] = pose.get_meta(idx)
        return sample

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
