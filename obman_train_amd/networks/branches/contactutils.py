"""Ray-parity inside test, HIP-backed.

Mirror of ``mano_train/networks/branches/contactutils.py:62-159`` (reference):
``batch_mesh_contains_points(ray_origins, obj_triangles) -> exterior [B,P] bool``.  The fast path
used by ``compute_contact_loss`` passes vertices + faces (``mesh_exterior``) so the [B,F,3,3]
triangle gather of ``contactloss.py:169`` is never materialised; the triangle-tensor signature of the
reference is kept for callers that already hold triangles.
"""
import torch

from obman_train_amd import ops


def mesh_exterior(points, obj_verts, faces_dev, patches=1):
    """points [B,P,3], obj_verts [B,Nv,3], faces_dev [F,3] int32 -> (exterior bool [B,P], hits int32 with the inside test in
    its parity).  ``patches > 1``: faces are that many equal groups, each a closed surface; interior = inside any of them."""
    hits = ops.mesh_contains_hits(points, obj_verts, faces_dev, patches=patches)
    return (hits & 1) == 0, hits


def batch_mesh_contains_points(ray_origins, obj_triangles, direction=None):
    if direction is not None:
        raise NotImplementedError("the ray direction is the reference's fixed constant (contactutils.py:65)")
    B, T = obj_triangles.shape[:2]
    verts = obj_triangles.reshape(B, T * 3, 3)
    faces = torch.arange(T * 3, dtype=torch.int32, device=obj_triangles.device).view(T, 3)
    return mesh_exterior(ray_origins, verts, faces)[0]
