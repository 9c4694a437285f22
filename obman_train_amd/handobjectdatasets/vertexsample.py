"""Area-weighted surface sampling of an object mesh (reference ``handobjectdatasets/vertexsample.py:6-30``): host numpy,
global ``np.random`` draws in the reference's order (choice, rand, rand)."""
import numpy as np


def tri_area(v):
    return 0.5 * np.linalg.norm(np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]), axis=1)


def points_from_mesh(faces, vertices, vertex_nb=600, show_cloud=False):
    if show_cloud:
        raise NotImplementedError("matplotlib preview of the sampled cloud is outside the hot path")
    areas = tri_area(vertices[faces])
    proba = areas / areas.sum()
    rand_idxs = np.random.choice(range(areas.shape[0]), size=vertex_nb, p=proba)
    u = np.random.rand(vertex_nb, 1)
    v = np.random.rand(vertex_nb, 1)
    outside = u + v > 1  # fold the unit square onto the lower triangle
    u[outside] = 1 - u[outside]
    v[outside] = 1 - v[outside]
    tris = vertices[faces[rand_idxs]]
    return tris[:, 0] + u * (tris[:, 1] - tris[:, 0]) + v * (tris[:, 2] - tris[:, 0])
