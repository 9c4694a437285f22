"""Surface samples of an object mesh for the Chamfer ground truth.

Behavioural mirror of ``points_from_mesh`` (reference ``handobjectdatasets/vertexsample.py:11-29``): triangles are drawn
with probability proportional to their area, then a point is drawn uniformly inside each.  The global ``np.random`` stream
is consumed exactly like the reference does (one ``choice`` of ``vertex_nb`` triangle ids, then two ``rand(vertex_nb, 1)``
blocks) and the float arithmetic keeps its order, so a seeded run samples the very same points.
"""
import numpy as np


def points_from_mesh(faces, vertices, vertex_nb=600, show_cloud=False):
    """faces ``[F,3]`` int, vertices ``[V,3]`` -> ``[vertex_nb,3]`` points on the surface."""
    if show_cloud:
        raise NotImplementedError("the matplotlib preview of the sampled cloud is outside the hot path")
    corners = vertices[faces]                                   # [F,3,3]
    normals = np.cross(corners[:, 1] - corners[:, 0], corners[:, 2] - corners[:, 0])
    area = 0.5 * np.linalg.norm(normals, axis=1)
    picked = np.random.choice(range(area.shape[0]), size=vertex_nb, p=area / area.sum())
    # barycentric weights of the two edges leaving corner 0: uniform on the unit square, mirrored into the lower triangle
    weights = [np.random.rand(vertex_nb, 1), np.random.rand(vertex_nb, 1)]
    mirrored = weights[0] + weights[1] > 1
    for w in weights:
        w[mirrored] = 1 - w[mirrored]
    tri = vertices[faces[picked]]
    origin = tri[:, 0]
    return origin + weights[0] * (tri[:, 1] - origin) + weights[1] * (tri[:, 2] - origin)
