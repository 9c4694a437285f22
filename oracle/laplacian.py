"""Oracle: template cotangent Laplacian and the Laplacian regulariser.

Follows ``mano_train/networks/branches/laplacianloss.py:24-185``: cotangent weights of the *template* sphere
(``cotangent`` :153-185, Heron's formula, columns = edges 23, 31, 12), symmetric sparse matrix minus its row sums
(:98-128), ``Lx`` per sample (:133), loss = mean over all B*N vertices of ``||(Lx)_i||_2`` (:36-41); backward of the
matrix product is ``L g`` (:137-150, L is symmetric).  The reference wraps this in a legacy ``autograd.Function`` that
raises on torch >= 1.5 when called; its ``forward`` / ``backward`` methods are still callable and pin this restatement."""
import numpy as np
import torch
from scipy import sparse


def cotangent_weights(verts, faces):
    """verts [N,3] float32 torch, faces [F,3] int -> C [F,3] (cot/2 of the angle opposite to edges 23, 31, 12)."""
    f = torch.as_tensor(np.asarray(faces)).long()
    v1, v2, v3 = verts[f[:, 0]], verts[f[:, 1]], verts[f[:, 2]]
    l1 = torch.sqrt(((v2 - v3) ** 2).sum(1))
    l2 = torch.sqrt(((v3 - v1) ** 2).sum(1))
    l3 = torch.sqrt(((v1 - v2) ** 2).sum(1))
    sp = (l1 + l2 + l3) * 0.5
    A = 2 * torch.sqrt(sp * (sp - l1) * (sp - l2) * (sp - l3))
    c23 = l2 ** 2 + l3 ** 2 - l1 ** 2
    c31 = l1 ** 2 + l3 ** 2 - l2 ** 2
    c12 = l1 ** 2 + l2 ** 2 - l3 ** 2
    return torch.stack([c23, c31, c12], 1) / A.unsqueeze(1) / 4


def laplacian_matrix(verts, faces):
    """scipy CSR [N,N]: L = (C + C^T) - diag(row sums), float32 like the reference."""
    faces = np.asarray(faces)
    n = verts.shape[0]
    C = cotangent_weights(verts, faces).numpy()
    rows = faces[:, [1, 2, 0]].reshape(-1)
    cols = faces[:, [2, 0, 1]].reshape(-1)
    L = sparse.csr_matrix((C.reshape(-1), (rows, cols)), shape=(n, n))
    L = L + L.T
    M = sparse.diags(np.array(np.sum(L, 1)).reshape(-1), format="csr")
    return (L - M).tocsr()


def laplacian_loss(L, verts):
    """verts [B,N,3] torch (autograd through a dense copy of L; test sizes only) -> scalar."""
    Ld = torch.from_numpy(L.toarray()).to(verts.dtype)
    Lx = torch.einsum("ij,bjc->bic", Ld, verts)
    return torch.norm(Lx.reshape(-1, 3), p=2, dim=1).mean(), Lx
