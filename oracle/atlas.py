"""Oracle: AtlasNet sphere decoder, atlas loss, edge regulariser.

Follows ``mano_train/networks/branches/atlasutils.py:42-75`` (PointGenCon),
``atlasbranch.py:110-150`` (forward_inference), ``:153-167`` (edge_loss) and
``:199-287`` (AtlasLoss.compute_loss), functional over state-dict-named tensors
(``decoder.conv{1..4}.*``, ``decoder.bn{1..3}.*``, ``decode_trans.{0,2}.*``,
``decode_scale.{0,2}.*``).  The [B,3+C,N] concat is materialised as in the reference.
"""
import torch
import torch.nn.functional as F

from .chamfer import chamfer_loss


def pointgen(params, x, training=True, out_factor=200.0, prefix="decoder.", momentum=0.1, eps=1e-5, mfma_round=None):
    """PointGenCon.forward (atlasutils.py:65-75): x [B,C,N] -> [B,3,N]; 3x(conv1d k=1, BN1d, ReLU), conv1d, x out_factor.

    ``mfma_round`` (not in the reference): the model of the build's bf16-MFMA flavour, e.g. ``lambda t: t.bfloat16().float()``:
    both operands of the layer-2 and layer-3 contractions are rounded, and so are the layer outputs h2 / h3 the build keeps
    in bf16 (rounded BEFORE BatchNorm, whose statistics are those of the stored values); fp32 accumulation, fp32 BatchNorm."""
    h = x
    for k in (1, 2, 3):
        w = params["%sconv%d.weight" % (prefix, k)]
        if mfma_round is not None and k in (2, 3):
            h, w = mfma_round(h), mfma_round(w)
        h = F.conv1d(h, w, params["%sconv%d.bias" % (prefix, k)])
        if mfma_round is not None and k in (2, 3):
            h = mfma_round(h)
        h = F.batch_norm(
            h, params.get("%sbn%d.running_mean" % (prefix, k)), params.get("%sbn%d.running_var" % (prefix, k)),
            params["%sbn%d.weight" % (prefix, k)], params["%sbn%d.bias" % (prefix, k)],
            training=training, momentum=momentum, eps=eps,
        )
        h = F.relu(h)
    h = F.conv1d(h, params["%sconv4.weight" % prefix], params["%sconv4.bias" % prefix])
    return out_factor * h


def _mlp2(params, prefix, x):
    h = F.relu(F.linear(x, params[prefix + "0.weight"], params[prefix + "0.bias"]))
    return F.linear(h, params[prefix + "2.weight"], params[prefix + "2.bias"])


def forward_inference(params, features, template_verts, faces, predict_trans=False, predict_scale=False,
                      separate_features=None, training=True, out_factor=200.0, mfma_round=None):
    """AtlasBranch.forward_inference (atlasbranch.py:110-150).  template_verts [N,3], features [B,C].
    ``mfma_round`` (not in the reference): see ``pointgen`` - the model of the build's bf16-MFMA decoder flavour."""
    B = features.shape[0]
    trans = _mlp2(params, "decode_trans.", features) if predict_trans else None
    scale = _mlp2(params, "decode_scale.", features) if predict_scale else None
    grid = template_verts.unsqueeze(0).repeat(B, 1, 1).transpose(2, 1)  # [B,3,N]
    dec_feat = separate_features if separate_features is not None else features
    x = torch.cat((grid, dec_feat.unsqueeze(2).repeat(1, 1, grid.shape[2])), 1)
    verts = pointgen(params, x, training=training, out_factor=out_factor, mfma_round=mfma_round).transpose(2, 1)
    if predict_scale:
        scaled = scale.unsqueeze(1) * verts
        if predict_trans:
            points = scaled + trans.unsqueeze(1)
    elif predict_trans:
        points = verts + trans.unsqueeze(1)
    if not predict_scale and not predict_trans:
        res = {"objpoints3d": verts, "objfaces": faces}
    if predict_trans:
        res = {"objpoints3d": points, "objtrans": trans, "objpointscentered3d": verts, "objfaces": faces}
    if predict_scale:
        res["objscale"] = scale
    return res


def forward_random(params, features, rand_grid, predict_trans=False, training=True, out_factor=200.0):
    """AtlasBranch.forward (atlasbranch.py:78-108) with the normal draws given: rand_grid [B,3,P] (un-normalised)."""
    trans = _mlp2(params, "decode_trans.", features) if predict_trans else None
    grid = rand_grid / torch.sqrt(torch.sum(rand_grid ** 2, dim=1, keepdim=True))
    x = torch.cat((grid, features.unsqueeze(2).repeat(1, 1, grid.size(2))), 1)
    verts = pointgen(params, x, training=training, out_factor=out_factor).transpose(2, 1)
    if predict_trans:
        return {"objpoints3d": verts + trans.unsqueeze(1), "objtrans": trans, "objpointscentered3d": verts}
    return {"objpoints3d": verts}


def edge_loss(verts, faces):
    """atlasbranch.py:153-167: mean |squared edge length - per-sample mean squared edge length|."""
    f = torch.as_tensor(faces).long()
    a, b, c = verts[:, f[:, 0]], verts[:, f[:, 1]], verts[:, f[:, 2]]
    la = ((b - a) ** 2).sum(2)
    lb = ((c - b) ** 2).sum(2)
    lc = ((a - c) ** 2).sum(2)
    edges = torch.cat([lc, lb, la], 1)
    return (edges - edges.mean(1, keepdim=True)).abs().mean()


def atlas_loss(preds, gt_points, lambda_atlas=None, final_lambda_atlas=None, trans_weight=0, scale_weight=0,
               edge_regul_lambda=None):
    """AtlasLoss.compute_loss (atlasbranch.py:199-287), Chamfer only (the reference removed EMD)."""
    out = {}
    if gt_points is not None and (lambda_atlas or final_lambda_atlas):
        if "objtrans" in preds and "objpointscentered3d" in preds:
            centroids = gt_points.mean(1)
            l_trans = F.mse_loss(preds["objtrans"], centroids)
            out["atlas_trans3d"] = l_trans
            centred = gt_points - centroids.unsqueeze(1)
            if "objscale" in preds:
                gt_scale = torch.norm(centred, 2, 2).max(1)[0]
                l_scale = F.mse_loss(preds["objscale"], gt_scale.unsqueeze(1))
                out["atlas_scale3d"] = l_scale
            else:
                l_scale = 0
            c1, c2 = chamfer_loss(preds["objpointscentered3d"], centred)
            sym = torch.mean(c1 + c2)
            mesh = preds["objpointscentered3d"]
            f1, f2 = chamfer_loss(preds["objpoints3d"], gt_points)
            sym_final = torch.mean(f1 + f2)
            out["final_chamfer_loss"] = sym_final
            final = lambda_atlas * sym + final_lambda_atlas * sym_final + trans_weight * l_trans + scale_weight * l_scale
        else:
            if "objpoints3d" in preds and lambda_atlas:
                c1, c2 = chamfer_loss(preds["objpoints3d"], gt_points)
                sym = torch.mean(c1 + c2)
                final = lambda_atlas * sym
                mesh = preds["objpoints3d"]
            # else: the reference raises UnboundLocalError below
        if edge_regul_lambda is not None and edge_regul_lambda > 0:
            l_edge = edge_loss(mesh, preds["objfaces"])
            out["atlas_edge_regul"] = l_edge
            final = final + edge_regul_lambda * l_edge
    else:
        sym = None
        final = torch.zeros(1)
    out["atlas_objpoints3d"] = sym
    return final, out
