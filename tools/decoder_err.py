"""Dev tool: per-parameter error of the HIP decoder against the fp32 oracle AND an fp64 rerun of the oracle."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import test_decoder_gpu as T
from obman_train_amd import ops
from obman_train_amd.icosphere import multi_patch

cases = [(35, 3, 1, 1), (515, 4, 3, 1), (131, 5, 2, 1), (515, 2, 1, 25), (515, 8, 1, 25), (515, 64, 3, 1)]
for c1, B, subdiv, patches in cases:
    grid = torch.from_numpy(multi_patch(subdiv, patches)[0].astype(np.float32))
    rng = np.random.RandomState(int(os.environ.get('SEED', '100')))
    feats = torch.from_numpy(rng.normal(0, 1, size=(B, c1 - 3)).astype(np.float32))
    cot = torch.from_numpy(rng.normal(0, 1, size=(B, grid.shape[0], 3)).astype(np.float32))
    dec64 = T._decoder(c1, 7).double(); dec64.train(True)
    want, f_o, params = T._oracle(dec64, feats.double(), grid.double(), True)
    (want * cot.double()).sum().backward()
    dec_g = T._decoder(c1, 7).cuda(); dec_g.train(True)
    f_g = feats.cuda().requires_grad_()
    got = ops.pointgen_decode(dec_g, f_g, grid.cuda())
    (got * cot.cuda()).sum().backward()
    line = ["out %.1e" % ((got.detach().cpu().double() - want).abs().max() / want.abs().max()).item(),
            "feat %.1e" % ((f_g.grad.cpu().double() - f_o.grad).abs().max() / f_o.grad.abs().max()).item()]
    for name, prm in dec_g.named_parameters():
        w = params["decoder." + name].grad.reshape(prm.grad.shape)
        if name.startswith("conv") and name.endswith("bias") and name != "conv4.bias":
            continue
        line.append("%s %.1e" % (name.replace("weight", "w").replace("bias", "b"), ((prm.grad.cpu().double() - w).abs().max() / w.abs().max()).item()))
    if c1 > 3:
        w = params["decoder.conv1.weight"].grad.reshape(dec_g.conv1.weight.shape)
        g = dec_g.conv1.weight.grad.cpu().double()
        line.append("c1w[:, :3] %.1e c1w[:, 3:] %.1e" % (((g[:, :3] - w[:, :3]).abs().max() / w[:, :3].abs().max()).item(),
                                                        ((g[:, 3:] - w[:, 3:]).abs().max() / w[:, 3:].abs().max()).item()))
    print((c1, B, subdiv, patches), " ".join(line))
