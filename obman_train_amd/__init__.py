"""obman_train_amd - MI355X-native (gfx950) mesh-loss hot path of hassony2/obman_train.

ResNet18 -> {MANO LBS, AtlasNet sphere decoder} -> Chamfer + contact/penetration losses,
behind the reference's ``HandNet.forward(sample) -> (total_loss, results, losses)`` API
(``mano_train/networks/handnet.py:198-392``).  The hot ops are hand-written HIP kernels
in ``csrc/`` behind a C-ABI (``include/obman_hip.h``); see DESIGN.md.
"""
__version__ = "0.1.0"
