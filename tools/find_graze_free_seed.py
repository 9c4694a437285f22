"""CPU helper for tests/test_fullsize_gpu.py: find a scene seed whose hand vertices do not graze any triangle border of the
25-patch object under the fp64 margin of tests/test_contact_gpu._margin_ok (both the HIP kernel and the oracle are
arbitrary for grazing rays, so the full-size parity test uses graze-free scenes).

    python tools/find_graze_free_seed.py 4 1      # subdivision, batch
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("OBMAN_MANO_SYNTHETIC", "1")


def main():
    from tests.test_fullsize_gpu import _c3_scene, _graze_free

    subdiv, batch = int(sys.argv[1]), int(sys.argv[2])
    for seed in range(12, 64):
        hand, obj, faces = _c3_scene(batch, seed, subdiv)
        ok = _graze_free(hand, obj, faces, 25)
        print("seed %d: %d grazing vertices" % (seed, int((~ok).sum())), flush=True)
        if bool(ok.all()):
            break


if __name__ == "__main__":
    main()
