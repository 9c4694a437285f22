"""Zero-edit drop-in: an unmodified script that imports ``mano_train.networks.handnet`` the way
``traineval.py:12-13`` does gets this package's mirrors, while un-mirrored ``mano_train.*`` modules still come from the
script's own tree.  Uses a stand-in tree with the reference's layout (namespace packages, no ``__init__.py``)."""
import json
import os
import subprocess
import sys

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

SCRIPT = '''
import json, sys
from mano_train.networks.handnet import HandNet
from mano_train.networks import netutils
from mano_train.options import someopts          # not mirrored: must come from this tree
import mano_train.networks.branches.atlasutils as au
json.dump({"handnet": HandNet.__module__, "netutils": netutils.__name__, "opts": someopts.WHO, "chamfer": au.ChamferLoss.__module__,
           "argv": sys.argv[1:], "main": __name__}, open(sys.argv[1], "w"))
'''


def test_unmodified_script_resolves_to_the_mirrors(tmp_path):
    tree = tmp_path / "ref"
    (tree / "mano_train" / "networks").mkdir(parents=True)
    (tree / "mano_train" / "options").mkdir(parents=True)
    (tree / "mano_train" / "networks" / "handnet.py").write_text("class HandNet:\n    pass\n")   # the 'reference' one: must lose
    (tree / "mano_train" / "networks" / "netutils.py").write_text("WHO = 'reference'\n")
    (tree / "mano_train" / "options" / "someopts.py").write_text("WHO = 'reference tree'\n")
    (tree / "train.py").write_text(SCRIPT)
    out = tmp_path / "out.json"
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""), OBMAN_MANO_SYNTHETIC="1")
    subprocess.run([sys.executable, "-m", "obman_train_amd.dropin", str(tree / "train.py"), str(out), "--flag"], check=True, env=env,
                   cwd=str(tmp_path))
    got = json.loads(out.read_text())
    assert got["handnet"] == "obman_train_amd.networks.handnet"
    assert got["netutils"] == "obman_train_amd.networks.netutils"
    assert got["chamfer"] == "obman_train_amd.networks.branches.atlasutils"
    assert got["opts"] == "reference tree"
    assert got["argv"] == [str(out), "--flag"] and got["main"] == "__main__"


def test_install_is_idempotent_and_lists_what_it_aliased():
    code = ("import obman_train_amd.dropin as d, sys; a = d.install(); b = d.install(extra=True); "
            "import mano_train.netscripts.epochpass3d as e; "
            "assert e.__name__ == 'obman_train_amd.netscripts.epochpass3d', e.__name__; "
            "assert 'mano_train.networks.handnet' in a and len(b) > len(a); print('ok')")
    env = dict(os.environ, PYTHONPATH=REPO, OBMAN_MANO_SYNTHETIC="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr
