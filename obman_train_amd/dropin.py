"""Zero-edit drop-in: run an UNMODIFIED reference script with this package behind the reference's module names.

    python -m obman_train_amd.dropin /path/to/obman_train/traineval.py --atlas_mesh --mano_use_pca ...

`traineval.py:12-13` does ``from mano_train.networks.handnet import HandNet`` / ``from mano_train.networks import netutils``.
The script's own directory is ``sys.path[0]``, so a shadow package on ``PYTHONPATH`` could never win against the reference's
files; instead the mirror modules are registered in ``sys.modules`` under the reference's dotted names BEFORE the script runs
(the import system looks there first), and the script is executed with ``runpy`` as ``__main__`` with its own directory at
the front of ``sys.path`` - exactly what ``python traineval.py`` sets up.  Everything that is not mirrored (``mano_train.options``,
``mano_train.exputils``, dataset readers ...) still resolves to the reference's own files: ``mano_train`` has no ``__init__.py``
(a namespace package), so its other sub-packages import normally.

``install()`` is the opt-in for a caller who prefers one line at the top of their own script:

    import obman_train_amd.dropin; obman_train_amd.dropin.install()
"""
import importlib
import os
import runpy
import sys
import types

# reference dotted name -> mirror in this package (same public names, signatures, dict keys, state-dict layout)
MIRRORS = {
    "mano_train.networks.handnet": "obman_train_amd.networks.handnet",
    "mano_train.networks.netutils": "obman_train_amd.networks.netutils",
    "mano_train.networks.bases.resnet": "obman_train_amd.networks.bases.resnet",
    "mano_train.networks.branches.manobranch": "obman_train_amd.networks.branches.manobranch",
    "mano_train.networks.branches.atlasbranch": "obman_train_amd.networks.branches.atlasbranch",
    "mano_train.networks.branches.atlasutils": "obman_train_amd.networks.branches.atlasutils",
    "mano_train.networks.branches.contactloss": "obman_train_amd.networks.branches.contactloss",
    "mano_train.networks.branches.contactutils": "obman_train_amd.networks.branches.contactutils",
    "mano_train.networks.branches.laplacianloss": "obman_train_amd.networks.branches.laplacianloss",
}
# the f-rows of SURVEY §8 (epoch loop, checkpoint I/O, meters): opt-in, the reference's own versions work too
EXTRA_MIRRORS = {
    "mano_train.netscripts.epochpass3d": "obman_train_amd.netscripts.epochpass3d",
    "mano_train.modelutils.modelio": "obman_train_amd.modelutils.modelio",
    "mano_train.evaluation.evalutils": "obman_train_amd.evaluation.evalutils",
    "mano_train.evaluation.zimeval": "obman_train_amd.evaluation.zimeval",
}


def _namespace(name):
    """A stand-in parent package that still finds the reference's un-mirrored sub-modules: a namespace-style module whose
    ``__path__`` lists every ``<sys.path entry>/<name as path>`` directory that exists."""
    mod = sys.modules.get(name)
    if mod is not None:
        return mod
    rel = name.replace(".", os.sep)
    mod = types.ModuleType(name)
    mod.__path__ = [os.path.join(p or ".", rel) for p in sys.path if os.path.isdir(os.path.join(p or ".", rel))]
    mod.__package__ = name
    sys.modules[name] = mod
    return mod


def install(extra=False):
    """Register the mirrors under the reference's module names.  Returns the list of names installed."""
    table = dict(MIRRORS)
    if extra:
        table.update(EXTRA_MIRRORS)
    done = []
    for ref_name, ours in table.items():
        module = importlib.import_module(ours)
        parts = ref_name.split(".")
        for k in range(1, len(parts)):  # parents first, so `from mano_train.networks import netutils` finds the attribute
            _namespace(".".join(parts[:k]))
        sys.modules[ref_name] = module
        setattr(sys.modules[".".join(parts[:-1])], parts[-1], module)
        done.append(ref_name)
    return done


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    extra = False
    if argv and argv[0] == "--all-mirrors":
        extra, argv = True, argv[1:]
    if not argv:
        raise SystemExit("usage: python -m obman_train_amd.dropin [--all-mirrors] <reference script.py> [script arguments ...]")
    script = os.path.abspath(argv[0])
    sys.path.insert(0, os.path.dirname(script))  # what `python script.py` does; must precede install() (namespace paths)
    install(extra=extra)
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
