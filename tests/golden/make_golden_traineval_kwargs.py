"""Dev-container generator: the keyword dictionaries `traineval.py` builds for `HandNet(**kwargs)`.

Runs the REFERENCE's own argument parsers (`mano_train/options/{datasetopts,nets3dopts,expopts}.py`, imported from
/root/reference) on three command lines and evaluates the keyword expressions of the `HandNet(...)` call in
`/root/reference/traineval.py:39-76` (read from its AST - `main()` itself is not run: it needs datasets) against the parsed
namespace.  The result - plain names and values, no reference source - is committed as `tests/golden/traineval_kwargs.json`;
`tests/test_traineval_kwargs.py` constructs this package's `HandNet` from every entry.

    python tests/golden/make_golden_traineval_kwargs.py
"""
import argparse
import ast
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, REF)

RECIPES = {
    "default_cli": [],
    # README.md:133 (the full ObMan recipe)
    "readme_full": "--atlas_predict_trans --atlas_predict_scale --atlas_mesh --mano_use_shape --mano_use_pca --freeze_batchnorm "
                   "--atlas_separate_encoder".split(),
    # BASELINE.json configs[0] / [1] flags
    "baseline_configs1": "--atlas_mesh --mano_use_pca --atlas_lambda 0.167".split(),
    # contact + penetration fine-tuning (README's contact options; BASELINE.json configs[2])
    "contact": "--atlas_predict_trans --atlas_predict_scale --atlas_mesh --atlas_lambda 0.167 --mano_use_shape --mano_use_pca --contact_lambda 1 "
               "--collision_lambda 1 --contact_thresh 10 --collision_thresh 20 --contact_mode dist_tanh --collision_mode dist_tanh "
               "--contact_zones zones --contact_target all".split(),
}


def main():
    from mano_train.options import datasetopts, expopts, nets3dopts

    tree = ast.parse(open(os.path.join(REF, "traineval.py")).read())
    call = None
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and getattr(node.func, "id", None) == "HandNet":
            call = node
            break
    assert call is not None and not call.args
    out = {"_source": "traineval.py:39-76 keyword expressions evaluated on the reference's parsers (tests/golden/make_golden_traineval_kwargs.py)"}
    for name, argv in RECIPES.items():
        parser = argparse.ArgumentParser()
        datasetopts.add_dataset_opts(parser)
        datasetopts.add_dataset3d_opts(parser)
        nets3dopts.add_nets3d_opts(parser)
        nets3dopts.add_train3d_opts(parser)
        expopts.add_exp_opts(parser)
        args = parser.parse_args(argv)
        kwargs = {}
        for kw in call.keywords:
            kwargs[kw.arg] = eval(compile(ast.Expression(kw.value), "traineval.py", "eval"), {"args": args})  # noqa: S307
        out[name] = {"argv": argv, "kwargs": kwargs,
                     "train_options": {k: getattr(args, k) for k in ("optimizer", "lr", "momentum", "weight_decay", "freeze_batchnorm",
                                                                     "freeze_encoder", "batch_size", "workers", "epochs")
                                       if hasattr(args, k)}}
    path = os.path.join(HERE, "traineval_kwargs.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(path, {k: len(v["kwargs"]) for k, v in out.items() if k != "_source"})


if __name__ == "__main__":
    main()
