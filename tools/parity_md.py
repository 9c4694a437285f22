"""gpurun_out/parity_measured.jsonl (tests/conftest.py:record_measurement, one full `pytest -m gpu` run) -> profiles/r06_parity_measured.md.
    python tools/parity_md.py gpurun_out/parity_measured.jsonl profiles/r06_parity_measured.md"""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = {}
for line in open(src):
    d = json.loads(line)
    rows[d.pop("test")] = d  # the last record of a test wins


def fmt(v):
    if isinstance(v, float):
        return "%.3g" % v
    if isinstance(v, dict):
        return "{" + ", ".join("%s: %s" % (k, fmt(x)) for k, x in v.items()) + "}"
    if isinstance(v, (list, tuple)):
        return "[" + ", ".join(fmt(x) for x in v) + "]"
    return str(v)


def table(names, cols):
    out = ["| test | " + " | ".join(cols) + " |", "|---|" + "---|" * len(cols)]
    for n in names:
        if n in rows:
            out.append("| `%s` | " % n + " | ".join(fmt(rows[n].get(c, "")) for c in cols) + " |")
    return "\n".join(out)


with open(dst, "w") as fh:
    fh.write("# Measured parity, round 6 (MI355X, one `pytest -m gpu` run of the final tree; written by `tests/conftest.py:record_measurement`, "
             "`tools/parity_md.py`)\n\nRelative errors unless a column says otherwise.\n\n")
    fh.write("## (a) inside test against exact arithmetic: adversarial scenes through the fp32 oracle and the kernels (VERDICT r05 task 2a)\n\n")
    adv = sorted(n for n in rows if n.startswith("contains_adversarial"))
    fh.write(table(adv, ["points", "with_a_crossing", "in_fixed_1e-5_margin", "in_error_bound", "kernel_vs_exact_disagree_decided",
                         "oracle_vs_exact_disagree_decided", "kernel_vs_oracle_disagree_decided", "kernel_vs_oracle_disagree_in_bound",
                         "kernel_vs_exact_disagree_in_bound", "oracle_vs_exact_disagree_in_bound", "kernel_vs_oracle_disagree_outside_fixed_margin"]) + "\n\n")
    fh.write("`[seed, scenes, points per scene, faces]`.  A point is DECIDED when no pair's exact u, v, 1 - u - v, t, det (fp64 on the fp32 inputs) lies within the fp32 "
             "forward-error bound 8 eps Q |tvec| / |e| of its threshold (Q = |e1||e2| / |det|): there every fp32 evaluation must return the exact crossing count, and does - kernel and "
             "oracle alike (0 everywhere).  Inside the bound both are arbitrary: they differ from each other on 12 - 13 % of those points, and each from exact arithmetic about as often.  "
             "The fixed 1e-5 margin of the smooth-scene tests does not separate the classes on ill-conditioned triangles (last column).\n\n")
    g = sorted(n for n in rows if n.startswith("contains_graze"))
    fh.write("Smooth scenes (random clouds against blobs), margin 1e-5 as in round 5:\n\n" + table(g, ["margin", "points", "in_margin", "in_margin_disagree"]) + "\n\n")
    fh.write("## (b) configs[2] in its stated precision against the oracle with the same roundings (task 2b)\n\n")
    fh.write(table(["configs2_dec_bf16_bs64_256_vs_bf16_oracle", "configs2_all_bf16_bs16_256_vs_autocast_oracle", "configs2_bs64_256_vs_oracle"],
                   ["total", "worst_soft", "worst_soft_term", "objpoints3d_of_scale", "objpoints3d_rms_of_scale", "verts_of_scale", "repulsion_hamming",
                    "attraction_hamming"]) + "\n\n")
    for n in ("configs2_dec_bf16_bs64_256_vs_bf16_oracle", "configs2_all_bf16_bs16_256_vs_autocast_oracle"):
        if n in rows:
            fh.write("* `%s`, every loss term: %s; gradients (relative L2 against the oracle's autograd, which does not round the gradient operands): %s\n"
                     % (n, fmt(rows[n].get("terms", {})), fmt(rows[n].get("grads_l2", {}))))
    fh.write("\n(last row: the fp32 model against the fp32 oracle, `worst_loss` %s on `%s`.)\n\n"
             % (fmt(rows.get("configs2_bs64_256_vs_oracle", {}).get("worst_loss")), rows.get("configs2_bs64_256_vs_oracle", {}).get("worst_term")))
    fh.write("## (c) the recorded step under a second capture and pool churn (task 5c)\n\n")
    n = "single_graph_second_capture_and_pool_churn"
    if n in rows:
        fh.write("* eager vs eager (run-to-run noise of the same ten steps): %s\n* disturbed graph vs eager: %s\n* disturbed vs undisturbed graph: %s\n* weights, relative L2 "
                 "between the two graph runs: %s\n\n" % (fmt(rows[n]["eager_vs_eager"]), fmt(rows[n]["disturbed_graph_vs_eager"]), fmt(rows[n]["disturbed_vs_undisturbed_graph"]),
                                                          fmt(rows[n]["weights_rel_l2"])))
    fh.write("## (d) every other record of the run\n\n")
    done = set(adv) | set(g) | {"configs2_dec_bf16_bs64_256_vs_bf16_oracle", "configs2_all_bf16_bs16_256_vs_autocast_oracle", n}
    for k in sorted(rows):
        if k in done:
            continue
        r = {a: b for a, b in rows[k].items() if not isinstance(b, dict) or len(b) <= 12}
        fh.write("* `%s`: %s\n" % (k, fmt(r)))
