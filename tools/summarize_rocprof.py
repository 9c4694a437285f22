"""rocprofv3 ``*_kernel_stats.csv`` -> committed markdown summary: every obman HIP kernel plus the top-N others, per step.

    python tools/summarize_rocprof.py gpurun_out/r02_c2_kernel_stats.csv profiles/r02_c2_kernel_stats.md auto "command line"
(steps: number of train steps the profiled process ran - precondition, warm-up, timed and PCIe-probe steps all count;
"auto" reads it off a once-per-step kernel; MIOpen's naive_conv find-phase kernels are excluded)"""
import csv
import sys

OURS = ("pairmin", "rowmean2", "mano_", "contains_", "contact_", "dec::", "edge_", "laplacian", "bnact::", "imgstream", "blur_kernel",
        "warp_kernel", "mean_kernel", "adam_kernel", "adam_tick", "affine_fwd", "affine_bwd", "mse_fwd", "mse_bwd", "mse_finalize", "gt_stats",
        "bf16_shadow", "obman_fill_u32")


def main(src, dst, steps="auto", cmd="", top=30, steady=""):
    rows = [r for r in csv.DictReader(open(src)) if "naive_conv" not in r["Name"]]
    if steps == "auto":  # train steps executed by the profiled process = launches of a once-per-step kernel (MANO forward)
        steps = max([float(r["Calls"]) for r in rows if "mano_fwd_kernel" in r["Name"]] or [1.0])
    steps = float(steps)
    if steady:  # drop kernels launched in fewer than half of the steps: MIOpen's find-phase candidates, probes after the timed region
        # ("steady" = 0.5; a number sets the fraction: a fresh box runs the find phase inside the profiled process, and its candidates
        # then reach > 50 % of the step count - 0.9 keeps the per-step kernels only)
        try:
            frac = float(steady)
        except ValueError:
            frac = 0.5
        rows = [r for r in rows if float(r["Calls"]) >= frac * steps]
    tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
    ours = [r for r in rows if any(k in r["Name"] for k in OURS)]
    others = [r for r in rows if r not in ours]
    # MIOpen / CK / rocBLAS / hipBLASLt: the convolution and GEMM kernels AND MIOpen's own helpers around them (SubTensorOpWithScalar1d =
    # the zero fills of its split-K weight-gradient workspaces, SubTensorOpWithCastTensor1d = their fp32 -> bf16 casts; r06: these were
    # booked as "other PyTorch kernels" until round 5)
    conv = [r for r in others if "igemm" in r["Name"] or "ck::" in r["Name"] or "_ZN2ck" in r["Name"] or "Cijk" in r["Name"]
            or "SubTensorOp" in r["Name"]]

    def ms(sel):
        return sum(float(r["TotalDurationNs"]) for r in sel) / steps / 1e6

    with open(dst, "w") as fh:
        fh.write("# rocprofv3 --kernel-trace --stats, per training step\n\ncommand: `%s`\n\n" % cmd)
        fh.write("%d distinct kernels; **%.2f ms of kernel time per step**: obman_train_amd HIP kernels %.2f ms, MIOpen/CK/rocBLAS "
                 "convolutions + GEMMs (incl. MIOpen's SubTensorOp helper kernels) %.2f ms, other PyTorch kernels %.2f ms.\n\n" % (len(rows), tot, ms(ours), ms(conv), ms(others) - ms(conv)))
        for title, sel in (("obman_train_amd HIP kernels", ours), ("top %d other kernels (MIOpen / PyTorch)" % top, others[:top])):
            fh.write("## %s\n\n| kernel | calls/step | avg us | ms/step |\n|---|---|---|---|\n" % title)
            for r in sel:
                name = r["Name"].replace("(anonymous namespace)::", "").replace("|", "/")
                fh.write("| `%s` | %.1f | %.1f | %.3f |\n" % (name[:120], float(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3,
                                                          float(r["TotalDurationNs"]) / steps / 1e6))
            fh.write("\n")


if __name__ == "__main__":
    main(*sys.argv[1:5], steady=sys.argv[5] if len(sys.argv) > 5 else "")
