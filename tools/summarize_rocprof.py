"""rocprofv3 ``*_kernel_stats.csv`` -> small committed summary (markdown): every obman HIP kernel plus the
top-N others.  ``python tools/summarize_rocprof.py gpurun_out/c2_kernel_stats.csv profiles/r01_c2_kernel_stats.md "cmd"``"""
import csv
import sys

OURS = ("pairmin", "rowmean2", "mano_", "contains_kernel", "contact_", "pointgen", "decoder_", "edge_")


def main(src, dst, cmd="", top=25):
    rows = list(csv.DictReader(open(src)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    ours = [r for r in rows if any(k in r["Name"] for k in OURS)]
    others = [r for r in rows if r not in ours][:top]
    with open(dst, "w") as fh:
        fh.write("# rocprofv3 --kernel-trace --stats summary\n\n")
        fh.write("command: `%s`\n\nsource: `%s` (%d distinct kernels, %.1f ms total kernel time incl. MIOpen find/tuning launches)\n\n"
                 % (cmd, src, len(rows), tot / 1e6))
        for title, sel in (("obman_train_amd HIP kernels", ours), ("top %d other kernels (MIOpen / PyTorch)" % top, others)):
            fh.write("## %s\n\n| kernel | calls | total ms | avg us | min us | max us | %% |\n|---|---|---|---|---|---|---|\n" % title)
            for r in sel:
                name = r["Name"].replace("(anonymous namespace)::", "").replace("|", "/")
                fh.write("| `%s` | %s | %.3f | %.2f | %.2f | %.2f | %.3f |\n" % (
                    name[:110], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                    float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
            fh.write("\n")


if __name__ == "__main__":
    main(*sys.argv[1:4])
