"""HBM traffic per launch and kernel class from two rocprofv3 counter passes.

    python tools/pmc_traffic.py <dir of --pmc FETCH_SIZE run> <dir of --pmc WRITE_SIZE run> "<command profiled>" > profiles/<name>.json

FETCH_SIZE / WRITE_SIZE are reported in KiB-sized units of 1 KB (rocprofv3 derived metric); on gfx950 FETCH_SIZE tallies a
128-B request as 64 B, so reads are doubled (MI355X_MICROARCH.md, HBM / rocprofv3 section).  WRITE_SIZE is taken as is.
"""
import csv, glob, json, re, sys, collections


def per_kernel(d, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = re.sub(r"<.*", "", r["Kernel_Name"].split("(")[0]).split("::")[-1].split(" ")[-1]
            tot[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return tot, cnt


def main():
    fd, wd, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    rt, rc = per_kernel(fd, "FETCH_SIZE")
    wt, wc = per_kernel(wd, "WRITE_SIZE")
    out = {}
    for k in rt:
        if not k.startswith(("gemm", "lstm", "conv", "stack", "embed", "attn", "beam", "greedy", "split")):
            continue
        out[k] = {"hbm_read_bytes_per_launch": int(2 * 1024 * rt[k] / rc[k]),
                  "hbm_write_bytes_per_launch": int(1024 * wt[k] / max(wc[k], 1)),
                  "launches_sampled": rc[k],
                  "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `%s`; FETCH_SIZE doubled "
                            "(gfx950 tallies 128-B requests as 64 B); mean over all launches of the kernel" % cmd}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
