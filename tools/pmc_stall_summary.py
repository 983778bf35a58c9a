"""Per-kernel means of every counter in rocprofv3 counter_collection CSVs (one row per dispatch and counter).
   python tools/pmc_stall_summary.py <csv> [<csv> ...]   -> table on stdout"""
import csv, sys, collections
for path in sys.argv[1:]:
    try:
        rows = list(csv.DictReader(open(path)))
    except OSError as e:
        print(path, e); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    seen = set()
    for r in rows:
        k = r["Kernel_Name"].replace("void ttr::", "").replace("(ttr::ConvParams)", "")[:52]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); cnt[k] += 1
    names = sorted({c for k in acc for c in acc[k]})
    print("==", path.split("/")[-1]); print(f"{'kernel':52s} {'n':>5s} " + " ".join(f"{n[-16:]:>16s}" for n in names))
    for k in sorted(acc, key=lambda k: -max(acc[k].values())):
        if "gemm" in k or "attn" in k or "layernorm_planes" in k or "conv" in k:
            print(f"{k:52s} {cnt[k]:5d} " + " ".join(f"{acc[k][n] / cnt[k]:16.0f}" for n in names))
