"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/prof_pages.py into profiles/<name>.json:
HBM bytes per launch for every kernel and for the CRAFT convolution kernels as a group.
  python tools/pmc_to_json.py <fetch_counter_collection.csv> <write_counter_collection.csv> <pages> <steps> <out.json>"""
import collections, csv, json, sys

fetch_csv, write_csv, pages, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
CONV = ("conv3p", "conv3h", "conv1u", "gemm2_kernel", "gemm_sp_kernel", "conv1_direct", "conv1_split", "igemm_kernel", "conv3s")   # (detector-only runs: every such launch is CRAFT's)


def fold(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg


f, w = fold(fetch_csv, "FETCH_SIZE"), fold(write_csv, "WRITE_SIZE")
kernels = {}
for k in sorted(set(f) | set(w)):
    n = f[k][0] or w[k][0]
    kernels[k[:120]] = {"launches_per_step": n / steps, "fetch_bytes_per_launch": 2 * 1024 * f[k][1] / max(1, f[k][0]),
                        "write_bytes_per_launch": 1024 * w[k][1] / max(1, w[k][0])}
# run prof_pages.py with 0 words per page (no crops: the detector only), so every conv / gemm2 launch seen is CRAFT's
# PMC_EXCLUDE="pat1;pat2": kernel-name fragments that do NOT belong to CRAFT although they match (the f16x4 engine's recogniser GEMMs are
# gemm2_kernel<..., 4> - a blank page still yields a stray crop or two)
import os
EXC = [x for x in os.environ.get("PMC_EXCLUDE", "").split(";") if x]
conv = {k: v for k, v in kernels.items() if any(c in k for c in CONV) and not any(x in k for x in EXC)}
tot_f = sum(v["fetch_bytes_per_launch"] * v["launches_per_step"] for v in conv.values())
tot_w = sum(v["write_bytes_per_launch"] * v["launches_per_step"] for v in conv.values())
n = sum(v["launches_per_step"] for v in conv.values())
# every kernel of the detector-only step but the copies and the CCL: the convolutions plus the elementwise kernels between them (pools, upsamples)
DET_EXC = ("copyBuffer", "fillBuffer", "ccl_", "rowext", "binarize", "candidates", "minmax")
det = {k: v for k, v in kernels.items() if not any(x in k for x in DET_EXC) and not any(x in k for x in EXC)}
det_f = sum(v["fetch_bytes_per_launch"] * v["launches_per_step"] for v in det.values())
det_w = sum(v["write_bytes_per_launch"] * v["launches_per_step"] for v in det.values())
json.dump({
    "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 tools/prof_pages.py %d %d 0  (second pass: --pmc WRITE_SIZE)" % (pages, steps),
    "corrections": "counter values are KiB; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE as is",
    "pages_per_step": pages, "pages": pages, "steps": steps, "kernels": kernels,
    "craft_conv_kernels": {"launches_per_step": n, "fetch_bytes_per_step": tot_f, "write_bytes_per_step": tot_w,
                               "hbm_bytes_per_launch": (tot_f + tot_w) / max(1, n)},
    "detector_all_kernels": {"launches_per_step": sum(v["launches_per_step"] for v in det.values()), "fetch_bytes_per_step": det_f, "write_bytes_per_step": det_w,
                             "hbm_GB_per_page": (det_f + det_w) / 1e9 / pages, "note": "resize + every CRAFT launch (convolutions, pools, upsamples); CCL and copies excluded"},
}, open(out, "w"), indent=1)
print(out, "conv+gemm2 launches/step", n, "GB/step", (tot_f + tot_w) / 1e9, "| all detector kernels GB/page", (det_f + det_w) / 1e9 / pages)
