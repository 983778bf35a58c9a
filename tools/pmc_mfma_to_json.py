"""Fold a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES / SQ_WAVE_CYCLES, GRBM_GUI_ACTIVE) + its kernel trace into
profiles/<name>.json: per kernel the matrix-core utilisation rocprof reports.
  python tools/pmc_mfma_to_json.py <counter_collection.csv> <kernel_trace.csv or ''> <out.json> [note]

MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8): the busy counter sums the cycles every
SIMD's matrix pipe was occupied (16 per v_mfma_f32_16x16x32_bf16, 32 per 32x32x16), GRBM_GUI_ACTIVE sums the 8 XCDs' active cycles
(MI355X_MICROARCH.md, DVFS give-back).  A kernel issuing MFMAs back to back on every SIMD reads 1.0."""
import collections, csv, json, sys

cc, kt, out = sys.argv[1], sys.argv[2], sys.argv[3]
note = sys.argv[4] if len(sys.argv) > 4 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(cc)):
    a = agg[r["Kernel_Name"]][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
dur = collections.defaultdict(lambda: [0, 0.0])
if kt:
    for r in csv.DictReader(open(kt)):
        d = dur[r["Kernel_Name"]]
        d[0] += 1
        d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
kernels = {}
for k, cs in agg.items():
    n = max(v[0] for v in cs.values())
    e = {"dispatches": n}
    for c, v in cs.items():
        e[c + "_per_dispatch"] = v[1] / max(1, v[0])
    mf, gui = cs.get("SQ_VALU_MFMA_BUSY_CYCLES"), cs.get("GRBM_GUI_ACTIVE")
    if mf and gui and gui[1] > 0:
        e["mfma_util"] = mf[1] / (4 * 256 * gui[1] / 8.0)
    if k in dur and dur[k][0]:
        e["avg_us_under_profiler"] = dur[k][1] / dur[k][0]
        if gui and gui[1] > 0:
            e["clock_ghz"] = (gui[1] / gui[0] / 8.0) / (dur[k][1] / dur[k][0] * 1e3)
    kernels[k[:140]] = e
top = dict(sorted(kernels.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES_per_dispatch", 0) * kv[1]["dispatches"])[:24])
json.dump({"note": note, "formula": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)", "kernels": top}, open(out, "w"), indent=1)
for k, e in top.items():
    print("%-70s n=%4d util=%s us=%s" % (k[:70], e["dispatches"], "%.3f" % e["mfma_util"] if "mfma_util" in e else "-", "%.1f" % e.get("avg_us_under_profiler", 0)))
