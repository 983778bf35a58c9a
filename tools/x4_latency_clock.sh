#!/bin/bash
# GPU box: effective shader clock per kernel of single-page calls (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 / duration; rough below ~0.3 ms per
# dispatch, MI355X_MICROARCH.md "DVFS give-back") - is the latency regime running at the clock the batch regime holds?
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export TTR_PREC=f16x4
rm -rf /tmp/lc; rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d /tmp/lc -o p -- python3 $R/tools/prof_pages.py ${PAGES:-1} 6 40 bench_grid_boxes=1 "$@" > /tmp/lc.log 2>&1
python3 - <<'PY'
import collections, csv, glob
cc = sorted(glob.glob("/tmp/lc/**/*counter_collection.csv", recursive=True))[-1]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    k = r["Kernel_Name"].replace("void ttr::", "")[:58]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) if "End_Timestamp" in r else 0
    agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"]); agg[k][2] += dur
for k, (n, c, d) in sorted(agg.items(), key=lambda kv: -kv[1][2])[:16]:
    print(f"{n:5d} x  {d / n / 1e3:8.1f} us   clock {c / 8 / d if d else 0:5.2f} GHz   {k}")
PY
