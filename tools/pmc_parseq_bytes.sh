#!/bin/bash
# GPU box: HBM bytes per launch of the recogniser's kernels at 1280 crops (FETCH_SIZE / WRITE_SIZE passes, each on its own; units and the
# gfx950 FETCH_SIZE correction as in tools/pmc_to_json.py).   tools/pmc_parseq_bytes.sh [key=value ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export TTR_PREC=f16x4
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pq_$c; rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pq_$c -o p -- python3 $R/tools/prof_parseq.py ${CROPS:-1280} 2 "$@" > /tmp/pq_$c.log 2>&1
done
python3 - <<'PY'
import collections, csv, glob
def fold(c):
    f = sorted(glob.glob(f"/tmp/pq_{c}/**/*counter_collection.csv", recursive=True))[-1]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            agg[r["Kernel_Name"][:70]][0] += 1; agg[r["Kernel_Name"][:70]][1] += float(r["Counter_Value"])
    return agg
f, w = fold("FETCH_SIZE"), fold("WRITE_SIZE")
rows = []
for k in set(f) | set(w):
    n = max(f[k][0], w[k][0], 1)
    rows.append((2 * 1024 * f[k][1] / n + 1024 * w[k][1] / n, n, 2 * 1024 * f[k][1] / n, 1024 * w[k][1] / n, k))
for tot, n, fb, wb, k in sorted(rows, reverse=True)[:14]:
    print(f"{n:5d} x  fetch {fb / 1e6:9.1f} MB  write {wb / 1e6:9.1f} MB   {k}")
PY
