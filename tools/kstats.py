import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
tot = sum(sum(v) for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print("%-72s n=%5d avg %9.1f us  total %9.1f us  %5.1f %%" % (k[:72], len(v), sum(v) / len(v), sum(v), 100 * sum(v) / tot))
print("total", tot)
