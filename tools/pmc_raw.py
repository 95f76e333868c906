"""Per-kernel means of every counter in a rocprofv3 counter_collection.csv."""
import collections, csv, re, sys
per = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("sara_hip::", "").replace("void ", "")
    per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[name].add((r["Dispatch_Id"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for name, c in sorted(per.items(), key=lambda kv: -sum(d for _, d in dur[kv[0]])):
    n = len(dur[name]); t = sum(d for _, d in dur[name]) / n / 1e3
    print(f"{name[:44]:44s} n={n:3d} avg {t:9.1f} us  " + "  ".join(f"{k.replace('SQ_','')}={sum(v)/len(v)/1e6:.2f}M" for k, v in sorted(c.items())))
