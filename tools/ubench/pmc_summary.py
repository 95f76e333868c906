import csv, collections, sys
for f in sys.argv[1:]:
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "march" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f)
    for k, v in sorted(d.items()):
        print(f"   {k:28s} {sum(v)/len(v):16.0f}")
