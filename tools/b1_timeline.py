"""Developer aid: prints the kernel timeline of one steady-state single-frame
call from a rocprofv3 --kernel-trace CSV (start offset, duration in us, queue)."""
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("sara_hip::", "").replace("void ", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id")))
rows.sort()
calls, cur = [], []
for x in rows:
    cur.append(x)
    if x[2].startswith("descriptor_kernel"):
        calls.append(cur)
        cur = []
c = calls[int(sys.argv[2]) if len(sys.argv) > 2 else 30]
t0 = c[0][0]
for st, en, n, q in c:
    print(f"{(st - t0) / 1e3:8.1f} {(en - st) / 1e3:7.1f}  q{q} {n[:60]}")
print("span", (c[-1][1] - t0) / 1e3)
