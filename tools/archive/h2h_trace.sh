# Timeline of kernels and copies of the float32 host -> host step (no torch)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/prof
mkdir -p $D
cd $R
env "$@" rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $D -o h2h -- python tools/h2h_notorch.py --kinds f32 > $D/h2h.log 2>&1
grep "per .*step" $D/h2h.log
python - <<'PY'
import csv, os
D = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/prof"
ev = []
for r in csv.DictReader(open(D + "/h2h_memory_copy_trace.csv")):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"], int(r.get("Bytes", 0) or 0)))
ks = []
for r in csv.DictReader(open(D + "/h2h_kernel_trace.csv")):
    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
ks.sort()
# group kernels into steps: a descriptor kernel ends a step
steps, cur = [], []
for k in ks:
    cur.append(k)
    if "descriptor_kernel" in k[2]:
        steps.append((cur[0][0], cur[-1][1]))
        cur = []
t0 = steps[5][0]
rows = [(a, b, "KERNELS", 0) for a, b in steps[5:10]]
rows += [e for e in ev if e[0] >= t0 - 20e6 and e[0] <= steps[9][1] + 5e6 and e[3] > 1 << 20]
rows.sort()
for a, b, d, n in rows:
    print("%9.3f .. %9.3f ms  %-28s %7.1f MB" % ((a - t0) / 1e6, (b - t0) / 1e6, d, n / 1e6))
PY
