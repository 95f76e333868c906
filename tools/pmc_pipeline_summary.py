"""Summarise tools/pmc_pipeline.sh: per kernel, VALU instructions per launch,
VALU pipe utilisation (2 cycles per wave64 instruction, 1024 SIMDs) and the
wave-cycle buckets."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
per = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("sara_hip::", "").replace("void ", "")
    per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        dur[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(f"{'kernel':46s} {'n':>4s} {'sum ms':>8s} {'VALU M':>9s} {'pipe%':>6s} {'act%':>5s} {'winst%':>6s} {'wany%':>6s} {'salu/valu':>9s} {'lds/valu':>8s}")
for name, c in sorted(per.items(), key=lambda kv: -sum(dur[kv[0]])):
    n = len(dur[name])
    t = sum(dur[name]) * 1e-9
    valu = sum(c["SQ_INSTS_VALU"])
    wc = sum(c["SQ_WAVE_CYCLES"]) or 1
    busy = sum(c["SQ_BUSY_CYCLES"])
    # SQ_BUSY_CYCLES is summed over the 32 SE-level SQs: cycles = busy / 32
    cyc = busy / 32.0
    pipe = valu * 2.0 / (cyc * 1024) * 100 if cyc else 0
    print(f"{name[:46]:46s} {n:4d} {t*1e3:8.3f} {valu/1e6:9.1f} {pipe:6.1f} "
          f"{sum(c['SQ_ACTIVE_INST_ANY'])/wc*100:5.1f} {sum(c['SQ_WAIT_INST_ANY'])/wc*100:6.1f} "
          f"{sum(c['SQ_WAIT_ANY'])/wc*100:6.1f} {sum(c['SQ_INSTS_SALU'])/max(valu,1):9.3f} {sum(c['SQ_INSTS_LDS'])/max(valu,1):8.3f}")
