"""Span summary of a multi-stream kernel trace: with the per-octave streams the
blur kernels of different octaves overlap, so the per-kernel statistics cannot
be summed to the stage time bench.py reports.  This prints, per step, the span
first-blur-start -> last-blur-end (the Gaussian-pyramid stage) and the spans of
the other stages, from rocprofv3's *_kernel_trace.csv.

  python tools/span_summary.py <kernel_trace.csv> [frames-per-step=64] > profiles/rNN_spans.txt
"""
import csv
import re
import sys

path = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rows = []
for r in csv.DictReader(open(path)):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("sara_hip::", "").replace("void ", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
# a step starts at its first blur of octave 0 = the kernel after the last descriptor kernel
steps, cur = [], []
for st, en, name in rows:
    if name.startswith("__amd") and not cur:
        continue
    cur.append((st, en, name))
    if name.startswith("descriptor_kernel"):
        steps.append(cur)
        cur = []
P = sum((1920 >> o) * (1080 >> o) for o in range(4))


def span(step, pred):
    k = [(s, e) for s, e, n in step if pred(n)]
    return (min(s for s, _ in k), max(e for _, e in k)) if k else None


print("# %s: %d steps; spans in us (first start -> last end of the stage's kernels)" % (path.split("/")[-1], len(steps)))
print("# pyramid = gaussian_blur_* (+ scale_kernel); achieved = 48*P*%d frames / span" % frames)
print("%4s %9s %9s %9s %9s %9s %9s %8s" % ("step", "pyramid", "extrema", "gradient", "orient.", "descr.", "step", "pyr TB/s"))
for i, stp in enumerate(steps):
    pyr = span(stp, lambda n: n.startswith("gaussian_blur") or n.startswith("scale_kernel"))
    ext = span(stp, lambda n: n.startswith("extrema") or n.startswith("finish_sites") or n.startswith("bucket") or n.startswith("rank_"))
    grd = span(stp, lambda n: n.startswith("gradient_polar"))
    ori = span(stp, lambda n: n.startswith("orientation") or n.startswith("scan_peaks"))
    dsc = span(stp, lambda n: n.startswith("descriptor"))
    whole = (min(s for s, _, _ in stp), max(e for _, e, _ in stp))
    us = lambda sp: (sp[1] - sp[0]) / 1e3 if sp else 0.0
    tb = 48.0 * P * frames / 1e12 / (us(pyr) * 1e-6) if pyr else 0.0
    print("%4d %9.1f %9.1f %9.1f %9.1f %9.1f %9.1f %8.2f" % (i, us(pyr), us(ext), us(grd), us(ori), us(dsc), us(whole), tb))
