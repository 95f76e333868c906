"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate
passes because both need TCC slots) into profiles/pyramid_traffic.json, the
file bench.py reads for roofline.traffic.

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d D -o f -- python bench.py --steps K --warmup W --cpu-frames 0
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d D -o w -- python bench.py --steps K --warmup W --cpu-frames 0
  python tools/pmc_traffic.py D <K+W> profiles/pyramid_traffic.json

Units and gfx950 corrections (/opt/skills/guides/MI355X_MICROARCH.md, section
HBM): both counters are in KiB; FETCH_SIZE reports exactly half of the bytes
of a wide (16 B/lane) coalesced streaming read, so it is doubled; WRITE_SIZE
is used as is - it matches the algorithmic store bytes of these kernels to
4 digits (4 B x P x 64 frames per plane set), which calibrates it here.
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(float)
    calls = collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sara_hip::", "")
        agg[k] += float(r["Counter_Value"])
        calls[k] += 1
    return agg, calls


def main():
    d, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    f, fc = per_kernel(d + "/f_counter_collection.csv", "FETCH_SIZE")
    w, wc = per_kernel(d + "/w_counter_collection.csv", "WRITE_SIZE")
    kernels = {}
    rd = wr = 0.0
    for k in sorted(set(f) | set(w)):
        r = 2.0 * f.get(k, 0.0) * 1024 / steps
        s = w.get(k, 0.0) * 1024 / steps
        kernels[k] = {"launches_per_step": fc.get(k, wc.get(k, 0)) / steps,
                      "hbm_read_bytes_per_step": r, "hbm_write_bytes_per_step": s}
        if "gaussian_blur" in k or k.startswith("scale_kernel"):
            rd += r
            wr += s
    json.dump({
        "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                  "FETCH_SIZE x2 (gfx950 wide-load correction), KiB -> bytes",
        "workload": "bench.py defaults (64 x 1920x1080 frames, 4 octaves)",
        "stage": "Gaussian pyramid (all gaussian_blur* + scale kernels)",
        "hbm_read_bytes_per_step": rd, "hbm_write_bytes_per_step": wr,
        "hbm_bytes_per_step": rd + wr,
        "whole_step_read_bytes": sum(v["hbm_read_bytes_per_step"] for v in kernels.values()),
        "whole_step_write_bytes": sum(v["hbm_write_bytes_per_step"] for v in kernels.values()),
        "kernels": kernels}, open(out, "w"),
        indent=1)
    print("pyramid stage: read %.3f GB + write %.3f GB per step" % (rd / 1e9, wr / 1e9))
    print("whole step: read %.3f GB + write %.3f GB" %
          (sum(v["hbm_read_bytes_per_step"] for v in kernels.values()) / 1e9,
           sum(v["hbm_write_bytes_per_step"] for v in kernels.values()) / 1e9))
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_read_bytes_per_step"]
                       - kv[1]["hbm_write_bytes_per_step"])[:14]:
        print("  %-58s %5.1f launches  read %6.3f GB  write %6.3f GB" %
              (k[:58], v["launches_per_step"], v["hbm_read_bytes_per_step"] / 1e9,
               v["hbm_write_bytes_per_step"] / 1e9))


if __name__ == "__main__":
    main()
