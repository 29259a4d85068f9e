"""How long the GPU runs no kernel at all inside one timed frame (rocprofv3 kernel trace of bench.py --steps 2): the sum of the
gaps between the union of kernel intervals of the last plain pass. usage (on the GPU box): python tools/idle_gaps.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gen = [i for i, r in enumerate(rows) if "k_extend<false" in r["Kernel_Name"] and r["Kernel_Name"].split("(")[0].rstrip().endswith("true>")]
i0 = gen[-1]
# the frame ends with the last film kernel after i0
last = max(i for i, r in enumerate(rows) if i >= i0 and ("k_film" in r["Kernel_Name"] or "k_scatter4" in r["Kernel_Name"] or "k_gather4" in r["Kernel_Name"]))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows[i0:last + 1])
t0, end = iv[0][0], iv[0][1]
gaps = []
for s, e in iv[1:]:
    if s > end:
        gaps.append((s - end, (end - t0) / 1e6))
    end = max(end, e)
total = (end - t0) / 1e6
print("frame %.3f ms, %d kernels, idle %.3f ms in %d gaps; largest: %s" % (total, len(iv), sum(g for g, _ in gaps) / 1e6, len(gaps),
      ", ".join("%.0f us at %.1f ms" % (g / 1e3, at) for g, at in sorted(gaps, reverse=True)[:8])))
