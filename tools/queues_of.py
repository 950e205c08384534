"""tools/queues_of.py TRACE.csv -- which hardware queue each HIP stream of a rocprofv3 kernel trace ran on, and the DP launches' start / end
(ms): two streams on one queue run their kernels one after the other whatever the host asked for.  Bring-up tool."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pairs = {}
for r in rows:
    pairs.setdefault((int(r["Stream_Id"]), int(r["Queue_Id"])), 0)
    pairs[(int(r["Stream_Id"]), int(r["Queue_Id"]))] += 1
print("stream -> queue (kernels):", ", ".join("%d -> %d (%d)" % (s, q, n) for (s, q), n in sorted(pairs.items())))
dp = [r for r in rows if "k_dp_" in r["Kernel_Name"] or "k_em_" in r["Kernel_Name"]]
if dp:
    t0 = min(int(r["Start_Timestamp"]) for r in dp)
    for r in dp[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
        name = r["Kernel_Name"].split("::")[-1].split("(")[0]
        print("%-28s stream %s queue %s  %8.2f -> %8.2f ms" % (name, r["Stream_Id"], r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6))
