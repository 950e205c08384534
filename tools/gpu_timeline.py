"""tools/gpu_timeline.py TRACE.csv [window_ms] -- what the GPU did during the last `window_ms` of a rocprofv3 kernel trace: time with a DP
kernel running, with only other kernels running, with none; and the busy time per kernel name inside the window.  Bring-up tool for the
files -> file job (bench.py --workload c3), whose phases share the device."""
import csv, re, sys, collections
def kname(r):
    m = re.search(r"(k_\w+|__amd_\w+)", r["Kernel_Name"])
    return m.group(1) if m else r["Kernel_Name"][:30]
rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else None
end = max(int(r["End_Timestamp"]) for r in rows)
t_lo = end - win if win else min(int(r["Start_Timestamp"]) for r in rows)
ev = []
per = collections.Counter()
for r in rows:
    s, e = max(int(r["Start_Timestamp"]), t_lo), int(r["End_Timestamp"])
    if e <= t_lo: continue
    name = kname(r)
    dp = name.startswith("k_dp_") or name.startswith("k_em_")
    ev.append((s, 1, dp)); ev.append((e, -1, dp))
    per[name] += e - s
ev.sort()
n_dp = n_other = 0
last = t_lo
acc = collections.Counter()
for t, d, dp in ev:
    acc["dp" if n_dp else ("other" if n_other else "idle")] += t - last
    last = t
    if dp: n_dp += d
    else: n_other += d
tot = sum(acc.values())
print("window %.1f ms: DP kernel running %.1f, only other kernels %.1f, idle %.1f" % (tot / 1e6, acc["dp"] / 1e6, acc["other"] / 1e6, acc["idle"] / 1e6))
for name, ns in per.most_common(14): print("  %-28s %8.1f ms" % (name, ns / 1e6))
if len(sys.argv) > 3:  # the launches of at least argv[3] ms, in order
    for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e > t_lo and e - s >= float(sys.argv[3]) * 1e6:
            print("  %8.1f -> %8.1f  %-22s stream %s queue %s" % ((s - t_lo) / 1e6, (e - t_lo) / 1e6, kname(r), r["Stream_Id"], r["Queue_Id"]))
