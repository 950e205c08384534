"""tools/c3_shares.py [out.json] -- the shares of the strong-scaling files -> file job (BASELINE.json configs[3]) on ONE GPU: bench.py
--workload c3 --reads n for the share of one rank at N = 8 / 4 / 2 / 1, with the box's host threads and with 2 (what a rank gets when
eight share a 16-core grant).  Writes profiles/r04_c3_shares_one_gpu.json's form.  Run on the GPU box."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for reads in (6250, 12500, 25000, 50000):
    row = {"reads": reads}
    for threads in (None, 2):
        env = dict(os.environ)
        if threads:
            env["NPR_HOST_THREADS"] = str(threads)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c3", "--reads", str(reads), "--steps", "7", "--warmup", "1"],
                           env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(p.stderr[-800:], file=sys.stderr)
            continue
        d = json.loads(line[-1])
        if threads is None:
            row.update(ms_per_step=d["ms_per_step"], reads_per_s=d["reads_per_s"], rank0_phase_seconds=d["rank0_phase_seconds"])
        else:
            row["ms_per_step_2_host_threads"] = d["ms_per_step"]
    rows.append(row)
    print(reads, round(row.get("ms_per_step", 0), 1), round(row.get("ms_per_step_2_host_threads", 0), 1), flush=True)
out = {"what": "bench.py --workload c3 --reads n --steps 7 --warmup 1 on ONE GPU: the share of one rank of the strong-scaling job (BASELINE.json "
               "configs[3]) at N = 8 / 4 / 2 / 1, files -> file, with the box's host threads and with NPR_HOST_THREADS=2", "rows": rows}
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
