#!/bin/bash
# Dev aid (GPU box): mean duration of igr_advance_kernel and of the network launch by query round (config 4)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/kt && timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python3 bench.py --config 4 --steps 40 --warmup 1 --no-cpu > gpurun_out/kt.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/kt/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
adv = collections.defaultdict(list); net = collections.defaultdict(list)
r = -1
for row in rows:
    n = row["Kernel_Name"]; d = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
    if "overlap_kernel" in n: r = -1
    elif "igr_advance_kernel" in n: r += 1; adv[r].append(d)
    elif "igr_query2_kernel" in n: net[r + 1].append(d)
tot_a = sum(sum(v) for v in adv.values()); tot_n = sum(sum(v) for v in net.values())
print("advance total %.1f ms, network total %.1f ms over %d detections" % (tot_a / 1e3, tot_n / 1e3, len(adv[0])))
for k in sorted(adv):
    a = adv[k]; b = net.get(k, [0])
    print("round %2d  advance mean %8.1f us (share %4.1f%%)   network mean %8.1f us (share %4.1f%%)" % (k, sum(a) / len(a), 100 * sum(a) / tot_a, sum(b) / max(1, len(b)), 100 * sum(b) / max(tot_n, 1)))
PY
rm -rf gpurun_out/kt
