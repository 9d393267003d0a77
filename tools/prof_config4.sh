cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/k4 && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/k4 -- python3 bench.py --config 4 --steps 45 --warmup 1 --no-cpu --batch 512 > gpurun_out/k4.log 2>&1
python3 - <<'PY'
import csv, glob
f=glob.glob("gpurun_out/k4/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:10]:
    print("%-58s calls %6s total %8.1f ms avg %8.1f us  %5.1f%%"%(r['Name'][:58], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
tail -c 400 gpurun_out/k4.log | head -c 300
rm -rf gpurun_out/k4
