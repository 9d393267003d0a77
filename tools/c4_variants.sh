for mode in nohint hint; do
  if [ $mode = nohint ]; then export DSS_NO_IGR_HINT=1; else unset DSS_NO_IGR_HINT; fi
  timeout -k 10 300 python bench.py --config 4 --no-cpu --batch 256 --steps 60 > gpurun_out/c4_$mode.json 2> gpurun_out/c4_$mode.err
  python -c "
import json
r=json.load(open('gpurun_out/c4_$mode.json'))
print('$mode', r['value'], r['config']['attempts'], r['config']['forward_s'], r['roofline']['avg_launch_ms'], r['roofline_second_kernel']['avg_launch_ms'])"
done
