# A/B of the grouped scan's query-tile size per phase at configs[4] on one GPU ("first,second"; default 32,32 there)
for qt in "32,32" "32,64" "64,64" "64,32"; do AMDREC_IVF_QTILE=$qt python bench.py --ads 10000000 --index ivf --nlist 4096 --nprobe 64 --steps 10 --no-cpu-baseline --no-search-sweep --no-strict-fp32 --no-latency-sweep > gpurun_out/r04_qtile.json 2> gpurun_out/r04_qtile.err || exit 1; python - "$qt" <<'PY'
import json, sys
d=json.loads(open('gpurun_out/r04_qtile.json').read().strip().split('\n')[-1])
print('qtile', sys.argv[1], d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items() if k.startswith('ivf')})
PY
done
