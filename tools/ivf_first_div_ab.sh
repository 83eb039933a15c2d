for cfg in "8 1" "16 1" "4 1" "32 1"; do set -- $cfg; AMDREC_IVF_FIRST_DIV=$1 AMDREC_IVF_MIXED=$2 python bench.py --ads 10000000 --index ivf --nlist 4096 --nprobe 64 --steps 10 --no-cpu-baseline > gpurun_out/r04_firstdiv_$1.json 2> gpurun_out/r04_firstdiv_$1.err || exit 1; python - <<PY
import json
d=json.loads(open('gpurun_out/r04_firstdiv_$1.json').read().strip().split('\n')[-1])
print('first_div=$1', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items() if k.startswith('ivf')}, d['search'].get('recall_at_500_vs_flat'))
PY
done
