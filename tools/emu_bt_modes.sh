mkdir -p gpurun_out/r06_prof
for v in ${BT_MODES:-"FSGPU_DEVICE_BACKTRACE=0" "X=auto"}; do echo "== $v"; env $v python bench.py --emulate-rank-share 8 --scaling weak 2>/dev/null > /tmp/emu_out.json; python - <<'PY'
import json
d=json.loads(open('/tmp/emu_out.json').read().strip().splitlines()[-1])
a=d['allvsall']; k=d['kmer_prefilter']
print('main ms/step %.2f | type2 %.2f | kmer qps %.0f | allvsall qps %.0f %s | module %s' % (d['ms_per_step'], d['align_type2']['ms_per_step'], k['queries_per_s'], a['queries_per_s'], {x:round(y,1) for x,y in a['host_wall_ms_per_batch'].items()}, a.get('native_module_end_to_end',{}).get('seconds')))
print(a.get('native_module_end_to_end',{}).get('module_timing'))
print(d.get('emulated_rank_share'))
PY
done
