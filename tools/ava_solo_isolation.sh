show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['allvsall']['align_roofline']['solo']; print('$1: all-vs-all k_sw3 solo %.3f ms frac %.3f' % (r['kernel_ms'], r['frac']))"; }
python bench.py --no-cpu-baseline --type2-steps 0 --fullrange-steps 0 --no-kmer --single-targets 0 2>/dev/null | show "main leg + all-vs-all"
python bench.py --no-cpu-baseline --steps 2 --warmup 1 --type2-steps 0 --fullrange-steps 0 --single-targets 0 2>/dev/null | show "short main + k-mer leg + all-vs-all"
python bench.py --no-cpu-baseline --steps 2 --warmup 1 --type2-steps 0 --fullrange-steps 0 --no-kmer --single-targets 0 2>/dev/null | show "short main + all-vs-all"
