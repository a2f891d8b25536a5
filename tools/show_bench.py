#!/usr/bin/env python3
"""Print the headline numbers of a bench.py JSON line (file argument)."""
import json, sys
d = json.load(open(sys.argv[1]))
ex = d.pop('extras', {})
print('ms/step %.3f  value %.3e loci/s  scaling=%s n_gpus=%d' % (d['ms_per_step'], d['value'], d['scaling'], d['n_gpus']))
print('roofline', {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d['roofline'].items() if k != 'traffic_source'})
print('kernels_ms', {k: round(v, 4) for k, v in d['kernels_ms'].items() if v})
print('count roofline', round(d['k_locus_count_roofline']['frac'], 3), ' parity', d.get('parity'))
print('cpu', {k: v for k, v in d.get('cpu_baseline', {}).items() if k != 'sample'})
for k, v in ex.get('strong_shard', {}).items():
    if k != 'note':
        print(' shard', k, 'ms %.3f eff %.3f' % (v['ms_per_step'], v['predicted_efficiency']), {a: round(b, 3) for a, b in v['kernels_ms'].items()})
for name in ('compact_outputs', 'config1', 'short_rows', 'config2', 'associatr_scan', 'cpu_baseline_c', 'end_to_end'):
    if name in ex:
        print(name, json.dumps(ex[name])[:900])
