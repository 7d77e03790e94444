"""The tables that accompany bench.py's headline (config.mid_occupancy, config.shapes, config.few_chains, cpu_baseline) in readable form.
usage: python tools/bench_tables.py BENCH_LINE.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
r = d['roofline']
print('headline: %.1f M evals/s, roofline.frac %.3f (frac_executed %.3f), %.2f ms per step'
      % (d['value'] / 1e6, r['frac'], r.get('frac_executed', float('nan')), d['ms_per_step']))
c = d['config']
if c.get('mid_occupancy'):
    print('mid occupancy (sampler kind 1: one chain per 512-thread workgroup, 3: one chain per wave, 0: sixteen chains per workgroup):')
    for m in c['mid_occupancy']:
        print('  %5d units: %6.1f M evals/s (kind %d)' % (m['units'], m['evals_per_s'] / 1e6, m['sampler_kind']))
if c.get('shapes'):
    print('shapes (frequencies x basis functions; evaluator code; evals/s and dense-formulation roofline fraction at 4096 and 2048 units):')
    for s in c['shapes']:
        if 'error' in s:
            print('  %3d x %3d  %s' % (s['nf'], s['K'], s['error'])); continue
        a, b = s['units_4096'], s['units_2048']
        print('  %3d x %3d  evaluator %d  4096 units: %6.1f M (kind %d, frac %.3f)   2048 units: %6.1f M (kind %d, frac %.3f)'
              % (s['nf'], s['K'], s['evaluator'], a['evals_per_s'] / 1e6, a['sampler_kind'], a['frac'], b['evals_per_s'] / 1e6,
                 b['sampler_kind'], b['frac']))
if c.get('few_chains'):
    print('few chains:', json.dumps(c['few_chains']))
if c.get('strong_scaling'):
    print('strong scaling:', json.dumps(c['strong_scaling']))
cb = d.get('cpu_baseline')
if cb:
    print('cpu baseline:', json.dumps({k: cb[k] for k in ('value', 'cores', 'kind', 'single_core') if k in cb}))
