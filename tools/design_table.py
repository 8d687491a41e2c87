#!/usr/bin/env python3
"""Prints the rows of DESIGN.md section 5's table (paste them; it no longer edits DESIGN.md) from profiles/r03_bench_lines.jsonl and
profiles/r03_kernel_stats_*.csv (so that every figure there can be recomputed from a file in profiles/).
usage: python tools/design_table.py [ROUND, default 03]"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = sys.argv[1] if len(sys.argv) > 1 else '04'
rows = [json.loads(l) for l in open(os.path.join(ROOT, 'profiles/r%s_bench_lines.jsonl' % N))]

def avg(cfg, s):
    for r in csv.DictReader(open(os.path.join(ROOT, 'profiles/r%s_kernel_stats_%s_s%d.csv' % (N, cfg, s)))):
        if 'resample_' in r['Name']:
            return float(r['AverageNs']) / 1e3

pct = lambda x: '%.1f' % (100 * x)
num = lambda v: format(round(v, -2), ',.0f').replace(',', ' ')
L = lambda d: d['roofline']['launch_us']
tr = lambda d: '%.2f' % (d['roofline']['traffic'] / d['roofline']['algorithmic_bytes_per_launch'])

def c(wl, s, mode='fast', io='int16'):
    return [d for d in rows if wl in d['config']['workload'] and d['config']['streams_per_gpu'] == s and
            d['config']['mode'] == mode and d['config']['io'] == io and 'configs[4]' not in d['config']['workload']][0]

out = []
d1, d32 = c('configs[1]', 1), c('configs[1]', 32)
d4 = [d for d in rows if 'configs[4]' in d['config']['workload']][0]
out.append('| cfg2, 1 stream × 2^20 frames (BASELINE configs[1], `bench.py` default) | %.2f µs | %.2f µs | %s | %s %% | %s %% | %s %% | %s× |' %
           (L(d1), avg('cfg2', 1), num(d1['value']), pct(d1['roofline']['frac']), pct(d1['roofline']['read_only_frac']), pct(d1['valu']['frac']), tr(d1)))
out.append('| cfg2, 32 streams (configs[4] per-GPU share; `--total-streams 32` gives the same launch: %.1f µs) | %.1f µs | %.1f µs | %s | %s %% | %s %% | %s %% | %s× |' %
           (L(d4), L(d32), avg('cfg2', 32), num(d32['value']), pct(d32['roofline']['frac']), pct(d32['roofline']['read_only_frac']), pct(d32['valu']['frac']), tr(d32)))

def two(label, wl, cfg):
    a, b = c(wl, 1), c(wl, 32)
    out.append('| %s | %.1f / %.1f µs | %.1f / %.1f | %s / %s | %s / %s %% | %s / %s %% | %s / %s %% | %s / %s× |' %
               (label, L(a), L(b), avg(cfg, 1), avg(cfg, 32), num(a['value']), num(b['value']), pct(a['roofline']['frac']), pct(b['roofline']['frac']),
                pct(a['roofline']['read_only_frac']), pct(b['roofline']['read_only_frac']), pct(a['valu']['frac']), pct(b['valu']['frac']), tr(a), tr(b)))

two('cfg3 24k→48k mono q10 (slide64 kernel: fp64 accumulate like the reference; fraction of the 78.6 TF fp64 vector peak), 1 / 32 streams', 'configs[2]', 'cfg3')
two('cfg4 48k→44.1k 8 ch q5 (period kernel, padded window), 1 / 32 streams', 'configs[3]', 'cfg4')
two('F3 24k→48k mono q5 (slide kernel), 1 / 32 streams', 'SURVEY F3', 'f3')
fa, fb = c('configs[1]', 1, 'fast', 'float'), c('configs[1]', 32, 'fast', 'float')
out.append('| cfg2, float I/O (`--io float`), 1 / 32 streams | %.1f / %.1f µs | %.1f / %.1f | %s / %s | %s / %s %% | %s / %s %% | %s / %s %% | — |' %
           (L(fa), L(fb), avg('cfg2float', 1), avg('cfg2float', 32), num(fa['value']), num(fb['value']), pct(fa['roofline']['frac']), pct(fb['roofline']['frac']),
            pct(fa['roofline']['read_only_frac']), pct(fb['roofline']['read_only_frac']), pct(fa['valu']['frac']), pct(fb['valu']['frac'])))
e = c('configs[1]', 1, 'exact')
out.append('| cfg2 EXACT mode (bit-identical to the reference), 1 stream | %.1f µs | %.1f µs | %s | %s %% | %s %% | — | — |' % (L(e), avg('cfg2exact', 1), num(e['value']), pct(e['roofline']['frac']), pct(e['roofline']['read_only_frac'])))
ea, eb = c('configs[2]', 1, 'exact'), c('configs[2]', 32, 'exact')
out.append('| cfg3 EXACT mode: configs[2] at the reference\'s precision (fp64 sums of fp32 products, `resample.c:409-417`), 1 / 32 streams | %.1f / %.1f µs | %.1f / %.1f | %s / %s | %s / %s %% | %s / %s %% | — | — |' %
           (L(ea), L(eb), avg('cfg3exact', 1), avg('cfg3exact', 32), num(ea['value']), num(eb['value']), pct(ea['roofline']['frac']), pct(eb['roofline']['frac']),
            pct(ea['roofline']['read_only_frac']), pct(eb['roofline']['read_only_frac'])))
for io in ('int16', 'float'):
    try:
        ma, mb = c('custom: 44100->48000 Hz, 1ch', 1, 'fast', io), c('custom: 44100->48000 Hz, 1ch', 32, 'fast', io)
    except IndexError:
        continue
    out.append('| 44.1k→48k mono q7, %s I/O (not a BASELINE config), 1 / 32 streams | %.1f / %.1f µs | — | %s / %s | %s / %s %% | %s / %s %% | %s / %s %% | — |' %
               (io, L(ma), L(mb), num(ma['value']), num(mb['value']), pct(ma['roofline']['frac']), pct(mb['roofline']['frac']),
                pct(ma['roofline']['read_only_frac']), pct(mb['roofline']['read_only_frac']), pct(ma['valu']['frac']), pct(mb['valu']['frac'])))
cp = [c(w, 1)['cpu_baseline']['value'] for w in ('configs[1]', 'configs[2]', 'configs[3]', 'SURVEY F3')]
out.append("| CPU baseline: the reference's own C natively compiled (`oracle/_ref`, not the WASM build), 1 host core, cfg2 / cfg3 / cfg4 / F3 | — | — | %.1f / %.1f / %.1f / %.1f | — | — | — | — |" % tuple(cp))
ee = [c(w, 1)['end_to_end'] for w in ('configs[1]', 'configs[2]', 'configs[3]', 'SURVEY F3')]
out.append('| `end_to_end` of the 1-stream lines: the host-buffer call (pageable buffers, H2D + kernel + D2H + wait), cfg2 / cfg3 / cfg4 / F3 | %.3f / %.3f / %.3f / %.3f ms per chunk | — | %s | — | — | — | — |' %
           (tuple(x['ms_per_chunk'] for x in ee) + (' / '.join(num(x['input_msamples_per_s']) for x in ee),)))
print('\n'.join(out))
