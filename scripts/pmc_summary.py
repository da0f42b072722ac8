"""Per-kernel HBM-side bytes per launch from the two counter passes of scripts/make_pmc.sh.

FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts wide (16 B per lane) reads at half
their size, so it is doubled (the kernels here read 16 B per lane throughout)."""
import csv, glob, json, sys, collections, re

def load(d):
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][0] += 1
        agg[r['Kernel_Name']][1] += float(r['Counter_Value'])
    return agg

def demangled(name):
    m = re.match(r'_ZN4stmi\d+(\w+?)I(DF16b|f)((?:Li\d+E)+)(Lb[01]E)?EEvNS_\d+\w+E', name)
    if not m:
        # the profiler's own demangler renders "__bf16, 1" (DF16bLi1E) as "bool _Accum, int, E"
        return name.replace('<bool _Accum, int, E', '<__bf16, 1')
    args = ', '.join(re.findall(r'Li(\d+)E', m.group(3)))
    if m.group(4):
        args += ', true' if m.group(4) == 'Lb1E' else ', false'
    return 'void stmi::%s<%s, %s>(stmi::%s)' % (m.group(1), '__bf16' if m.group(2) == 'DF16b' else 'float', args,
                                                 'WgradParams' if 'wgrad' in m.group(1) else 'ConvGemmParams')

fe, wr = load('gpurun_out/pmc_fetch'), load('gpurun_out/pmc_write')
out = {}
for k, (n, v) in fe.items():
    if 'stmi' not in k:
        continue
    w = wr.get(k, [1, 0.0])
    out[demangled(k)] = {'launches': n, 'fetch_bytes_per_launch': round(2 * v / n * 1024),
                         'write_bytes_per_launch': round(w[1] / max(w[0], 1) * 1024),
                         'bytes_per_launch': round((2 * v / n + w[1] / max(w[0], 1)) * 1024)}
json.dump({'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --steps 3 --warmup 1; '
                     'FETCH_SIZE doubled (gfx950 counts 128-byte requests as 64)', 'kernels': out},
          open(sys.argv[1], 'w'), indent=1, sort_keys=True)
print(len(out), 'kernels ->', sys.argv[1])
