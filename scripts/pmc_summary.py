"""Per-kernel HBM-side bytes per launch from the two counter passes of scripts/make_pmc.sh.

FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts wide (16 B per lane) reads at half
their size, so it is doubled (the kernels here read 16 B per lane throughout)."""
import csv, glob, json, os, sys, collections, re

def step_rows(path):
    """Rows of the last two training steps only (a step ends with the multi-tensor weight repack): the run
    also holds the one-off style-image pass, whose launches are many times larger than a step's."""
    rows = list(csv.DictReader(open(path)))
    marks = sorted({int(r['Dispatch_Id']) for r in rows if 'pack_weights_multi' in r['Kernel_Name']})
    if len(marks) < 3:
        return rows
    lo, hi = marks[-3], marks[-1]
    return [r for r in rows if lo < int(r['Dispatch_Id']) <= hi]


def load(d):
    f = max(glob.glob(d + '/*/*counter_collection.csv'), key=os.path.getmtime)   # the newest pass (merged copies keep old ones)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in step_rows(f):
        agg[r['Kernel_Name']][0] += 1
        agg[r['Kernel_Name']][1] += float(r['Counter_Value'])
    return agg

def demangled(name):
    t = lambda c: '__bf16' if c == 'DF16b' else 'float'
    m = re.match(r'_ZN4stmi15conv_rgb_kernelI(DF16b|f)Li(\d+)ELi(\d+)E(DF16b|f)EEv', name)
    if m:
        return ('void stmi::conv_rgb_kernel<%s, %s, %s, %s>(%s const*, float, float const*, %s const*, stmi::ConvGemmParams)'
                % (t(m.group(1)), m.group(2), m.group(3), t(m.group(4)), t(m.group(4)), t(m.group(1))))
    if name.startswith('void stmi::conv_rgb_kernel<bool _Accum'):  # the profiler's own rendering of <__bf16, 9, 2, __bf16>
        return ('void stmi::conv_rgb_kernel<__bf16, 9, 2, __bf16>(__bf16 const*, float, float const*, __bf16 const*, '
                'stmi::ConvGemmParams)')
    m = re.match(r'_ZN4stmi15conv_to3_kernelI(DF16b|f)Li(\d+)ELi(\d+)ELi(\d+)ELb([01])EEEv', name)
    if m:
        a = t(m.group(1))
        return ('void stmi::conv_to3_kernel<%s, %s, %s, %s, %s>(%s const*, %s const*, %s const*, %s*, int, int, int, stmi::To3Clamp)'
                % (a, m.group(2), m.group(3), m.group(4), 'true' if m.group(5) == '1' else 'false', a, a, a, a))
    m = re.match(r'_ZN4stmi\d+(\w+?)I(DF16b|f)((?:Li\d+E)+)((?:Lb[01]E)*)EEvNS_\d+\w+E', name)
    if not m:
        # the profiler's own demangler renders "__bf16, 1" (DF16bLi1E) as "bool _Accum, int, E"
        return name.replace('<bool _Accum, int, E', '<__bf16, 1')
    args = ', '.join(re.findall(r'Li(\d+)E', m.group(3)))
    for b in re.findall(r'Lb([01])E', m.group(4)):
        args += ', true' if b == '1' else ', false'
    return 'void stmi::%s<%s, %s>(stmi::%s)' % (m.group(1), '__bf16' if m.group(2) == 'DF16b' else 'float', args,
                                                 'WgradParams' if 'wgrad' in m.group(1) else 'ConvGemmParams')

def load_mfma(d):
    """Per kernel: launches, MFMA busy cycles, MFMA ops (x512 = FLOP), chip-active cycles (summed over the 8 XCDs)."""
    files = glob.glob(d + '/*/*counter_collection.csv')
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    if not files:
        return agg
    seen = set()
    for r in step_rows(max(files, key=os.path.getmtime)):
        k = r['Kernel_Name']
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id'])
            agg[k]['launches'] += 1
    return agg

fe, wr, mf = load('gpurun_out/pmc_fetch'), load('gpurun_out/pmc_write'), load_mfma('gpurun_out/pmc_mfma')
out = {}
for k, (n, v) in fe.items():
    if 'stmi' not in k:
        continue
    w = wr.get(k, [1, 0.0])
    out[demangled(k)] = {'launches': n, 'fetch_bytes_per_launch': round(2 * v / n * 1024),
                         'write_bytes_per_launch': round(w[1] / max(w[0], 1) * 1024),
                         'bytes_per_launch': round((2 * v / n + w[1] / max(w[0], 1)) * 1024)}
    m = mf.get(k)
    if m and m['launches'] and m['GRBM_GUI_ACTIVE']:
        # 1024 SIMDs; GRBM_GUI_ACTIVE is the sum of the 8 XCDs' active cycles
        out[demangled(k)].update({
            'mfma_flop_per_launch': round(m['SQ_INSTS_VALU_MFMA_MOPS_BF16'] * 512 / m['launches']),
            'mfma_busy_frac': round(m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8 * 1024), 4)})
# Groups as bench.py's roofline names its bandwidth-bound calls: one instance-norm forward = its statistics, finalize
# and apply launches; bytes per call = the traffic of all of them / the number of apply launches.
def group(pattern, calls_of):
    ks = [k for k in out if re.search(pattern, k)]
    calls = sum(out[k]['launches'] for k in ks if re.search(calls_of, k))
    if not ks or not calls:
        return None
    tot = lambda f: sum(out[k][f] * out[k]['launches'] for k in ks)
    return {'launches': calls, 'fetch_bytes_per_launch': round(tot('fetch_bytes_per_launch') / calls),
            'write_bytes_per_launch': round(tot('write_bytes_per_launch') / calls),
            'bytes_per_launch': round(tot('bytes_per_launch') / calls), 'kernels': sorted(ks)}
for gname, pat, calls_of in (('instance_norm_fwd', r'in_stats|in_apply', r'in_apply'),
                             ('instance_norm_bwd', r'in_bwd_reduce|in_bwd_apply', r'in_bwd_apply')):
    g_ = group(pat, calls_of)
    if g_:
        out[gname] = g_
json.dump({'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / (SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 '
                     'GRBM_GUI_ACTIVE), separate passes, bench.py --steps 3 --warmup 1 with STMI_SERIAL=gwc, launches of the last two steps only (no concurrent '
                     'kernels: the counters are chip-wide); FETCH_SIZE doubled (gfx950 counts 128-byte requests as 64); '
                     'mfma_busy_frac = MFMA busy cycles / (active cycles x 1024 SIMDs)', 'kernels': out},
          open(sys.argv[1], 'w'), indent=1, sort_keys=True)
print(len(out), 'kernels ->', sys.argv[1])
