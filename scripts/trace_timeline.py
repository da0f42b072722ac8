"""Timeline of the last graph-replayed step in a rocprofv3 kernel trace: per-queue busy time, gaps
between consecutive kernels, and the kernels with the largest in-step duration.
usage: python scripts/trace_timeline.py gpurun_out/trace1/*/*_kernel_trace.csv"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
# a step ends with pack_weights_multi; take the span between the last two of them
marks = [i for i, r in enumerate(rows) if 'pack_weights_multi' in r['Kernel_Name']]
a, b = marks[-2] + 1, marks[-1] + 1
step = rows[a:b]
t0, t1 = step[0]['s'], step[-1]['e']
print('kernels in step: %d   span %.1f us' % (len(step), (t1 - t0) / 1e3))
byq = collections.defaultdict(list)
for r in step:
    byq[r['Queue_Id']].append(r)
for q, ks in byq.items():
    busy = sum(k['e'] - k['s'] for k in ks)
    gaps = [ks[i + 1]['s'] - ks[i]['e'] for i in range(len(ks) - 1)]
    print('queue %s: %d kernels, busy %.1f us, sum of gaps %.1f us (median gap %.2f us)' % (
        q, len(ks), busy / 1e3, sum(g for g in gaps if g > 0) / 1e3, sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0))
# union of busy intervals over all queues
iv = sorted((r['s'], r['e']) for r in step)
cov, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        cov += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
cov += cur_e - cur_s
print('some kernel running: %.1f us; nothing running: %.1f us' % (cov / 1e3, (t1 - t0 - cov) / 1e3))
def short(n):
    m = re.search(r'stmi::?(\w+)|stmi\d+(\w+?)I', n)
    return n[:70]
if len(sys.argv) > 2:
    for r in step:
        print('%8.1f %7.1f q%s %s' % ((r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, r['Queue_Id'], r['Kernel_Name'][:90]))
