"""Kernel / H2D-copy timeline of the last `window_ms` of a rocprofv3 --kernel-trace --memory-copy-trace run (CSV directory):
busy tenths per time bin, overall and per stream.  usage: timeline.py <dir with *_kernel_trace.csv> [window_ms] [bins]"""
import csv, glob, os, sys
from collections import Counter
d = sys.argv[1]; win = float(sys.argv[2]) if len(sys.argv) > 2 else 82.0; bins = int(sys.argv[3]) if len(sys.argv) > 3 else 44
K = list(csv.DictReader(open(glob.glob(os.path.join(d, '*_kernel_trace.csv'))[0])))
M = list(csv.DictReader(open(glob.glob(os.path.join(d, '*_memory_copy_trace.csv'))[0])))
for r in K + M: r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
t_end = max(r['e'] for r in K); w0 = t_end - int(win * 1e6)
Kw = [r for r in K if r['s'] >= w0]
Mw = [r for r in M if r['s'] >= w0 and r['Direction'].endswith('HOST_TO_DEVICE') and r['e'] - r['s'] > 50000]
def union(iv):
    iv = sorted(iv)
    if not iv: return 0
    tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
sq = {}
for r in Kw: sq.setdefault(r['Stream_Id'], Counter())[r['Queue_Id']] += 1
print('stream->queue', {k: dict(v) for k, v in sq.items()}, 'copy streams', dict(Counter(r['Stream_Id'] for r in Mw)))
print('kernel union %.1f ms, sum %.1f, copy union %.1f' % (union([(r['s'], r['e']) for r in Kw]) / 1e6, sum(r['e'] - r['s'] for r in Kw) / 1e6, union([(r['s'], r['e']) for r in Mw]) / 1e6))
t0 = min(r['s'] for r in Kw + Mw); bw = (t_end - t0) / bins
def binfrac(iv):
    out = []
    for b in range(bins):
        bs = t0 + b * bw; be = bs + bw
        out.append(union([(max(s, bs), min(e, be)) for s, e in iv if e > bs and s < be]) / bw)
    return ' '.join('%2d' % round(x * 10) for x in out)
print('bin %.2f ms' % (bw / 1e6))
print('kern   ', binfrac([(r['s'], r['e']) for r in Kw]))
print('copy   ', binfrac([(r['s'], r['e']) for r in Mw]))
for sid in sorted(set(r['Stream_Id'] for r in Kw), key=int): print('k %4s' % sid, binfrac([(r['s'], r['e']) for r in Kw if r['Stream_Id'] == sid]))
for sid in sorted(set(r['Stream_Id'] for r in Mw), key=int): print('c %4s' % sid, binfrac([(r['s'], r['e']) for r in Mw if r['Stream_Id'] == sid]))
