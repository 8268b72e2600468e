"""Aggregate rocprofv3 --pmc CSV output per kernel name: python tools/pmc_summary.py <dir>"""
import csv, glob, sys, collections
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-48:]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(k, r['Counter_Name'])] += 1
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:10]:
    print(k)
    for c, val in sorted(v.items()):
        n = cnt[(k, c)]
        print('    %-28s total=%.4g  per-dispatch=%.4g  (n=%d)' % (c, val, val / n, n))
