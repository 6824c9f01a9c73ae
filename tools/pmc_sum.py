"""Sums every counter of a rocprofv3 --pmc csv per kernel: python tools/pmc_sum.py <dir>"""
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0][:40]
        tot[k][row['Counter_Name']] += float(row['Counter_Value']); n[(k, row['Counter_Name'])] += 1
for k, d in tot.items():
    if d.get('SQ_INSTS_VALU', 0) < 1e8: continue
    print(k, {c: '%.4g' % v for c, v in d.items()}, 'dispatches', max(n[(k, c)] for c in d), flush=True)
