"""Averages rocprofv3 --pmc counters per spmm kernel: python tools/pmcsum.py DIR"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'spmm' in k:
        acc[k.split('::')[-1][:22]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print('   ', k, {c: round(sum(x) / len(x)) for c, x in v.items()})
