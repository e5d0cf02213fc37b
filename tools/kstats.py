"""Prints the spmm rows of a rocprofv3 kernel_stats.csv found under the given directory."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
out = []
for r in csv.DictReader(open(f)):
    if 'spmm' in r['Name']:
        out.append('%s %.1f' % (r['Name'].split('::')[-1][:28], float(r['AverageNs']) / 1e3))
print(' | '.join(out))
