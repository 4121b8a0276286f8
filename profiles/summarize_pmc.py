"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for path in sorted(glob.glob(root + '/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(path)):
        name = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    if not any(s in k for s in ('k_georef', 'k_bin_frame', 'k_bbox', 'k_hist')):
        continue
    print(k)
    for c, v in sorted(acc[k].items()):
        print('   %-28s mean %.4g  (n=%d)' % (c, sum(v) / len(v), len(v)))
