"""Average PMC counters per kernel from rocprofv3 --pmc CSV output: python tools/pmc_sum.py <dir> <kernel substring>."""
import csv, glob, collections, sys
f = glob.glob(f'{sys.argv[1]}/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(float); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    if sys.argv[2] not in r['Kernel_Name']: continue
    agg[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
for k in sorted(agg): print(f"{k:32s} {agg[k] / cnt[k]:.4e}  (n={cnt[k]})")
