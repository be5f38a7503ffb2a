"""Average PMC counter values per launch of the kernels whose name contains argv[2] (rocprofv3 counter_collection.csv under gpurun_out/argv[1])."""
import collections, csv, glob, sys
rows = []
for f in glob.glob(f'gpurun_out/{sys.argv[1]}/**/*counter_collection.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
agg, cnt = collections.defaultdict(float), collections.Counter()
for r in rows:
    if sys.argv[2] in r['Kernel_Name']:
        agg[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
print(sys.argv[1], ' '.join(f"{k}={agg[k] / cnt[k]:.4e}" for k in sorted(agg)))
