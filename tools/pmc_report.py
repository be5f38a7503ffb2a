import csv, glob, collections, sys
d = sys.argv[1]
f = glob.glob(f'gpurun_out/{d}/*/*counter_collection.csv')[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(float)
for r in rows:
    n = r['Kernel_Name']
    if 'gemm_mfma' not in n: continue
    var = 'TN' if '<true, true' in n else ('NN' if '<false, true' in n else 'NT')
    agg[var][r['Counter_Name']] += float(r['Counter_Value']); cnt[(var, r['Counter_Name'])] += 1
for var in sorted(agg):
    print(var, ' '.join(f"{k}={v / cnt[(var, k)]:.3e}" for k, v in sorted(agg[var].items())))
