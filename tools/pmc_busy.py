"""Per kernel symbol: SQ_BUSY_CYCLES (summed over the shader engines) per microsecond of launch duration, from one rocprofv3
--kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES run (gpurun_out/argv[1]).  A kernel whose compute units idle for part of its
duration (tail, imbalance across XCDs) shows a lower ratio than one that keeps them all busy to the end."""
import collections, csv, glob, sys
d = sys.argv[1]
cc = list(csv.DictReader(open(glob.glob(f'gpurun_out/{d}/**/*counter_collection.csv', recursive=True)[0])))
kt = {r['Dispatch_Id']: r for r in csv.DictReader(open(glob.glob(f'gpurun_out/{d}/**/*kernel_trace.csv', recursive=True)[0]))}
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in cc:
    k = kt.get(r['Dispatch_Id'])
    if k is None:
        continue
    name = r['Kernel_Name'][:70]
    agg[name][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_BUSY_CYCLES':
        agg[name]['us'] += (int(k['End_Timestamp']) - int(k['Start_Timestamp'])) / 1e3
        agg[name]['n'] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1]['us'])
for name, a in rows[:24]:
    print(f"{a['us'] / a['n']:9.1f} us x{int(a['n']):4d}  busy/us {a['SQ_BUSY_CYCLES'] / a['us']:9.0f}  wave/us {a['SQ_WAVE_CYCLES'] / a['us']:9.0f}  {name}")
