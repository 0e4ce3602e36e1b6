"""Summarise a rocprofv3 --pmc counter_collection.csv for the env kernel (per wave-step averages)."""
import csv, glob, sys, collections
d = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
f = glob.glob(d + '/**/*_counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'dcc_env' in r['Kernel_Name']:
        acc[(r['Kernel_Name'][-40:], r['Counter_Name'])].append((float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp']), int(r['Grid_Size']), r['VGPR_Count'], r['SGPR_Count']))
for (kn, k), v in sorted(acc.items()):
    big = [x for x in v if x[1] > 3e5]
    if not big: continue
    waves = big[0][2] / 64
    avg = sum(x[0] for x in big) / len(big)
    print("%-42s %-24s n=%d  total=%.4g  per_wave_step=%.1f  dur_us=%.0f vgpr=%s sgpr=%s" % (kn, k, len(big), avg, avg / waves / steps, sum(x[1] for x in big) / len(big) / 1e3, big[0][3], big[0][4]))
