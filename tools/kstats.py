"""Print a rocprofv3 kernel_stats.csv as a compact table (development helper)."""
import csv, glob, sys
path = sys.argv[1]
f = glob.glob(path + '/*/*kernel_stats.csv')[0] if not path.endswith('.csv') else path
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel time %.2f ms' % (tot / 1e6))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    name = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print('%-58s calls=%6s total_ms=%9.3f avg_us=%10.2f pct=%5.1f' % (name[:58], r['Calls'], float(r['TotalDurationNs']) / 1e6,
          float(r['AverageNs']) / 1e3, float(r['Percentage'])))
