"""Print per-kernel average durations from a rocprofv3 kernel_stats/trace csv dir."""
import csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)):
    for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 8]:
        print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f} min_us={float(r['MinNs'])/1e3:8.1f} {r['Percentage']}%")
