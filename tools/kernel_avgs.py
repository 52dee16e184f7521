"""average duration per kernel from a rocprofv3 --stats kernel_stats csv: python tools/kernel_avgs.py <stats_kernel_stats.csv>"""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{float(r['AverageNs']) / 1e3:9.2f} us  x{int(r['Calls']):6d}  {r['Percentage']:>6s} %  {r['Name'][:90]}")
