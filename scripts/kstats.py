"""Prints the first rows of a rocprofv3 kernel_stats.csv.  usage: kstats.py <csv> [rows=20]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    name = r["Name"].split("(")[0][-44:]
    print("%-46s calls=%6s avg_us=%9.1f pct=%s" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
