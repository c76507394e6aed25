"""rocprofv3 --kernel-trace --stats summary (the `top_kernels` view of its SQLite output) as CSV: name, calls, total us, average us, % (the view holds microseconds)."""
import csv
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
for r in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    w.writerow([r[0], r[1], int(r[2]), round(r[3], 1), round(r[4], 3)])
