"""Per-kernel stats (calls / total / avg / min / max ns) out of a rocprofv3 rocpd sqlite database,
written in the same column order as rocprofv3's kernel_stats.csv."""
import csv
import sqlite3
import sys


def main(db, out):
  c = sqlite3.connect(db)
  cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
  name = 'name' if 'name' in cols else 'kernel_name'
  rows = c.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                   "from kernels group by %s order by 3 desc" % (name, name)).fetchall()
  tot = float(sum(r[2] for r in rows))
  with open(out, 'w', newline='') as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
    for r in rows:
      w.writerow([r[0], r[1], r[2], round(r[3], 2), round(100.0 * r[2] / tot, 2), r[4], r[5]])


if __name__ == '__main__':
  main(sys.argv[1], sys.argv[2])
