"""Turns a rocprofv3 results .db (rocpd sqlite, `rocprofv3 --kernel-trace --stats`) into a small text summary
for profiles/.  Usage: python profiles/summarize_db.py <results.db> <out.md> "<command line that was profiled>" """
import sqlite3
import sys


def main(db, out, cmd):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary\n\ncommand: `%s`\n\n" % cmd)
        f.write("durations in microseconds (rocpd `top_kernels` view: total_duration/average in ns / 1000)\n\n")
        f.write("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|\n")
        for name, calls, tot, avg, pct in rows:
            short = name if len(name) < 90 else name[:87] + "..."
            f.write("| `%s` | %d | %.1f | %.1f | %.2f |\n" % (short, calls, tot, avg, pct))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
