"""Summarise a rocprofv3 --kernel-trace --stats result (.db, rocpd format) into a text table:
per kernel: calls, total / mean / min / max duration.  Used to produce the files under profiles/."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = [f"# rocprofv3 --kernel-trace --stats summary of {path}", f"# columns of `kernels` view: {cols}",
             f"{'kernel':70s} {'calls':>8s} {'total_us':>12s} {'mean_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}"]
    for name, n, tot, avg, mn, mx in rows:
        short = name if len(name) <= 70 else name[:67] + "..."
        lines.append(f"{short:70s} {n:8d} {tot/1e3:12.1f} {avg/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*tot/total:6.1f}")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
