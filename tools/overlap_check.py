import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
t0 = rows[0][1]
# find a window in the middle
mid = len(rows) // 2
for name, st, en, q, sid in rows[mid:mid + 40]:
    print("%-22s q=%s s=%s  start %9.1f us  dur %6.1f us" % (name.split("(")[0][-22:], q, sid, (st - t0) / 1e3, (en - st) / 1e3))
