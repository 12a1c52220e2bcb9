"""Print mean PMC counter value per kernel from rocprofv3 --pmc result databases."""
import sqlite3, sys
for path in sys.argv[1:]:
    db = sqlite3.connect(path); cur = db.cursor()
    rows = cur.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection "
                       "group by kernel_name, counter_name").fetchall()
    for name, c, v, n, dur in rows:
        if "rocclr" in name: continue
        print("%-28s %-11s mean %.1f (KiB units) over %d dispatches, mean duration %.1f us   [%s]" % (name.split("(")[0][-28:], c, v, n, dur / 1e3, path))
