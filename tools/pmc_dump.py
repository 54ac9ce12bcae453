import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select k.name, p.name, avg(e.value), count(*) from pmc_events e join pmc_info p on e.pmc_id = p.id "
                  "join kernels k on e.event_id = k.event_id where k.name like ? group by k.name, p.name", (sys.argv[2],)).fetchall() \
    if False else None
cur = db.cursor()
print([r[1] for r in cur.execute("pragma table_info('pmc_events')")])
print([r[1] for r in cur.execute("pragma table_info('counters_collection')")])
for r in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like ? group by kernel_name, counter_name", (sys.argv[2],)):
    print(r)
