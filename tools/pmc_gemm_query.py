import sqlite3, sys
for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    for r in db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%gemm_wp%' group by kernel_name, counter_name"):
        print(r[0][:60], r[1], f"{r[2]/1e6:.2f}M", r[3])
