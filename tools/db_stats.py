#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats wrote a rocpd SQLite file: dump its per-kernel summary (the `top_kernels` view) as CSV.
    python tools/db_stats.py gpurun_out/prof_x/x_results.db > profiles/....csv"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-fused   (durations in us)")
print("Name,Calls,TotalDurationUs,AverageUs,Percentage")
for n, c, t, a, p in rows:
    print(f"\"{n}\",{c},{t:.1f},{a:.2f},{p:.3f}")
