#!/usr/bin/env python3
"""tools/copy_timeline.py <rocprofv3 .db>: what the copy engines did -- per direction: copies, bytes, busy time, GB/s while busy, share of the traced span; and the kernels' busy time beside it."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
mc = [t for t in tabs if "memory_copy" in t.lower() or "memory_copies" in t.lower()]
print("tables:", [t for t in tabs if "copy" in t.lower() or "kernel" in t.lower()][:12])
for t in mc[:3]:
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % t)]
    print(t, cols)
