#!/usr/bin/env python3
"""Per-DISPATCH counter values out of the rocpd databases of several rocprofv3 --pmc passes over the same command (tools/r05_pmc.sh micro): the n-th dispatch of a
kernel is the same call in every pass, so the passes are joined by dispatch order.  Prints one line per dispatch: index, kernel, then every counter."""
import glob
import os
import sqlite3
import sys

rows = {}          # dispatch order -> {counter: value}
names = {}
for d in sys.argv[1:]:
    for db in sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)):
        con = sqlite3.connect(db)
        cur = con.execute("select * from counters_collection limit 1")
        cols = [c[0] for c in cur.description]
        kcol = next((c for c in cols if "kernel" in c and "name" in c), None) or "name"
        ccol = next(c for c in cols if "counter_name" in c)
        vcol = next(c for c in cols if c in ("value", "counter_value"))
        dcol = next((c for c in cols if c in ("dispatch_id", "dispatch_index")), None) or next(c for c in cols if "dispatch" in c)
        disp = {}
        for k, c, v, di in con.execute("select %s, %s, %s, %s from counters_collection" % (kcol, ccol, vcol, dcol)):
            if "hevcdl" not in str(k):
                continue
            disp.setdefault(di, {}).setdefault(c, 0.0)
            disp[di][c] += v
            names[di] = k
        for order, di in enumerate(sorted(disp)):
            rows.setdefault(order, {}).update(disp[di])
            rows[order]["_kernel"] = names[di]
        con.close()
ctrs = sorted({c for r in rows.values() for c in r if not c.startswith("_")})
print("dispatch | kernel | " + " | ".join(ctrs))
for o in sorted(rows):
    print("%3d | %s | " % (o, rows[o]["_kernel"][:40]) + " | ".join("%.0f" % rows[o].get(c, float("nan")) for c in ctrs))
