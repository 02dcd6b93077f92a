#!/usr/bin/env python3
"""Summarise the rocpd SQLite databases rocprofv3 writes on this image: kernel-trace -> top_kernels view,
--pmc -> counters_collection rows of the two product kernels.  usage: rocpd_summary.py <dir> [...]"""
import glob
import os
import sqlite3
import sys

for d in sys.argv[1:]:
    for db in sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)):
        con = sqlite3.connect(db)
        names = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
        print("##", db)
        if "top_kernels" in names:
            cur = con.execute("select * from top_kernels")
            print("columns:", ", ".join(c[0] for c in cur.description))
            for r in cur.fetchall():
                if "hevcdl" in str(r[0]):          # the product's kernels only (torch's generator kernels of the synthetic input are left out)
                    print(" | ".join(str(x) for x in r))
        if "counters_collection" in names:
            cur = con.execute("select * from counters_collection limit 1")
            cols = [c[0] for c in cur.description]
            kcol = next((c for c in cols if "kernel" in c and "name" in c), None) or next((c for c in cols if c == "name"), None)
            ccol = next((c for c in cols if "counter_name" in c), None)
            vcol = next((c for c in cols if c in ("value", "counter_value")), None)
            if kcol and ccol and vcol:
                q = "select %s, %s, sum(%s), count(*) from counters_collection group by %s, %s" % (kcol, ccol, vcol, kcol, ccol)
                for r in con.execute(q):
                    if "hevcdl" in str(r[0]):
                        print("%s | %s = %s (sum over %d rows)" % r)
            else:
                print("counters_collection columns:", cols)
        con.close()
