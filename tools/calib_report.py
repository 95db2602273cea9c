"""Per-dispatch FETCH_SIZE / WRITE_SIZE (KiB -> bytes) of tools/calib_fetch.hip from the two rocprofv3 --pmc passes."""
import glob, os, sqlite3, sys
for sub, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    dbs = glob.glob(os.path.join(sys.argv[1], sub, "**", "*.db"), recursive=True)
    if not dbs:
        print(sub, "no db")
        continue
    c = sqlite3.connect(dbs[0])
    for name, val in c.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        print("%-10s %-28s %.0f bytes" % (counter, name.split("(")[0], val * 1024.0))
