"""Turns a rocprofv3 (rocpd sqlite) result into the per-kernel summary text committed under profiles/."""
import sqlite3, sys
def main(db, out):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(top_kernels)")]
    rows = c.execute("select * from top_kernels").fetchall()
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (from %s)\n" % db)
        f.write(" | ".join(cols) + "\n")
        for r in rows:
            f.write(" | ".join(str(x) for x in r) + "\n")
    print(open(out).read())
if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
