import sqlite3, sys, glob
for path in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    suf = [t for t in tabs if t.startswith("rocpd_metadata")][0][len("rocpd_metadata"):]
    q = ("select s.kernel_name, d.grid_size_x, d.workgroup_size_x, (d.end-d.start)/1e3 from rocpd_kernel_dispatch%s d join rocpd_info_kernel_symbol%s s on d.kernel_id = s.id where s.kernel_name like '%%mfft%%' order by d.start" % (suf, suf))
    rows = list(c.execute(q))
    print(len(rows))
    for r in rows[-40:]:
        print(r[1], r[2], round(r[3], 1))
