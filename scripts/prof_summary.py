"""Turns a rocprofv3 rocpd database (--kernel-trace --stats) into the CSV summary kept under profiles/."""
import sqlite3
import sys

db_path, out_path = sys.argv[1], sys.argv[2]
header = sys.argv[3:] 
db = sqlite3.connect(db_path)
cur = db.cursor()
suffix = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace('rocpd_kernel_dispatch', '')
q = """select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), sum(d.end-d.start), max(s.arch_vgpr_count), max(s.accum_vgpr_count),
       max(s.sgpr_count), max(d.group_segment_size), max(d.private_segment_size), max(d.grid_size_x), max(d.workgroup_size_x)
       from rocpd_kernel_dispatch%s d join rocpd_info_kernel_symbol%s s on d.kernel_id=s.id group by s.kernel_name order by 6 desc""" % (suffix, suffix)
rows = list(cur.execute(q))
total = sum(r[5] for r in rows)
lines = ["# " + h for h in header]
lines.append("kernel,calls,avg_us,min_us,max_us,total_ms,percent,arch_vgpr,accum_vgpr,sgpr,lds_bytes,scratch_bytes,grid,workgroup")
for r in rows:
    lines.append("%s,%d,%.1f,%.1f,%.1f,%.2f,%.1f,%d,%d,%d,%d,%d,%d,%d" % (r[0].replace('.kd', ''), r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e6,
                                                                       100.0 * r[5] / total, r[6], r[7], r[8], r[9], r[10], r[11], r[12]))
open(out_path, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
