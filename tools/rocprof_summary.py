#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (bench_results.db) into the plain-text per-kernel summary we commit under profiles/.

usage: tools/rocprof_summary.py <kernel-trace db> [<pmc db> ...] > profiles/<name>.txt
       tools/rocprof_summary.py --traffic-json profiles/pmc_traffic.json --frames N <FETCH_SIZE db> <WRITE_SIZE db>
         (HBM bytes per launch of every kernel = FETCH_SIZE + WRITE_SIZE, KiB -> bytes; bench.py reads the file for roofline.traffic)
"""
import json, re, sqlite3, sys


def short_name(mangled):
    m = re.match(r"_ZN4cfhd3dev(\d+)", mangled)
    if m:
        start = m.end(); return mangled[start:start + int(m.group(1))]
    return mangled.replace(".kd", "")


# kernels whose global loads are 16 bytes per lane: on gfx950 FETCH_SIZE counts such streaming reads at half their size
# (MI355X_MICROARCH.md, HBM section; confirmed here by k_ent_pack, a plain copy: WRITE_SIZE = 2 x FETCH_SIZE)
WIDE_LOADS = ("k_ent_count", "k_ent_emit", "k_ent_pack")


def traffic_json(out, frames, dbs):
    acc = {}
    for path in dbs:
        cur = sqlite3.connect(path).cursor()
        t = tables(cur)
        rows = cur.execute("select s.kernel_name, p.name, count(*), max(e.value) from %s e join %s p on e.pmc_id=p.id join %s k on e.event_id=k.event_id "
                           "join %s s on k.kernel_id=s.id group by s.kernel_name, p.name" % (t["pmc_event"], t["info_pmc"], t["kernel_dispatch"], t["info_kernel_symbol"])).fetchall()
        for name, counter, n, mx in rows:
            # kernels launched once per wavelet level (k_fwd_plane / k_inv_plane) differ per launch: keep the largest launch
            acc.setdefault(short_name(name), {})[counter] = mx * 1024.0
    kernels = {}
    for k, v in acc.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            fetch = v["FETCH_SIZE"] * (2 if k in WIDE_LOADS else 1)
            kernels[k] = {"fetch_bytes_per_launch": int(fetch), "write_bytes_per_launch": int(v["WRITE_SIZE"]),
                          "hbm_bytes_per_launch": int(fetch + v["WRITE_SIZE"]), "fetch_counter_scale": 2 if k in WIDE_LOADS else 1}
    json.dump({"frames_per_launch": frames, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), KiB * 1024, largest launch of each kernel; "
               "calibration: k_fwd_yuv422 (dword loads) WRITE_SIZE equals its band bytes exactly and FETCH_SIZE = packed input + 3-5 % halo, so dword-load kernels are taken as reported; "
               "kernels with 16-byte-per-lane loads (k_ent_count, k_ent_emit, k_ent_pack) are doubled as MI355X_MICROARCH.md prescribes (k_ent_pack, a plain copy, reports FETCH = WRITE / 2)",
               "kernels": kernels}, open(out, "w"), indent=1, sort_keys=True)


def tables(cur):
    names = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    suffix = [n for n in names if n.startswith("rocpd_metadata_")][0][len("rocpd_metadata_"):]
    return {k: "rocpd_%s_%s" % (k, suffix) for k in ("kernel_dispatch", "info_kernel_symbol", "pmc_event", "info_pmc", "memory_copy")}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--traffic-json":
        return traffic_json(sys.argv[2], int(sys.argv[4]), sys.argv[5:])
    for path in sys.argv[1:]:
        cur = sqlite3.connect(path).cursor()
        t = tables(cur)
        print("== %s" % path)
        rows = cur.execute("select s.kernel_name, count(*), avg(k.end-k.start), min(k.end-k.start), max(k.end-k.start), sum(k.end-k.start), "
                           "max(k.grid_size_x/k.workgroup_size_x), max(k.grid_size_y), max(k.grid_size_z), max(k.group_segment_size) "
                           "from %s k join %s s on k.kernel_id=s.id group by s.kernel_name order by 6 desc" % (t["kernel_dispatch"], t["info_kernel_symbol"])).fetchall()
        total = sum(r[5] for r in rows) or 1
        print("%-58s %6s %12s %12s %12s %12s %7s  %s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "total_us", "pct", "grid(x,y,z) lds"))
        for r in rows:
            print("%-58s %6d %12.2f %12.2f %12.2f %12.2f %6.2f%%  (%d,%d,%d) %d" % (r[0][:58], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[5] / total, r[6], r[7], r[8], r[9]))
        n = cur.execute("select count(*) from %s" % t["pmc_event"]).fetchone()[0]
        if n:
            print("-- PMC (average per dispatch; FETCH_SIZE / WRITE_SIZE are in KiB as reported by rocprofv3)")
            rows = cur.execute("select s.kernel_name, p.name, count(*), avg(e.value), min(e.value), max(e.value) from %s e join %s p on e.pmc_id=p.id "
                               "join %s k on e.event_id=k.event_id join %s s on k.kernel_id=s.id group by s.kernel_name, p.name order by 1"
                               % (t["pmc_event"], t["info_pmc"], t["kernel_dispatch"], t["info_kernel_symbol"])).fetchall()
            for r in rows:
                print("%-58s %-12s n=%-4d avg=%14.1f min=%14.1f max=%14.1f" % (r[0][:58], r[1], r[2], r[3], r[4], r[5]))
        print()


if __name__ == "__main__":
    main()
