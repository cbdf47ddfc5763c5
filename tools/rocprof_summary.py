#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (bench_results.db) into the plain-text per-kernel summary we commit under profiles/.

usage: tools/rocprof_summary.py <kernel-trace db> [<pmc db> ...] > profiles/<name>.txt
       tools/rocprof_summary.py --traffic-json profiles/pmc_traffic.json --frames N [--workload W] <FETCH_SIZE db> <WRITE_SIZE db>
         (HBM bytes per launch of every kernel = FETCH_SIZE + WRITE_SIZE, KiB -> bytes; bench.py reads the file for roofline.traffic)
"""
import json, re, sqlite3, sys


def short_name(mangled):
    m = re.match(r"_ZN4cfhd3dev(\d+)", mangled)
    if m:
        start = m.end(); return mangled[start:start + int(m.group(1))]
    return mangled.replace(".kd", "")


# FETCH_SIZE on gfx950 tallies the 128-byte requests of a coalesced streaming read at 64 bytes (MI355X_MICROARCH.md, HBM section: "double
# it before comparing with a byte count").  Calibration in our own access patterns: k_ent_pack is a plain copy and reports FETCH = WRITE / 2;
# every other kernel's doubled FETCH lands 0-4 % above the bytes it must read at least once (k_fwd_yuv422 1.001 x its packed input,
# k_inv_yuv422 1.000 x its twelve bands, k_ent_count 1.03 x the coded bands), so the factor 2 is applied to all of them.  WRITE_SIZE is exact
# (k_fwd_yuv422 writes its band bytes to the byte).
FETCH_SCALE = 2


def traffic_json(out, frames, dbs, workload="1080p"):
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
            fetch = v["FETCH_SIZE"] * (FETCH_SCALE if k.startswith("k_") else 1)
            kernels[k] = {"fetch_bytes_per_launch": int(fetch), "write_bytes_per_launch": int(v["WRITE_SIZE"]),
                          "hbm_bytes_per_launch": int(fetch + v["WRITE_SIZE"]), "fetch_counter_scale": FETCH_SCALE if k.startswith("k_") else 1}
    json.dump({"frames_per_launch": frames, "workload": workload, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), KiB * 1024, largest launch of each kernel; "
               "FETCH_SIZE of our kernels doubled (gfx950 counts coalesced 128-byte read requests at 64 bytes; calibrated on k_ent_pack, a plain copy, and on the read-once lower bound of every kernel), WRITE_SIZE as reported",
               "kernels": kernels}, open(out, "w"), indent=1, sort_keys=True)


def tables(cur):
    names = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    suffix = [n for n in names if n.startswith("rocpd_metadata_")][0][len("rocpd_metadata_"):]
    return {k: "rocpd_%s_%s" % (k, suffix) for k in ("kernel_dispatch", "info_kernel_symbol", "pmc_event", "info_pmc", "memory_copy")}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--traffic-json":
        rest = sys.argv[5:]
        workload = "1080p"
        if rest and rest[0] == "--workload": workload = rest[1]; rest = rest[2:]
        return traffic_json(sys.argv[2], int(sys.argv[4]), rest, workload)
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
