"""Fills the tables of DESIGN.md (the blocks between <!--RESULTS_TABLE--> ... markers, the HOSTFED / CABI / CPU lines) from a bench line and the PMC traffic file:
   python tools/fill_design_tables.py profiles/r06_v_default_bench_line.json profiles/pmc_traffic.json
The VALU column is SQ_INSTS_VALU / (8 x SQ_BUSY_CYCLES) of profiles/r06_u_1080p_batch512_sq_counters.txt (tools/profile_sq.sh)."""
import json, sys
bench = sys.argv[1]; pmcf = sys.argv[2]
d = json.loads(open(bench).read().strip().splitlines()[-1]); c = d["config"]
k = c["kernel_ms_one_step_at_a_time"]; kr = c["kernel_ms_per_step"]
pm = json.load(open(pmcf))["kernels"]
S = 4147200; P = 4147200; sb = c["sample_bytes_per_frame"]; coded = (S - S // 64) * 2
def name(prefix): return next(n for n in k if n.startswith(prefix))
rows = [("k_fwd_yuv422_strip_blocks", "P + 2S = 12 441 600", P + 2 * S, "k_fwd_yuv422_strip_blocks", "85 %"),
        ("k_fwd_plane_strip[L2]", "S", S, None, ""), ("k_fwd_plane_strip[L3]", "S ÷ 4", S // 4, None, ""),
        (name("k_ent_count_blocks"), "level-1 bands as block lists (3S ÷ 4 coefficients)", S * 3 // 4 * 2, "k_ent_count_blocks", "80 %"),
        ("k_ent_count", "coded bands of levels 2, 3", coded - S * 3 // 4 * 2, "k_ent_count", "75 %"),
        ("k_ent_emit", "token strings + sample", coded // 2 + sb, "k_ent_emit", "31 % (70 % of wave cycles wait)"),
        ("k_dec_index", "sample", sb, "k_dec_index", "59 % (71 % of wave cycles wait: LDS lookups)"),
        ("k_dec_tiles", "sample + coded bands", sb + coded, "k_dec_tiles", "54 % (63 % of wave cycles wait: latency)"),
        ("k_inv_plane_strip[L3]", "S ÷ 4", S // 4, None, ""), ("k_inv_plane_strip[L2]", "S", S, None, ""),
        ("k_inv_yuv422_strip_blocks", "2S + P = 12 441 600", 2 * S + P, "k_inv_yuv422_strip_blocks", "85 %")]
out = ["| kernel | algorithmic bytes / frame | ms per 512 frames (one step at a time; as run with four in flight) | GB/s | % of 8 TB/s | PMC traffic / algorithmic | VALU issue |", "|---|---|---|---|---|---|---|"]
for n, what, b, pk, valu in rows:
    ms = k[n]; gbs = b * 512 / (ms * 1e-3) / 1e9
    tr = "%.2f" % (pm[pk]["hbm_bytes_per_launch"] / (b * 512)) if pk and pk in pm else ""
    bold = "**" if n == d["roofline"]["kernel"] else ""
    out.append("| `%s` | %s | %s%.2f%s (%.2f) | %.0f | %s%.0f %%%s | %s | %s |" % (n.split("[L1")[0].strip(), what, bold, ms, bold, kr[n], gbs, bold, 100 * gbs / 8000, bold, tr, valu))
small = {n: k[n] for n in ("k_ent_scan", "k_ent_layout", "k_dec_parse", "k_dec_plan", "k_dec_chain", "k_dec_lowpass")}
tot_traffic = sum(v["hbm_bytes_per_launch"] for n, v in pm.items() if n.startswith("k_"))
txt = "\n".join(out)
txt += "\n\n(" + ", ".join("`%s` %.2f" % (n, v) for n, v in small.items()) + " ms; `k_dec_plan` is one workgroup that numbers the chunks in front of `k_dec_index`, `k_dec_chain` includes repair, re-index and the tile records.)  Σ PMC traffic of a step %.1f GB = %.2f × its algorithmic bytes.\n" % (tot_traffic / 1e9, tot_traffic / (24883200 * 512))
r = d["roofline"]; wp = c["whole_path"]; w16 = c.get("with_16_hardware_queues", {})
txt += "\n**The step: %.2f ms = %.1f k fps with four steps in flight on the runtime's default hardware queues** (round 5: 50.8 k with that setting, 59.0 k with 16 queues; this tree with `GPU_MAX_HW_QUEUES=16`: %s k, `config.with_16_hardware_queues`) — the whole path at %.2f TB/s = %.1f %% of the peak; Σ kernels one step at a time %.1f ms (round 5: 11.6).  `roofline`: `%s`, %.2f ms, %.0f GB/s = **%.3f** of the peak (%.3f in real HBM bytes: PMC traffic %.2f GB).  Round 5's dominant kernel `k_dec_tiles` is now %.2f ms (0.34 → %.2f nominal), the two level-1 transforms %.2f / %.2f ms (1.19 / 1.44 before this round's cuts in their instruction count: %.2f / %.2f of the peak in algorithmic bytes) — which leaves the longest launch of the step to the two entropy kernels that are bound by neither bytes nor arithmetic (`k_ent_emit`, `k_dec_index`: §5.1).\n" % (
    d["ms_per_step"], d["value"] / 1e3, ("%.1f" % (w16.get("value", 0) / 1e3)) if w16.get("value") else "-", wp["gbs"] / 1e3, 100 * wp["frac_of_hbm_peak"], wp["sum_of_kernels_ms"], r["kernel"], r["launch_ms"], r["achieved"], r["frac"],
    r.get("hbm_frac_from_pmc_traffic") or 0, (r.get("traffic") or 0) / 1e9, k["k_dec_tiles"], (sb + coded) * 512 / (k["k_dec_tiles"] * 1e-3) / 1e9 / 8000,
    k["k_fwd_yuv422_strip_blocks"], k["k_inv_yuv422_strip_blocks"], (P + 2 * S) * 512 / (k["k_fwd_yuv422_strip_blocks"] * 1e-3) / 1e9 / 8000, (P + 2 * S) * 512 / (k["k_inv_yuv422_strip_blocks"] * 1e-3) / 1e9 / 8000)
o = c.get("other_workloads", {})
txt += "\n| workload (BASELINE config) | frames per step | fps (round 5) | longest kernel, % of 8 TB/s (one step at a time) |\n|---|---|---|---|\n"
prev = {"2160p": "17 900", "rg48-2160p": "14 200-15 900", "b64a-4320p": "1 870", "byr4-2160p": "17 500-19 100", "1080i": "55 200"}
for n in ("2160p", "rg48-2160p", "b64a-4320p", "byr4-2160p", "1080i"):
    v = o.get(n, {})
    if "value" in v: txt += "| `%s` | %d | **%d** (%s) | `%s` %.0f %% |\n" % (n, v["frames_per_step"], round(v["value"], -1), prev[n], v["roofline"]["kernel"], 100 * v["roofline"]["frac"])
hf = d.get("host_fed", {}); a = hf.get("registered_buffers", {}); b_ = hf.get("plain_buffers", {})
hostfed = "**%.1f k fps** from page-locked buffers (%.0f GB/s over the link, both directions together), %.1f k from plain buffers (staged by eight threads)" % (a.get("fps", 0) / 1e3, a.get("pcie_gbs_both_directions", 0), b_.get("fps", 0) / 1e3)
ca = c.get("c_abi_fps", {}); pl = ca.get("plain_buffers", {}); p16 = ca.get("plain_buffers_16_threads", {}); rg = ca.get("buffers_registered_by_the_caller_16_threads", {})
g = lambda dct, pre: next((v for kk, v in dct.items() if kk.startswith(pre)), 0)
cabi = "round trip **%d fps** (the three runs: %s), pool encode %d, decode on 8 handles %d, synchronous %d / %d encode / decode; 16 + 16 threads: round trip %d; registered buffers (16 + 16): %d / %d synchronous, %d decode, %d pool encode" % (
    g(pl, "round_trip"), " / ".join("%d" % g(r_, "round_trip") for r_ in pl.get("runs", [])), g(pl, "pool_encode"), g(pl, "decode_fps"), pl.get("sync_encode_fps", 0), pl.get("sync_decode_fps", 0), g(p16, "round_trip"),
    rg.get("sync_encode_fps", 0), rg.get("sync_decode_fps", 0), g(rg, "decode_fps"), g(rg, "pool_encode"))
cb = d.get("cpu_baseline", {}); cpu = "**%s fps** round trip (%s)" % (cb.get("value"), cb.get("sample", "")[:160])
s = open("DESIGN.md").read()
import re
def put(tag, text):
    global s
    a = s.find("<!--%s-->" % tag); b = s.find("<!--/%s-->" % tag)
    if a >= 0 and b > a: s = s[:a + len(tag) + 7] + text + s[b:]
    else: s = s.replace(tag, "<!--%s-->%s<!--/%s-->" % (tag, text, tag))
put("RESULTS_TABLE", "\n" + txt + "\n"); put("HOSTFED_LINE", hostfed); put("CABI_LINE", cabi); put("CPU_LINE", cpu)
open("DESIGN.md", "w").write(s)
print(txt[:3000]); print(hostfed); print(cabi); print(cpu)
