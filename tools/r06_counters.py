#!/usr/bin/env python3
"""gpurun_out/<tag>/ (tools/r06_collect_counters.sh) -> profiles/counters.json + profiles/<tag>.txt

    python tools/r06_counters.py r06_counters

Every entry of counters.json is one (workload, kernel of the bench line) with what the passes measured for it and WHERE it
comes from: the obtg_source_hash of the compile unit the kernel lives in, as the library on the GPU box reported it
(meta.json, written by the collect script before the passes), the tree (git HEAD here; the hashes must be this tree's or the
tool refuses), and the text file the per-kernel means are kept in.  bench.py (counters_for) reports an entry only when the
RUNNING library's hash equals the entry's.

  hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024     MI355X_MICROARCH.md: KB units, gfx950 FETCH_SIZE x 2
  valu_busy_frac       = SQ_INSTS_VALU x 4 clocks / (1024 SIMDs x launch clocks), launch clocks = the stats pass's average
                         duration x 2.4 GHz (the counter passes themselves run the kernels several times slower)
  lds_bank_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import csv
import glob
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

# (workload of the pass, key in the bench line, needle in the kernel symbol, compile unit)
KERNELS = (
    ("C3", "C3", "pair_sweep", "k_pair_sweep<11>", "gjk_kernels"),
    ("C4", "C4", "pair_sweep", "k_pair_sweep_tiled<16>", "gjk_kernels"),
    ("C5", "C5", "temporal_sep", "k_sep_dynamics_elev<11>", "bern_kernels"),
    ("C5", "C5", "gjk", "k_gjk_swarm_planar<11, 0>", "gjk_kernels"),
    ("C2", "C2", "pair_sweep", "k_pair_sweep_3d<11>", "gjk_kernels"),
    ("C2_file", "C2_file", "pair_sweep", "k_pair_sweep_3d<6>", "gjk_kernels"),
    ("C3_fd_structured", "C3_fd_structured", "pair_sweep", "k_step_fd_structured<11, false", "gjk_kernels"),
    ("C5_fd_structured", "C5_fd_structured", "pair_sweep", "k_step_fd_structured<11, true", "gjk_kernels"),
    ("C4_fd_structured", "C4_fd_structured", "pair_sweep", "k_step_fd_structured<16, false", "gjk_kernels"),
    ("C5_mindist", "C5_mindist", "reference_algorithm", "k_min_dist_quad<true, 2, 11>", "gjk_kernels"),
    ("C5_mindist", "C5_mindist", "jacobian_list", "k_min_dist_quad<true, 3, 11>", "gjk_kernels"),
    ("C5_mindist", "C5_mindist", "curve_polygon_reference_algorithm", "k_min_dist2poly_quad<true, 11>", "gjk_kernels"),
)


def pmc_means(dirname):
    """(kernel symbol, counter) -> (mean per dispatch, dispatches)"""
    acc = {}
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                a = acc.setdefault((row["Kernel_Name"], row["Counter_Name"]), [0.0, 0])
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def stats_avg_ns(dirname):
    out = {}
    for path in glob.glob(os.path.join(dirname, "**", "*kernel_stats.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                out[row["Name"]] = (float(row["AverageNs"]), int(row["Calls"]))
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06_counters"
    src = os.path.join(REPO, "gpurun_out", tag)
    meta = json.load(open(os.path.join(src, "meta.json")))
    from optimalbeziertrajectorygeneration_amd import build
    local = build.unit_hashes()
    for u, h in meta.items():
        if local.get(u) != h:
            raise SystemExit("the passes under %s ran a library built from other sources than this tree's (%s: %s there, %s here): "
                             "collect them again on this tree" % (src, u, h, local.get(u)))
    head = subprocess.run(["git", "-C", REPO, "rev-parse", "--short", "HEAD"], stdout=subprocess.PIPE, universal_newlines=True).stdout.strip()
    dirty = subprocess.run(["git", "-C", REPO, "status", "--porcelain", "--", "optimalbeziertrajectorygeneration_amd/csrc", "include"],
                           stdout=subprocess.PIPE, universal_newlines=True).stdout.strip()
    tree = head + ("+uncommitted kernel edits" if dirty else "")
    txt_name = "profiles/%s.txt" % tag
    lines = ["# hardware counters of the bench's kernels, one tree, one gpurun call (tools/r06_collect_counters.sh); tree %s" % tree,
             "# obtg_source_hash on the box: %s" % json.dumps(meta),
             "# per kernel: mean per dispatch over the dispatches of the pass; FETCH_SIZE / WRITE_SIZE in KB; SQ_* are chip totals",
             "# (SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles); avg_ns from the --kernel-trace --stats pass of the same command"]
    entries = []
    cache = {}
    for wl_dir, wl, key, needle, unit in KERNELS:
        if wl_dir not in cache:
            cache[wl_dir] = {p: pmc_means(os.path.join(src, "%s__%s" % (wl_dir, p))) for p in ("fetch", "write", "issue", "lds")}
            cache[wl_dir]["stats"] = stats_avg_ns(os.path.join(src, "%s__stats" % wl_dir))
        c = cache[wl_dir]
        syms = sorted({k for (k, _) in c["issue"] if needle in k} | {k for (k, _) in c["write"] if needle in k})
        if not syms:
            print("no dispatches of %s in the %s passes" % (needle, wl_dir))
            continue
        sym = syms[0]

        def val(p, ctr):
            return c[p].get((sym, ctr), (None, 0))[0]
        avg = next((v for k, v in c["stats"].items() if needle in k), (None, 0))
        e = {"workload": wl, "kernel": key, "kernel_symbol": sym.replace("void obtg::", "").split("(")[0], "unit": unit,
             "source_hash": meta[unit], "tree": tree, "source": txt_name}
        fe, wr = val("fetch", "FETCH_SIZE"), val("write", "WRITE_SIZE")
        if fe is not None and wr is not None:
            e["fetch_kb"], e["write_kb"] = round(fe, 1), round(wr, 1)
            e["hbm_bytes_per_launch"] = int(round((2.0 * fe + wr) * 1024))
        iv = val("issue", "SQ_INSTS_VALU")
        if iv is not None:
            e["valu_wave_insts"] = iv
            e["salu_wave_insts"] = val("issue", "SQ_INSTS_SALU")
            e["lds_wave_insts"] = val("issue", "SQ_INSTS_LDS")
            e["waves"] = val("issue", "SQ_WAVES")
            if avg[0]:
                e["avg_ns_stats_pass"] = round(avg[0], 1)
                e["launches_stats_pass"] = avg[1]
                e["valu_busy_frac"] = round(iv * 4.0 / (1024.0 * avg[0] * 2.4), 4)
        la, lc = val("lds", "SQ_LDS_IDX_ACTIVE"), val("lds", "SQ_LDS_BANK_CONFLICT")
        if la:
            e["lds_bank_conflict_frac"] = round(lc / la, 4)
        entries.append(e)
        lines.append("")
        lines.append("## %s / %s: %s   (%s, obtg_source_hash %s)" % (wl, key, sym, unit, meta[unit]))
        if avg[0]:
            lines.append("   avg_ns %.1f over %d launches (stats pass)" % avg)
        for p in ("fetch", "write", "issue", "lds"):
            for (k, ctr), (m, cnt) in sorted(c[p].items()):
                if k == sym:
                    lines.append("   %-24s n=%-4d mean=%.6g" % (ctr, cnt, m))
        for k in ("hbm_bytes_per_launch", "valu_busy_frac", "lds_bank_conflict_frac"):
            if k in e:
                lines.append("   -> %s = %s" % (k, e[k]))
    out = {"_note": "written by tools/r06_counters.py from gpurun_out/%s; bench.py reports an entry only when the running library's "
                    "obtg_source_hash(unit) equals the entry's source_hash (bench.counters_for)" % tag,
           "entries": entries}
    json.dump(out, open(os.path.join(REPO, "profiles", "counters.json"), "w"), indent=1)
    open(os.path.join(REPO, txt_name), "w").write("\n".join(lines) + "\n")
    for e in entries:
        print(e["workload"], e["kernel"], e["kernel_symbol"], {k: e.get(k) for k in ("hbm_bytes_per_launch", "valu_busy_frac", "lds_bank_conflict_frac", "avg_ns_stats_pass")})


if __name__ == "__main__":
    main()
