"""Turn the raw rocprofv3 output of tools/collect_profiles.sh into the files kept under profiles/.

    python tools/parse_profiles.py r01_d [--workload C3]

* profiles/<tag>_kernel_stats.csv, <tag>_fd_dedup_kernel_stats.csv   (rocprofv3 --stats summaries)
* profiles/<tag>_bench.json, <tag>_bench_with_cpu_baseline.json       (bench.py lines of the same runs)
* profiles/traffic.json    HBM bytes per launch from the two PMC passes:
      bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024     (MI355X_MICROARCH.md: gfx950 FETCH_SIZE x2, KB units)
"""
import csv, glob, json, os, shutil, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAMILY = (("pair_sweep", "k_pair_sweep"), ("pair_sweep", "k_step_fd_structured"), ("temporal_sep", "k_sep_dynamics_elev"), ("temporal_sep", "k_normsq_elev"), ("ang_rate", "k_dynamics"), ("ang_rate", "k_ang_rate"),
          ("speed", "k_speed"), ("gjk", "k_gjk_swarm"))


def pmc_mean(dirname, counter):
    """kernel name -> mean counter value per dispatch."""
    acc = {}
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                a = acc.setdefault(row["Kernel_Name"], [0.0, 0])
                a[0] += float(row["Counter_Value"]); a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}


def main():
    tag = sys.argv[1]
    workload = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "C3"
    src = os.path.join(REPO, "gpurun_out", tag)
    dst = os.path.join(REPO, "profiles")
    for sub, name in (("stats", f"{tag}_kernel_stats.csv"), ("stats_dedup", f"{tag}_fd_dedup_kernel_stats.csv"),
                      ("stats_2s", f"{tag}_two_streams_kernel_stats.csv"), ("stats_c5", f"{tag}_C5_kernel_stats.csv"),
                      ("stats_structured", f"{tag}_fd_structured_kernel_stats.csv"),
                      ("stats_structured_c5", f"{tag}_C5_fd_structured_kernel_stats.csv"),
                      ("stats_c4", f"{tag}_C4_kernel_stats.csv"),
                      ("stats_structured_c4", f"{tag}_C4_fd_structured_kernel_stats.csv")):
        hits = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
        if hits:
            shutil.copy(hits[0], os.path.join(dst, name))
    for a, b in (("bench_stats.json", f"{tag}_bench.json"), ("bench_default.json", f"{tag}_bench_with_cpu_baseline.json"),
                 ("bench_stats_2s.json", f"{tag}_two_streams_bench.json"), ("bench_stats_c5.json", f"{tag}_C5_bench.json"),
                 ("bench_stats_c4.json", f"{tag}_C4_bench.json")):
        if os.path.exists(os.path.join(src, a)):
            shutil.copy(os.path.join(src, a), os.path.join(dst, b))
    tpath = os.path.join(dst, "traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    for wl, suffix in ((workload, ""), ("C5", "_c5"), (workload + "_fd_structured", "_structured"),
                       ("C5_fd_structured", "_structured_c5"), ("C4_fd_structured", "_structured_c4")):
        fetch = pmc_mean(os.path.join(src, "fetch" + suffix), "FETCH_SIZE")
        write = pmc_mean(os.path.join(src, "write" + suffix), "WRITE_SIZE")
        per, detail = {}, {}
        for fam, needle in FAMILY:
            for k in fetch:
                if needle in k and k in write and fam not in per:
                    b = int(round((2.0 * fetch[k] + write[k]) * 1024))
                    per[fam] = b
                    detail[fam] = dict(kernel=k, FETCH_SIZE_KB=round(fetch[k], 1), WRITE_SIZE_KB=round(write[k], 1),
                                       hbm_bytes_per_launch=b)
        if per:
            traffic[wl] = per
            traffic.setdefault("_detail", {})[wl] = detail
            traffic["_source"] = tag
        print(wl, json.dumps(detail, indent=1))
    json.dump(traffic, open(tpath, "w"), indent=1)


if __name__ == "__main__":
    main()
