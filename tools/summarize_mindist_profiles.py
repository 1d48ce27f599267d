"""gpurun_out/r05_l2 (tools/r05_gpu_l2.sh: bench.py --mode mindist under rocprofv3 --stats and two --pmc passes) ->
profiles/r05_mindist_kernel_stats_quad.csv, r05_mindist_bench_quad.json, r05_mindist_pmc_quad.txt, mindist_pmc.json"""
import csv, glob, collections, json, os, shutil
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(REPO, "gpurun_out", "r05_l2")
P = os.path.join(REPO, "profiles")
shutil.copy(os.path.join(O, "mindist_stats", "run_kernel_stats.csv"), os.path.join(P, "r05_mindist_kernel_stats_quad.csv"))
open(os.path.join(P, "r05_mindist_bench_quad.json"), "w").write(open(os.path.join(O, "mindist.json")).read().strip().splitlines()[-1] + "\n")
lines = ["# bench.py --mode mindist on the tree as round 5 ends: k_min_dist_quad / k_min_dist2poly_quad (a 16-lane row of the wavefront per child,",
         "# the children's gjkNew calls in lockstep) and the robust searches; rocprofv3 --pmc, two passes (tools/r05_gpu_l2.sh);",
         "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles"]
vals = {}
for d in ("mindist_pmc", "mindist_pmc2"):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(O, d, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].split("::")[-1]
            if "min_dist" in k:
                acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        lines.append("%-28s %-22s launches=%d mean=%.5g" % (k, c, len(v), sum(v) / len(v)))
        vals[(k, c)] = sum(v) / len(v)
dur = {}
for r in csv.DictReader(open(os.path.join(P, "r05_mindist_kernel_stats_quad.csv"))):
    if "min_dist" in r["Name"]:
        dur[r["Name"].split("(")[0].split("::")[-1].replace("void ", "")] = float(r["AverageNs"])
out = {}
for k, name in (("k_min_dist_quad", "reference_algorithm"), ("k_min_dist_robust<11>", "robust"),
                ("k_min_dist2poly_quad", "curve_polygon_reference_algorithm"), ("k_min_dist2poly_robust<11>", "curve_polygon_robust")):
    if k not in dur or (k, "SQ_INSTS_VALU") not in vals:
        continue
    ms = dur[k] / 1e6
    clk = ms * 1e-3 * 2.4e9
    busy = vals[(k, "SQ_INSTS_VALU")] * 4 / (1024 * clk)
    lines.append("# %s: %.3f ms per launch (rocprofv3 --kernel-trace --stats, profiles/r05_mindist_kernel_stats_quad.csv) = %.1f M clocks at 2.4 GHz;" % (k, ms, clk / 1e6))
    lines.append("#   %.3g VALU wave-instructions x 4 clocks / 1024 SIMDs = %.1f %% of the issue slots" % (vals[(k, "SQ_INSTS_VALU")], 100 * busy))
    out[name] = {"valu_busy": round(busy, 3), "kernel": k, "source": "profiles/r05_mindist_pmc_quad.txt: SQ_INSTS_VALU x 4 clocks / (1024 SIMDs x launch clocks)"}
lines += ["# curve <-> curve before (profiles/r05_mindist_pmc_after.txt): k_min_dist_wave 22.37 ms under the profiler (19.05 ms in the bench), 4.34 G VALU",
          "#   wave-instructions, 31.6 % of the issue slots: the quad form issues 0.45 x the wave instructions (one instruction advances the gjkNew calls of",
          "#   four children) and its slowest pair's chain is 0.4 x as long; the first quad form counted (before own points in registers and the batched",
          "#   convergence test): 7.744 ms, 1.926 G VALU, 40.5 %",
          "# curve <-> polygon before (same bench, k_min_dist2poly_wave, one wavefront per gjkNew call): 5.301 ms per launch, 1.367e8 VALU wave-instructions",
          "#   (2.7 % of the issue slots); both forms are bound by the 3 of 4096 pairs whose inner gjkNew runs to md_cap = 4096 rounds -- a launch is as long",
          "#   as one such call, and a lockstep trip of 16-lane rows is shorter than a wavefront-wide one"]
open(os.path.join(P, "r05_mindist_pmc_quad.txt"), "w").write("\n".join(lines) + "\n")
json.dump(out, open(os.path.join(P, "mindist_pmc.json"), "w"), indent=1)
print("\n".join(l for l in lines if l.startswith("# k_") or l.startswith("#   ")))
