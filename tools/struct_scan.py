"""Interleaved scan of the structured step's launch parameters in ONE process (the launch reads OBTG_STRUCT_* per call):
    python tools/struct_scan.py C5 "OBTG_STRUCT_PER16=7,4,4,1" "OBTG_STRUCT_PER16=8,3,3,2;OBTG_STRUCT_SEP_WGS=4096" ...
every configuration ROUNDS times in turn, N launches each; prints median / min ms per launch.  "" = the defaults."""
import os
import sys
import time

sys.path.insert(0, ".")
sys.argv, ARGS = sys.argv[:2], sys.argv[2:]
exec(open("tools/timeline_structured_run.py").read().split("import time")[0])     # the workload's context and buffers (WL = argv[1])
import numpy as np

KEYS = ("OBTG_STRUCT_PER16", "OBTG_STRUCT_SEP_WGS", "OBTG_STRUCT_GJK_WGS", "OBTG_STRUCT_GJK_CHUNK", "OBTG_STRUCT_DYN_ROWS", "OBTG_STRUCT_DYN_SPLIT")
ROUNDS, N = int(os.environ.get("SCAN_ROUNDS", "5")), int(os.environ.get("SCAN_N", "60"))


def run(n):
    for _ in range(n):
        ctx.constraint_sweep_fd_structured_dev(d0.data_ptr(), 1, synth.FD_STEP, dtf.data_ptr(), B, 0.9, sep.data_ptr(), 5.0, True, 1.0,
                                               sp.data_ptr(), an.data_ptr(), flag.data_ptr(), p1.data_ptr(), p2.data_ptr(), dist.data_ptr(), None, st.data_ptr(), 128, 256)


run(300)
torch.cuda.synchronize()
res = {a: [] for a in ARGS}
for r in range(ROUNDS):
    for a in ARGS:
        for k in KEYS:
            os.environ.pop(k, None)
        for kv in filter(None, a.split(";")):
            k, v = kv.split("=")
            os.environ[k] = v
        run(10)
        torch.cuda.synchronize()
        t = time.perf_counter()
        run(N)
        torch.cuda.synchronize()
        res[a].append((time.perf_counter() - t) * 1e3 / N)
for a in ARGS:
    print("%-70s median %.4f  min %.4f ms" % (a or "(defaults)", float(np.median(res[a])), min(res[a])))
