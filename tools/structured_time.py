import sys, time
sys.path.insert(0, ".")
exec(open("tools/timeline_structured_run.py").read().split("for i in range(600")[0])
def run(n):
    for i in range(n):
        ctx.constraint_sweep_fd_structured_dev(d0.data_ptr(), 1, synth.FD_STEP, dtf.data_ptr(), B, 0.9, sep.data_ptr(), 5.0, True, 1.0,
                                               sp.data_ptr(), an.data_ptr(), flag.data_ptr(), p1.data_ptr(), p2.data_ptr(), dist.data_ptr(), None, st.data_ptr(), 128, 256)
run(2500 if WL == "C3" else (300 if WL == "C5" else 10)); torch.cuda.synchronize()
n = 500 if WL == "C3" else (100 if WL == "C5" else 10)
t=time.perf_counter(); run(n); torch.cuda.synchronize(); print("%.4f ms" % ((time.perf_counter()-t)*1e3/n))
