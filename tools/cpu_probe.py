import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from bench import cpu_baseline
from fvta_memexqa_amd.synth import CONFIGS
nt = int(sys.argv[1]); n = int(sys.argv[2])
import bench
bench.host_cores = lambda: nt
t0 = time.time()
r = cpu_baseline(dict(CONFIGS["metric"], dense=True), n, False)
print(nt, n, r, "wall %.1f" % (time.time() - t0), flush=True)
