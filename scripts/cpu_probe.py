import sys, os, json
sys.path.insert(0, '.')
import bench
print("cores", bench.host_cores(), "affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
try: print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e: print("no cpu.max", e)
for n in (16, 18):
    print(n, bench.cpu_baseline(n, 8, 1, budget_s=120))
