import os
aff = sorted(os.sched_getaffinity(0))
print("affinity", len(aff), aff)
seen = {}
for c in aff:
    try:
        l3 = open("/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list" % c).read().strip()
        sib = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip()
        l2 = open("/sys/devices/system/cpu/cpu%d/cache/index2/size" % c).read().strip()
        l3s = open("/sys/devices/system/cpu/cpu%d/cache/index3/size" % c).read().strip()
    except OSError as e:
        l3 = sib = l2 = l3s = str(e)
    print(c, "L3 shared with", l3, "| SMT siblings", sib, "| L2", l2, "L3", l3s)
os.system("lscpu | head -30; cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc")
