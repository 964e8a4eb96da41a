import ctypes as C, os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minidiff_amd import _capi, workloads
from minidiff_amd.tape import hip_engine
md = hip_engine(); lib = _capi.load()
st, step = workloads.make_cfg3(md, n=int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000)
s = (C.c_int64 * 4)()
for i in range(30):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); lib.sync(); t2 = time.perf_counter()
    lib.mem_stats(s)
    print(i, "enqueue %.3f ms total %.3f ms" % ((t1-t0)*1e3, (t2-t0)*1e3), "in_use %.1f GB cached %.1f GB nmalloc %d" % (s[0]/1e9, s[1]/1e9, s[3]), "gc", gc.get_count())
