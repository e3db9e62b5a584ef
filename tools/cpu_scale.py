import ctypes, threading, time, os, sys
lib = ctypes.CDLL(os.path.join(sys.argv[1], "tests/native/libhost_hooks.so"))
n = 100000
def work():
    out = ctypes.create_string_buffer(32 * n)
    lib.hh_transcript_rng(out, ctypes.c_size_t(n), b"x", ctypes.c_size_t(1), bytes(32), bytes(32))
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for T in (1, 2, 4, 8, 16, 32):
    th = [threading.Thread(target=work) for _ in range(T)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; dt = time.perf_counter() - t0
    print("threads %2d: %.1f ns per scalar per thread, aggregate %.1f M scalars/s" % (T, dt / n * 1e9, T * n / dt / 1e6))
