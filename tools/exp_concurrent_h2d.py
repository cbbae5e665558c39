"""What stalls the first concurrent pageable host-to-device copies of a process?  N host threads, each with its own stream, host
buffer and device buffer, copy `size` bytes per round (hipMemcpyAsync from pageable memory + hipStreamSynchronize), all threads
released together per round.  Prints per round the slowest thread's time: a stall shows as a round of several ms.
python tools/exp_concurrent_h2d.py N size_bytes [rounds]"""
import ctypes, sys, threading, time
hip = ctypes.CDLL("libamdhip64.so")
N = int(sys.argv[1]); size = int(sys.argv[2]); rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 14
vp = ctypes.c_void_p
hip.hipMalloc.argtypes = [ctypes.POINTER(vp), ctypes.c_size_t]
hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(vp), ctypes.c_uint]
hip.hipMemcpyAsync.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_int, vp]
hip.hipStreamSynchronize.argtypes = [vp]
assert hip.hipSetDevice(0) == 0
bufs = []
for i in range(N):
    d = vp(); s = vp()
    assert hip.hipMalloc(ctypes.byref(d), size) == 0 and hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
    h = ctypes.create_string_buffer(bytes(size))          # pageable, touched
    bufs.append((d, s, h))
times = [[0.0] * N for _ in range(rounds)]
bar = threading.Barrier(N)
def worker(i):
    hip.hipSetDevice(0)
    d, s, h = bufs[i]
    for r in range(rounds):
        bar.wait()
        t0 = time.perf_counter()
        hip.hipMemcpyAsync(d, ctypes.cast(h, vp), size, 1, s)
        hip.hipStreamSynchronize(s)
        times[r][i] = (time.perf_counter() - t0) * 1e3
th = [threading.Thread(target=worker, args=(i,)) for i in range(N)]
[t.start() for t in th]; [t.join() for t in th]
print("N=%d size=%d KB: slowest thread per round (ms): %s" % (N, size >> 10, " ".join("%.2f" % max(r) for r in times)))
