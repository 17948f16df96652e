"""Does hipExtStreamCreateWithCUMask restrict a stream to a CU subset on this box?  (time a VALU-heavy torch op)"""
import ctypes, os, sys, time, torch
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(ncu_enabled, total=256, pattern="low"):
    words = (ctypes.c_uint32 * (total // 32))()
    for i in range(total):
        on = (i < ncu_enabled) if pattern == "low" else (i % (total // ncu_enabled) == 0)
        if on: words[i // 32] |= (1 << (i % 32))
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(total // 32), words)
    print("create rc", rc, "mask", [hex(w) for w in words])
    return s
torch.cuda.init()
x = torch.randn(64 * 1024 * 1024, device="cuda")
def work():
    y = x
    for _ in range(4): y = torch.erf(torch.sin(y) * 1.01)      # VALU heavy elementwise
    return y
def timeit(stream):
    with torch.cuda.stream(stream):
        work(); stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): work()
        stream.synchronize()
    return (time.perf_counter() - t0) / 5 * 1e3
print("default stream ms", timeit(torch.cuda.current_stream()))
for n, pat in ((256, "low"), (128, "low"), (64, "low"), (64, "spread"), (32, "low")):
    s = masked_stream(n, pattern=pat)
    ext = torch.cuda.ExternalStream(s.value)
    print(n, pat, "CUs: ms", timeit(ext), flush=True)
